import os, sys, time
sys.path.insert(0,"3d_recognizer_amd"); sys.path.insert(0,".")
import numpy as np, torch
import bench
from randlanet._train import TrainStep
dev=torch.device("cuda",0)
model=bench.build_model(dev); model.train()
st=TrainStep(model,4,40960,use_graph=True)
xyz,labels=bench.synthetic_batch(4,40960,2,1234)
st.set_batch(torch.from_numpy(xyz).to(dev), torch.from_numpy(labels).to(dev))
np.random.seed(1234)
st.capture()
def run(n, mode):
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    t0=time.perf_counter(); e0.record()
    for i in range(n):
        if mode=="replay_only":
            st._g_main.replay()
        elif mode=="fixed_perm":
            st.perm.copy_(st._perm_ring[0], non_blocking=True); st._g_main.replay()
        else:
            st.step(np.random.permutation(40960))
    e1.record(); torch.cuda.synchronize()
    print(mode, "wall %.2f ms/step  gpu %.2f ms/step"%((time.perf_counter()-t0)*1e3/n, e0.elapsed_time(e1)/n), "loss", st.last_metrics()["loss"], flush=True)
for mode in ("step","replay_only","fixed_perm","step","replay_only"):
    run(30, mode)

import time as _t
def timed_step(perm):
    t=[_t.perf_counter()]
    slot = st._perm_slot; st._perm_slot=(slot+1)%4
    if st._perm_events[slot] is not None: st._perm_events[slot].synchronize()
    t.append(_t.perf_counter())
    staging = st._perm_ring[slot]
    staging.copy_(torch.from_numpy(np.ascontiguousarray(perm, dtype=np.int64)))
    t.append(_t.perf_counter())
    st.perm.copy_(staging, non_blocking=True)
    t.append(_t.perf_counter())
    ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream(st.dev)); st._perm_events[slot]=ev
    t.append(_t.perf_counter())
    st._g_main.replay()
    t.append(_t.perf_counter())
    st.out_host.copy_(st.out, non_blocking=True)
    t.append(_t.perf_counter())
    return np.diff(t)*1e3
torch.cuda.synchronize()
acc=np.zeros(6); n=30
t0=_t.perf_counter()
for i in range(n):
    tp=_t.perf_counter(); p=np.random.permutation(40960); tperm=(_t.perf_counter()-tp)*1e3
    acc+=timed_step(p)
torch.cuda.synchronize()
print("wall %.2f ms/step; host ms: ev.sync %.3f staging %.3f h2d %.3f evrec %.3f replay %.3f d2h %.3f ; np.perm %.3f"%(((_t.perf_counter()-t0)*1e3/n,)+tuple(acc/n)+(tperm,)))
