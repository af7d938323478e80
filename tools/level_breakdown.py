#!/usr/bin/env python3
"""Where a training step of the metric's configuration (N = 40960, bs = 8) spends its time, BY ENCODER LEVEL and kernel
(HIP events around every launch of an eager pass, the engine's level tag):  python tools/level_breakdown.py [batch]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "3d_recognizer_amd"))
import numpy as np
import torch
import bench
from randlanet import _hip as H
from randlanet import _ops as ops
from randlanet._train import TrainStep

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda")
m = bench.build_model(dev, 0)
m.train()
N = 40960
st = TrainStep(m, B, N, loss="dice", use_graph=False)
x, y = bench.synthetic_batch(B, N, 2, 1)
st.set_batch(torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev))
for _ in range(3):
    st.step(np.random.permutation(N))
torch.cuda.synchronize()
ops.TIMER = ops.KernelTimer()
steps = 5
for _ in range(steps):
    st.step(np.random.permutation(N))
torch.cuda.synchronize()
rec = ops.TIMER.records
ops.TIMER = None
by = {}
for cat, key, nbytes, flops, e0, e1, kern, lvl in rec:
    k = (lvl, bench.function_name(kern))
    a = by.setdefault(k, [0.0, 0])
    a[0] += e0.elapsed_time(e1)
    a[1] += 1
levels = sorted({k[0] for k in by})
tot = sum(v[0] for v in by.values()) / steps
print(f"batch {B}: eager kernel time {tot:.3f} ms per step, {sum(v[1] for v in by.values()) / steps:.0f} timed launches")
for lvl in levels:
    rows = sorted(((k[1], v) for k, v in by.items() if k[0] == lvl), key=lambda kv: -kv[1][0])
    lt = sum(v[0] for _, v in rows) / steps
    ln = sum(v[1] for _, v in rows) / steps
    print(f"\n== level {lvl}: {lt:.3f} ms, {ln:.0f} launches")
    for name, v in rows:
        print(f"   {name:34s} {v[0] / steps:7.3f} ms  x{v[1] / steps:5.1f}  {1e3 * v[0] / v[1]:7.1f} us")
