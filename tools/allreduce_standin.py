#!/usr/bin/env python3
"""What the pipelined preparation (TrainStep(pipeline=True): the next step's coordinate-only kernels on a second stream) hides
beside a gradient all-reduce - measured on ONE GPU with a spin kernel of the collective's length standing in for it
(rl_spin_us: one idle wavefront on the step's stream between the network graph and the Adam graph, exactly where
sync_gradients sits in the multi-rank schedule).  Per (per-GPU batch, stand-in length): ms per step of the split schedule
with the plain order and with the pipelined preparation, and the single-graph step for reference.
usage: python tools/allreduce_standin.py [steps]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "3d_recognizer_amd"))
import numpy as np
import torch
from randlanet import _hip as H
from randlanet._train import TrainStep
from randlanet.utils.modules import RandLANet, RandLANetSettings

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
N, dev = 40960, torch.device("cuda", 0)
perm = np.random.RandomState(0).permutation(N)


def run(B, split, pipeline, spin_us):
    torch.manual_seed(0)
    net = RandLANet(RandLANetSettings(n_classes=2, n_points=N, n_neighbors=16, layer_sizes=[16, 64, 128, 256]), dev)
    net.train()
    st = TrainStep(net, B, N, loss="dice", use_graph=True, split_schedule=split, pipeline=pipeline)
    rs = np.random.RandomState(1)
    x = rs.uniform(0, 1, (B, N, 3)).astype(np.float32)
    y = (x[..., 2] > 0.5).astype(np.int64)
    st.set_batch(torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev))
    if split:
        st._allreduce = lambda: H.check(H.lib().rl_spin_us(spin_us, H.stream_ptr()), "rl_spin_us") if spin_us else None
    st.capture()
    for _ in range(20):
        st.step(perm)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            st.step(perm)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps * 1e3)
    return best


print(f"{'clouds/GPU':>10s} {'stand-in us':>11s} {'one graph':>10s} {'split plain':>12s} {'split pipelined':>16s}   (ms per step)")
for B in (1, 2, 4, 8):
    one = run(B, False, False, 0)
    for us in (0, 60, 120, 250):
        plain = run(B, True, False, us)
        piped = run(B, True, True, us)
        print(f"{B:10d} {us:11d} {one:10.3f} {plain:12.3f} {piped:16.3f}", flush=True)
