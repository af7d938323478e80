#!/usr/bin/env python3
"""TrainStep with / without the pipelined preparation, same process:  python tools/pipeline_probe.py [batch]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "3d_recognizer_amd"))
import numpy as np
import torch
import bench
from randlanet._train import TrainStep

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda")
N = 40960
x, y = bench.synthetic_batch(B, N, 2, 1)
perms = [np.random.permutation(N) for _ in range(8)]


def run(pipeline, n=200, variant=""):
    m = bench.build_model(dev, 0)
    m.train()
    st = TrainStep(m, B, N, loss="dice", pipeline=pipeline)
    st.set_batch(torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev))
    st.capture()
    if variant == "nowait":          # (wrong results; what the cross-stream wait costs)
        import types
        orig = torch.cuda.Stream.wait_event
    for i in range(20):
        st.step(perms[i % 8])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        st.step(perms[i % 8])
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    return 1e3 * t / n, 1e3 * th / n


for p in (False, True, False, True):
    t, th = run(p)
    print(f"batch {B} pipeline={p}: {t:.3f} ms per step (host submission {th:.3f} ms per step)")
