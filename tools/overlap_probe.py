#!/usr/bin/env python3
"""Does the coordinate-only work of a step (KNN of all levels + graph transposes: it depends on the input alone) hide beside the
network's kernels when it runs as its OWN graph on a second stream?  Times N replays of (a) the full step graph, (b) a graph with
knn_multi + csr_build alone, (c) both submitted to two streams, one (b) per (a).  python tools/overlap_probe.py [batch]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "3d_recognizer_amd"))
import numpy as np
import torch
import bench
from randlanet import _ops as ops
from randlanet._train import TrainStep

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda")
m = bench.build_model(dev, 0)
m.train()
N, K, L, dec = 40960, 16, 4, 4
st = TrainStep(m, B, N, loss="dice")
x, y = bench.synthetic_batch(B, N, 2, 1)
st.set_batch(torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev))
st.capture()

xyz = torch.from_numpy(x).to(dev).contiguous()
tasks, ratio = [], 1
for _ in range(L):
    tasks.append((N // ratio, N // ratio, K)); ratio *= dec
for _ in range(L):
    tasks.append((N // ratio, dec * N // ratio, 1)); ratio //= dec


def side_work():
    s = ops.knn_multi(xyz, tasks)
    return s, ops.csr_build([(s[i][0], tasks[i][0]) for i in range(2 * L)])


side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for _ in range(2):
        keep = side_work()
torch.cuda.synchronize()
g_side = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    with torch.cuda.graph(g_side, stream=side):
        keep = side_work()
torch.cuda.synchronize()
perm = np.random.permutation(N)


def timeit(fn, n=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


def main_only():
    st.step(perm)


def side_only():
    with torch.cuda.stream(side):
        g_side.replay()


def both():
    with torch.cuda.stream(side):
        g_side.replay()
    st.step(perm)


def both_event():
    # the dependency a pipelined step would have: main graph t waits for side graph t (issued one step earlier)
    ev = torch.cuda.Event()
    with torch.cuda.stream(side):
        g_side.replay()
        ev.record(side)
    st.step(perm)
    torch.cuda.current_stream().wait_event(ev)


a, b = timeit(main_only), timeit(side_only)
c, d = timeit(both), timeit(both_event)
print(f"batch {B}: full step graph {a:.3f} ms; knn + csr graph alone {b:.3f} ms; both on two streams {c:.3f} ms per pair "
      f"(with an event edge per step {d:.3f}); serial sum {a + b:.3f}; hidden {a + b - c:.3f} ms of {b:.3f}")
