#!/usr/bin/env python3
"""All eight neighbour searches of one bs = 8 forward (rl_knn_multi) for one value of RL_KNN_OCC (points per grid cell / k):
    for o in 0.25 0.35 0.5 0.65 0.8 1.0; do RL_KNN_OCC=$o python tools/knn_occ_sweep.py; done      (0.5, the default, is the optimum)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "3d_recognizer_amd"))
import torch
from randlanet import _ops as ops
B, N = 8, 40960
torch.manual_seed(0)
xyz = torch.rand(B, N, 3, device="cuda")
tasks, ratio = [], 1
for _ in range(4):
    tasks.append((N // ratio, N // ratio, 16)); ratio *= 4
for _ in range(4):
    tasks.append((N // ratio, 4 * N // ratio, 1)); ratio //= 4
for _ in range(3): ops.knn_multi(xyz, tasks)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.knn_multi(xyz, tasks)
e1.record(); torch.cuda.synchronize()
print(os.environ.get("RL_KNN_OCC", "default"), round(e0.elapsed_time(e1) / 20 * 1e3, 1), "us per knn_multi")
