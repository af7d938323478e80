#!/usr/bin/env python3
"""Micro-benchmark of rl_knn_i32 on the eight searches of one config-A forward (B=4)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "3d_recognizer_amd"))
import torch
from randlanet import _ops as ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B, N = 4, 40960
xyz = torch.rand(B, N, 3, device="cuda")
cases = [(N, N, 16), (N // 4, N // 4, 16), (N // 16, N // 16, 16), (N // 64, N // 64, 16),
         (N // 256, N // 64, 1), (N // 64, N // 16, 1), (N // 16, N // 4, 1), (N // 4, N, 1)]
tot = 0.0
for Ns, Nq, k in cases:
    for brute in (False, True):
        if brute and Ns > 4096:
            continue
        for _ in range(3):
            ops.knn_i32(xyz, xyz, Ns, Nq, k, brute=brute)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.knn_i32(xyz, xyz, Ns, Nq, k, brute=brute)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        if not brute:
            tot += us
        print(f"Ns={Ns:6d} Nq={Nq:6d} k={k:2d} {'brute' if brute else 'grid '} {us:8.1f} us  {B*Nq/us:8.1f} Mquery/s")
print(f"sum of grid searches {tot:.1f} us")
