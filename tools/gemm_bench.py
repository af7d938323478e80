#!/usr/bin/env python3
"""Micro-benchmark of rl_gemm / rl_wgrad on the layer shapes of config A (fp32).
usage: python tools/gemm_bench.py [reps]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "3d_recognizer_amd"))
import torch
from randlanet import _ops as ops

SHAPES = [(163840, 128, 128), (40960, 256, 256), (10240, 128, 256), (2560, 512, 256), (640, 512, 512),
          (2560, 1024, 256), (10240, 256, 128), (655360, 64, 64), (2621440, 16, 16)]
NARROW = [(2621440, 8, 8), (2621440, 16, 16), (655360, 32, 32), (655360, 16, 32), (163840, 64, 64), (163840, 32, 64),
          (163840, 64, 32), (163840, 16, 64), (40960, 64, 64)]
if os.environ.get("GB_NARROW"):
    SHAPES = NARROW
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = "cuda"
for (M, K, N) in SHAPES:
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) / K ** 0.5
    dY = torch.randn(M, N, device=dev)
    dW = torch.empty_like(W)
    a = ops.plain(A, 1, M)
    sc, sh = torch.rand(K, device=dev) + .5, torch.randn(K, device=dev)
    a.scale, a.shift, a.act, a.slope = sc, sh, 1, 0.0
    stats = ops.new_stats(dev, N)
    res = {}
    for name, fn in (("gemm", lambda: ops.gemm(a, W, 1, K, N, None, stats=stats)),
                     ("wgrad", lambda: ops.wgrad(a, dY, M, N, dW, 1, K, None))):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        res[name] = (ms, 2.0 * M * K * N / ms / 1e9, 4.0 * M * (K + N) / ms / 1e6)
    print(f"M={M:8d} K={K:5d} N={N:4d} | gemm {res['gemm'][0]*1e3:8.1f} us {res['gemm'][1]:6.1f} TF/s {res['gemm'][2]:7.0f} GB/s"
          f" | wgrad {res['wgrad'][0]*1e3:8.1f} us {res['wgrad'][1]:6.1f} TF/s")
