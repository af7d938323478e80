#!/usr/bin/env python3
"""Micro-benchmark of rl_gemm / rl_wgrad on the layer shapes of config A (fp32).
Operands rotate through enough buffer sets to exceed the 256 MB Infinity Cache, so the numbers are HBM-cold like the
launches inside a training step (a single re-used buffer set reads 1.3-1.5x faster than anything the step sees).
usage: python tools/gemm_bench.py [reps]      GB_NARROW=1 for the streaming shapes, GB_HOT=1 for one buffer set"""
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
if os.environ.get("GB_SHAPES"):
    SHAPES = [tuple(int(v) for v in t.split("x")) for t in os.environ["GB_SHAPES"].split(",")]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = "cuda"
for (M, K, N) in SHAPES:
    per_set = 4 * M * (K + 2 * N)
    nset = 1 if os.environ.get("GB_HOT") else max(2, min(8, (600 << 20) // per_set + 1))
    sets = []
    for _ in range(nset):
        A = torch.randn(M, K, device=dev)
        dY = torch.randn(M, N, device=dev)
        Y = torch.empty(M, N, device=dev)
        a = ops.plain(A, 1, M)
        a.scale, a.shift, a.act, a.slope = torch.rand(K, device=dev) + .5, torch.randn(K, device=dev), 1, 0.0
        sets.append((a, dY, Y))
    W = torch.randn(N, K, device=dev) / K ** 0.5        # (out, in): forward reads it k-contiguous
    Wt = W.t().contiguous()                             # dgrad-style operand: n contiguous
    dW = torch.empty_like(W)
    stats = ops.new_stats(dev, N)
    res = {}
    it = [0]
    def nxt():
        it[0] += 1
        return sets[it[0] % nset]
    def f_gemm():
        a, dY, Y = nxt(); ops.gemm(a, W, 1, K, N, None, out=Y, out_bstride=M, stats=stats)
    def f_dgrad():
        a, dY, Y = nxt(); ops.gemm(a, Wt, N, 1, N, None, out=Y, out_bstride=M, stats=stats)
    def f_wgrad():
        a, dY, Y = nxt(); ops.wgrad(a, dY, M, N, dW, 1, K, None)
    for name, fn in (("gemm", f_gemm), ("dgrad", f_dgrad), ("wgrad", f_wgrad)):
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
    print(f"M={M:8d} K={K:5d} N={N:4d} | " + " | ".join(
        f"{n} {res[n][0]*1e3:7.1f} us {res[n][1]:5.1f} TF/s {res[n][2]:5.0f} GB/s" for n in ("gemm", "dgrad", "wgrad")), flush=True)
