#!/usr/bin/env python3
"""A/B of the wide-GEMM epilogue variants: time rl_gemm (pre-split weights -> wgemm_kernel) with and without bias /
accumulate on two wide shapes.  usage: python tools/epilogue_bench.py [reps]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "3d_recognizer_amd"))
import torch
import randlanet._hip as H
if os.environ.get("EB_LIB"):            # A/B: a second build of the library (path relative to the repo root)
    H._LIB_PATH = os.path.join(REPO, os.environ["EB_LIB"])
from randlanet import _ops as ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = "cuda"
for (M, K, N) in [(81920, 256, 256), (20480, 128, 256), (81920, 128, 128), (327680, 64, 128)]:
    nset = 4
    sets = []
    for _ in range(nset):
        A = torch.randn(M, K, device=dev)
        Y = torch.zeros(M, N, device=dev)
        a = ops.plain(A, 1, M)
        sets.append((a, Y))
    W = torch.randn(N, K, device=dev) / K ** 0.5
    bias = torch.randn(N, device=dev)
    ws = ops.split_weights([(W, 1, K, K, N)])
    stats = ops.new_stats(dev, N)
    line = []
    for name, kw in (("plain", {}), ("bias", {"bias": bias}), ("bias+stats", {"bias": bias, "stats": stats}),
                     ("accumulate", {"accumulate": True})):
        b = kw.pop("bias", None)
        def fn(i):
            a, Y = sets[i % nset]
            ops.gemm(a, W, 1, K, N, b, out=Y, out_bstride=M, wsplit=ws, **kw)
        for i in range(3):
            fn(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        line.append(f"{name} {e0.elapsed_time(e1) / reps * 1e3:7.1f} us")
    print(f"M={M:7d} K={K:4d} N={N:4d} | " + " | ".join(line), flush=True)
