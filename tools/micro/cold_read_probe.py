#!/usr/bin/env python3
"""What a streaming READ reaches on this GPU when its operands come cold from HBM (rotating through more than the 256 MB Infinity
Cache) and when they are resident: torch.sum over 84 / 168 MB, and rl_bn_bwd_reduce on [327680, 64] (G + Y = 168 MB)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "3d_recognizer_amd"))
import torch
from randlanet import _ops as ops

dev = "cuda"
M, C = 327680, 64
NB = 8                                   # 8 x 84 MB per operand: nothing survives a round
Gs = [torch.randn(M, C, device=dev) for _ in range(NB)]
Ys = [torch.randn(M, C, device=dev) for _ in range(NB)]


def timed(fn, reps):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def bn_lazy(i):
    y = ops.plain(Ys[i], 1, M)
    y.scale, y.shift = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    y.mean, y.invstd = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    y.act, y.slope = 2, 0.2
    return y


lazies = [bn_lazy(i) for i in range(NB)]
dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
for name, rot in (("cold", NB), ("resident", 1)):
    t = timed(lambda i: Gs[i % rot].sum(), 40)
    print(f"torch.sum 84 MB {name:9s} {t:7.1f} us  {84e6 * 1.0 / t / 1e6 * 1.0:7.2f} TB/s".replace("TB/s", "GB/ms"))
    t = timed(lambda i: (Gs[i % rot].sum(), Ys[i % rot].sum()), 40)
    print(f"2 x torch.sum   {name:9s} {t:7.1f} us")
    ops.TIMER = ops.KernelTimer()
    for i in range(20):
        ops.bn_backward(Gs[i % rot], lazies[i % rot], dg, db, True)
    torch.cuda.synchronize()
    acc = {}
    for cat, key, nbytes, flops, e0, e1, kern, lvl in ops.TIMER.records:
        acc.setdefault(cat, []).append(e0.elapsed_time(e1) * 1e3)
    ops.TIMER = None
    for k, v in acc.items():
        v = sorted(v)
        print(f"{k:16s} {name:9s} median {v[len(v) // 2]:7.1f} us")
