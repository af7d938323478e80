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

# Do G and Y at the SAME offset from equally aligned bases collide in the memory system (same channel / bank, different DRAM
# rows)?  Y carved out of one big buffer at base + shift: time of the cold reduce sweep per shift.
print("stagger test: bn_bwd_reduce cold, Y at a 2 MB-aligned base + shift")
per = (M * C * 4 + (4 << 20)) // 4          # floats per slot: the tensor + 4 MB of room for the shift
bigG = torch.randn(NB * per, device=dev)
bigY = torch.randn(NB * per, device=dev)
assert bigG.data_ptr() % (2 << 20) == 0 and bigY.data_ptr() % (2 << 20) == 0, (bigG.data_ptr(), bigY.data_ptr())
for shift in (0, 256, 1024, 4096, 8192, 16384, 65536, 262144, 1 << 20, (1 << 20) + 4096):
    G2 = [bigG[i * per: i * per + M * C].view(M, C) for i in range(NB)]
    Y2 = [bigY[i * per + shift // 4: i * per + shift // 4 + M * C].view(M, C) for i in range(NB)]
    lz = []
    for i in range(NB):
        y = ops.plain(Y2[i], 1, M)
        y.scale, y.shift = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        y.mean, y.invstd = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        y.act, y.slope = 2, 0.2
        lz.append(y)
    ops.TIMER = ops.KernelTimer()
    for i in range(24):
        ops.bn_backward(G2[i % NB], lz[i % NB], dg, db, True)
    torch.cuda.synchronize()
    acc = {}
    for cat, key, nbytes, flops, e0, e1, kern, lvl in ops.TIMER.records:
        acc.setdefault(cat, []).append(e0.elapsed_time(e1) * 1e3)
    ops.TIMER = None
    print(f"shift {shift:8d} B: " + "  ".join(f"{k} {sorted(v)[len(v) // 2]:6.1f} us" for k, v in acc.items()))
