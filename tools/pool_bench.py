#!/usr/bin/env python3
"""The tile kernels of one encoder level on their own: fused pooling forward / backward (virtual rpe branch), the rpe
weight-gradient kernel and the gather backward, at the benchmark's shapes (B = 8, N = 40960), random operands.
    python tools/pool_bench.py [level ...]        RL_HIP_LIB=<other build> to time a kernel variant"""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "3d_recognizer_amd"))
from randlanet import _hip as H  # noqa: E402
from randlanet import _ops as ops  # noqa: E402

DEV = torch.device("cuda")


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def level(l, B=8, N=40960):
    d = [16, 64, 128, 256][l]
    n, h = N // 4 ** l, d // 2
    torch.manual_seed(l)
    xyz = torch.rand((B, n, 4), device=DEV)
    # RL_BENCH_LOCAL=w: neighbours within +-w of the point's own index (gathers that stay in L1 / L2) instead of uniformly random
    # ones (what the network sees: its points are randomly permuted) - the kernels' time without their gather misses
    w = int(os.environ.get("RL_BENCH_LOCAL", "0"))
    if w:
        idx = ((torch.arange(n, device=DEV).view(1, n, 1) + torch.randint(-w, w + 1, (B, n, 16), device=DEV)) % n).to(torch.int32)
    else:
        idx = torch.randint(0, n, (B, n, 16), dtype=torch.int32, device=DEV)
    d2 = torch.rand((B, n, 16), device=DEV) * 0.01
    r = lambda *s: torch.randn(*s, device=DEV) * 0.3
    vr = ops.VirtualRpe(xyz, idx, d2, B, n, h, r(h, 10), r(h), r(h, h), r(h))
    bn = lambda: ops.Lazy(d2, B, n * 16, n * 16, h, torch.rand(h, device=DEV) + 0.5, r(h), H.ACT_RELU, 0.0, r(h), torch.rand(h, device=DEV) + 0.5, "x")
    vr.bn1, vr.bn2 = bn(), bn()
    g = ops.plain(r(B * n, h), B, n)
    W = r(d, d)
    dP = r(B * n, d)
    dW = torch.zeros(d, d, device=DEV)
    rdt = ops.row_dtype()
    GU = torch.zeros((B * n * 16, h), dtype=rdt, device=DEV)
    csr = ops.csr_build([(idx, n)])[0]
    gg = torch.zeros((B * n, h), device=DEV)
    out = {}
    out["pool_fwd s1"] = timeit(lambda: ops.pool_fwd(vr, g, idx, W, n, d, 1))
    out["pool_fwd s2"] = timeit(lambda: ops.pool_fwd(vr, g, idx, W, n, d, 2))
    nslots = H.lib().rl_pool_bwd_slots(B * n, d)
    bst = torch.empty((nslots, 2, h), dtype=torch.float64, device=DEV)
    DG = [None]

    def bwd(stage, acc):
        DG[0] = ops.pool_bwd(vr, g, idx, W, n, d, dP, GU, acc, dW, pending=None, stage=stage, bn_bwd_stats=bst)
    out["pool_bwd s2 (store)"] = timeit(lambda: bwd(2, False))
    out["pool_bwd s1 (accumulate)"] = timeit(lambda: bwd(1, True))
    coef = r(2 * h)
    GU1 = torch.empty_like(GU)
    pend = []
    out["rpe_wgrad s2"] = timeit(lambda: (pend.clear(), ops.rpe_wgrad(vr, 2, GU, coef, r(h, h), r(h), pend, GU1)))
    out["rpe_wgrad s1"] = timeit(lambda: (pend.clear(), ops.rpe_wgrad(vr, 1, GU, coef, r(h, 10), r(h), pend, None)))
    out["segment_sum"] = timeit(lambda: ops.segment_sum_rows(DG[0], (0, h), n * 16, csr, gg, n))
    out["rpe_stats s1"] = timeit(lambda: ops.rpe_stats(vr, 1))
    print(f"level {l} (d = {d}, {B * n} points, storage {ops.get_storage()}): " + "  ".join(f"{k} {v:.0f} us" for k, v in out.items()), flush=True)


# RL_BENCH_POINTS=p: p points per launch instead of the benchmark's (e.g. 5120 = one point per resident wavefront: what a launch
# costs before its first point - the prologue that stages W and loads the per-lane constants)
for l in ([int(a) for a in sys.argv[1:]] or [0, 1]):
    pts = int(os.environ.get("RL_BENCH_POINTS", "0"))
    if pts:
        level(l, 8, (pts // 8) * 4 ** l)
    else:
        level(l)
