#!/usr/bin/env python3
"""BatchNorm-backward sweeps on their own at the benchmark's shapes, HBM-cold: the operand pairs rotate through more bytes than
the 256 MB memory-side cache holds, so a launch finds neither G nor Y on chip (inside a step G was just written, Y is cold).
    python tools/bn_bwd_bench.py            reduce / finalize / apply, and the residual-junction pair, us per launch and GB/s"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "3d_recognizer_amd"))
from randlanet import _hip as H  # noqa: E402
from randlanet import _ops as ops  # noqa: E402

DEV = torch.device("cuda")
SHAPES = [(327680, 64), (327680, 32), (327680, 8), (81920, 128), (81920, 32), (20480, 256), (20480, 64), (5120, 512), (5120, 128), (1280, 512)]


def timed(fns, reps):
    """fns: one closure per operand set; they are called round-robin."""
    for f in fns:
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fns[i % len(fns)]()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    for rows, C in SHAPES:
        nbytes = rows * C * 4
        sets = max(2, min(16, int(600e6 // (2 * nbytes)) + 1))
        red, app, both, rred, rapp = [], [], [], [], []
        for s in range(sets):
            torch.manual_seed(s)
            Y = torch.randn(rows, C, device=DEV)
            G = torch.randn(rows, C, device=DEV)
            r = lambda: torch.rand(C, device=DEV) + 0.5
            y = ops.Lazy(Y, 1, rows, rows, C, r(), r() - 1.0, H.ACT_LRELU, 0.2, r() - 1.0, r(), "x")
            d = ops._bn_bwd_desc(G, rows, y)
            stats = ops.new_stats(DEV, C)
            coef = torch.rand(2 * C, device=DEV) * 1e-3
            d.stats = stats.data_ptr()
            d2 = ops._bn_bwd_desc(G, rows, y)
            d2.coef = coef.data_ptr()
            keep = (Y, G, y, stats, coef)
            red.append(lambda d=d, k=keep: H.check(H.lib().rl_bn_bwd_reduce(H.C.byref(d), ops._st()), "reduce"))
            app.append(lambda d=d2, k=keep: H.check(H.lib().rl_bn_bwd_apply(H.C.byref(d), ops._st()), "apply"))
            dg, db = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
            both.append(lambda G=G, y=y, dg=dg, db=db: ops.bn_backward(G, y, dg, db, True))
        out = [f"reduce {timed(red, 40):6.1f} us ({2 * nbytes / timed(red, 40) / 1e3:5.0f} GB/s)",
               f"apply {timed(app, 40):6.1f} us ({3 * nbytes / timed(app, 40) / 1e3:5.0f} GB/s)",
               f"reduce+finalize+apply {timed(both, 40):6.1f} us"]
        if ops.resid_bn_supported is not None and C >= 32:
            fns = []
            for s in range(sets):
                Y1, Y2 = torch.randn(rows, C, device=DEV), torch.randn(rows, C, device=DEV)
                G, O = torch.randn(rows, C, device=DEV), torch.randn(rows, C, device=DEV)
                r = lambda: torch.rand(C, device=DEV) + 0.5
                y1 = ops.Lazy(Y1, 1, rows, rows, C, r(), r(), H.ACT_NONE, 0.0, r(), r(), "a")
                y2 = ops.Lazy(Y2, 1, rows, rows, C, r(), r(), H.ACT_NONE, 0.0, r(), r(), "b")
                g4 = [torch.empty(C, device=DEV) for _ in range(4)]
                fns.append(lambda G=G, O=O, y1=y1, y2=y2, g4=g4: ops.resid_bn_backward(G, O, 0.01, y1, y2, *g4))
            out.append(f"residual pair {timed(fns[:max(2, sets // 2)], 20):6.1f} us")
        print(f"{rows:7d} x {C:3d} ({sets} operand sets): " + "   ".join(out), flush=True)


main()
