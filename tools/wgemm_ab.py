#!/usr/bin/env python3
"""A/B of the two operand-staging variants of the wide GEMM (rl_set_wgemm_staging "registers" / "dma") on the wide shapes of
config A at bs=8: results must be bitwise equal (same products, same order); timings are HBM-cold (operands rotate through
more than the 256 MB Infinity Cache).  usage: python tools/wgemm_ab.py [reps]       WAB_CHECK_ONLY=1 skips the timing"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "3d_recognizer_amd"))
import torch
from randlanet import _ops as ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
if os.environ.get("WAB_LIB"):            # a second build of the library (path relative to the repo root)
    import randlanet._hip as H
    H._LIB_PATH = os.path.join(REPO, os.environ["WAB_LIB"])
dev = "cuda"
torch.manual_seed(0)

# (M, K, N, launches per step) - the wgemm rows of profiles/r03_v3_event_breakdown.json
STEP = [(81920, 256, 256, 4), (81920, 128, 128, 2), (20480, 256, 128, 2), (5120, 256, 512, 2), (1280, 512, 512, 2),
        (5120, 512, 256, 2), (20480, 128, 256, 2), (5120, 256, 256, 2), (5120, 256, 128, 2), (5120, 128, 256, 2),
        (20480, 512, 128, 1), (20480, 128, 128, 2), (20480, 64, 128, 2), (81920, 32, 256, 1), (5120, 1024, 256, 1),
        (81920, 32, 128, 1), (20480, 128, 512, 1), (81920, 64, 128, 1), (5120, 256, 1024, 1)]


def run(how, fn):
    ops.set_wgemm_staging(how)
    out = fn()
    torch.cuda.synchronize()
    return out


def check():
    """Every epilogue / operand variant, odd sizes included, both stagings: bitwise equal."""
    bad = 0
    cases = []
    for (M, K, N) in [(1000, 64, 128), (5120, 256, 256), (4099, 128, 192), (640, 512, 512), (2560, 1024, 256), (130, 32, 128),
                      (20480, 128, 128)]:
        for variant in ("plain", "lazy_relu", "lazy_leaky", "stats", "bias_stats", "accumulate", "split", "addend", "dgrad",
                        "batched_y"):
            cases.append((M, K, N, variant))
    for (M, K, N, variant) in cases:
        A = torch.randn(M, K, device=dev)
        W = torch.randn(N, K, device=dev) / K ** 0.5
        w_ks, w_ns = 1, K
        if variant == "dgrad":                       # the weight read n-contiguous (input-gradient orientation)
            Wt = torch.randn(K, N, device=dev) / K ** 0.5
            W, w_ks, w_ns = Wt, N, 1
        ws = ops.split_weights([(W, w_ks, w_ns, K, N)])
        a = ops.plain(A, 1, M)
        kw = {}
        if variant.startswith("lazy"):
            a.scale, a.shift = torch.rand(K, device=dev) + .5, torch.randn(K, device=dev)
            a.act, a.slope = (1, 0.0) if variant == "lazy_relu" else (2, 0.2)
        bias = torch.randn(N, device=dev) if variant == "bias_stats" else None
        outs = []
        for how in ("registers", "dma"):
            def fn():
                Y = torch.full((M, N), 0.5, device=dev)
                st = ops.new_stats(dev, N) if "stats" in variant else None
                extra = {}
                res = [Y]
                if variant == "accumulate":
                    extra["accumulate"] = True
                if variant in ("split", "addend"):
                    extra["addend"] = torch.arange(M * N, device=dev, dtype=torch.float32).reshape(M, N) * 1e-3
                if variant == "split":
                    h = N // 2
                    Y = torch.zeros(M, h, device=dev)
                    o2 = torch.zeros(M, N - h, device=dev)
                    extra.update(out2=o2, split_col=h)
                    res = [Y, o2]
                if variant == "batched_y" and M % 4 == 0:
                    ab = ops.plain(A, 4, M // 4)
                    ab.scale, ab.shift, ab.act, ab.slope = a.scale, a.shift, a.act, a.slope
                    Yb = torch.zeros(4 * (M // 4 + 7), N, device=dev)
                    ops.gemm(ab, W, w_ks, w_ns, N, bias, out=Yb, out_bstride=M // 4 + 7, stats=st, wsplit=ws)
                    return [Yb]
                ops.gemm(a, W, w_ks, w_ns, N, bias, out=Y, out_bstride=M, stats=st, wsplit=ws, **extra)
                if st is not None:
                    res.append(st[:(M + 127) // 128].clone())      # the slots this launch owns
                return res
            outs.append(run(how, fn))
        def eq(x, y):
            if x.dtype == torch.float64:        # partial statistics: the persistent kernel groups other tiles per slot
                return torch.allclose(x.sum(0), y.sum(0), rtol=1e-12, atol=0.0)
            return torch.equal(x, y)
        same = all(eq(x, y) for x, y in zip(outs[0], outs[1]))
        ref = (torch.relu(A * a.scale + a.shift) if variant == "lazy_relu" else A).double() @ (W.double().t() if w_ks == 1 else W.double())
        err = float((outs[1][0][:, :8].double() - ref[:, :8]).abs().max()) if variant in ("plain", "lazy_relu", "dgrad") else 0.0
        if not same or err > 1e-3:
            bad += 1
        if not same or err > 1e-3 or not os.environ.get("WAB_REPEAT"):
            print(f"M={M:6d} K={K:4d} N={N:4d} {variant:11s} {'bitwise equal' if same else 'DIFFERENT'}" + (f"  |err vs fp64| {err:.1e}" if err else ""), flush=True)
    return bad


def bench():
    tot = {"registers": 0.0, "dma": 0.0}
    for (M, K, N, launches) in STEP:
        per_set = 4 * M * (K + N)
        nset = max(2, min(12, (600 << 20) // per_set + 1))
        sets = []
        for _ in range(nset):
            A = torch.randn(M, K, device=dev)
            a = ops.plain(A, 1, M)
            a.scale, a.shift, a.act, a.slope = torch.rand(K, device=dev) + .5, torch.randn(K, device=dev), 1, 0.0
            sets.append((a, torch.empty(M, N, device=dev)))
        W = torch.randn(N, K, device=dev) / K ** 0.5
        ws = ops.split_weights([(W, 1, K, K, N)])
        stats = None if os.environ.get("WAB_NOSTATS") else ops.new_stats(dev, N)
        split = bool(os.environ.get("WAB_SPLIT"))
        if split:
            addend, out2 = torch.randn(M, N, device=dev), torch.empty(M, N - N // 2, device=dev)
            sets = [(a, torch.empty(M, N // 2, device=dev)) for a, _ in sets]
        line = []
        for how in ("registers", "dma", "registers", "dma"):
            ops.set_wgemm_staging(how)
            def fn(i):
                a, Y = sets[i % nset]
                if split:       # the input-gradient GEMM of a concat: addend + the two halves to two tensors
                    ops.gemm(a, W, 1, K, N, None, out=Y[:, :N // 2], out_bstride=M, wsplit=ws, addend=addend, out2=out2, split_col=N // 2)
                else:
                    ops.gemm(a, W, 1, K, N, None, out=Y, out_bstride=M, stats=stats, wsplit=ws)
            for i in range(3):
                fn(i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(reps):
                fn(i)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / reps * 1e3
            line.append(us)
            tot[how] += 0.5 * us * launches
        print(f"M={M:6d} K={K:4d} N={N:4d} x{launches} | registers {line[0]:6.1f} {line[2]:6.1f} us | dma {line[1]:6.1f} {line[3]:6.1f} us | "
              f"{4.0 * M * (K + N) / min(line[1], line[3]) / 1e3:5.0f} GB/s", flush=True)
    print(f"per step: registers {tot['registers'] / 1e3:.3f} ms, dma {tot['dma'] / 1e3:.3f} ms")


if os.environ.get("WAB_SHAPES"):
    STEP = [tuple(int(v) for v in t.split("x")) + (1,) for t in os.environ["WAB_SHAPES"].split(",")]
if not os.environ.get("WAB_BENCH_ONLY"):
    # WAB_REPEAT=n: the race screen - the same 70 cases n times over with fresh random operands (a misplaced wait shows as a
    # rare wrong tile, not as a steady failure)
    for rep in range(int(os.environ.get("WAB_REPEAT", "1"))):
        bad = check()
        print(f"check (pass {rep}):", "ok" if not bad else f"{bad} FAILED", flush=True)
        if bad:
            sys.exit(1)
if not os.environ.get("WAB_CHECK_ONLY"):
    bench()
