#!/usr/bin/env python3
"""A/B of the wide GEMM's output tile (rl_set_wgemm_tile "128" / "auto", round 6) on the wide shapes of a step of config A at a
given batch size, each timed as a replayed hipGraph of `reps` dependent launches over rotating operand sets (HBM-cold-ish,
no host gaps): forward form (lazy operand + statistics) and input-gradient form (plain operand, accumulate, no statistics).
usage: python tools/wgemm_tile_bench.py [reps] [batch]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "3d_recognizer_amd"))
import torch
from randlanet import _ops as ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = "cuda"
torch.manual_seed(0)
# (points per cloud, K, N, launches per step): the wide products of config A ([16, 64, 128, 256], N = 40960)
STEP = [(10240, 256, 256, 4), (10240, 128, 128, 2), (2560, 256, 128, 2), (640, 256, 512, 2), (160, 512, 512, 2),
        (640, 512, 256, 2), (2560, 128, 256, 2), (640, 256, 256, 2), (640, 256, 128, 2), (640, 128, 256, 2),
        (2560, 512, 128, 1), (2560, 128, 128, 2), (2560, 64, 128, 2), (10240, 32, 256, 1), (640, 1024, 256, 1),
        (10240, 32, 128, 1), (2560, 128, 512, 1), (10240, 64, 128, 1), (640, 256, 1024, 1)]


def timed(fn):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps):
            fn(i)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


TILES = ("128", "auto", "auto/nosplit")
tot = {}
for (n, K, N, launches) in STEP:
    M = n * bs
    per_set = 4 * M * (K + N)
    nset = max(2, min(12, (600 << 20) // per_set + 1))
    sets = []
    for _ in range(nset):
        A = torch.randn(M, K, device=dev)
        a = ops.plain(A, 1, M)
        a.scale, a.shift, a.act, a.slope = torch.rand(K, device=dev) + .5, torch.randn(K, device=dev), 1, 0.0
        sets.append((a, ops.plain(A, 1, M), torch.zeros(M, N, device=dev)))
    W = torch.randn(N, K, device=dev) / K ** 0.5
    ws = ops.split_weights([(W, 1, K, K, N)])
    stats = ops.new_stats(dev, N)
    line = []
    for form in ("fwd", "dgrad"):
        for tile in TILES:
            ops.set_wgemm_tile(tile.split("/")[0])
            ops.set_gemm_ksplit(not tile.endswith("nosplit"))

            def fn(i):
                a, ap, Y = sets[i % nset]
                if form == "fwd":
                    ops.gemm(a, W, 1, K, N, None, out=Y, out_bstride=M, stats=stats, wsplit=ws)
                else:
                    ops.gemm(ap, W, 1, K, N, None, out=Y, out_bstride=M, accumulate=True, wsplit=ws)
            us = timed(fn)
            line.append(us)
            tot[(form, tile)] = tot.get((form, tile), 0.0) + us * launches / 2
    ops.set_wgemm_tile("auto")
    ops.set_gemm_ksplit(True)
    nt = len(TILES)
    print(f"M={M:6d} K={K:4d} N={N:4d} x{launches} | fwd+stats " + " ".join(f"{t}: {line[i]:5.1f}" for i, t in enumerate(TILES)) +
          " us | dgrad+acc " + " ".join(f"{t}: {line[nt + i]:5.1f}" for i, t in enumerate(TILES)) + " us", flush=True)
print("per step (half the launches in each form): " + ", ".join(f"{k[0]}/{k[1]} {v / 1e3:.3f} ms" for k, v in tot.items()))
