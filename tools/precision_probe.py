#!/usr/bin/env python3
"""Per-kernel arithmetic error of the matrix kernels in each RL_WIDE_GEMM mode, against fp64:
error = max |y - y64| / rms(y64).  bf16x3 should sit near 2^-17 * sqrt-ish growth (~1e-5), bf16 near 2^-9 (~3e-3)."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "3d_recognizer_amd"))
from randlanet import _ops as ops  # noqa: E402

DEV = torch.device("cuda")
torch.manual_seed(0)


def err(y, ref):
    return float((y.double() - ref).abs().max() / ref.pow(2).mean().sqrt())


def probe(M, K, N):
    A = torch.randn(M, K, device=DEV)
    W = torch.randn(N, K, device=DEV) / K ** 0.5          # Conv2d layout (N, K): ks = 1, ns = K
    dY = torch.randn(M, N, device=DEV)
    sc = torch.rand(K, device=DEV) + 0.5
    sh = torch.randn(K, device=DEV) * 0.1
    ref = A.double() @ W.double().t()
    ref_lazy = torch.relu(A.double() * sc.double() + sh.double()) @ W.double().t()
    ref_dA = dY.double() @ W.double()
    ref_dW = dY.double().t() @ A.double()
    a = ops.plain(A, 1, M)
    al = ops.Lazy(A, 1, M, M, K, sc, sh, 1, 0.0)          # ACT_RELU = 1
    g = ops.plain(dY, 1, M)
    out = {}
    for mode in ("fp32", "bf16x3", "bf16"):
        ops.set_wide_gemm(mode)
        ws = ops.split_weights([(W, 1, K, K, N), (W, K, 1, N, K)])
        r = {}
        r["fwd"] = err(ops.gemm(a, W, 1, K, N), ref)
        r["fwd+planes"] = err(ops.gemm(a, W, 1, K, N, wsplit=ws), ref)
        r["fwd lazy+planes"] = err(ops.gemm(al, W, 1, K, N, wsplit=ws), ref_lazy)
        r["dgrad"] = err(ops.gemm(g, W, K, 1, K), ref_dA)
        r["dgrad+planes"] = err(ops.gemm(g, W, K, 1, K, wsplit=ws), ref_dA)
        acc = torch.zeros(M, K, device=DEV)
        ops.gemm(g, W, K, 1, K, out=acc, accumulate=True, wsplit=ws)
        r["dgrad acc+planes"] = err(acc, ref_dA)
        dW = torch.empty(N, K, device=DEV)
        ops.wgrad(a, dY, M, N, dW, 1, K, None)
        r["wgrad"] = err(dW, ref_dW)
        out[mode] = r
    ops.set_wide_gemm("bf16x3")
    print(f"M={M} K={K} N={N}")
    for k in out["fp32"]:
        print(f"   {k:18s} fp32 {out['fp32'][k]:.2e}   bf16x3 {out['bf16x3'][k]:.2e}   bf16 {out['bf16'][k]:.2e}")


for shape in ((5000, 128, 128), (1250, 256, 256), (5000, 256, 128), (5000, 64, 128), (312, 512, 512), (20000, 32, 128), (5000, 128, 64)):
    probe(*shape)
