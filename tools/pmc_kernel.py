#!/usr/bin/env python3
"""Runs ONE kernel shape of the training step a few times on random data so that
`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes) can read its HBM traffic.
usage: python3 tools/pmc_kernel.py gemm|wgrad|wgradb M K N [reps]      (wgradb: through the grouped launch, rl_wgrad_batch)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "3d_recognizer_amd"))
import torch
from randlanet import _ops as ops
kind, M, K, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
dev = "cuda"
A = torch.randn(M, K, device=dev)
W = torch.randn(N, K, device=dev) / K ** 0.5
a = ops.plain(A, 1, M)
a.scale, a.shift, a.act, a.slope = torch.rand(K, device=dev) + .5, torch.randn(K, device=dev), 1, 0.0
stats = ops.new_stats(dev, N)
dY = torch.randn(M, N, device=dev)
dW = torch.empty_like(W)
big = torch.empty(1 << 28, dtype=torch.float32, device=dev)      # 1 GiB: evicts the 256 MiB Infinity Cache between launches
ws = ops.split_weights([(W, 1, K, K, N)])      # the 8-wavefront kernel, as inside a step (empty in the fp32 mode)
use_stats = len(sys.argv) > 6 and sys.argv[6] == "stats"
for _ in range(reps):
    big.fill_(1.0)
    if kind == "gemm":
        ops.gemm(a, W, 1, K, N, None, stats=stats if use_stats else None, wsplit=ws)
    elif kind == "wgradb":
        pend, batch = [], []
        ops.wgrad(a, dY, M, N, dW, 1, K, None, pending=pend, batch=batch)
        ops.wgrad_batch_flush(batch)
        ops.wgrad_flush(pend)
    else:
        ops.wgrad(a, dY, M, N, dW, 1, K, None)
torch.cuda.synchronize()
print("done")
