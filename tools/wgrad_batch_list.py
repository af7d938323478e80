#!/usr/bin/env python3
"""The layers of one backward pass's grouped weight-gradient launches (rl_wgrad_batch) at the benchmark's configuration:
rows, K (input channels), N (output channels), kind (1: 128 x 128-tile kernel, 2: streaming kernel), operand megabytes.
    python tools/wgrad_batch_list.py"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "3d_recognizer_amd"))
from randlanet import _ops as ops  # noqa: E402
from randlanet._train import TrainStep  # noqa: E402

orig = ops.wgrad_batch_flush


def logged(batch):
    tot = {1: 0.0, 2: 0.0}
    for d, nbytes, flops, a, dY, slab, kind in batch:
        K = a.C if hasattr(a, "C") else 10
        rows = a.rows if hasattr(a, "rows") else a.B * a.n * a.K
        gy, gz = -(-d.N // 128), -(-K // 128)
        mb = rows * (d.N * (gz if kind == 1 else 1) + K * (gy if kind == 1 else 1)) * dY.element_size() / 1e6
        tot[kind] += mb
        print(f"kind {kind}: rows {rows:7d}  K {K:4d}  N {d.N:4d}  tiles {gy}x{gz}  operand reads {mb:7.1f} MB")
    print(f"wide: {tot[1]:.0f} MB, streaming: {tot[2]:.0f} MB")
    orig(batch)


ops.wgrad_batch_flush = logged
sys.path.insert(0, REPO)
import bench  # noqa: E402

dev = torch.device("cuda")
B, N, C = 8, bench.CFG["n_points"], bench.CFG["n_classes"]
model = bench.build_model(dev, seed=0)
model.train()
step = TrainStep(model, B, N, loss="dice", lr=1e-2, use_graph=False)
xyz, labels = bench.synthetic_batch(B, N, C, 1234)
step.set_batch(torch.from_numpy(xyz).to(dev), torch.from_numpy(labels).to(dev))
step.step(np.random.default_rng(0).permutation(N))
torch.cuda.synchronize()
