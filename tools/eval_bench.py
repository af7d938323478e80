#!/usr/bin/env python3
"""Trainer.evaluate passes/s on the MI355X: the on-device loop (InferStep graph per batch, fused loss + counts, one
read-back per evaluate) against the per-batch eager loop (RL_EVAL_EAGER=1: two read-backs per batch).
    python tools/eval_bench.py [n_clouds] [n_points] [batch]"""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "3d_recognizer_amd"))

from randlanet import Model, RandLANetSettings  # noqa: E402

n_clouds, n_pts, bs = (int(a) for a in (sys.argv[1:4] + ["16", "40960", "8"][len(sys.argv) - 1:]))
rs = np.random.RandomState(0)
data = []
for i in range(n_clouds):
    xyz = rs.uniform(0, 1, (n_pts + 1000, 3)).astype(np.float32)
    data.append((xyz, np.zeros((xyz.shape[0], 0), np.float32), (xyz[:, 2] > 0.5).astype(np.int64)))
torch.manual_seed(0)
model = Model(RandLANetSettings(n_classes=2, n_points=n_pts, n_neighbors=16, layer_sizes=[16, 64, 128, 256], knn="kdtree"))
res = {}
for mode in ("device", "eager", "device"):
    os.environ["RL_EVAL_EAGER"] = "1" if mode == "eager" else "0"
    model.evaluate(data, ["a", "b"], batch_size=bs)                      # warm-up (captures the graphs)
    torch.cuda.synchronize()
    t = time.perf_counter()
    out = model.evaluate(data, ["a", "b"], batch_size=bs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    res[mode] = (10 / dt, 10 * n_clouds / dt, out["mIoU"])
    print(f"{mode:7s}: {10 / dt:7.2f} passes/s  {10 * n_clouds / dt:8.1f} clouds/s  mIoU {out['mIoU']:.6f}", flush=True)
assert res["device"][2] == res["eager"][2], "the two loops must give the same numbers"
