#!/usr/bin/env python3
"""Latency of Model.predict (the number the reference UI's 250 ms timer sees): train.py's settings
(2500 points, K = 32, 4 layers), one raw cloud of ~150k points, seed-0 down-sample -> forward -> full-resolution
up-sampling -> confidences.  usage: python tools/predict_bench.py [reps]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "3d_recognizer_amd"))
import numpy as np, torch
from randlanet import Model, RandLANetSettings

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
# a one-GPU box shows 256 cores but grants 16: torch's CPU thread pool must not be sized for 256 (the few host-side
# tensor ops of predict - fancy indexing, .cpu() - then cost 25 ms instead of 0.2 ms)
torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
rs = np.random.RandomState(0)
cloud = rs.rand(150000, 3).astype(np.float32)
for up in ("nni", "idw"):
    model = Model(RandLANetSettings(n_classes=2, n_points=2500, n_neighbors=32, upsampling=up))
    for _ in range(3):
        conf = model.predict(cloud)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        conf = model.predict(cloud)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    assert conf.shape == (2, 150000) and np.isfinite(conf).all()
    print(f"predict ({up}): {dt * 1e3:7.2f} ms per 150k-point cloud", flush=True)
