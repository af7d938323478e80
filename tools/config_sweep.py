#!/usr/bin/env python3
"""Functional + timing check of the fused training step on BASELINE.json's other single-GPU shapes (not bench lines):
  S   65536 pts, 13 classes, 5 encoder layers [16,64,128,256,512], batch 8
  Kt  122880 pts, 20 classes, 4 encoder layers, batch 2 (the per-GPU share of bs 16 over 8 GPUs)
  A8  config A at batch 8 (the `metric` field's bs=8 on one GPU)
usage: python tools/config_sweep.py [steps]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "3d_recognizer_amd"))
import numpy as np, torch
import bench
from randlanet._train import TrainStep
from randlanet.utils.modules import RandLANet, RandLANetSettings

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda")
torch.set_num_threads(bench.host_cores())
for tag, N, C, layers, B in (("A8", 40960, 2, [16, 64, 128, 256], 8), ("S", 65536, 13, [16, 64, 128, 256, 512], 8),
                             ("Kt", 122880, 20, [16, 64, 128, 256], 2)):
    torch.manual_seed(0)
    s = RandLANetSettings(n_classes=C, n_points=N, n_neighbors=16, layer_sizes=layers, knn="naive")
    model = RandLANet(s, dev); model.train()
    st = TrainStep(model, B, N, loss="dice", lr=1e-2, use_graph=True)
    xyz, labels = bench.synthetic_batch(B, N, C, 1234)
    st.set_batch(torch.from_numpy(xyz).to(dev), torch.from_numpy(labels).to(dev))
    np.random.seed(1)
    st.capture()
    for _ in range(3):
        st.step(np.random.permutation(N))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        st.step(np.random.permutation(N))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    m = st.last_metrics()
    assert np.isfinite(m["loss"]), tag
    print(f"{tag:3s} B={B} N={N} C={C} L={len(layers)}: {dt*1e3:8.2f} ms/step {B/dt:8.1f} clouds/s  loss {m['loss']:.4f} mIoU {m['mIoU']:.3f} "
          f"HBM {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
    del st, model
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
