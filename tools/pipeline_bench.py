#!/usr/bin/env python3
"""Input-pipeline throughput: clouds/s of one augmented, shuffled epoch stream at config A's shape
(40960 of ~120k points per cloud, batch 4) for
  host    : the reference-style numpy pipeline + DataLoader (randlanet.utils.dataset.get_data_loader), incl. the H2D copy
  device  : DeviceDataLoader, random numbers from numpy in the reference's order
  device* : DeviceDataLoader, sample indices / jitter noise drawn on the GPU
usage: python tools/pipeline_bench.py [epochs]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "3d_recognizer_amd"))
import numpy as np, torch
from randlanet.utils.augmentation import AugmentationSettings
from randlanet.utils.dataset import get_data_loader
from randlanet.utils.device_dataset import get_device_data_loader

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
rs = np.random.RandomState(0)
ds = [(rs.rand(120000, 3).astype(np.float32), np.zeros((120000, 0), np.float32), rs.randint(0, 2, 120000).astype(np.int64))
      for _ in range(16)]
aug = AugmentationSettings()
dev = torch.device("cuda")

def run(make, name):
    loader = make()
    for inp, lab, _ in loader:      # warm-up epoch
        inp = inp.to(dev); lab = lab.to(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0
    for _ in range(epochs):
        for inp, lab, _ in loader:
            inp = inp.to(dev); lab = lab.to(dev); n += inp.shape[0]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name:8s} {n / dt:8.1f} clouds/s  ({dt / n * 1e3:.2f} ms/cloud)", flush=True)

run(lambda: get_data_loader(ds, 40960, 4, shuffle=True, consistent_sampling=False, augmentation_settings=aug), "host")
run(lambda: get_device_data_loader(ds, 40960, 4, shuffle=True, consistent_sampling=False, augmentation_settings=aug, device=dev), "device")
run(lambda: get_device_data_loader(ds, 40960, 4, shuffle=True, consistent_sampling=False, augmentation_settings=aug, device=dev, rng="device"), "device*")
