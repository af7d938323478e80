#!/usr/bin/env python3
"""Per-kernel HIP-event breakdown of one training step of BASELINE.json's other single-GPU configurations (bench.py's
instrumented eager pass):  python tools/config_breakdown.py S | Kt"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "3d_recognizer_amd"))
import torch
import bench
from randlanet._train import TrainStep
cfgname = sys.argv[1] if len(sys.argv) > 1 else 'S'
cfg = bench.CFG_S if cfgname == 'S' else bench.CFG_KT
dev = torch.device('cuda')
m = bench.build_model(dev, 0, cfg); m.train()
B, N = cfg['per_gpu_batch'], cfg['n_points']
st = TrainStep(m, B, N, loss='dice', use_graph=True)
x, y = bench.synthetic_batch(B, N, cfg['n_classes'], 1)
st.set_batch(torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev))
st.capture()
roof, bd = bench.roofline_pass(st)
print(cfgname, 'step kernel ms', bd['step_kernel_ms'])
for k, v in list(bd['kernels'].items())[:22]:
    print(f"{k:30s} {v['ms_per_step']:7.3f} ms {v['launches_per_step']:5.1f} {v['avg_launch_us']:8.1f} us {v['GBps']:8.1f} GB/s {v['TFLOPs']:7.1f} TF")
for r in bd['top_shapes'][:16]:
    print(f"   {r['kernel']:24s} {r['op']:40s} {r['ms_per_step']:.3f} x{r['launches_per_step']:.0f} {r['GBps']:8.1f} GB/s {r['TFLOPs']:6.1f}")
