#!/usr/bin/env python3
"""Run a few pipelined steps (to be traced by rocprofv3 --kernel-trace):  rocprofv3 --kernel-trace -d DIR -- python3 tools/pipeline_trace.py"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "3d_recognizer_amd"))
import numpy as np
import torch
import bench
from randlanet._train import TrainStep

B, N = 8, 40960
dev = torch.device("cuda")
x, y = bench.synthetic_batch(B, N, 2, 1)
m = bench.build_model(dev, 0)
m.train()
st = TrainStep(m, B, N, loss="dice", pipeline=True)
st.set_batch(torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev))
st.capture()
for i in range(30):
    st.step(np.random.permutation(N))
torch.cuda.synchronize()
