#!/usr/bin/env python3
"""Runs the largest KNN search of config A (B=4, 40960 points, k=16) a few times for a rocprofv3 --pmc pass."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "3d_recognizer_amd"))
import torch
from randlanet import _ops as ops
B, N = 4, 40960
xyz = torch.rand(B, N, 3, device="cuda")
for _ in range(5):
    ops.knn_i32(xyz, xyz, N, N, 16)
torch.cuda.synchronize()
print("done")
