#!/usr/bin/env python3
"""How large are the gradients whose TRUE value is exactly 0 (conv / Linear biases in front of a BatchNorm) on the HIP path,
against Adam's eps = 1e-8?  One training step of the G6 configuration (tests/golden/train_run.npz clouds), both arithmetic modes."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "3d_recognizer_amd"))
from randlanet import _ops as ops
from randlanet.utils.losses import get_loss
from randlanet.utils.modules import RandLANet, RandLANetSettings

z = np.load(os.path.join(REPO, "tests", "golden", "train_run.npz"))
dev = torch.device("cuda")
for mode in ("bf16x3", "fp32"):
    ops.set_wide_gemm(mode)
    torch.manual_seed(0)
    net = RandLANet(RandLANetSettings(n_classes=3, n_points=1024, n_neighbors=16, layer_sizes=[8, 16, 32, 32]), dev)
    net.fc_end[2].p = 0.0
    net.train()
    rs = np.random.RandomState(0)
    sel = [rs.choice(3000, 1024, replace=False) for _ in range(4)]
    x = torch.from_numpy(np.stack([z["clouds"][i][s] for i, s in enumerate(sel)]).astype(np.float32)).to(dev)
    y = torch.from_numpy(np.stack([z["labels"][i][s] for i, s in enumerate(sel)]).astype(np.int64)).to(dev)
    np.random.seed(0)
    get_loss("dice")(net(x), y).backward()
    zero, other = [], []
    for name, p in net.named_parameters():
        g = p.grad.abs()
        if (name.endswith("conv.bias") and not name.startswith("fc_end.3")) or name == "fc_start.bias":
            zero.append((name, float(g.max())))
        else:
            other.append(float(g.max()))
    zs = np.array([v for _, v in zero])
    print(f"[{mode}] zero-true-gradient biases ({len(zero)} tensors): max |g| median {np.median(zs):.2e}, largest {zs.max():.2e} "
          f"({zero[int(zs.argmax())][0]}), smallest {zs.min():.2e}; fc_start.bias {dict(zero)['fc_start.bias']:.2e}; "
          f"other parameters: max |g| median {np.median(other):.2e}")
