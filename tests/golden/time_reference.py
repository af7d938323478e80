#!/usr/bin/env python3
"""how faithful is bench.py's CPU baseline ("kind": "port") to the real thing?

Times the synthetic config-A training step (forward + dice + backward + Adam) on this container's cores
  (a) with the REFERENCE's own randlanet.utils.modules.RandLANet (imported from /root/reference with the inert faiss /
      tensorboard placeholders of make_golden.py and its C++ KNN, oracle/_ref/knn_tpk.so, behind knn_approximate), and
  (b) with the CPU restatement bench.py times on the GPU node (oracle/randlanet_oracle.py + oracle/knn_oracle.c),
same batch, same thread count, and prints both and their ratio.  Runs only where /root/reference exists (never on the
GPU box); results are recorded in BASELINE.md / DESIGN.md.

usage: python tests/golden/time_reference.py [B] [timed_steps]"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
import numpy as np
import torch

import make_golden as G          # imports the reference (placeholders + knn_tpk) exactly like the fixture generator
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cores = len(os.sched_getaffinity(0))
torch.set_num_threads(cores)
CFG = bench.CFG
N, C, K, layers = CFG["n_points"], CFG["n_classes"], CFG["n_neighbors"], list(CFG["layer_sizes"])
xyz, labels = bench.synthetic_batch(B, N, C, 1234)
x, y = torch.from_numpy(xyz), torch.from_numpy(labels)


def reference_step():
    from randlanet.utils.losses import FocalTverskyLoss
    torch.manual_seed(0)
    net = G.build_ref_net(C, N, K, layers)
    net.train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    crit = FocalTverskyLoss(alpha=0.5, gamma=1.0)          # "dice" (trainer.py:244-269)
    times = []
    for it in range(STEPS + 1):
        t0 = time.perf_counter()
        logits = net(x)
        loss = crit(logits, y)
        opt.zero_grad()
        loss.backward()
        opt.step()
        times.append(time.perf_counter() - t0)
    return float(np.mean(times[1:])), float(loss)


def port_step():
    from oracle import randlanet_oracle as O
    from oracle.loss_metrics_oracle import loss_by_name
    net = bench.build_model(torch.device("cpu"))
    P = {k: v.detach().clone() for k, v in net.state_dict().items()}
    params = [v.requires_grad_(True) for k, v in P.items() if v.is_floating_point() and "running" not in k]
    opt = torch.optim.Adam(params, lr=1e-2)
    times = []
    for it in range(STEPS + 1):
        t0 = time.perf_counter()
        perm = np.random.permutation(N)
        buffers = {}
        logits = O.forward(P, x, perm, layer_sizes=layers, n_neighbors=K, training=True, buffers=buffers)
        loss = loss_by_name("dice", logits, y)
        opt.zero_grad()
        loss.backward()
        opt.step()
        for k, v in buffers.items():
            P[k] = v
        times.append(time.perf_counter() - t0)
    return float(np.mean(times[1:])), float(loss)


if __name__ == "__main__":
    np.random.seed(0)
    tr, lr = reference_step()
    np.random.seed(0)
    tp, lp = port_step()
    print(f"cores {cores}  B {B}  N {N}  steps {STEPS}")
    print(f"reference  {tr:7.3f} s/step  {B / tr:6.3f} clouds/s  (loss {lr:.4f})")
    print(f"port       {tp:7.3f} s/step  {B / tp:6.3f} clouds/s  (loss {lp:.4f})")
    print(f"port / reference time = {tp / tr:.3f}")
