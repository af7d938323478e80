#!/usr/bin/env python3
"""How much of the step is random-gather latency?  Times the bench step with (a) a random permutation (the real thing) and
(b) a permutation of the SAME kind of random subsets whose bands [N/4^(l+1), N/4^l) are each sorted along a Morton curve,
so that most neighbour gathers land near the gathering row.  (b) is a measurement device, not a mode."""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "3d_recognizer_amd"))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from randlanet import _ops as ops  # noqa: E402
from randlanet._train import TrainStep  # noqa: E402


def morton(xyz, bits=10):
    q = np.clip((xyz * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    code = np.zeros(len(xyz), dtype=np.int64)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return code


def band_sorted(perm, xyz, L=4, dec=4, bits=10):
    N = len(perm)
    out = perm.copy()
    edges = [0] + [N // dec ** l for l in range(L, -1, -1)]
    for a, b in zip(edges[:-1], edges[1:]):
        seg = out[a:b]
        out[a:b] = seg[np.argsort(morton(xyz[seg], bits), kind="stable")]
    return out


dev = torch.device("cuda")
B, N = 8, 40960
model = bench.build_model(dev)
model.train()
st = TrainStep(model, B, N, loss="dice", lr=1e-2, use_graph=True)
# one cloud shape for all batch elements would make the sort exact for all; here every cloud is its own uniform draw, the
# permutation is shared (modules.py:571), so sort along cloud 0 - the others are uncorrelated: use the SAME cloud 8 times
xyz, labels = bench.synthetic_batch(1, N, 2, 1234)
xyz, labels = np.repeat(xyz, B, 0), np.repeat(labels, B, 0)
st.set_batch(torch.from_numpy(xyz).to(dev), torch.from_numpy(labels).to(dev))
st.capture()
rs = np.random.RandomState(0)
BITS = [int(b) for b in os.environ.get("LP_BITS", "10").split(",")]       # Morton bits per axis (3: 512 cells - a coarse, stable bucket order)
variants = [("random", lambda p: p)] + [(f"band-sorted/{b}b", (lambda bb: lambda p: band_sorted(p, xyz[0], bits=bb))(b)) for b in BITS] + [("random", lambda p: p)]
for name, make in variants:
    perms = [make(rs.permutation(N)) for _ in range(8)]
    for i in range(10):
        st.step(perms[i % 8])
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(100):
        st.step(perms[i % 8])
    torch.cuda.synchronize()
    print(f"{name:16s}: {(time.perf_counter() - t) * 10:.3f} ms/step", flush=True)
# measured on one MI355X: round 3 (the kernels of that time) random 7.97 ms/step, band-sorted 7.90; round 5 (virtual rpe kernels,
# clouds on XCDs) random 6.60, band-sorted 6.52 / 6.47 / 6.43 / 6.42 / 6.43 at 2 / 3 / 4 / 5 / 10 bits per axis - which is why
# Engine.prepare now sorts the bands itself (ops.band_sort, 4 bits per axis); run this probe with RL_NO_BAND_SORT=1 to see the
# host-sorted permutation against the random one
