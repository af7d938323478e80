#!/usr/bin/env python3
"""How well conditioned is a train-step test point?  The same step with its BatchNorm partial sums merely RE-GROUPED (streaming GEMM
on a third of its workgroups: bitwise the same products, other lanes add other rows) and with the sampling bands in cell order,
against the run as drawn: worst gradient difference per tensor relative to the largest gradient entry.  Round 6 used it to find
that `test_cell_order_...`'s test point moves by 1e-3 under ANY re-grouping (exact-product mode included) while the same net with
other formula weights moves by 3e-7 - the bound of that test is now measured against the re-grouped run.
usage: python tools/conditioning_probe.py [fp32|bf16x3] [N]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "3d_recognizer_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
from randlanet import _ops as ops
from randlanet._train import TrainStep
from oracle import randlanet_oracle as O
from oracle.init_formula import formula_state_dict
from randlanet.utils.modules import RandLANet, RandLANetSettings
DEV = torch.device("cuda", 0)
mode = sys.argv[1] if len(sys.argv) > 1 else "fp32"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ops.set_wide_gemm(mode)
C, K, layers = 2, 16, [16, 64, 128, 256]
rs = np.random.RandomState(21)
x = rs.uniform(0, 1, (3, N, 3)).astype(np.float32)
y = (np.linalg.norm(x - 0.5, axis=-1) < 0.3).astype(np.int64)
np.random.seed(4)
perm = np.random.permutation(N)
sd = formula_state_dict(O.state_dict_layout(C, 0, layers), seed=7)
lib = ops.H.lib()
def run(sort, div):
    ops.NO_BAND_SORT = not sort
    net = RandLANet(RandLANetSettings(n_classes=C, n_points=N, n_neighbors=K, layer_sizes=layers), DEV)
    net.load_state_dict(sd)
    net.train(); net.fc_end[2].p = 0.0
    st = TrainStep(net, 3, N, loss="dice", use_graph=False)
    st.set_batch(torch.from_numpy(x).to(DEV), torch.from_numpy(y).to(DEV))
    st.perm.copy_(torch.from_numpy(perm))
    lib.rl_set_sgemm_grid_div(div)
    with torch.cuda.device(0):
        st._fwd_bwd()
    torch.cuda.synchronize()
    lib.rl_set_sgemm_grid_div(1)
    return {n: g.detach().cpu().clone() for n, g in st.flat.grads.items()}
def cmp(a, b, tag):
    rows = []
    scale = max(float(v.abs().max()) for v in a.values())
    for n in a:
        d = float((a[n] - b[n]).abs().max()); m = float(a[n].abs().max())
        rows.append((d / scale, d, m, n))
    rows.sort(reverse=True)
    print(tag, "global scale", f"{scale:.3e}")
    for e, d, m, n in rows[:5]:
        print(f"   {n:45s} diff/global-scale {e:.2e} abs {d:.2e} own scale {m:.2e}")
u1 = run(False, 1)
cmp(u1, run(False, 3), f"[{mode} N={N}] unsorted div1 vs unsorted div3 (pure regrouping)")
cmp(u1, run(True, 1), f"[{mode} N={N}] unsorted vs cell-sorted")
