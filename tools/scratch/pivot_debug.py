import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "3d_recognizer_amd"))
import numpy as np, torch
from oracle import randlanet_oracle as O
from oracle.init_formula import formula_state_dict
from randlanet import _engine as E
from randlanet import _ops as ops
from randlanet.utils.losses import get_loss
from randlanet.utils.modules import RandLANet, RandLANetSettings
DEV = torch.device("cuda", 0)
C, N, K, F, layers, B = 3, 1029, 8, 1, [16, 32, 64], 1
sd = formula_state_dict(O.state_dict_layout(C, F, layers), seed=C + N)
rs = np.random.RandomState(N)
x = torch.from_numpy(rs.uniform(0, 1, (B, N, 3 + F)).astype(np.float32)).to(DEV)
y = torch.from_numpy(np.minimum((x[..., 2].cpu().numpy() * C).astype(np.int64), C - 1)).to(DEV)
lib = ops.H.lib()
ops.set_wide_gemm("fp32")

def grads(div, running):
    E.BN_PIVOT_RUNNING = running
    net = RandLANet(RandLANetSettings(n_classes=C, n_points=N, n_features=F, n_neighbors=K, layer_sizes=list(layers)), DEV)
    net.load_state_dict(sd)
    net.fc_end[2].p = 0.0
    net.train()
    for it in range(2):
        net.zero_grad()
        lib.rl_set_sgemm_grid_div(div if it else 1)
        np.random.seed(21)
        logits = net(x)
        get_loss("cross_entropy")(logits, y).backward()
        out = {n: p.grad.detach().cpu().clone() for n, p in net.named_parameters()}
        if it == 0:
            eng = net.engine()
            k = "encoder.0.mlp_rpe1.batch_norm"
            print("after step 1", "running" if running else "Pv", "Pv", eng.Pv[k][:4].tolist(), "rm", eng.Bf[k + ".running_mean"][:4].tolist(),
                  "bias", eng.P["encoder.0.mlp_rpe1.conv.bias"][:4].tolist())
    lib.rl_set_sgemm_grid_div(1)
    return out

for running in (False, True):
    base = grads(1, running)
    g = grads(3, running)
    rows = []
    for name, r in base.items():
        d = float((g[name] - r).abs().max()); m = float(r.abs().max())
        rows.append((d / (m + 1e-12), d, m, name))
    rows.sort(reverse=True)
    print("pivot on", "running mean" if running else "Pv")
    for e, d, m, n in rows[:8]:
        print(f"   {n:45s} rel {e:.2e} abs {d:.2e} scale {m:.2e}")
