import os, sys
import numpy as np, torch
REPO = os.environ.get("RAG_REPO") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "3d_recognizer_amd")); sys.path.insert(0, REPO)
DEV = "cuda"
from oracle import randlanet_oracle as O
from oracle.init_formula import formula_state_dict
from oracle.loss_metrics_oracle import loss_by_name
from randlanet.utils.losses import get_loss
from randlanet.utils.modules import RandLANet, RandLANetSettings
C, N, K, F, layers, B, loss_name = 3, 1029, 8, 1, [16, 32, 64], 1, "cross_entropy"
sd = formula_state_dict(O.state_dict_layout(C, F, layers), seed=C + N)
net = RandLANet(RandLANetSettings(n_classes=C, n_points=N, n_features=F, n_neighbors=K, layer_sizes=list(layers)), DEV)
net.load_state_dict(sd); net.fc_end[2].p = 0.0; net.train()
rs = np.random.RandomState(N)
x = rs.uniform(0, 1, (B, N, 3 + F)).astype(np.float32)
y = np.minimum((x[..., 2] * C).astype(np.int64), C - 1)
np.random.seed(21); perm = np.random.permutation(N); np.random.seed(21)
logits = net(torch.from_numpy(x).to(DEV))
get_loss(loss_name)(logits, torch.from_numpy(y).to(DEV)).backward()
def oracle(dt):
    P = {k: ((v.to(dt) if v.is_floating_point() else v).clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else (v.to(dt) if v.is_floating_point() else v).clone()) for k, v in sd.items()}
    ref = O.forward(P, torch.from_numpy(x).to(dt), perm, layer_sizes=layers, n_neighbors=K, training=True, dropout_p=0.0)
    loss_by_name(loss_name, ref, torch.from_numpy(y)).backward()
    return P
P32, P64 = oracle(torch.float32), oracle(torch.float64)
w_h = w_o = 0.0
for name, p in net.named_parameters():
    r = P64[name].grad; m = float(r.abs().max()) + 1e-30
    eh = float((p.grad.cpu().double() - r).abs().max()) / m
    eo = float((P32[name].grad.double() - r).abs().max()) / m
    w_h, w_o = max(w_h, eh), max(w_o, eo)
    if eh > 2e-3 or eo > 2e-3: print(f"  {name:45s} hip-vs-f64 {eh:.2e}   f32oracle-vs-f64 {eo:.2e}")
print("worst: hip", w_h, "f32 oracle", w_o)
