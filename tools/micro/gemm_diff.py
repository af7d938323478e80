"""Every rl_gemm call of one forward + backward of the ragged test configuration against an fp64 torch product."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "3d_recognizer_amd")); sys.path.insert(0, REPO)
DEV = "cuda"
from oracle import randlanet_oracle as O
from oracle.init_formula import formula_state_dict
from randlanet.utils.losses import get_loss
from randlanet.utils.modules import RandLANet, RandLANetSettings
from randlanet import _ops as ops
from randlanet import _hip as H
orig = ops.gemm
def checked(a, W, w_ks, w_ns, N, bias=None, **kw):
    out = kw.get("out"); acc = kw.get("accumulate", False)
    old = out.clone() if (out is not None and acc) else None
    simple = isinstance(a, ops.Lazy) and a.B == 1 and kw.get("out2") is None and kw.get("addend") is None
    Y = orig(a, W, w_ks, w_ns, N, bias, **kw)
    torch.cuda.synchronize()
    if not simple:
        print("unchecked", type(a).__name__, getattr(a, "B", None), getattr(a, "n", None), getattr(a, "bstride", None), N, kw.keys())
    if simple:
        K = a.C
        A = a.raw[:a.B * a.n, :K].double()
        if a.scale is not None:
            z = A * a.scale.double() + a.shift.double()
            A = z if a.act == 0 else (torch.relu(z) if a.act == 1 else torch.nn.functional.leaky_relu(z, a.slope))
        flat = W.flatten().double()
        kk = torch.arange(K, device=DEV)[:, None] * w_ks; nn = torch.arange(N, device=DEV)[None, :] * w_ns
        Wm = flat[(kk + nn)]
        ref = A @ Wm
        if bias is not None: ref = ref + bias.double()
        rows = a.B * a.n
        if old is not None: ref = ref + old[:rows, :N].double()
        e = float((Y[:rows, :N].double() - ref).abs().max()); m = float(ref.abs().max())
        st = kw.get("stats")
        es = 0.0
        if st is not None:
            slots = H.row_blocks(rows, 128)
            tot = st.view(-1, 2, N)[:slots].sum(0)
            Yd = Y[:rows, :N].double()
            es = max(float((tot[0] - Yd.sum(0)).abs().max() / (Yd.abs().sum(0).max() + 1e-30)), float((tot[1] - (Yd * Yd).sum(0)).abs().max() / ((Yd * Yd).sum(0).max() + 1e-30)))
        tag = "BAD " if (e > 1e-4 * m + 1e-7 or es > 1e-5) else "ok  "
        print(f"{tag} M {rows} K {K} N {N} ks {w_ks} ns {w_ns} acc {acc} lazy {a.scale is not None} act {a.act} stats {kw.get('stats') is not None} ldy {Y.shape[1]} err {e:.2e} max {m:.2e} stats_err {es:.2e} kernel {H.lib().rl_last_kernel().decode()}")
    return Y
ops.gemm = checked
import randlanet._engine as E
C, N, K, F, layers, B = 3, 1029, 8, 1, [16, 32, 64], 1
sd = formula_state_dict(O.state_dict_layout(C, F, layers), seed=C + N)
net = RandLANet(RandLANetSettings(n_classes=C, n_points=N, n_features=F, n_neighbors=K, layer_sizes=list(layers)), DEV)
net.load_state_dict(sd); net.fc_end[2].p = 0.0; net.train()
rs = np.random.RandomState(N)
x = rs.uniform(0, 1, (B, N, 3 + F)).astype(np.float32)
y = np.minimum((x[..., 2] * C).astype(np.int64), C - 1)
np.random.seed(21)
logits = net(torch.from_numpy(x).to(DEV))
get_loss("cross_entropy")(logits, torch.from_numpy(y).to(DEV)).backward()
from oracle.loss_metrics_oracle import loss_by_name
P = {k: (v.double().clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else (v.double() if v.is_floating_point() else v).clone()) for k, v in sd.items()}
np.random.seed(21); perm = np.random.permutation(N)
ref = O.forward(P, torch.from_numpy(x).double(), perm, layer_sizes=layers, n_neighbors=K, training=True, dropout_p=0.0)
loss_by_name("cross_entropy", ref, torch.from_numpy(y)).backward()
w = 0.0
for name, p in net.named_parameters():
    if name.endswith("conv.bias") or name == "fc_start.bias": continue
    r = P[name].grad; w = max(w, float((p.grad.cpu().double() - r).abs().max()) / (float(r.abs().max()) + 1e-30))
print("worst gradient error vs fp64 oracle (checked run):", w)
