import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "3d_recognizer_amd"))
from randlanet import _ops as ops
DEV = "cuda"
for rows in (2047, 2048, 2049, 2050, 3000, 4097):
    C = 64
    torch.manual_seed(C)
    Y = (torch.randn(rows, C, device=DEV) * 2 + 1)
    gamma = torch.rand(C, device=DEV) + 0.5
    beta = torch.randn(C, device=DEV)
    G = torch.randn(rows, C, device=DEV)
    Yd = Y.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    mean = Yd.mean(0); var = Yd.var(0, unbiased=False)
    ref = (Yd - mean) / torch.sqrt(var + 1e-6) * gd + bd
    torch.nn.functional.leaky_relu(ref, 0.2).backward(G.double())
    Yf = Y.clone().requires_grad_(True)
    r32 = torch.nn.functional.batch_norm(Yf.t()[None], None, None, gamma, beta, True, 0.99, 1e-6)[0].t()
    torch.nn.functional.leaky_relu(r32, 0.2).backward(G)
    m32 = Y.mean(0); is32 = 1.0 / torch.sqrt(Y.var(0, unbiased=False) + 1e-6)
    sc = gamma * is32
    lz = ops.Lazy(Y.contiguous(), 1, rows, rows, C, sc, beta - m32 * sc, 2, 0.2, m32, is32)
    out = {}
    for small in (False, True):
        ops.NO_BN_SMALL = not small
        g = G.clone(); dg, db = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
        ops.bn_backward(g, lz, dg, db, True)
        out[small] = float((g.double() - Yd.grad).abs().max())
    print(rows, "three launches", out[False], "one launch (if supported)", out[True], "torch fp32", float((Yf.grad.double() - Yd.grad).abs().max()), flush=True)
