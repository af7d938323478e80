import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "3d_recognizer_amd"))
import torch
from randlanet import _hip as H, _ops as ops
DEV = torch.device("cuda")
d, B, n_parent, n = 16, int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
acc = bool(int(sys.argv[4])) if len(sys.argv) > 4 else True
torch.manual_seed(d + n)
K, h = 16, d // 2
xyz = torch.rand(B, n_parent, 3, device=DEV)
idx, d2 = ops.knn_i32(xyz, xyz, n, n, K)
W1, b1 = torch.randn(h, 10, device=DEV) * 0.5, torch.randn(h, device=DEV) * 0.1
W2, b2 = torch.randn(h, h, device=DEV) / h ** 0.5, torch.randn(h, device=DEV) * 0.1
g1w, g1b = torch.rand(h, device=DEV) + 0.5, torch.randn(h, device=DEV) * 0.2
g2w, g2b = torch.rand(h, device=DEV) + 0.5, torch.randn(h, device=DEV) * 0.2
Gf = torch.randn(B * n_parent, h, device=DEV)
g = ops.Lazy(Gf, B, n, n_parent, h, torch.rand(h, device=DEV) + 0.5, torch.randn(h, device=DEV) * 0.3, 2, 0.2)
Ws = torch.randn(d, d, device=DEV) / d ** 0.5
rows, P = B * n * K, B * n
dP = torch.randn(P, d, device=DEV)
def bn(stats, nslots, gamma, beta, c):
    rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
    return ops.bn_finalize(stats, rows, 128, c, gamma, beta, rm, rv, None, 0.99, 1e-6, True, nslots=nslots)
rpe = ops.rpe_build(ops.Rpe(xyz, idx, d2, B, n, K))
st1 = ops.new_stats(DEV, h)
Y1 = ops.gemm(rpe, W1, 1, 10, h, b1, stats=st1)
s1 = bn(st1, H.row_blocks(rows, 128), g1w, g1b, h)
u1 = ops.Lazy(Y1, B, n * K, n * K, h, s1[0], s1[1], 1, 0.0, s1[2], s1[3])
st2 = ops.new_stats(DEV, h)
Y2 = ops.gemm(u1, W2, 1, h, h, b2, stats=st2)
s2 = bn(st2, H.row_blocks(rows, 128), g2w, g2b, h)
u2 = ops.Lazy(Y2, B, n * K, n * K, h, s2[0], s2[1], 1, 0.0, s2[2], s2[3])
vr = ops.VirtualRpe(xyz, idx, d2, B, n, h, W1, b1, W2, b2)
vr.bn1 = ops.Lazy(d2, B, n * K, n * K, h, s1[0], s1[1], 1, 0.0, s1[2], s1[3])
vr.bn2 = ops.Lazy(d2, B, n * K, n * K, h, s2[0], s2[1], 1, 0.0, s2[2], s2[3])
for stage, u in ((2, u2), (1, u1)):
    GUs = torch.full((rows, h), 0.25, device=DEV); dWs = torch.empty(d, d, device=DEV)
    DGs = ops.pool_bwd(u, g, idx, Ws, n, d, dP, GUs, acc, dWs)
    GUv = torch.full((rows, h), 0.25, device=DEV); dWv = torch.empty(d, d, device=DEV)
    nslots = H.lib().rl_pool_bwd_slots(P, d)
    bst = torch.empty((nslots, 2, h), dtype=torch.float64, device=DEV)
    DGv = ops.pool_bwd(vr, g, idx, Ws, n, d, dP, GUv, acc, dWv, stage=stage, bn_bwd_stats=bst)
    for name, a, b_ in (("DG", DGv, DGs), ("GU", GUv, GUs)):
        err = (a - b_).abs().view(P, 16 * h).max(1).values
        bad = torch.nonzero(err > 1e-3).flatten()
        print(f"stage {stage} {name}: max err {float(err.max()):.3e}, bad points {bad.numel()} of {P}; first {bad[:12].tolist()} last {bad[-6:].tolist()}")
    print(f"stage {stage} dW err {float((dWv - dWs).abs().max()):.3e}", "grid", nslots)
