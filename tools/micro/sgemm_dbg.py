import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "3d_recognizer_amd"))
from randlanet import _ops as ops
from randlanet import _hip as H
DEV = "cuda"
torch.manual_seed(0)
for M in (16, 64, 257, 1029, 8232, 2056):
    for K, N in ((8, 8), (16, 16), (10, 8), (16, 32), (32, 32), (64, 64), (32, 64), (64, 16), (4, 8), (8, 3)):
        for acc in (False, True):
            for lazy in (False, True):
                A = torch.randn(M, K, device=DEV)
                W = torch.randn(N, K, device=DEV) / K ** 0.5
                a = ops.plain(A, 1, M)
                Ad = A.double()
                if lazy:
                    a.scale, a.shift, a.act, a.slope = torch.rand(K, device=DEV) + .5, torch.randn(K, device=DEV), 2, 0.2
                    Ad = torch.nn.functional.leaky_relu(Ad * a.scale.double() + a.shift.double(), 0.2)
                bias = torch.randn(N, device=DEV)
                Y0 = torch.randn(M, N, device=DEV)
                Y = Y0.clone()
                ops.gemm(a, W, 1, K, N, bias, out=Y, accumulate=acc)
                ref = Ad @ W.double().t() + bias.double() + (Y0.double() if acc else 0)
                e = float((Y.double() - ref).abs().max())
                if e > 2e-5 * float(ref.abs().max()):
                    print("BAD", M, K, N, "acc", acc, "lazy", lazy, e, H.lib().rl_last_kernel().decode() if hasattr(H.lib(), "rl_last_kernel") else "")
print("done")
