"""UpSampler.forward of the reference (randlanet/utils/modules.py:416-456) on the HIP kernels:
exact K-NN (rl_knn_i32) + rl_upsample_cf.  Features are (B,F,N1,1), coordinates (B,N,3)."""
import torch

from .. import _hip as H
from .. import _ops as ops

_POWER = {"nni": 0, "nna": 1, "idw": 1, "isdw": 2}   # 'nna' == 'idw': modules.py:434-437 passes no flag


def upsample_features(approach: str, features: torch.Tensor, xyz: torch.Tensor, xyz_upsampled: torch.Tensor,
                      device: torch.device) -> torch.Tensor:
    if approach == "none":
        return features
    if approach not in _POWER:
        raise ValueError(f"Upsampling approach {approach} not understood!")
    if torch.device(device).type != "cuda":
        from .._cpu import upsample_host          # the module lives on the CPU device (model.py:38-40): host twin
        return upsample_host(approach, features, xyz, xyz_upsampled)
    power = _POWER[approach]
    k = 1 if power == 0 else 8                                        # modules.py:371
    f = features.to(device, torch.float32)
    B, F, N1 = f.shape[0], f.shape[1], f.shape[2]
    s = xyz.to(device, torch.float32).contiguous()
    q = xyz_upsampled.to(device, torch.float32).contiguous()
    with torch.cuda.device(device):
        idx, d2 = ops.knn_i32(s, q, s.shape[1], q.shape[1], k)
        out = ops.upsample_cf(f.reshape(B, F, N1).contiguous(), idx, d2, power)
    return out.unsqueeze(-1)
