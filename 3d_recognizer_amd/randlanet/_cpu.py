"""The package's CPU device: eval-mode RandLA-Net forward, K-NN and up-sampling on the host.

The reference picks the CPU when no GPU is present (randlanet/model.py:38-40) and config P of BASELINE.json is
`predict.py` run that way, so `Model(use_gpu=False)` / a GPU-less box must keep working.  This module is that path and
nothing else: it is chosen by the DEVICE of the model (an explicit `use_gpu=False`, or no HIP device at all) - never as
a fallback for a missing or failing HIP library on a GPU, which still raises.  Inference only; training needs the GPU.

Neighbour search is the library's own host twin rl_knn_f32_cpu (csrc/knn_host.hip: uniform grid, the reference's fp32
distance expression, (d2, index) order); the per-point arithmetic is plain PyTorch-CPU in the channel-last layout of the
HIP schedule (rows = points, a 1x1 convolution = one matmul), following modules.py:542-611 op for op.
"""
import ctypes as C
from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import _hip as H

BN_EPS = 1e-6   # modules.py:87, :497


def knn_host(support: torch.Tensor, query: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """(B,Ns,3), (B,Nq,3) host tensors -> idx int64 (B,Nq,k), d2 fp32 (B,Nq,k): rl_knn_f32_cpu."""
    s = support.detach().to("cpu", torch.float32).contiguous()
    q = query.detach().to("cpu", torch.float32).contiguous()
    B, Ns, _ = s.shape
    Nq = q.shape[1]
    idx = torch.empty((B, Nq, k), dtype=torch.int64)
    d2 = torch.empty((B, Nq, k), dtype=torch.float32)
    H.check(H.lib().rl_knn_f32_cpu(s.data_ptr(), q.data_ptr(), B, Ns, Nq, int(k), idx.data_ptr(), d2.data_ptr()),
            "rl_knn_f32_cpu")
    return idx, d2


def upsample_host(approach: str, features: torch.Tensor, xyz: torch.Tensor, xyz_up: torch.Tensor) -> torch.Tensor:
    """UpSampler.forward (modules.py:416-456) on the host: features (B,F,N1,1) -> (B,F,N2,1)."""
    power = {"nni": 0, "nna": 1, "idw": 1, "isdw": 2}[approach]       # 'nna' == 'idw': modules.py:434-437 passes no flag
    f = features.detach().to("cpu", torch.float32)
    B, Fc, N1 = f.shape[0], f.shape[1], f.shape[2]
    f = f.reshape(B, Fc, N1)
    k = 1 if power == 0 else 8                                         # modules.py:371
    idx, d2 = knn_host(xyz, xyz_up, k)
    gathered = torch.gather(f.unsqueeze(2).expand(B, Fc, idx.shape[1], N1), 3,
                            idx.unsqueeze(1).expand(B, Fc, idx.shape[1], k))          # (B,F,N2,k)
    if power == 0:
        return gathered[..., 0].unsqueeze(-1)
    dist = torch.sqrt(d2)
    dp = dist * dist if power == 2 else dist
    w = (1.0 + 1e-7) / (dp + 1e-7)                                    # modules.py:398-408
    w = w / w.sum(-1, keepdim=True)
    return (gathered * w.unsqueeze(1)).sum(-1).unsqueeze(-1)


class HostForward:
    """Eval-mode forward over a reference-layout state_dict (parameters are read in place, nothing is copied)."""

    def __init__(self, layer_sizes: List[int], n_neighbors: int, decimation: int, params: Dict[str, torch.Tensor],
                 buffers: Dict[str, torch.Tensor]):
        self.layers, self.K, self.dec = list(layer_sizes), int(n_neighbors), int(decimation)
        self.P, self.Bf = params, buffers

    # y = act(BN_eval(x . W + b)): SharedMLP (modules.py:93-104) on (..., Cin) rows
    def _mlp(self, x: torch.Tensor, name: str, act: str = "", slope: float = 0.0, transposed: bool = False, bn: bool = True):
        W = self.P[f"{name}.conv.weight"]
        W2 = W.view(W.shape[0], W.shape[1])
        y = x @ (W2 if transposed else W2.t()) + self.P[f"{name}.conv.bias"]
        if bn:
            y = self._bn(y, f"{name}.batch_norm")
        return self._act(y, act, slope)

    def _bn(self, y: torch.Tensor, name: str) -> torch.Tensor:
        scale = self.P[f"{name}.weight"] / torch.sqrt(self.Bf[f"{name}.running_var"] + BN_EPS)
        return y * scale + (self.P[f"{name}.bias"] - self.Bf[f"{name}.running_mean"] * scale)

    @staticmethod
    def _act(y, act, slope):
        if act == "relu":
            return torch.relu(y)
        if act == "lrelu":
            return F.leaky_relu(y, slope)
        return y

    @staticmethod
    def _gather(feat: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        """feat (B,n,C), idx (B,n,K) -> (B,n,K,C)"""
        B, n, K = idx.shape
        return torch.gather(feat, 1, idx.reshape(B, n * K, 1).expand(-1, -1, feat.shape[2])).view(B, n, K, feat.shape[2])

    def _pool(self, name: str, x: torch.Tensor) -> torch.Tensor:
        """AttentivePooling (modules.py:246-253): x (B,n,K,d)"""
        scores = torch.softmax(x @ self.P[f"{name}.score_fn.0.weight"].t(), dim=2)
        return self._mlp((scores * x).sum(2), f"{name}.mlp", "relu")

    def _lfa(self, l: int, x: torch.Tensor, xyz: torch.Tensor) -> torch.Tensor:
        """LocalFeatureAggregation (modules.py:298-325): x (B,n,Cin), xyz (B,n,3)"""
        e = f"encoder.{l}"
        idx, d2 = knn_host(xyz, xyz, self.K)
        nb = self._gather(xyz, idx)
        ctr = xyz.unsqueeze(2).expand_as(nb)
        rpe = torch.cat([ctr, nb, ctr - nb, torch.sqrt(d2).unsqueeze(-1)], dim=-1)      # modules.py:173-186
        f0 = self._mlp(x, f"{e}.mlp1", "lrelu", 0.2)
        r1 = self._mlp(rpe, f"{e}.mlp_rpe1", "relu")
        q1 = self._pool(f"{e}.pool1", torch.cat([r1, self._gather(f0, idx)], dim=-1))
        r2 = self._mlp(r1, f"{e}.mlp_rpe2", "relu")
        q2 = self._pool(f"{e}.pool2", torch.cat([r2, self._gather(q1, idx)], dim=-1))
        return F.leaky_relu(self._mlp(q2, f"{e}.mlp2") + self._mlp(x, f"{e}.shortcut"), 0.01)

    def __call__(self, inp: torch.Tensor, perm: np.ndarray) -> torch.Tensor:
        """(B,N,3+F) host fp32, perm = the forward's np.random.permutation(N) -> logits (B,C,N)."""
        B, N, _ = inp.shape
        L, dec = len(self.layers), self.dec
        with torch.no_grad():
            p = torch.from_numpy(np.asarray(perm))
            x_in = inp[:, p]                                                   # modules.py:571-573
            xyz = x_in[..., :3].contiguous()
            x = x_in @ self.P["fc_start.weight"].t() + self.P["fc_start.bias"]
            x = F.leaky_relu(self._bn(x, "bn_start.0"), 0.2)
            skips, ratio = [], 1
            for l in range(L):
                n = N // ratio
                x = self._lfa(l, x[:, :n], xyz[:, :n].contiguous())
                skips.append(x)
                ratio *= dec
            x = self._mlp(x[:, : N // ratio], "mlp", "relu")
            for j in range(L):
                n_c, n_f = N // ratio, dec * N // ratio
                nn, _ = knn_host(xyz[:, :n_c].contiguous(), xyz[:, :n_f].contiguous(), 1)       # modules.py:358
                up = torch.gather(x, 1, nn.expand(-1, -1, x.shape[2]))
                x = self._mlp(torch.cat([up, skips.pop()], dim=-1), f"decoder.{j}", "relu", transposed=True)
                ratio //= dec
            x = self._mlp(x, "fc_end.0", "relu")
            x = self._mlp(x, "fc_end.1", "relu")
            x = self._mlp(x, "fc_end.3", bn=False)                              # Dropout is the identity in eval mode
            out = torch.empty_like(x)
            out[:, p] = x                                                      # modules.py:608
            return out.permute(0, 2, 1).contiguous()
