"""RandLA-Net for MI355X behind the reference's module surface.

Mirrors randlanet/utils/modules.py of matthiasverstraete/3d_recognizer: the same settings
dataclass, class names, constructor signatures, parameter names / shapes (state_dict is
interchangeable both ways) and forward contract.  On a HIP device all arithmetic is done by the
kernels of librandla_hip.so, scheduled by `_engine.Engine`; a missing library or a failing launch
raises - nothing falls back.  A model PLACED on the CPU (the reference's own device choice,
model.py:38-40: no GPU, or use_gpu=False) runs the host inference path of `_cpu.py`
(rl_knn_f32_cpu + PyTorch-CPU rows; inference only, training needs the MI355X).
"""
import ctypes
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
from torch import nn

from .. import _hip as H
from .. import _ops as ops
from .._engine import Engine

_KNN_CHOICES = ("kdtree", "approximate", "naive")
_UPSAMPLING_CHOICES = ("none", "nni", "nna", "idw", "isdw")


@dataclass
class RandLANetSettings:
    """Model settings; field for field the reference's dataclass (modules.py:10-39).

    `knn` is accepted for compatibility: on this build every choice runs the same exact HIP
    search (rl_knn_*), which is what "kdtree" computes and what "approximate"/"naive"
    approximate."""
    n_classes: int
    n_points: int = 10000
    n_features: int = 0
    n_neighbors: int = 32
    decimation: int = 4
    layer_sizes: List[int] = field(default_factory=lambda: [16, 64, 128, 256])
    knn: str = "approximate"
    upsampling: str = "nni"

    def __post_init__(self):
        # same checks and wording as the reference's __new__ hook (modules.py:41-52)
        assert self.knn in _KNN_CHOICES, (
            f'knn value "{self.knn}" not understood, should be "kdtree", "approximate" or "naive"')
        assert self.upsampling in _UPSAMPLING_CHOICES, (
            f'upsampling value "{self.upsampling}" not understood, '
            'should be "none", "nni", "nna", "idw", or "isdw"')

    def update(self, **kwargs):
        for key, value in kwargs.items():
            if hasattr(self, key):
                setattr(self, key, value)


# ---------------------------------------------------------------------------------------------------------------
# Stand-alone forwards of the sub-modules (reference modules.py:93-104, 159-186, 199-221, 246-253, 298-325).  Inside
# RandLANet.forward the blocks run as one fused launch schedule (_engine.py); called on their own they take and return the
# reference's (B, C, N, K) tensors and run the same HIP kernels one block at a time.  SharedMLP is differentiable on its own
# (input, conv weight / bias, BatchNorm weight / bias: `_SharedMLPRows`, the kernels of the network's backward); the other
# blocks are forward only when called alone - training goes through RandLANet / TrainStep.  Layout changes are views / copies;
# all arithmetic is in librandla_hip.so.
def _act_code(activation) -> tuple:
    if activation is None:
        return H.ACT_NONE, 0.0
    if isinstance(activation, nn.LeakyReLU):
        return H.ACT_LRELU, float(activation.negative_slope)
    if isinstance(activation, nn.ReLU):
        return H.ACT_RELU, 0.0
    raise H.HipKernelError(f"activation {type(activation).__name__} has no HIP implementation")


def _to_rows(x: torch.Tensor) -> torch.Tensor:
    """(B, C, N, K) -> (B*N*K, C) fp32 contiguous"""
    B, Cc, N, K = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * N * K, Cc).to(torch.float32).contiguous()


def _from_rows(rows: torch.Tensor, B: int, N: int, K: int) -> torch.Tensor:
    return rows.view(B, N, K, rows.shape[1]).permute(0, 3, 1, 2)


def _require_device(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise H.HipKernelError(f"{what} runs on an MI355X (HIP) device: move the module and its input there")


class _SharedMLPRows(torch.autograd.Function):
    """rows (M, n_in) -> act(BN(rows . W + b)) as an autograd node over the HIP kernels: forward = rl_gemm (+ batch
    statistics) + rl_bn_finalize + the lazy application; backward = rl_bn_bwd_* + rl_wgrad + rl_gemm with swapped strides."""

    @staticmethod
    def forward(ctx, rows, weight, bias, gamma, beta, mod):
        M = rows.shape[0]
        transposed = isinstance(mod.conv, nn.ConvTranspose2d)
        n_in, n_out = (weight.shape[0], weight.shape[1]) if transposed else (weight.shape[1], weight.shape[0])
        W2 = weight.detach().view(weight.shape[0], weight.shape[1])
        ks, ns = ops.weight_strides(W2, transposed, n_in, n_out)
        bn = mod.batch_norm
        train_stats = bn is not None and mod.training
        stats = ops.new_stats(rows.device, n_out) if train_stats else None
        a = ops.plain(rows.detach(), 1, M)
        # batch statistics as shifted sums around the running mean (rl_gemm_desc.stats_pivot_*)
        Y = ops.gemm(a, W2, ks, ns, n_out, bias.detach(), stats=stats, pivot=(bn.running_mean, None) if train_stats else None)
        act, slope = _act_code(mod.activation)
        y = ops.Lazy(Y, 1, M, M, n_out, None, None, act, slope)
        if bn is not None:
            y.scale, y.shift, y.mean, y.invstd = ops.bn_finalize(
                stats, M, 128, n_out, gamma.detach(), beta.detach(), bn.running_mean, bn.running_var,
                bn.num_batches_tracked if train_stats else None, 0.99, 1e-6, train_stats, pivoted=train_stats,
                nslots=ops.gemm_stat_slots(M, n_out, n_in) if train_stats else None)
            if not train_stats:      # eval mode: the "batch" statistics of the backward formulas are the running ones
                y.mean, y.invstd = bn.running_mean.clone(), torch.rsqrt(bn.running_var + 1e-6)
        elif act != H.ACT_NONE:
            y.scale, y.shift = torch.ones(n_out, device=rows.device), torch.zeros(n_out, device=rows.device)
        ctx.saved = (a, W2, ks, ns, y, bn is not None, train_stats)
        if y.scale is None:
            return Y
        out = torch.empty_like(Y)
        ops.copy_rows(Y, (0, n_out), M, out, (0, n_out), M, M, lazy=y)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        a, W2, ks, ns, y, has_bn, train_stats = ctx.saved
        M, n_out, dev = y.rows, y.C, grad_out.device
        with torch.cuda.device(dev):
            G = grad_out.to(torch.float32).contiguous().clone()
            dgamma = torch.zeros(n_out, device=dev) if has_bn else None
            dbeta = torch.zeros(n_out, device=dev) if has_bn else None
            if has_bn and not train_stats:
                # eval mode: dgamma / dbeta are plain sums (no batch-statistics terms in the input gradient)
                st = ops.new_stats(dev, n_out)
                d = ops._bn_bwd_desc(G, y.bstride, y)
                d.stats = st.data_ptr()
                H.check(H.lib().rl_bn_bwd_reduce(ctypes.byref(d), ops._st()), "rl_bn_bwd_reduce")
                ops._bn_bwd_finalize(st, H.lib().rl_bn_bwd_slots(M), M, n_out, dgamma, dbeta, torch.empty(2 * n_out, device=dev), None)
                y.mean = None                               # -> bn_backward applies the activation derivative and the scale only
            if y.scale is not None:
                ops.bn_backward(G, y, dgamma, dbeta, train_stats)
            dW, db = torch.empty_like(W2), torch.empty(n_out, device=dev)
            ops.wgrad(a, G, M, n_out, dW, ks, ns, db)
            d_rows = ops.gemm(ops.plain(G, 1, M), W2, ns, ks, a.C, None)
        return d_rows, dW.view(W2.shape[0], W2.shape[1], 1, 1), db, dgamma, dbeta, None


class SharedMLP(nn.Module):
    """Shared MLP (reference modules.py:60-104): 1x1 conv or transposed conv, optional BatchNorm2d(eps=1e-6,
    momentum=0.99), optional activation.  forward: (B, n_in, N, K) -> (B, n_out, N, K)."""

    def __init__(self, n_in: int, n_out: int, transpose: bool = False, bn: bool = True,
                 activation: Optional[nn.Module] = None):
        super().__init__()
        conv = nn.ConvTranspose2d if transpose else nn.Conv2d
        self.conv = conv(n_in, n_out, kernel_size=1, stride=1, padding_mode="zeros")
        self.batch_norm = nn.BatchNorm2d(n_out, eps=1e-6, momentum=0.99) if bn else None
        self.activation = activation

    def _rows_forward(self, rows: torch.Tensor) -> torch.Tensor:
        """(M, n_in) rows -> (M, n_out) rows: GEMM + (batch-statistics or running) BatchNorm + activation."""
        M = rows.shape[0]
        W = self.conv.weight.detach()
        transposed = isinstance(self.conv, nn.ConvTranspose2d)
        n_in, n_out = (W.shape[0], W.shape[1]) if transposed else (W.shape[1], W.shape[0])
        W2 = W.view(W.shape[0], W.shape[1])
        ks, ns = ops.weight_strides(W2, transposed, n_in, n_out)
        bn = self.batch_norm
        train_stats = bn is not None and self.training
        stats = ops.new_stats(rows.device, n_out) if train_stats else None
        Y = ops.gemm(ops.plain(rows, 1, M), W2, ks, ns, n_out, self.conv.bias.detach(), stats=stats,
                     pivot=(bn.running_mean, None) if train_stats else None)
        act, slope = _act_code(self.activation)
        if bn is not None:
            scale, shift, _, _ = ops.bn_finalize(stats, M, 128, n_out, bn.weight.detach(), bn.bias.detach(), bn.running_mean,
                                                 bn.running_var, bn.num_batches_tracked if train_stats else None,
                                                 0.99, 1e-6, train_stats, pivoted=train_stats,
                                                 nslots=ops.gemm_stat_slots(M, n_out, n_in) if train_stats else None)
        elif act != H.ACT_NONE:
            scale, shift = torch.ones(n_out, device=rows.device), torch.zeros(n_out, device=rows.device)
        else:
            return Y
        out = torch.empty_like(Y)
        ops.copy_rows(Y, (0, n_out), M, out, (0, n_out), M, M, lazy=ops.Lazy(Y, 1, M, M, n_out, scale, shift, act, slope))
        return out

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        _require_device(input, "SharedMLP")
        B, _, N, K = input.shape
        with torch.cuda.device(input.device):
            if torch.is_grad_enabled() and (input.requires_grad or self.conv.weight.requires_grad):
                bn = self.batch_norm
                rows = _SharedMLPRows.apply(_to_rows(input), self.conv.weight, self.conv.bias,
                                            bn.weight if bn is not None else None, bn.bias if bn is not None else None, self)
                return _from_rows(rows, B, N, K)
            with torch.no_grad():
                return _from_rows(self._rows_forward(_to_rows(input)), B, N, K)


class RelativePositionEncoding(nn.Module):
    """[x_i, x_nbr, x_i - x_nbr, dist] per neighbour (reference modules.py:156-186): (B, 10, N, K)."""

    def forward(self, xyz: torch.Tensor, neighbors: torch.Tensor, distances: torch.Tensor) -> torch.Tensor:
        _require_device(xyz, "RelativePositionEncoding")
        B, N, K = neighbors.shape
        with torch.cuda.device(xyz.device), torch.no_grad():
            rpe = ops.rpe_build(ops.Rpe(xyz.to(torch.float32).contiguous(), neighbors.to(torch.int32).contiguous(),
                                        distances.to(torch.float32).contiguous(), B, N, K), distances=True)
            return rpe.raw.view(B, N, K, 12)[..., :10].permute(0, 3, 1, 2)


class PointFeatureAugmentation(nn.Module):
    """cat[relative position encoding, gathered neighbour features] (reference modules.py:194-221): (B, 2h, N, K)."""

    def forward(self, relative_position_encoding: torch.Tensor, features: torch.Tensor, neighbors: torch.Tensor) -> torch.Tensor:
        _require_device(features, "PointFeatureAugmentation")
        B, N, K = neighbors.shape
        h = features.shape[1]
        with torch.cuda.device(features.device), torch.no_grad():
            r = _to_rows(relative_position_encoding)
            f = _to_rows(features)                                      # (B*N, h)
            X = torch.empty((B * N * K, r.shape[1] + h), dtype=torch.float32, device=f.device)
            ops.copy_rows(r, (0, r.shape[1]), N * K, X, (0, r.shape[1]), B * N * K, N * K)
            ops.copy_rows(f, (0, h), N, X, (r.shape[1], h), B * N * K, N * K, index=neighbors.to(torch.int32).contiguous())
            return _from_rows(X, B, N, K)


class AttentivePooling(nn.Module):
    """Attentive pooling (reference modules.py:224-253): (B, n_in, N, K) -> (B, n_out, N, 1)."""

    def __init__(self, n_in: int, n_out: int):
        super().__init__()
        self.score_fn = nn.Sequential(nn.Linear(n_in, n_in, bias=False), nn.Softmax(dim=-2))
        self.mlp = SharedMLP(n_in, n_out, activation=nn.ReLU())

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        _require_device(input, "AttentivePooling")
        B, d, N, K = input.shape
        with torch.cuda.device(input.device), torch.no_grad():
            X = _to_rows(input)
            S = ops.gemm(ops.plain(X, B, N * K), self.score_fn[0].weight.detach(), 1, d, d, None)
            pooled = ops.attpool_fwd(X, S, B * N, K)
            return _from_rows(self.mlp._rows_forward(pooled), B, N, 1)


class LocalFeatureAggregation(nn.Module):
    """One encoder block (reference modules.py:256-325); parameters registered in the reference's order so that
    state_dict() enumerates identically.  forward(xyz (B,N,3), input (B,n_in,N,1), knn_approach) -> (B, 2*n_out, N, 1)."""

    def __init__(self, n_in: int, n_out: int, n_neighbors: int, device: torch.device):
        super().__init__()
        self._n_neighbors = n_neighbors
        self._device = device
        self._n_out = n_out
        self.mlp1 = SharedMLP(n_in, n_out // 2, activation=nn.LeakyReLU(0.2))
        self.mlp2 = SharedMLP(n_out, 2 * n_out)
        self.shortcut = SharedMLP(n_in, 2 * n_out)
        self.knn = KNN(device)
        self.rpe = RelativePositionEncoding()
        self.pfa = PointFeatureAugmentation()
        self.mlp_rpe1 = SharedMLP(10, n_out // 2, activation=nn.ReLU())
        self.mlp_rpe2 = SharedMLP(n_out // 2, n_out // 2, activation=nn.ReLU())
        self.pool1 = AttentivePooling(n_out, n_out // 2)
        self.pool2 = AttentivePooling(n_out, n_out)
        self.lrelu = nn.LeakyReLU()

    def forward(self, xyz: torch.Tensor, input: torch.Tensor, knn_approach: str = "approximate") -> torch.Tensor:
        if knn_approach not in _KNN_CHOICES:
            raise ValueError(f"KNN approach {knn_approach} not understood!")
        dev = torch.device(self._device)
        if dev.type != "cuda":
            raise H.HipKernelError("LocalFeatureAggregation runs on an MI355X (HIP) device")
        from .._engine import Context
        B, n_in, N, _ = input.shape
        d, K = self._n_out, self._n_neighbors
        with torch.cuda.device(dev), torch.no_grad():
            pts = xyz.to(dev, torch.float32).contiguous()
            rows = _to_rows(input.to(dev))
            idx, d2 = ops.knn_i32(pts, pts, N, N, K)
            eng = Engine([d], K, 4, 2, 0, {f"encoder.0.{k}": v.detach() for k, v in self.named_parameters()},
                         {f"encoder.0.{k}": v for k, v in self.named_buffers()})
            ctx = Context()
            ctx.training, ctx.B, ctx.N = self.training, B, N
            out = eng._lfa(ctx, 0, ops.plain(rows, B, N), pts, N, d, idx, d2)
            return _from_rows(out.raw, B, N, 1)


class KNN(nn.Module):
    """K nearest neighbours (reference modules.py:107-150): returns (indices int64, distances),
    distances being sqrt of the squared L2.  Every `approach` runs the exact HIP search."""

    def __init__(self, device: torch.device):
        super().__init__()
        self._device = device

    def forward(self, xyz: torch.Tensor, xyz_query: torch.Tensor, n_neighbors: int,
                approach: str = "approximate") -> Tuple[torch.Tensor, torch.Tensor]:
        if approach not in _KNN_CHOICES:
            raise ValueError(f"KNN approach {approach} not understood!")
        from .knn import knn_exact
        neighbors, d2 = knn_exact(xyz.to(self._device), xyz_query.to(self._device), n_neighbors, device=self._device)
        return neighbors, torch.sqrt(d2)


class UpSampler(nn.Module):
    """Feature up-sampling (reference modules.py:328-456) on (B,F,N1,1) features."""

    def __init__(self, upsampling_approach: str, device: torch.device):
        super().__init__()
        self._upsampling_approach = upsampling_approach
        self._device = device
        self.knn = KNN(device)

    def forward(self, features: torch.Tensor, xyz: torch.Tensor, xyz_upsampled: torch.Tensor) -> torch.Tensor:
        from .upsample import upsample_features
        return upsample_features(self._upsampling_approach, features, xyz, xyz_upsampled, self._device)


class RandLANet(nn.Module):
    """RandLA-Net (reference modules.py:459-611): same constructor, parameters and forward
    contract; forward/backward run on the HIP kernels."""

    def __init__(self, settings: RandLANetSettings, device: Optional[torch.device] = None):
        super().__init__()
        self._settings = settings
        k = settings.n_neighbors
        if device is None:
            device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self._device = torch.device(device)
        sizes = list(settings.layer_sizes)
        L, dec = len(sizes), settings.decimation
        self._min_n_points = max(k * dec ** (L - 1), 2 * dec ** L)

        # registration order == reference (modules.py:494-530) -> identical state_dict order and,
        # under the same torch seed, identical default initialisation
        self.fc_start = nn.Linear(settings.n_features + 3, 8)
        self.bn_start = nn.Sequential(nn.BatchNorm2d(8, eps=1e-6, momentum=0.99), nn.LeakyReLU(0.2))
        self.encoder = nn.ModuleList()
        width = 8
        for d in sizes:
            self.encoder.append(LocalFeatureAggregation(width, d, k, self._device))
            width = 2 * d
        self.mlp = SharedMLP(width, width, activation=nn.ReLU())
        self.upsampling = UpSampler("nni", self._device)
        self.decoder = nn.ModuleList()
        width *= 2
        for d in sizes[::-1][1:]:
            self.decoder.append(SharedMLP(width, 2 * d, transpose=True, activation=nn.ReLU()))
            width = 4 * d
        self.decoder.append(SharedMLP(width, 8, transpose=True, activation=nn.ReLU()))
        self.fc_end = nn.Sequential(
            SharedMLP(8, 64, activation=nn.ReLU()),
            SharedMLP(64, 32, activation=nn.ReLU()),
            nn.Dropout(),
            SharedMLP(32, settings.n_classes, bn=False),
        )
        self.to(self._device)
        self._engine: Optional[Engine] = None
        self._engine_key = None
        self._infer_steps = {}
        # new weights: the engine's statistic pivots (the previous batch's means under the OLD weights) start over
        self.register_load_state_dict_post_hook(RandLANet._weights_replaced)

    @staticmethod
    def _weights_replaced(module, incompatible_keys) -> None:
        if getattr(module, "_engine", None) is not None:
            module._engine.reset_pivots()

    # -- reference properties (modules.py:534-540)
    @property
    def device(self) -> torch.device:
        return self._device

    @property
    def settings(self) -> RandLANetSettings:
        return self._settings

    # -- engine plumbing
    def engine(self) -> Engine:
        """The launch schedule bound to the current parameter / buffer tensors."""
        params = dict(self.named_parameters())
        buffers = dict(self.named_buffers())
        key = tuple(p.data_ptr() for p in params.values()) + tuple(b.data_ptr() for b in buffers.values())
        if self._engine is None or key != self._engine_key:
            s = self._settings
            self._engine = Engine(s.layer_sizes, s.n_neighbors, s.decimation, s.n_classes, s.n_features,
                                  {k: v.detach() for k, v in params.items()}, buffers)
            self._engine_key = key
        return self._engine

    _MAX_INFER_STEPS = 4

    def infer_step(self, B: int, N: int):
        """The eval forward for one batch shape as a replayable hipGraph (`_train.InferStep`), captured on first use and
        kept while the parameters stay where they are (Trainer.evaluate's passes run through these)."""
        from .._train import InferStep
        eng = self.engine()
        step = self._infer_steps.get((B, N))
        if step is None or step.engine is not eng:
            was_training = self.training
            step = InferStep(self, B, N)
            step.capture()
            self.train(was_training)
            self._infer_steps[(B, N)] = step
            # a captured graph + its static buffers per batch shape: keep the few most recent ones (a loader's full batch and
            # its ragged last batch are two shapes; a sweep over shapes must not pile graphs up)
            while len(self._infer_steps) > self._MAX_INFER_STEPS:
                self._infer_steps.pop(next(iter(self._infer_steps)))
        else:
            self._infer_steps[(B, N)] = self._infer_steps.pop((B, N))       # most recently used last
        return step

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        """(B, N, 3+F) -> logits (B, C, N) in the original point order (modules.py:542-611).
        Draws np.random.permutation(N) from the global numpy RNG exactly once, as the reference."""
        B, N, dim = input.size()
        assert dim == 3 + self._settings.n_features, "Input should have shape (B, N, 3 + F)!"
        assert N >= self._min_n_points, \
            f"Input point cloud should have at least {self._min_n_points} points!"
        if self._device.type != "cuda":
            # the model was PLACED on the CPU (use_gpu=False, or a box without a HIP device - the reference's device
            # choice, model.py:38-40): host inference path; never a fallback for a failing HIP library on a GPU
            if self.training:
                raise H.HipKernelError("training runs on an MI355X (HIP) device only; the CPU device does inference")
            from .._cpu import HostForward
            s = self._settings
            host = HostForward(s.layer_sizes, s.n_neighbors, s.decimation,
                               {k: v.detach() for k, v in self.named_parameters()}, dict(self.named_buffers()))
            return host(input.to("cpu", torch.float32), np.random.permutation(N))
        inp = input.to(self._device, torch.float32).contiguous()
        perm = torch.from_numpy(np.random.permutation(N)).to(self._device)
        p_drop = float(self.fc_end[2].p)
        with torch.cuda.device(self._device):       # launches go to the CURRENT device's stream
            if self.training and torch.is_grad_enabled():
                names = [n for n, _ in self.named_parameters()]
                return _NetFunction.apply(self, inp, perm, p_drop, names, *self.parameters())
            logits, _ = self.engine().forward(inp, perm, self.training, p_drop)
        return logits


class _NetFunction(torch.autograd.Function):
    """One autograd node for the whole network: forward and backward are HIP launch schedules."""

    @staticmethod
    def forward(fctx, module: RandLANet, inp, perm, p_drop, names, *params):
        eng = module.engine()
        logits, ectx = eng.forward(inp, perm, True, p_drop)
        fctx.eng, fctx.ectx, fctx.names = eng, ectx, names
        fctx.shapes = [(p.shape, p.device) for p in params]
        return logits

    @staticmethod
    def backward(fctx, dlogits):
        grads: Dict[str, torch.Tensor] = {
            n: torch.empty(s, dtype=torch.float32, device=d) for n, (s, d) in zip(fctx.names, fctx.shapes)}
        with torch.cuda.device(dlogits.device):
            fctx.eng.backward(fctx.ectx, dlogits, grads)
        return (None, None, None, None, None) + tuple(grads[n] for n in fctx.names)
