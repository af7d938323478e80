"""RandLA-Net for MI355X behind the reference's module surface.

Mirrors randlanet/utils/modules.py of matthiasverstraete/3d_recognizer: the same settings
dataclass, class names, constructor signatures, parameter names / shapes (state_dict is
interchangeable both ways) and forward contract - but the sub-modules are parameter containers
only.  All arithmetic is done by the HIP kernels of librandla_hip.so, scheduled by
`_engine.Engine`; there is no PyTorch or CPU implementation of the forward pass in this
package, and a missing library or a non-GPU device raises instead of falling back.
"""
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


def _no_forward(self, *a, **k):
    raise H.HipKernelError(
        f"{type(self).__name__} holds parameters only; RandLANet.forward runs the fused HIP schedule")


class SharedMLP(nn.Module):
    """Parameters of one shared MLP (reference modules.py:60-91): 1x1 conv or transposed conv,
    optional BatchNorm2d(eps=1e-6, momentum=0.99), optional activation (kept for printing)."""

    def __init__(self, n_in: int, n_out: int, transpose: bool = False, bn: bool = True,
                 activation: Optional[nn.Module] = None):
        super().__init__()
        conv = nn.ConvTranspose2d if transpose else nn.Conv2d
        self.conv = conv(n_in, n_out, kernel_size=1, stride=1, padding_mode="zeros")
        self.batch_norm = nn.BatchNorm2d(n_out, eps=1e-6, momentum=0.99) if bn else None
        self.activation = activation

    forward = _no_forward


class AttentivePooling(nn.Module):
    """Parameters of attentive pooling (reference modules.py:227-237)."""

    def __init__(self, n_in: int, n_out: int):
        super().__init__()
        self.score_fn = nn.Sequential(nn.Linear(n_in, n_in, bias=False), nn.Softmax(dim=-2))
        self.mlp = SharedMLP(n_in, n_out, activation=nn.ReLU())

    forward = _no_forward


class LocalFeatureAggregation(nn.Module):
    """Parameters of one encoder block, registered in the reference's order (modules.py:275-296)
    so that state_dict() enumerates identically."""

    def __init__(self, n_in: int, n_out: int, n_neighbors: int, device: torch.device):
        super().__init__()
        self._n_neighbors = n_neighbors
        self._device = device
        self.mlp1 = SharedMLP(n_in, n_out // 2, activation=nn.LeakyReLU(0.2))
        self.mlp2 = SharedMLP(n_out, 2 * n_out)
        self.shortcut = SharedMLP(n_in, 2 * n_out)
        self.mlp_rpe1 = SharedMLP(10, n_out // 2, activation=nn.ReLU())
        self.mlp_rpe2 = SharedMLP(n_out // 2, n_out // 2, activation=nn.ReLU())
        self.pool1 = AttentivePooling(n_out, n_out // 2)
        self.pool2 = AttentivePooling(n_out, n_out)
        self.lrelu = nn.LeakyReLU()

    forward = _no_forward


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
