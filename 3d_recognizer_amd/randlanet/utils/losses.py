"""Training losses of the reference (randlanet/utils/losses.py:7-87, trainer.py:244-269) on the
fused HIP loss kernel (rl_loss_forward / rl_loss_backward): one pass over the (B,C,N) logits
for softmax + the class sums, a second pass for the gradient.  Same class names, constructor
arguments and values as the reference; there is no PyTorch implementation behind them.
"""
import torch

from .. import _hip as H
from .. import _ops as ops

eps = 1e-7  # reference losses.py:4 (compiled into the kernel as LS_EPS)


class _HipLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, kind, alpha, gamma, neglect_background):
        if not logits.is_cuda:
            raise H.HipKernelError("losses run on the MI355X only: there is no CPU path in this build")
        lg = logits.detach().to(torch.float32).contiguous()
        lb = labels.to(lg.device, torch.int64).contiguous()
        with torch.cuda.device(lg.device):
            out, work = ops.loss_forward(lg, lb, kind, alpha, gamma, neglect_background)
        ctx.save_for_backward(lg, lb, work)
        ctx.cfg = (kind, alpha, gamma, neglect_background)
        return out[0].to(torch.float32)

    @staticmethod
    def backward(ctx, grad_out):
        lg, lb, work = ctx.saved_tensors
        kind, alpha, gamma, neglect = ctx.cfg
        with torch.cuda.device(lg.device):
            dlogits = ops.loss_backward(lg, lb, kind, alpha, gamma, neglect, work)
        return dlogits * grad_out, None, None, None, None, None


class FocalTverskyLoss(torch.nn.Module):
    """Dice (alpha .5, gamma 1), Tversky (gamma 1) and focal Tversky loss (losses.py:37-87)."""

    def __init__(self, alpha: float = 0.7, gamma: float = 4.0 / 3.0, neglect_background: bool = True):
        super().__init__()
        self._alpha, self._gamma, self._neglect_background = alpha, gamma, neglect_background

    def forward(self, logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        return _HipLoss.apply(logits, labels, 2, float(self._alpha), float(self._gamma),
                              bool(self._neglect_background))


class FocalLoss(torch.nn.Module):
    """Focal loss (losses.py:7-34)."""

    def __init__(self, gamma: float = 2):
        super().__init__()
        self._gamma = gamma

    def forward(self, logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        return _HipLoss.apply(logits, labels, 1, 0.0, float(self._gamma), False)


class CrossEntropyLoss(torch.nn.Module):
    """Mean cross entropy over all points - what torch.nn.CrossEntropyLoss() computes for the
    reference's "cross_entropy" choice (trainer.py:251-252)."""

    def forward(self, logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        return _HipLoss.apply(logits, labels, 0, 0.0, 0.0, False)


def get_loss(loss_function: str) -> torch.nn.Module:
    """Name -> loss module with the reference's standard parameters (trainer.py:244-269)."""
    if loss_function == "cross_entropy":
        return CrossEntropyLoss()
    if loss_function == "focal":
        return FocalLoss(gamma=2)
    if loss_function == "dice":
        return FocalTverskyLoss(alpha=0.5, gamma=1.0, neglect_background=True)
    if loss_function == "tversky":
        return FocalTverskyLoss(alpha=0.7, gamma=1.0, neglect_background=True)
    if loss_function == "focal_tversky":
        return FocalTverskyLoss(alpha=0.7, gamma=(4.0 / 3.0), neglect_background=True)
    raise ValueError(f"Loss function {loss_function} not known!")
