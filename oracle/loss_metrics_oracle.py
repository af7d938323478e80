"""ORACLE - TEST INFRASTRUCTURE ONLY.

numpy/torch-CPU restatement of the per-step loss and metric reductions of the reference:
  FocalTverskyLoss.forward  /root/reference/randlanet/utils/losses.py:59-87
  FocalLoss.forward         losses.py:17-34
  Trainer._get_loss         trainer.py:244-269   (name -> (alpha, gamma))
  accuracy / iou            metrics.py:8-59
Pinned by tests/golden/loss_metrics.npz (outputs of the reference's own classes).
"""
import numpy as np
import torch
import torch.nn.functional as F

EPS = 1e-7  # losses.py:4

LOSS_PARAMS = {  # trainer.py:253-267
    "dice": (0.5, 1.0),
    "tversky": (0.7, 1.0),
    "focal_tversky": (0.7, 4.0 / 3.0),
}


def tversky_loss(logits: torch.Tensor, labels: torch.Tensor, alpha: float, gamma: float,
                 neglect_background: bool = True) -> torch.Tensor:
    """losses.py:66-87; logits (B,C,N), labels (B,N) int64."""
    B, C, N = logits.shape
    p = F.softmax(logits, dim=1).permute(1, 0, 2).reshape(C, -1)
    y = F.one_hot(labels, C).to(p.dtype).permute(2, 0, 1).reshape(C, -1)
    if neglect_background:
        p, y = p[1:], y[1:]
    tp = (y * p).sum(1)
    fn = (y * (1 - p)).sum(1)
    fp = ((1 - y) * p).sum(1)
    ti = (tp + EPS) / (tp + alpha * fn + (1 - alpha) * fp + EPS)
    return ((1 - ti) ** gamma).mean()


def focal_loss(logits: torch.Tensor, labels: torch.Tensor, gamma: float = 2.0) -> torch.Tensor:
    """losses.py:24-34."""
    B, C, N = logits.shape
    y = F.one_hot(labels, C).to(logits.dtype).permute(0, 2, 1).clamp(EPS, 1.0 - EPS)
    p = F.softmax(logits, dim=1).clamp(EPS, 1.0 - EPS)
    return (-y * torch.log(p) * (1 - p) ** gamma).sum() / (B * N)


def loss_by_name(name: str, logits, labels):
    if name == "cross_entropy":
        return F.cross_entropy(logits, labels)
    if name == "focal":
        return focal_loss(logits, labels, 2.0)
    if name in LOSS_PARAMS:
        a, g = LOSS_PARAMS[name]
        return tversky_loss(logits, labels, a, g, True)
    raise ValueError(f"Loss function {name} not known!")


def accuracy(logits: np.ndarray, labels: np.ndarray):
    """metrics.py:18-32: OA and per-class accuracy; class absent from labels -> 1.0 when no
    point was (vacuously) 'correct', which is always, so 1.0 (metrics.py:27-28)."""
    C = logits.shape[-2]
    pred = np.argmax(logits, axis=-2)
    ok = pred == labels
    oa = float(ok.astype(np.float32).mean())
    per = []
    for c in range(C):
        m = labels == c
        n = float(m.sum())
        per.append(1.0 if n == 0 else float((ok & m).sum()) / n)
    return oa, per


def iou(logits: np.ndarray, labels: np.ndarray):
    """metrics.py:45-59: per-class IoU, union 0 -> 1.0; mIoU = nanmean."""
    C = logits.shape[-2]
    pred = np.argmax(logits, axis=-2)
    per = []
    for c in range(C):
        lm, pm = labels == c, pred == c
        union = float((lm | pm).sum())
        per.append(1.0 if union == 0 else float((lm & pm).sum()) / union)
    return float(np.nanmean(per)), per
