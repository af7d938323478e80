"""Segmentation metrics of the reference (randlanet/utils/metrics.py:8-255).

`accuracy` and `iou` are ratios of four per-class counts that the fused HIP loss kernel
produces in one pass (rl_loss_forward), so a call costs one launch and ONE device->host copy
instead of the reference's 2C+2 `.item()` round trips.  The collectors are host bookkeeping and
keep the reference's keys ("loss", "OA", "mAcc", "mIoU", "<class> IoU") and averaging order
(per batch, then per evaluation pass).
"""
from collections import OrderedDict
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import _hip as H
from .. import _ops as ops


def class_counts(logits: torch.Tensor, labels: torch.Tensor) -> np.ndarray:
    """(3, C) float64: [#(pred==c & label==c), #(label==c), #(pred==c)] for logits (B?,C,N)."""
    if logits.dim() == 2:
        logits, labels = logits.unsqueeze(0), labels.unsqueeze(0)
    if not logits.is_cuda:
        if not torch.cuda.is_available():
            raise H.HipKernelError("metrics run on the MI355X only: there is no CPU path in this build")
        logits = logits.cuda()
    lg = logits.detach().to(torch.float32).contiguous()
    lb = labels.to(lg.device, torch.int64).contiguous()
    with torch.cuda.device(lg.device):
        out, _ = ops.loss_forward(lg, lb, 0, 0.0, 0.0, False)
    C = lg.shape[1]
    return out[1:1 + 3 * C].cpu().numpy().reshape(3, C)


def accuracy_from_counts(cnt: np.ndarray) -> Tuple[float, List[float]]:
    inter, lab = cnt[0], cnt[1]
    # overall accuracy in fp32 like the reference's accuracy_mask.float().mean() (metrics.py:21)
    overall = float(np.float32(inter.sum()) / np.float32(lab.sum()))
    per_class = [1.0 if lab[c] == 0 else float(np.float32(inter[c]) / np.float32(lab[c]))
                 for c in range(cnt.shape[1])]       # absent class -> 1.0 (metrics.py:27-28)
    return overall, per_class


def iou_from_counts(cnt: np.ndarray) -> Tuple[float, List[float]]:
    inter, lab, pred = cnt
    per_class = []
    for c in range(cnt.shape[1]):
        union = lab[c] + pred[c] - inter[c]
        per_class.append(1.0 if union == 0 else float(np.float32(inter[c]) / np.float32(union)))
    return float(np.nanmean(per_class)), per_class   # empty union -> 1.0 (metrics.py:53-54)


def accuracy(logits: torch.Tensor, labels: torch.Tensor) -> Tuple[float, List[float]]:
    """Overall accuracy and per-class accuracies (reference metrics.py:8-32)."""
    return accuracy_from_counts(class_counts(logits, labels))


def iou(logits: torch.Tensor, labels: torch.Tensor) -> Tuple[float, List[float]]:
    """Mean IoU and per-class IoUs (reference metrics.py:35-59)."""
    return iou_from_counts(class_counts(logits, labels))


def _summary(prefix: str, loss, oa, macc, miou, class_ious, class_names) -> OrderedDict:
    d = OrderedDict([(f"{prefix}loss", loss), (f"{prefix}OA", oa), (f"{prefix}mAcc", macc),
                     (f"{prefix}mIoU", miou)])
    for c, v in enumerate(class_ious):
        name = (prefix + class_names[c]) if class_names else f"class {c}"
        d[name + " IoU"] = v
    return d


class MetricCollector:
    """Per-batch metrics of one pass over a dataset, averaged on request (metrics.py:62-156)."""

    def __init__(self, class_names: Optional[List[str]] = None):
        self._class_names = class_names
        self.reset()

    def reset(self) -> None:
        self._losses: List[float] = []
        self._oas: List[float] = []
        self._accs: List[np.ndarray] = []
        self._mious: List[float] = []
        self._ious: List[np.ndarray] = []

    def push(self, loss: float, overall_accuracy: float, per_class_accuracies: Sequence[float], miou: float,
             per_class_ious: Sequence[float]) -> None:
        self._losses.append(loss)
        self._oas.append(overall_accuracy)
        self._accs.append(np.asarray(per_class_accuracies))
        self._mious.append(miou)
        self._ious.append(np.asarray(per_class_ious))

    loss = property(lambda self: float(np.mean(self._losses)))
    overall_accuracy = property(lambda self: float(np.nanmean(self._oas)))
    per_class_accuracies = property(lambda self: list(np.nanmean(self._accs, axis=0)))
    mean_class_accuracy = property(lambda self: float(np.mean(self.per_class_accuracies)))
    miou = property(lambda self: float(np.nanmean(self._mious)))
    per_class_ious = property(lambda self: list(np.nanmean(self._ious, axis=0)))

    def as_dict(self, tag: str = "") -> OrderedDict:
        return _summary("" if tag == "" else f"{tag}_", self.loss, self.overall_accuracy,
                        self.mean_class_accuracy, self.miou, self.per_class_ious, self._class_names)


class MetricCollectorBag:
    """Mean and standard deviation over several seeded evaluation passes (metrics.py:159-255)."""

    def __init__(self, metric_collectors: List[MetricCollector], class_names: Optional[List[str]] = None):
        self._class_names = class_names
        self._mcs = metric_collectors

    @staticmethod
    def _ms(values) -> Tuple[float, float]:
        return np.mean(values), np.std(values)

    def _per_class(self, attr: str) -> List[Tuple[float, float]]:
        rows = [getattr(mc, attr) for mc in self._mcs]
        if not rows:
            return []
        return [self._ms([r[c] for r in rows]) for c in range(len(rows[0]))]

    loss = property(lambda self: self._ms([mc.loss for mc in self._mcs]))
    overall_accuracy = property(lambda self: self._ms([mc.overall_accuracy for mc in self._mcs]))
    mean_class_accuracy = property(lambda self: self._ms([mc.mean_class_accuracy for mc in self._mcs]))
    per_class_accuracies = property(lambda self: self._per_class("per_class_accuracies"))
    miou = property(lambda self: self._ms([mc.miou for mc in self._mcs]))
    per_class_ious = property(lambda self: self._per_class("per_class_ious"))

    def as_dict(self, tag: str = "", include_stdev: bool = False) -> OrderedDict:
        d = _summary("" if tag == "" else f"{tag}_", self.loss, self.overall_accuracy,
                     self.mean_class_accuracy, self.miou, self.per_class_ious, self._class_names)
        if include_stdev:
            return d
        return OrderedDict((k, v[0]) for k, v in d.items())
