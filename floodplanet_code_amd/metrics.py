"""Segmentation metrics of the reference plugin (water_seg_model.py:46-63): a torchmetrics
``MetricCollection[F1Score, JaccardIndex, Accuracy]`` with task="multiclass", average="micro" and
``ignore_index``, cloned with ``train_`` / ``val_`` / ``test_`` prefixes.

torchmetrics is third-party and absent here, so this is a restatement of its documented reductions from a
confusion matrix (PARITY UNPINNED -- no in-repo test or fixture of the reference pins these numbers):
  * pixels whose target == ignore_index are dropped,
  * F1 / Accuracy (stat-scores based, micro):  tp = #correct, fp = fn = N - tp,
  * Jaccard (confusion-matrix based, micro): sum(diag) / (sum(union) - union[ignore_index])  when
    0 <= ignore_index < n_classes  (torchmetrics' _jaccard_index_reduce).
The HIP loss kernel emits the confusion counts fused with the cross entropy; ``update_from_counts`` feeds
them in without touching the logits again.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch


class SegmentationMetrics:
    KEYS = ("MulticlassF1Score", "MulticlassJaccardIndex", "MulticlassAccuracy")

    def __init__(self, num_classes: int, ignore_index: Optional[int] = None, prefix: str = ""):
        self.num_classes = num_classes
        self.ignore_index = ignore_index
        self.prefix = prefix
        self._total = torch.zeros(num_classes, num_classes, dtype=torch.int64)

    def clone(self, prefix: str = "") -> "SegmentationMetrics":
        return SegmentationMetrics(self.num_classes, self.ignore_index, prefix)

    # -- confusion matrix ---------------------------------------------------------------------------
    def _counts(self, pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        n = self.num_classes
        pred = pred.reshape(-1).long()
        target = target.reshape(-1).long()
        keep = (target >= 0) & (target < n)
        if self.ignore_index is not None:
            keep &= target != self.ignore_index
        idx = target[keep] * n + pred[keep].clamp(0, n - 1)
        return torch.bincount(idx, minlength=n * n).view(n, n).to("cpu")

    def _reduce(self, m: torch.Tensor) -> Dict[str, torch.Tensor]:
        m = m.double()
        tp = m.diag().sum()
        tot = m.sum()
        fp = fn = tot - tp
        zero = torch.zeros((), dtype=torch.float64)
        f1 = 2 * tp / (2 * tp + fp + fn) if tot > 0 else zero
        acc = tp / tot if tot > 0 else zero
        union = m.sum(0) + m.sum(1) - m.diag()
        denom = union.sum()
        if self.ignore_index is not None and 0 <= self.ignore_index < self.num_classes:
            denom = denom - union[self.ignore_index]
        jac = tp / denom if denom > 0 else zero
        p = self.prefix
        return {f"{p}MulticlassF1Score": f1.float(), f"{p}MulticlassJaccardIndex": jac.float(),
                f"{p}MulticlassAccuracy": acc.float()}

    # -- torchmetrics-like protocol -----------------------------------------------------------------
    def update(self, pred: torch.Tensor, target: torch.Tensor) -> None:
        self._total += self._counts(pred, target)

    def update_from_counts(self, counts: torch.Tensor) -> Dict[str, torch.Tensor]:
        c = counts.detach().to("cpu").view(self.num_classes, self.num_classes)
        self._total += c
        return self._reduce(c)

    def __call__(self, pred: torch.Tensor, target: torch.Tensor) -> Dict[str, torch.Tensor]:
        c = self._counts(pred, target)
        self._total += c
        return self._reduce(c)

    def compute(self) -> Dict[str, torch.Tensor]:
        return self._reduce(self._total)

    def reset(self) -> None:
        self._total.zero_()

    def confusion(self) -> torch.Tensor:
        return self._total.clone()
