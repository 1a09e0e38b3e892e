"""Micro-averaged multilabel average precision with torchmetrics' semantics (K16, a14).

Mirrors ``MultilabelAveragePrecision(num_labels=C, average="micro")`` as used by
TFAM/train_and_eval.py:49,87,94,122-124: ``update(preds, target)`` applies a sigmoid to that batch iff any
value lies outside [0, 1]; ``compute()`` flattens everything over labels and integrates the distinct-threshold
precision-recall curve, AP = sum_k (R_k - R_{k-1}) P_k.  Scores are kept on the device they arrive on and
sorted there (torch.sort -> rocPRIM radix sort, a library primitive); under data
parallelism the per-rank score/target rows are all-gathered first (parallel.all_gather_rows).
"""
from __future__ import annotations

import torch

from . import parallel


class MultilabelAveragePrecision:
    def __init__(self, num_labels: int, average: str = "micro"):
        if average != "micro":
            raise NotImplementedError("only average='micro' is used by the reference path")
        self.num_labels = num_labels
        self.reset()

    def to(self, device):
        return self

    def reset(self):
        self._scores, self._targets = [], []

    def update(self, preds: torch.Tensor, target: torch.Tensor):
        p = preds.detach().float()
        if bool(((p < 0) | (p > 1)).any()):
            p = torch.sigmoid(p)
        self._scores.append(p.reshape(-1, self.num_labels))
        self._targets.append(target.detach().reshape(-1, self.num_labels).to(torch.int64))

    def compute(self, distributed: bool = False) -> torch.Tensor:
        if not self._scores:
            return torch.tensor(float("nan"))
        s = torch.cat(self._scores, dim=0)
        y = torch.cat(self._targets, dim=0)
        if distributed:
            s, y = parallel.all_gather_rows(s), parallel.all_gather_rows(y)
        return micro_average_precision(s, y)

    __call__ = update


class Accuracy:
    """Top-1 accuracy for the single-label MammalNet variants (TFAM/train_and_eval_frame_diff_MN.py:49,87,94:
    ``Accuracy(num_classes=C)``; ``update(logits [N,C], labels.int() [N,C])``).  The reference hands the metric one-hot
    integer rows; torchmetrics is not installed here and its legacy ``Accuracy(num_classes=...)`` call form no longer
    exists in the version SURVEY.md pins (1.7.1 requires ``task=``), so the intended statistic — fraction of rows whose
    arg-max logit is the labelled class — is what is computed ("parity unpinned")."""

    def __init__(self, num_classes: int, **unused):
        self.num_classes = num_classes
        self.reset()

    def to(self, device):
        return self

    def reset(self):
        self._correct = self._total = None

    def update(self, preds: torch.Tensor, target: torch.Tensor):
        p = preds.detach().reshape(-1, self.num_classes).argmax(dim=1)
        t = target.detach()
        t = t.reshape(-1, self.num_classes).argmax(dim=1) if t.dim() >= 2 or t.numel() != p.numel() else t.reshape(-1).to(torch.int64)
        c = (p == t).sum().to(torch.float64).reshape(1)
        n = torch.tensor([float(p.numel())], dtype=torch.float64, device=c.device)
        self._correct = c if self._correct is None else self._correct + c
        self._total = n if self._total is None else self._total + n

    def compute(self, distributed: bool = False) -> torch.Tensor:
        if self._total is None:
            return torch.tensor(float("nan"))
        c, n = self._correct, self._total
        if distributed:
            c, n = parallel.all_gather_rows(c.reshape(1, 1)).sum().reshape(1), parallel.all_gather_rows(n.reshape(1, 1)).sum().reshape(1)
        return (c / n).to(torch.float32).reshape(())

    __call__ = update


def micro_average_precision(scores: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
    """scores in [0,1] [N,C], targets {0,1} [N,C] -> scalar AP (float64 accumulation)."""
    s = scores.reshape(-1)
    y = targets.reshape(-1).to(torch.float64)
    s, order = torch.sort(s, descending=True, stable=True)
    y = y[order]
    tp = torch.cumsum(y, 0)
    fp = torch.cumsum(1.0 - y, 0)
    last = torch.ones_like(s, dtype=torch.bool)
    last[:-1] = s[1:] != s[:-1]                      # last element of every run of equal scores
    tp, fp = tp[last], fp[last]
    npos = y.sum()
    precision = tp / (tp + fp)
    recall = tp / npos
    prev = torch.cat([torch.zeros(1, dtype=recall.dtype, device=recall.device), recall[:-1]])
    return ((recall - prev) * precision).sum().to(torch.float32)
