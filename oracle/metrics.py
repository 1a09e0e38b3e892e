"""Oracle: micro-averaged multilabel average precision (test infrastructure only).

Follows the estimator of torchmetrics 1.7.1 ``MultilabelAveragePrecision(average="micro")``
(requirements.txt:60; call sites TFAM/train_and_eval.py:49,87,94,122-124): per ``update`` call the
scores get a sigmoid iff any of that batch's values lies outside [0,1]; at ``compute`` everything is
flattened over labels and AP = sum_k (R_k - R_{k-1}) P_k over distinct thresholds.  torchmetrics is
not installed offline, so this restatement is pinned against sklearn's average_precision_score (the
same estimator) in make_golden.py.
"""
from __future__ import annotations

import numpy as np


def maybe_sigmoid(batch_scores: np.ndarray) -> np.ndarray:
    s = np.asarray(batch_scores, dtype=np.float32)
    if ((s < 0) | (s > 1)).any():
        s = (1.0 / (1.0 + np.exp(-s.astype(np.float32)))).astype(np.float32)
    return s


def micro_average_precision(scores: np.ndarray, targets: np.ndarray) -> float:
    """scores [N,C] already in [0,1]; targets [N,C] in {0,1}.  O(n log n), float64 accumulation."""
    s = np.asarray(scores, dtype=np.float32).ravel()
    y = np.asarray(targets).ravel().astype(np.int64)
    order = np.argsort(-s, kind="stable")
    s, y = s[order], y[order]
    tp = np.cumsum(y)
    fp = np.cumsum(1 - y)
    last = np.r_[s[1:] != s[:-1], True]            # last element of each distinct-score run
    tp, fp = tp[last].astype(np.float64), fp[last].astype(np.float64)
    npos = float(y.sum())
    if npos == 0:
        return float("nan")
    precision = tp / (tp + fp)
    recall = tp / npos
    prev = np.r_[0.0, recall[:-1]]
    return float(np.sum((recall - prev) * precision))
