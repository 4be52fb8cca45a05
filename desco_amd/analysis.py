"""Accuracy metrics of the reference (subgraph_counting/analysis.py:22-83): the parity metric the
README's accuracy table is quoted in.  Pinned by tests/golden/metrics.json."""
from __future__ import annotations

import numpy as np


def _groups(pred, groupby):
    return [list(range(pred.shape[1]))] if groupby is None else groupby


def norm_mse(pred, truth, groupby=None):
    """MSE / Var(truth) per query group, float64 (analysis.py:22-43)."""
    pred, truth = np.asarray(pred, dtype=np.float64), np.asarray(truth, dtype=np.float64)
    return [float(np.mean((pred[:, g] - truth[:, g]) ** 2) / np.var(truth[:, g]))
            for g in _groups(pred, groupby)]


def mse(pred, truth, groupby=None):
    pred, truth = np.asarray(pred, dtype=np.float64), np.asarray(truth, dtype=np.float64)
    return [float(np.mean((pred[:, g] - truth[:, g]) ** 2)) for g in _groups(pred, groupby)]


def mae(pred, truth, groupby):
    pred, truth = np.asarray(pred), np.asarray(truth)
    return [float(np.mean(np.abs(pred[:, g] - truth[:, g]))) for g in groupby]
