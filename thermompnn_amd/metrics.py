"""Regression / rank metrics the reference's benchmarking script reports with torchmetrics
(/root/reference/analysis/thermompnn_benchmarking.py:68-75: R2Score, MSE, RMSE, SpearmanCorrCoef, PearsonCorrCoef),
restated with numpy so that predicted-vs-measured ddG tables can be scored without torchmetrics.
Host-side bookkeeping: not part of the device path."""
from __future__ import annotations

from typing import Dict

import numpy as np


def _rank_average(x: np.ndarray) -> np.ndarray:
    """1-based ranks with ties sharing their average rank (what Spearman needs)."""
    order = np.argsort(x, kind="mergesort")
    ranks = np.empty(len(x), dtype=np.float64)
    sx = x[order]
    i = 0
    while i < len(x):
        j = i
        while j + 1 < len(x) and sx[j + 1] == sx[i]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    return ranks


def pearson(pred, target) -> float:
    p, t = np.asarray(pred, np.float64), np.asarray(target, np.float64)
    pc, tc = p - p.mean(), t - t.mean()
    den = np.sqrt((pc * pc).sum() * (tc * tc).sum())
    return float((pc * tc).sum() / den) if den > 0 else float("nan")


def spearman(pred, target) -> float:
    return pearson(_rank_average(np.asarray(pred, np.float64)), _rank_average(np.asarray(target, np.float64)))


def get_metrics(pred, target) -> Dict[str, float]:
    """-> {'r2', 'mse', 'rmse', 'spearman', 'pearson'} for 1-D arrays of predictions and measurements
    (NaN targets are dropped, as the reference's per-dataset loops skip missing ddG values)."""
    p, t = np.asarray(pred, np.float64).ravel(), np.asarray(target, np.float64).ravel()
    ok = np.isfinite(p) & np.isfinite(t)
    p, t = p[ok], t[ok]
    mse = float(np.mean((p - t) ** 2)) if len(p) else float("nan")
    ss_tot = float(((t - t.mean()) ** 2).sum()) if len(p) else 0.0
    r2 = 1.0 - float(((p - t) ** 2).sum()) / ss_tot if ss_tot > 0 else float("nan")
    return {"r2": r2, "mse": mse, "rmse": float(np.sqrt(mse)), "spearman": spearman(p, t) if len(p) > 1 else float("nan"),
            "pearson": pearson(p, t) if len(p) > 1 else float("nan"), "n": int(len(p))}
