"""Oracle: peak-mask precision / recall / F1 (reference row a13 of SURVEY.md §8a).

testing/metrics.py:10-192 walks torch.nonzero(mask) in a Python loop and, for
each peak (b, f, t), multiplies a clipped 3x3 neighbourhood of the other mask
with a kernel whose only non-zero tap is the centre.  Because the kernel is
sliced ``[:2]`` on the low borders too, the tap lands on (f + [f == 0],
t + [t == 0]): peaks in row 0 / column 0 are compared one cell further in.
High borders are exact.  That quirk is reproduced here (axes of length >= 2).
"""
from __future__ import annotations

import math

import numpy as np


def _hits(walk: np.ndarray, other: np.ndarray):
    b, f, t = np.nonzero(walk)
    if len(b) == 0:
        return 0.0, 0
    ff = f + (f == 0)
    tt = t + (t == 0)
    return float(np.sum(other[b, ff, tt], dtype=np.float64)), len(b)


def precision(predicted: np.ndarray, gt: np.ndarray) -> float:
    """testing/metrics.py:88-163: mean over predicted peaks of the ground truth at the tap."""
    s, n = _hits(np.asarray(predicted), np.asarray(gt))
    return 0.0 if n == 0 else s / n


def recall(predicted: np.ndarray, gt: np.ndarray) -> float:
    """testing/metrics.py:10-85: mean over ground-truth peaks of the prediction at the tap."""
    s, n = _hits(np.asarray(gt), np.asarray(predicted))
    return 0.0 if n == 0 else s / n


def f1score(predicted: np.ndarray, gt: np.ndarray) -> float:
    """testing/metrics.py:166-192."""
    p = precision(predicted, gt)
    r = recall(predicted, gt)
    if math.isclose(p + r, 0.0):
        return 0.0
    return float(2.0 * (p * r) / (p + r))


def counts(predicted: np.ndarray, gt: np.ndarray) -> np.ndarray:
    """Per-clip integer counts [hit_p, n_p, hit_r, n_r] for 0/1 masks (what the device kernel emits)."""
    predicted = np.asarray(predicted)
    gt = np.asarray(gt)
    out = np.zeros((predicted.shape[0], 4), dtype=np.int64)
    for b in range(predicted.shape[0]):
        hp, npk = _hits(predicted[b : b + 1] != 0, (gt[b : b + 1] != 0).astype(np.int64))
        hr, nr = _hits(gt[b : b + 1] != 0, (predicted[b : b + 1] != 0).astype(np.int64))
        out[b] = (int(hp), npk, int(hr), nr)
    return out
