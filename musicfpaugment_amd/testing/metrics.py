"""Peak-mask Precision / Recall / F1 on MI355X -- mirror of the reference's testing/metrics.py:10-192.

Same classes, same ``forward(predicted, gt, device) -> float``.  The reference walks
torch.nonzero(mask) in a Python loop with three ``.item()`` syncs per peak; here one kernel
(csrc/metrics.hip) sweeps both masks and returns four integers per clip, so there is a single
device->host copy per call.  The reference's low-border quirk (a peak in row/column 0 is matched
one cell further in, metrics.py:44-83) is reproduced.  Masks are treated as binary (non-zero =
peak), which is what every caller passes (testing/audfprint_exps.py:119-121).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .. import ops


def _counts(predicted: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    if predicted.shape != gt.shape or predicted.dim() != 3:
        raise ValueError("masks must both be (B, N1, N2)")
    dev = predicted.device if predicted.is_cuda else (gt.device if gt.is_cuda else torch.device("cuda"))
    p = (predicted.to(dev) != 0).to(torch.uint8)
    g = (gt.to(dev) != 0).to(torch.uint8)
    return ops.peak_metrics_counts(p, g).sum(dim=0).cpu()     # [hits_p, n_p, hits_r, n_r]


class Precision(nn.Module):
    def forward(self, predicted: torch.Tensor, gt: torch.Tensor, device: str = "cpu") -> float:
        c = _counts(predicted, gt)
        return 0.0 if int(c[1]) == 0 else float(int(c[0])) / int(c[1])


class Recall(nn.Module):
    def forward(self, predicted: torch.Tensor, gt: torch.Tensor, device: str = "cpu") -> float:
        c = _counts(predicted, gt)
        return 0.0 if int(c[3]) == 0 else float(int(c[2])) / int(c[3])


class F1score(nn.Module):
    def __init__(self) -> None:
        super().__init__()
        self.prec = Precision()
        self.rec = Recall()

    def forward(self, predicted: torch.Tensor, gt: torch.Tensor, device: str = "cpu") -> float:
        c = _counts(predicted, gt)
        p = 0.0 if int(c[1]) == 0 else float(int(c[0])) / int(c[1])
        r = 0.0 if int(c[3]) == 0 else float(int(c[2])) / int(c[3])
        if math.isclose(p + r, 0.0):
            return 0.0
        return float(2.0 * (p * r) / (p + r))
