"""Dejavu peak picking on MI355X -- mirror of afp/dejavu/fingerprint.py:34-171 (up to the peak list).

``get_2D_peaks(arr2D, plot=False, amp_min=50) -> (peak_coordinates, peak_mask)`` keeps the reference
contract (coordinates [(freq, time)] in row-major order, mask float64).  ``fingerprint_peaks_batch``
is the batched device path: mlab.specgram-style PSD -> /max -> 10 ln -> -mean -> 21x21 local maxima.
Hash generation (generate_hashes, :174-213) is a next-tier row.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np
import torch

from ... import ops
from ...constants import afp_settings

PEAK_NEIGHBORHOOD_SIZE = 10  # afp/dejavu/variables.py:19
CONNECTIVITY_MASK = 2        # variables.py:18: full square footprint


def get_2D_peaks(arr2D, plot: bool = False, amp_min: int = afp_settings["dejavu"]["amp_min"],
                 device="cuda") -> Tuple[List[Tuple[int, int]], np.ndarray]:
    if plot:
        raise NotImplementedError("plotting is outside the hot path")
    a = torch.as_tensor(np.asarray(arr2D) if not isinstance(arr2D, torch.Tensor) else arr2D)
    a = a.to(device, torch.float64).reshape(1, *a.shape[-2:])
    mask, _ = ops.localmax2d(a, PEAK_NEIGHBORHOOD_SIZE, float(amp_min))
    m = mask[0]
    freqs, times = torch.nonzero(m, as_tuple=True)
    return list(zip(freqs.tolist(), times.tolist())), m.to(torch.float64).cpu().numpy()


def fingerprint_peaks_batch(wav: torch.Tensor, amp_min: float = afp_settings["dejavu"]["amp_min"],
                            scale_in: float = 32767.0):
    """(B, T) float32 on the GPU -> (mask (B,257,nF) uint8, npeaks (B,), specgram (B,257,nF) float64)."""
    psd, cmax = ops.specgram_psd(wav, scale_in=scale_in)
    arr = ops.dejavu_prepare(psd, cmax, 10.0, mean_order=1)
    mask, npeaks = ops.localmax2d(arr, PEAK_NEIGHBORHOOD_SIZE, float(amp_min))
    return mask, npeaks, ops.normalize_(psd, cmax, per_clip=True)
