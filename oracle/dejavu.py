"""Oracle: Dejavu 2-D local-maximum peak picker (reference rows a9, a10 of SURVEY.md §8a)."""
from __future__ import annotations

import numpy as np

from . import stft as ostft

# afp/dejavu/variables.py:18-19, testing/parameters.py:27-34
PEAK_NEIGHBORHOOD_SIZE = 10
AMP_MIN = 50


def _reflect_index(i: np.ndarray, n: int) -> np.ndarray:
    """scipy.ndimage mode='reflect' (half-sample symmetric: d c b a | a b c d | d c b a)."""
    period = 2 * n
    i = np.mod(i, period)
    return np.where(i >= n, period - 1 - i, i)


def _window_reduce(a: np.ndarray, radius: int, axis: int, reduce_fn, border: str, border_value=None) -> np.ndarray:
    n = a.shape[axis]
    out = None
    for d in range(-radius, radius + 1):
        idx = np.arange(n) + d
        if border == "reflect":
            shifted = np.take(a, _reflect_index(idx, n), axis=axis)
        else:
            inside = (idx >= 0) & (idx < n)
            shifted = np.take(a, np.clip(idx, 0, n - 1), axis=axis)
            shape = [1, 1]
            shape[axis] = n
            shifted = np.where(inside.reshape(shape), shifted, border_value)
        out = shifted if out is None else reduce_fn(out, shifted)
    return out


def maximum_filter_square(a: np.ndarray, radius: int = PEAK_NEIGHBORHOOD_SIZE) -> np.ndarray:
    """scipy.ndimage.maximum_filter(a, footprint=ones((2r+1, 2r+1))) with mode='reflect'.

    afp/dejavu/fingerprint.py:118-128.  The footprint
    iterate_structure(generate_binary_structure(2, 2), 10) is the full 21x21
    square, so the filter separates into two 1-D passes.
    """
    return _window_reduce(_window_reduce(a, radius, 0, np.maximum, "reflect"), radius, 1, np.maximum, "reflect")


def erode_square(b: np.ndarray, radius: int = PEAK_NEIGHBORHOOD_SIZE) -> np.ndarray:
    """binary_erosion(b, structure=ones(21x21), border_value=1).  fingerprint.py:131-134."""
    return _window_reduce(_window_reduce(b, radius, 0, np.logical_and, "const", True), radius, 1,
                          np.logical_and, "const", True)


def get_2d_peaks(arr2d: np.ndarray, amp_min: float = AMP_MIN, radius: int = PEAK_NEIGHBORHOOD_SIZE):
    """afp/dejavu/fingerprint.py:94-171 -> (peak_coordinates [(freq, time)], peak_mask float64).

    local_max = (21x21 max == value), XOR erosion of the exact-zero background,
    keep amplitude > amp_min (strict); coordinates in row-major order.
    """
    a = np.asarray(arr2d)
    local_max = maximum_filter_square(a, radius) == a
    eroded_bg = erode_square(a == 0, radius)
    detected = local_max != eroded_bg
    keep = detected & (a > amp_min)
    freqs, times = np.nonzero(keep)
    mask = np.zeros(a.shape)
    mask[keep] = 1
    return list(zip(freqs.tolist(), times.tolist())), mask


def preprocess(psd: np.ndarray) -> np.ndarray:
    """fingerprint.py:68,78-79 without denoising: /max, 10 ln(max(a, max/1e6)), minus mean."""
    a = psd / psd.max()
    a = 10 * np.log(np.maximum(a, np.max(a) / 1e6))
    return a - np.mean(a)


def fingerprint_peaks(samples: np.ndarray, amp_min: float = AMP_MIN):
    """fingerprint(..., denoising=False) up to the peak list: returns (coords, mask, specgram)."""
    psd = ostft.specgram_psd(samples)
    spec = psd / psd.max()
    coords, mask = get_2d_peaks(preprocess(psd), amp_min)
    return coords, mask, spec


def preprocess_denoised(unet_out: np.ndarray) -> np.ndarray:
    """fingerprint.py:74-79 after the UNet: the float32 output squared, then the same log / mean steps, which numpy keeps
    in float32 (a C-contiguous array: np.mean sums bin-major)."""
    a = np.ascontiguousarray(unet_out, dtype=np.float32) ** 2
    spec = a.copy()
    a = 10 * np.log(np.maximum(a, np.max(a) / 1e6))
    return a - np.mean(a), spec


def fingerprint_peaks_unet(samples: np.ndarray, unet_sd, amp_min: float = AMP_MIN):
    """fingerprint(..., denoising=True, denoising_model="unet") up to the peak list (fingerprint.py:58-84): the normalised
    PSD goes through the UNet as float32 (1, 1, 257, T); returns (coords, mask, specgram float32)."""
    import torch
    from . import unet as ounet
    psd = ostft.specgram_psd(samples)
    x = torch.tensor(psd / psd.max()).unsqueeze(0).unsqueeze(0).float()
    y = ounet.forward(x, unet_sd).squeeze().numpy()
    arr, spec = preprocess_denoised(y)
    coords, mask = get_2d_peaks(arr, amp_min)
    return coords, mask, spec
