"""Thin, typed wrappers over the C ABI (include/mfpa.h): torch tensors in, torch tensors out.

Each function allocates its outputs with torch on the input's device, checks shapes on the
host (a kernel must never see a shape its grid does not assume) and enqueues on torch's
current HIP stream.  Nothing here computes on the CPU.
"""
from __future__ import annotations

import ctypes
from functools import lru_cache
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import F32, F64, check, lib, ptr, require_gpu, stream

N_FFT, N_HOP, N_BINS = 512, 256, 257


def _dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.float64:
        return F64
    raise TypeError(f"float32 or float64 expected, got {t.dtype}")


# ----------------------------------------------------------------------------- STFT
def audfprint_window() -> np.ndarray:
    """np.hanning(514)[1:-1]  (training/visualisation.py:18, afp/audfprint/peak_extractor.py:257)."""
    return np.hanning(N_FFT + 2)[1:-1]


def dejavu_window() -> np.ndarray:
    """mlab.window_hanning on 512 samples = np.hanning(512)  (afp/dejavu/fingerprint.py:64)."""
    return np.hanning(N_FFT)


@lru_cache(maxsize=16)
def _tables_cached(kind: str, device_index: int) -> torch.Tensor:
    win = audfprint_window() if kind == "audfprint" else dejavu_window()
    return stft_tables(win, torch.device("cuda", device_index))


def stft_tables(window: np.ndarray, device) -> torch.Tensor:
    """Host-side table build (mfpa_stft_tables) + upload: window and FFT twiddles, float64."""
    w = np.ascontiguousarray(window, dtype=np.float64)
    if w.shape != (N_FFT,):
        raise ValueError("window must have 512 points")
    out = np.empty(_lib.STFT_TABLE_LEN, dtype=np.float64)
    check(lib().mfpa_stft_tables(w.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p)),
          "mfpa_stft_tables")
    return torch.from_numpy(out).to(device)


def default_tables(device, kind: str = "audfprint") -> torch.Tensor:
    device = torch.device(device)
    return _tables_cached(kind, device.index if device.index is not None else torch.cuda.current_device())


def stft_frames(n_samples: int) -> int:
    return 1 + n_samples // N_HOP


def stft_mag(wav: torch.Tensor, out_dtype=torch.float64, tables: Optional[torch.Tensor] = None,
             want_max: bool = True) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """(B, T_w) float32 -> |STFT| (B, 257, 1 + T_w//256) and per-clip float64 maxima."""
    require_gpu(wav, "waveform")
    if wav.dim() != 2 or wav.dtype != torch.float32:
        raise ValueError("waveform must be (B, T) float32")
    wav = wav.contiguous()
    B, T_w = wav.shape
    if T_w <= N_HOP:
        raise ValueError("reflect padding needs more than 256 samples per clip")
    if tables is None:
        tables = default_tables(wav.device)
    nF = stft_frames(T_w)
    mag = torch.empty((B, N_BINS, nF), dtype=out_dtype, device=wav.device)
    cmax = torch.empty((B,), dtype=torch.float64, device=wav.device) if want_max else None
    check(lib().mfpa_stft_mag(ptr(wav), B, T_w, ptr(tables), ptr(mag), _dtype_code(mag), ptr(cmax), stream()),
          "mfpa_stft_mag")
    return mag, cmax


def specgram_psd(wav: torch.Tensor, scale_in: float = 1.0, tables: Optional[torch.Tensor] = None):
    """(B, T_w) float32 -> mlab.specgram-style PSD (B, 257, (T_w-256)//256) float64 (unscaled) + clip maxima."""
    require_gpu(wav, "waveform")
    if wav.dim() != 2 or wav.dtype != torch.float32:
        raise ValueError("waveform must be (B, T) float32")
    wav = wav.contiguous()
    B, T_w = wav.shape
    if T_w < N_FFT:
        raise ValueError("need at least one full 512-sample frame")
    if tables is None:
        tables = default_tables(wav.device, "dejavu")
    nF = (T_w - 256) // 256
    psd = torch.empty((B, N_BINS, nF), dtype=torch.float64, device=wav.device)
    cmax = torch.empty((B,), dtype=torch.float64, device=wav.device)
    check(lib().mfpa_specgram_psd(ptr(wav), B, T_w, float(scale_in), ptr(tables), ptr(psd), ptr(cmax), stream()),
          "mfpa_specgram_psd")
    return psd, cmax


def normalize_(data: torch.Tensor, clip_max: torch.Tensor, per_clip: bool) -> torch.Tensor:
    """In-place data[b] /= (clip_max[b] if per_clip else max(clip_max))."""
    require_gpu(data, "data")
    B = data.shape[0]
    if clip_max.shape != (B,) or clip_max.dtype != torch.float64:
        raise ValueError("clip_max must be (B,) float64")
    n = data[0].numel() if B else 0
    check(lib().mfpa_normalize(ptr(data), _dtype_code(data), B, n, ptr(clip_max), int(per_clip), stream()),
          "mfpa_normalize")
    return data


def f64_to_f32(x: torch.Tensor) -> torch.Tensor:
    require_gpu(x)
    if x.dtype != torch.float64:
        raise TypeError("float64 expected")
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    check(lib().mfpa_f64_to_f32(ptr(x), ptr(out), x.numel(), stream()), "mfpa_f64_to_f32")
    return out


# ----------------------------------------------------------------------------- Audfprint picker
# testing/parameters.py:17-26 (afp_settings["audfprint"])
AUDFPRINT_DENSITY = 20
AUDFPRINT_MAX_PKS = 5
AUDFPRINT_F_SD = 30.0
AUDFPRINT_POLE = 0.98


def audfprint_a_dec(density: float = AUDFPRINT_DENSITY, n_hop: int = N_HOP) -> float:
    """peak_extractor.py:295."""
    return float(1 - 0.01 * (density * np.sqrt(n_hop / 352.8) / 35))


@lru_cache(maxsize=16)
def _gauss_cached(npoints: int, width: float, device_index: int) -> torch.Tensor:
    # peak_extractor.py:163-165 -- computed with numpy on the host so the table is numpy's, bit for bit
    tab = np.exp(-0.5 * ((np.arange(-npoints, npoints + 1) / width) ** 2))
    return torch.from_numpy(tab).to(torch.device("cuda", device_index))


def gauss_table(npoints: int, width: float, device) -> torch.Tensor:
    device = torch.device(device)
    return _gauss_cached(int(npoints), float(width),
                         device.index if device.index is not None else torch.cuda.current_device())


def audfprint_prepare(spec: torch.Tensor, denom: Optional[torch.Tensor] = None, mean_order: int = 0,
                      log_input: bool = False, pole: float = AUDFPRINT_POLE, denom_is_clip_max: bool = False) -> torch.Tensor:
    """(B, F, T) spectrogram -> frame-major filtered log-spectrogram (B, T, F-1) float64.  `denom_is_clip_max`: denom[b] is the
    maximum of the float64 spec[b] itself (what stft_mag returned with it), so the kernel skips its max pass."""
    require_gpu(spec, "spectrogram")
    if spec.dim() != 3:
        raise ValueError("spectrogram must be (B, F, T)")
    spec = spec.contiguous()
    B, F, T = spec.shape
    if F < 2 or F - 1 > 256 or T < 1:
        raise ValueError("need 2 <= F <= 257 bins and T >= 1 frames")
    if denom is not None and (denom.shape != (B,) or denom.dtype != torch.float64):
        raise ValueError("denom must be (B,) float64")
    filtered = torch.empty((B, T, F - 1), dtype=torch.float64, device=spec.device)
    scratch = torch.empty((B, F * T), dtype=torch.float64, device=spec.device)
    check(lib().mfpa_audfprint_prepare(ptr(spec), _dtype_code(spec), B, F, T, ptr(denom), int(mean_order),
                                       int(bool(log_input)) | (2 if (denom_is_clip_max and denom is not None and spec.dtype == torch.float64) else 0),
                                       float(pole), ptr(filtered), ptr(scratch), stream()),
          "mfpa_audfprint_prepare")
    return filtered


def audfprint_prune(filtered: torch.Tensor, a_dec: Optional[float] = None, maxpks: int = AUDFPRINT_MAX_PKS,
                    f_sd: float = AUDFPRINT_F_SD):
    """Frame-major filtered (B, T, R) float64 -> (mask (B, R, T) uint8, npeaks (B,) int32)."""
    require_gpu(filtered, "filtered spectrogram")
    if filtered.dim() != 3 or filtered.dtype != torch.float64:
        raise ValueError("filtered must be (B, T, R) float64")
    filtered = filtered.contiguous()
    B, T, R = filtered.shape
    if R % 4 or R < 4 or R > 256 or T < 1 or T > 1500 or not (1 <= maxpks <= 8):
        raise ValueError("unsupported pruner shape (R % 4 == 0, R <= 256, T <= 1500, maxpks <= 8)")
    if a_dec is None:
        a_dec = audfprint_a_dec()
    gauss = gauss_table(R, f_sd, filtered.device)
    mask = torch.empty((B, R, T), dtype=torch.uint8, device=filtered.device)
    npeaks = torch.empty((B,), dtype=torch.int32, device=filtered.device)
    check(lib().mfpa_audfprint_prune(ptr(filtered), B, R, T, ptr(gauss), float(a_dec), int(maxpks), ptr(mask),
                                     ptr(npeaks), stream()), "mfpa_audfprint_prune")
    return mask, npeaks


def audfprint_pick(mag: torch.Tensor, clip_max: torch.Tensor, a_dec: Optional[float] = None, maxpks: int = AUDFPRINT_MAX_PKS,
                   f_sd: float = AUDFPRINT_F_SD, pole: float = AUDFPRINT_POLE):
    """find_peaks without a denoiser, stages 1 + 2 in one call (mfpa_audfprint_pick): raw float64 |STFT| (B, F, T) and its per-clip
    maxima (what stft_mag returned) -> (mask (B, F-1, T) uint8, npeaks (B,) int32).  Same arithmetic as
    audfprint_prepare(mag, clip_max, mean_order=1, denom_is_clip_max=True) + audfprint_prune; the filtered spectrogram is never
    written (the pruner filters the frames as it walks them)."""
    require_gpu(mag, "spectrogram")
    if mag.dim() != 3 or mag.dtype != torch.float64:
        raise ValueError("spectrogram must be (B, F, T) float64")
    mag = mag.contiguous()
    B, F, T = mag.shape
    if clip_max.shape != (B,) or clip_max.dtype != torch.float64:
        raise ValueError("clip_max must be (B,) float64")
    if F < 141 or F > 257 or (F - 1) % 4 or T < 1 or T > 512 or ((F - 1) * T) % 16 or not (1 <= maxpks <= 8):
        raise ValueError("unsupported shape for the fused picker (141 <= F <= 257, (F - 1) % 4 == 0, T <= 512, maxpks <= 8)")
    if a_dec is None:
        a_dec = audfprint_a_dec()
    gauss = gauss_table(F - 1, f_sd, mag.device)
    work = torch.empty(B * F * T + B * 128, dtype=torch.float64, device=mag.device)
    mask = torch.empty((B, F - 1, T), dtype=torch.uint8, device=mag.device)
    npeaks = torch.empty((B,), dtype=torch.int32, device=mag.device)
    check(lib().mfpa_audfprint_pick(ptr(mag), ptr(clip_max), B, F, T, float(pole), ptr(gauss), float(a_dec), int(maxpks), ptr(work),
                                    ptr(mask), ptr(npeaks), stream()), "mfpa_audfprint_pick")
    return mask, npeaks


# ----------------------------------------------------------------------------- Dejavu picker
DEJAVU_RADIUS = 10   # afp/dejavu/variables.py:19 PEAK_NEIGHBORHOOD_SIZE
DEJAVU_AMP_MIN = 50  # testing/parameters.py:32


def dejavu_prepare(psd: torch.Tensor, denom: Optional[torch.Tensor], scale: float = 10.0,
                   mean_order: int = 1) -> torch.Tensor:
    require_gpu(psd, "psd")
    if psd.dim() != 3 or psd.dtype != torch.float64:
        raise ValueError("psd must be (B, F, T) float64")
    psd = psd.contiguous()
    B, F, T = psd.shape
    arr = torch.empty_like(psd)
    check(lib().mfpa_dejavu_prepare(ptr(psd), B, F, T, ptr(denom), float(scale), int(mean_order), ptr(arr), stream()),
          "mfpa_dejavu_prepare")
    return arr


def dejavu_pick(psd: torch.Tensor, clip_max: torch.Tensor, scale: float = 10.0, mean_order: int = 1, radius: int = DEJAVU_RADIUS,
                amp_min: float = DEJAVU_AMP_MIN):
    """The un-denoised Dejavu chain after the spectrogram in one call (mfpa_dejavu_pick): PSD (B, F, T) float64 and its per-clip
    maxima (what specgram_psd returned) -> (mask (B, F, T) uint8, npeaks (B,) int32).  Same arithmetic as
    dejavu_prepare(psd, clip_max, scale, mean_order) + localmax2d; the mean-subtracted array is never written."""
    require_gpu(psd, "psd")
    if psd.dim() != 3 or psd.dtype != torch.float64:
        raise ValueError("psd must be (B, F, T) float64")
    psd = psd.contiguous()
    B, F, T = psd.shape
    if clip_max.shape != (B,) or clip_max.dtype != torch.float64:
        raise ValueError("clip_max must be (B,) float64")
    per = ctypes.c_longlong(0)
    check(lib().mfpa_dejavu_pick_work_doubles(F, T, ctypes.byref(per)), "mfpa_dejavu_pick_work_doubles")
    work = torch.empty(max(B, 1) * per.value, dtype=torch.float64, device=psd.device)
    mask = torch.empty((B, F, T), dtype=torch.uint8, device=psd.device)
    npeaks = torch.empty(B, dtype=torch.int32, device=psd.device)
    check(lib().mfpa_dejavu_pick(ptr(psd), ptr(clip_max), B, F, T, float(scale), int(mean_order), int(radius), float(amp_min),
                                 ptr(work), ptr(mask), ptr(npeaks), stream()), "mfpa_dejavu_pick")
    return mask, npeaks


def dejavu_prepare_f32(x: torch.Tensor, square: bool = True, scale: float = 10.0, mean_order: int = 0) -> torch.Tensor:
    """The denoised branch of Dejavu's pre-processing (fingerprint.py:70-79): float32 (B, F, T) network output -> x**2 ->
    10*log(max(., max/1e6)) - mean in float32, widened to float64 for the picker."""
    require_gpu(x, "x")
    if x.dim() != 3 or x.dtype != torch.float32:
        raise ValueError("x must be (B, F, T) float32")
    x = x.contiguous()
    B, F, T = x.shape
    arr = torch.empty((B, F, T), dtype=torch.float64, device=x.device)
    check(lib().mfpa_dejavu_prepare_f32(ptr(x), B, F, T, int(bool(square)), float(scale), int(mean_order), ptr(arr),
                                        stream()), "mfpa_dejavu_prepare_f32")
    return arr


def localmax2d(arr: torch.Tensor, radius: int = DEJAVU_RADIUS, amp_min: float = DEJAVU_AMP_MIN):
    require_gpu(arr, "arr2D")
    if arr.dim() != 3 or arr.dtype != torch.float64:
        raise ValueError("arr must be (B, F, T) float64")
    arr = arr.contiguous()
    B, F, T = arr.shape
    if not (0 <= radius <= 16) or F < 1 or T < 1:
        raise ValueError("radius must be in [0, 16]")
    mask = torch.empty((B, F, T), dtype=torch.uint8, device=arr.device)
    npeaks = torch.empty((B,), dtype=torch.int32, device=arr.device)
    check(lib().mfpa_localmax2d(ptr(arr), B, F, T, int(radius), float(amp_min), ptr(mask), ptr(npeaks), stream()),
          "mfpa_localmax2d")
    return mask, npeaks


# ----------------------------------------------------------------------------- metrics
def peak_metrics_counts(predicted: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    """0/1 uint8 masks (B, N1, N2) -> (B, 4) int64 [hits_p, n_pred, hits_r, n_gt]."""
    require_gpu(predicted, "predicted")
    require_gpu(gt, "gt")
    if predicted.shape != gt.shape or predicted.dim() != 3:
        raise ValueError("masks must both be (B, N1, N2)")
    if predicted.dtype != torch.uint8 or gt.dtype != torch.uint8:
        raise TypeError("masks must be uint8")
    predicted, gt = predicted.contiguous(), gt.contiguous()
    B, N1, N2 = predicted.shape
    if N1 < 2 or N2 < 2:
        raise ValueError("mask axes must have length >= 2")
    counts = torch.empty((B, 4), dtype=torch.int64, device=predicted.device)
    check(lib().mfpa_peak_metrics(ptr(predicted), ptr(gt), B, N1, N2, ptr(counts), stream()), "mfpa_peak_metrics")
    return counts


# ----------------------------------------------------------------------------- landmarks / hashes
def audfprint_landmarks(mask: torch.Tensor, cap: int = 4096, mindt: int = 2, targetdt: int = 63, targetdf: int = 31,
                        maxpairs: int = 3):
    """Peak masks (B, R, T) uint8 -> (landmarks (B,cap,4), hashes (B,cap,2), unique sorted hashes (B,cap,2),
    counts (B,2) = [n_landmarks, n_unique]); all int32 on the device.  peak_extractor.py:313-346, :40-58, :443-460."""
    require_gpu(mask, "mask")
    if mask.dim() != 3 or mask.dtype != torch.uint8:
        raise ValueError("mask must be (B, R, T) uint8")
    mask = mask.contiguous()
    B, R, T = mask.shape
    if not (1 <= cap <= 8192) or R > 256:
        raise ValueError("cap must be in [1, 8192] and R <= 256")
    dev = mask.device
    lm = torch.zeros((B, cap, 4), dtype=torch.int32, device=dev)
    hs = torch.zeros((B, cap, 2), dtype=torch.int32, device=dev)
    uq = torch.zeros((B, cap, 2), dtype=torch.int32, device=dev)
    counts = torch.zeros((B, 2), dtype=torch.int32, device=dev)
    check(lib().mfpa_audfprint_landmarks(ptr(mask), B, R, T, cap, mindt, targetdt, targetdf, maxpairs, ptr(lm), ptr(hs),
                                         ptr(uq), ptr(counts), stream()), "mfpa_audfprint_landmarks")
    return lm, hs, uq, counts


def dejavu_hashes(mask: torch.Tensor, cap: int = 4096, peak_cap: int = 4096, fan_value: int = 3, min_dt: int = 0,
                  max_dt: int = 200):
    """Peak masks (B, F, T) uint8 -> (digests (B,cap,10) uint8 = sha1("f1|f2|dt")[:20 hex], t1 (B,cap) int32,
    counts (B,) int32).  afp/dejavu/fingerprint.py:174-213."""
    require_gpu(mask, "mask")
    if mask.dim() != 3 or mask.dtype != torch.uint8:
        raise ValueError("mask must be (B, F, T) uint8")
    mask = mask.contiguous()
    B, F, T = mask.shape
    dev = mask.device
    dig = torch.zeros((B, cap, 10), dtype=torch.uint8, device=dev)
    t1 = torch.zeros((B, cap), dtype=torch.int32, device=dev)
    counts = torch.zeros((B,), dtype=torch.int32, device=dev)
    check(lib().mfpa_dejavu_hashes(ptr(mask), B, F, T, cap, peak_cap, fan_value, min_dt, max_dt, ptr(dig), ptr(t1),
                                   ptr(counts), stream()), "mfpa_dejavu_hashes")
    return dig, t1, counts


def psnr_stats(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """(B, ...) pred (float32|float64) vs float64 target -> (B, 3) float64 [sse, min(target), max(target)]."""
    require_gpu(pred, "pred")
    require_gpu(target, "target")
    if pred.shape != target.shape or target.dtype != torch.float64:
        raise ValueError("pred/target must have the same shape, target float64")
    pred, target = pred.contiguous(), target.contiguous()
    B = pred.shape[0]
    n = pred[0].numel() if B else 0
    out = torch.empty((B, 3), dtype=torch.float64, device=pred.device)
    check(lib().mfpa_psnr_stats(ptr(pred), _dtype_code(pred), ptr(target), B, n, ptr(out), stream()), "mfpa_psnr_stats")
    return out
