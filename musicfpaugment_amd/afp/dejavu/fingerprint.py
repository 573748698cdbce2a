"""Dejavu peak picking on MI355X -- mirror of afp/dejavu/fingerprint.py:34-171 (up to the peak list).

``fingerprint(channel_samples, Fs, wsize, n_hop, fan_value, amp_min, denoising, denoising_model, get_masks)`` is the reference's
single-clip entry point with its return forms; ``get_2D_peaks(arr2D, plot=False, amp_min=50) -> (peak_coordinates, peak_mask)`` keeps the reference
contract (coordinates [(freq, time)] in row-major order, mask float64).  ``fingerprint_peaks_batch``
is the batched device path: mlab.specgram-style PSD -> /max -> 10 ln -> -mean -> 21x21 local maxima.
`generate_hashes` / `fingerprint_batch` run the pairing + SHA-1 on the device (mfpa_dejavu_hashes).
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np
import torch

from ... import ops
from ...constants import afp_settings

PEAK_NEIGHBORHOOD_SIZE = 10  # afp/dejavu/variables.py:19
CONNECTIVITY_MASK = 2        # variables.py:18: full square footprint


_DENOISERS = {"unet": None, "demucs": None}


def set_denoisers(unet=None, demucs=None) -> None:
    """The reference builds its denoisers at import time from checkpoint files (fingerprint.py:17-31); here the caller hands the
    modules over once (training.unet.UNet in eval mode on the GPU, training.model.Demucs) and `fingerprint` finds them."""
    if unet is not None:
        _DENOISERS["unet"] = unet
    if demucs is not None:
        _DENOISERS["demucs"] = demucs


def fingerprint(channel_samples, Fs: float = afp_settings["dejavu"]["samplerate"], wsize: int = afp_settings["dejavu"]["n_fft"],
                n_hop: int = afp_settings["dejavu"]["n_hop"], fan_value: int = afp_settings["dejavu"]["fan_value"],
                amp_min: int = afp_settings["dejavu"]["amp_min"], denoising: bool = False, denoising_model: str = "unet",
                get_masks: bool = "False", *, unet=None, device="cuda"):
    """afp/dejavu/fingerprint.py:34-91 for ONE clip, same signature and return forms: the list [(sha1 hex[:20], t1)] of hashes, or
    `(hashes, peak_mask (257, nF) float64, specgram (257, nF))` when `get_masks is True` -- the default is the STRING "False" and
    the test is `is True`, as in the reference (:43,88).  `channel_samples` are the raw (x 32767) samples Dejavu passes in.
    `denoising=True, denoising_model="unet"`: the spectrogram denoiser on the max-normalised PSD, output squared (:68-75), module
    from `unet=` or set_denoisers(); with "demucs" nothing happens HERE, as in the reference -- Dejavu denoises the waveform
    before it calls fingerprint (dejavu.py:85-106).  Everything runs on the device (fingerprint_batch with a batch of one)."""
    if denoising:
        assert denoising_model in ["unet", "demucs"]
    if (int(wsize), int(n_hop)) != (ops.N_FFT, ops.N_HOP):
        raise NotImplementedError("the device spectrogram is built for NFFT 512 / noverlap 256 (afp/parameters.py)")
    x = torch.as_tensor(np.asarray(channel_samples, dtype=np.float32) if not isinstance(channel_samples, torch.Tensor) else channel_samples)
    x = x.to(device, torch.float32).reshape(1, -1)
    net = None
    if denoising is True and denoising_model == "unet":
        net = unet if unet is not None else _DENOISERS["unet"]
        if net is None:
            raise ValueError("denoising with the UNet needs the module: pass unet=... or call set_denoisers(unet=...)")
    n_frames = (x.shape[1] - 256) // 256
    # at most one peak per 21 x 21 neighbourhood (it would need ~64 cells per peak to pack them), each paired with fan_value - 1
    # later peaks: `peaks` bounds the peak list (the kernel's limit is 16384 per call), `cap` only sizes the hash output -- a full
    # song has more than 16384 hashes long before it has 16384 peaks
    peaks = max(16, 257 * max(n_frames, 1) // 64)
    cap = max(16, min(peaks, 16384) * max(int(fan_value) - 1, 1))      # more hashes can never come out under the kernel's peak limit
    dig, t1, counts, mask, spec = fingerprint_batch(x, amp_min=amp_min, fan_value=fan_value, cap=cap, scale_in=1.0,
                                                    peak_cap=min(peaks, 16384),
                                                    denoising=net is not None, denoising_model="unet", unet=net)
    n = int(counts[0])
    if n < 0:
        raise ValueError("more than 16384 peaks in one recording: outside the device kernel's limit (split the recording)")
    hashes = _hashes_to_list(dig[0], t1[0], n)
    if get_masks is True:
        return hashes, mask[0].to(torch.float64).cpu().numpy(), spec[0].cpu().numpy()
    return hashes


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
                            scale_in: float = 32767.0, denoising: bool = False, denoising_model: str = "unet",
                            unet=None, demucs=None, want_spec: bool = True):
    """(B, T) float32 on the GPU -> (mask (B,257,nF) uint8, npeaks (B,), specgram (B,257,nF); None with want_spec=False on the
    un-denoised path, whose normalised specgram is one more pass over the PSD that the peaks do not need).

    ``denoising`` / ``denoising_model`` follow fingerprint.py:34-79 and dejavu.py:85-106: "unet" runs the spectrogram
    denoiser on the max-normalised PSD (cast to float32), squares its output and keeps float32 for the log / mean steps
    (specgram is then float32); "demucs" denoises the waveform before the x 32767 scaling, the rest is the float64 path.
    The networks are passed in (``unet`` = training.unet.UNet in eval mode, ``demucs`` = training.model.Demucs); the
    reference builds them at import time from checkpoint files."""
    if denoising:
        if denoising_model not in ("unet", "demucs"):
            raise AssertionError("denoising_model must be 'unet' or 'demucs'")
        net = unet if denoising_model == "unet" else demucs
        if net is None:
            raise ValueError(f"denoising_model={denoising_model!r} needs the {denoising_model} module")
        if denoising_model == "demucs":
            wav = demucs(wav)[:, 0]
    psd, cmax = ops.specgram_psd(wav, scale_in=scale_in)
    if denoising and denoising_model == "unet":
        y = unet.denoise_spectrogram(psd, cmax, per_clip=True)               # (B, 257, nF) float32
        arr = ops.dejavu_prepare_f32(y, square=True, scale=10.0, mean_order=0)
        mask, npeaks = ops.localmax2d(arr, PEAK_NEIGHBORHOOD_SIZE, float(amp_min))
        return mask, npeaks, y * y
    F, T = psd.shape[1:]
    if 141 <= F <= 257 and T <= 512:      # the fused pair of launches (mfpa_dejavu_pick); same bits as the two calls below
        mask, npeaks = ops.dejavu_pick(psd, cmax, 10.0, 1, PEAK_NEIGHBORHOOD_SIZE, float(amp_min))
    else:
        arr = ops.dejavu_prepare(psd, cmax, 10.0, mean_order=1)
        mask, npeaks = ops.localmax2d(arr, PEAK_NEIGHBORHOOD_SIZE, float(amp_min))
    return mask, npeaks, (ops.normalize_(psd, cmax, per_clip=True) if want_spec else None)


def _hashes_to_list(dig: torch.Tensor, t1: torch.Tensor, n: int):
    d = dig[:n].cpu().numpy()
    return [(bytes(d[i]).hex(), int(t)) for i, t in enumerate(t1[:n].cpu().tolist())]


def generate_hashes(peaks: List[Tuple[int, int]], fan_value: int = afp_settings["dejavu"]["fan_value"], device="cuda"):
    """[(freq, time)] -> [(sha1("f1|f2|dt")[:20], t1)], afp/dejavu/fingerprint.py:174-213, on the device."""
    if len(peaks) == 0:
        return []
    pk = np.asarray(peaks, dtype=np.int64).reshape(-1, 2)
    F, T = int(pk[:, 0].max()) + 1, int(pk[:, 1].max()) + 1
    mask = torch.zeros((1, F, T), dtype=torch.uint8)
    mask[0, pk[:, 0], pk[:, 1]] = 1
    cap = max(16, (fan_value - 1) * len(pk))
    dig, t1, counts = ops.dejavu_hashes(mask.to(device), cap=cap, peak_cap=max(16, min(16384, len(pk))), fan_value=fan_value)
    n = int(counts[0])
    if n < 0:
        raise ValueError("too many peaks for the device kernel (peak_cap 16384)")
    return _hashes_to_list(dig[0], t1[0], n)


def fingerprint_batch(wav: torch.Tensor, amp_min: float = afp_settings["dejavu"]["amp_min"],
                      fan_value: int = afp_settings["dejavu"]["fan_value"], cap: int = 4096, scale_in: float = 32767.0,
                      peak_cap: int = None, **denoise):
    """fingerprint(...) for a batch (afp/dejavu/fingerprint.py:34-91): (digests (B,cap,10) uint8, t1 (B,cap), counts (B,),
    peak mask, normalised specgram), everything on the device.  ``denoise``: the denoising arguments of
    fingerprint_peaks_batch."""
    mask, _, spec = fingerprint_peaks_batch(wav, amp_min, scale_in=scale_in, **denoise)
    # only the PEAK list is bounded by the kernel (<= 16384 peaks per clip, sorted in LDS); the hash capacity is just output memory
    dig, t1, counts = ops.dejavu_hashes(mask, cap=cap, peak_cap=min(cap, 16384) if peak_cap is None else peak_cap, fan_value=fan_value)
    return dig, t1, counts, mask, spec
