"""Oracle: AugmentFP signal chain, torch-CPU float32, with EXPLICIT parameters.

Restates the arithmetic of augmentation/transformations/*.py of the reference for given random draws (the draws
themselves are host-side torch.distributions calls, not the hot path):
  HighPass/LowPass (pass_filters.py:84-171, via julius 0.2.7 lowpass_filter(fft=False) -- NOT in the reference tree,
  restated from its published design: zeros=8, Hann-windowed sinc, unit DC gain, replicate padding => PARITY UNPINNED),
  ApplyImpulseResponse (impulse_response.py:73-164, FFT convolution, peak normalise, truncate),
  AddBackgroundNoise (background_noise.py:183-215), Gain (gain.py:62-70), Clipping (clipping.py:67-100),
  PeakNormalization (peak_normalization.py:38-67).  Everything except the julius filter is pinned by golden g10.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

ZEROS = 8


def lowpass_taps(cutoff: float, zeros: int = ZEROS) -> torch.Tensor:
    """julius.LowPassFilters (0.2.7) for one cutoff (fraction of the sample rate): half = int(zeros/cutoff/2),
    taps = 2c * hann(2*half+1) * sinc(2c*pi*t), normalised to unit sum."""
    if cutoff <= 0:
        raise ValueError("Buggy cutoff freq.")           # pass_filters.py:103-110
    if cutoff > 0.5:
        raise ValueError("A cutoff above 0.5 does not make sense.")
    half = int(zeros / cutoff / 2)
    window = torch.hann_window(2 * half + 1, periodic=False)
    t = torch.arange(-half, half + 1, dtype=torch.float32)
    arg = 2 * cutoff * math.pi * t
    sinc = torch.where(t == 0, torch.ones_like(arg), torch.sin(arg) / arg)
    taps = 2 * cutoff * window * sinc
    return taps / taps.sum()


def lowpass(x: torch.Tensor, cutoff: float) -> torch.Tensor:
    """(C, T) -> (C, T): replicate-pad by half, correlate with the (symmetric) taps."""
    taps = lowpass_taps(cutoff)
    half = (len(taps) - 1) // 2
    xp = F.pad(x[None], (half, half), mode="replicate")[0]
    return F.conv1d(xp[:, None], taps[None, None])[:, 0]


def highpass(x: torch.Tensor, cutoff: float) -> torch.Tensor:
    """pass_filters.py:158-171: x - lowpass(x)."""
    return x - lowpass(x, cutoff)


def next_fast_len(size: int) -> int:
    """impulse_response.py:168-195: next 5-smooth number."""
    while True:
        r = size
        for p in (2, 3, 5):
            while r % p == 0:
                r //= p
        if r == 1:
            return size
        size += 1


def convolve_full(signal: torch.Tensor, kernel: torch.Tensor) -> torch.Tensor:
    """impulse_response.py:119-164, mode='full', FFT based."""
    m, n = signal.size(-1), kernel.size(-1)
    size = next_fast_len(m + n - 1)
    res = torch.fft.irfft(torch.fft.rfft(signal, n=size) * torch.fft.rfft(kernel, n=size), n=size)
    return res[..., : m + n - 1]


def apply_ir(samples: torch.Tensor, ir: torch.Tensor) -> torch.Tensor:
    """(B,1,T), (B,1,L) -> (B,1,T): full convolution, divide by the peak of the FULL result, keep the first T samples."""
    T = samples.shape[2]
    conv = convolve_full(samples, ir)
    conv = conv / conv.abs().amax(dim=2, keepdim=True)
    return conv[..., :T]


def add_background(samples: torch.Tensor, background: torch.Tensor, snr_db: torch.Tensor) -> torch.Tensor:
    """(B,1,T), rms-normalised noise (B,T), (B,) -> x + rms(x)/10^(snr/20) * noise, then peak normalise."""
    rms = torch.sqrt(torch.mean(torch.square(samples), dim=-1))                   # (B,1)
    bg_rms = rms / (10 ** (snr_db.unsqueeze(-1) / 20))
    y = samples + bg_rms.unsqueeze(-1) * background[:, None, :]
    return y / y.abs().amax(dim=2, keepdim=True)


def gain(samples: torch.Tensor, gain_db: torch.Tensor) -> torch.Tensor:
    return samples * (10 ** (gain_db / 20)).view(-1, 1, 1)


def clipping(samples: torch.Tensor, percentile: torch.Tensor) -> torch.Tensor:
    """clipping.py:67-100: clamp to the per-example quantiles (p/2, 1 - p/2)."""
    lo_q = percentile / 2
    out = []
    for b in range(samples.shape[0]):
        x = samples[b, 0]
        lo = torch.quantile(x, lo_q[b])
        hi = torch.quantile(x, 1 - lo_q[b])
        out.append(torch.clip(x, min=lo, max=hi))
    return torch.stack(out)[:, None]


def clipping_flat(samples: torch.Tensor, percentile: torch.Tensor) -> torch.Tensor:
    """clipping.py:67-100 as batch_augment runs it on B > 1 selected examples: torch.quantile has no dim argument there, so
    example b is clamped to the p_b/2 and 1 - p_b/2 quantiles of the whole flattened sub-batch."""
    flat = samples[:, 0, :].reshape(-1)
    lo = torch.quantile(flat, percentile.reshape(-1) / 2)
    hi = torch.quantile(flat, 1 - percentile.reshape(-1) / 2)
    return torch.clip(samples[:, 0, :], min=lo[:, None], max=hi[:, None]).unsqueeze(1)


def peak_normalize(samples: torch.Tensor) -> torch.Tensor:
    peak = samples.abs().amax(dim=(1, 2), keepdim=True)
    return torch.where(peak > 0, samples / torch.where(peak > 0, peak, torch.ones_like(peak)), samples)


def rms_normalize(x: torch.Tensor) -> torch.Tensor:
    """utils.py:190-205."""
    return x / (x.square().mean(dim=-1, keepdim=True).sqrt() + 1e-8)
