"""ORACLE (test infrastructure only -- never imported by musicfpaugment_amd): CPU restatement of the reference's
waveform-domain losses, training/loss.py:10-186 (MultiResolutionSTFTLoss of the Demucs branch, training/train.py:292-297).

Pinned by tests/golden/g11_mrstft_loss.npz, generated from the REAL reference (tools/make_goldens.py, G11)."""
from __future__ import annotations

from typing import List, Tuple

import torch


def stft_mag(x: torch.Tensor, fft_size: int, hop_size: int, win_length: int) -> torch.Tensor:
    """loss.py:10-41: torch.stft (center, reflect pad, hann(win_length) periodic, zero-padded to fft_size) ->
    sqrt(clamp(re^2 + im^2, 1e-7)), transposed to (B, frames, bins)."""
    window = torch.hann_window(win_length)
    z = torch.view_as_real(torch.stft(x, fft_size, hop_size, win_length, window, return_complex=True))
    return torch.sqrt(torch.clamp(z[..., 0] ** 2 + z[..., 1] ** 2, min=1e-7)).transpose(2, 1)


def stft_loss(x: torch.Tensor, y: torch.Tensor, fft_size: int, hop_size: int, win_length: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """loss.py:44-125: spectral convergence ||y - x||_F / ||y||_F and log-magnitude L1, over the whole batch tensor."""
    xm, ym = stft_mag(x, fft_size, hop_size, win_length), stft_mag(y, fft_size, hop_size, win_length)
    sc = torch.norm(ym - xm, p="fro") / torch.norm(ym, p="fro")
    mag = torch.nn.functional.l1_loss(torch.log(ym), torch.log(xm))
    return sc, mag


def multi_resolution_stft_loss(x: torch.Tensor, y: torch.Tensor, fft_sizes: List[int] = (1024, 2048, 512),
                               hop_sizes: List[int] = (120, 240, 50), win_lengths: List[int] = (600, 1200, 240),
                               factor_sc: float = 0.1, factor_mag: float = 0.1):
    """loss.py:128-186: mean over the resolutions, times the factors.  Returns (sc, mag, per-resolution [(sc, mag)])."""
    per = [stft_loss(x, y, f, h, w) for f, h, w in zip(fft_sizes, hop_sizes, win_lengths)]
    sc = sum(p[0] for p in per) / len(per)
    mag = sum(p[1] for p in per) / len(per)
    return factor_sc * sc, factor_mag * mag, per
