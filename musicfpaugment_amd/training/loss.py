"""Waveform-domain spectral losses on MI355X -- mirror of the reference's training/loss.py:10-186 (forward values).

``stft(x, fft_size, hop_size, win_length, window)``, ``STFTLoss`` and ``MultiResolutionSTFTLoss`` keep the reference's
names, arguments and return conventions ((B, frames, bins) magnitudes; ``(sc_loss, mag_loss)`` scalars, the multi-resolution
mean multiplied by ``factor_sc`` / ``factor_mag``).  The CQT losses of the file (loss.py:189-322, nnAudio) are never
instantiated by train.py and are not built.  `forward` gives the loss VALUES (validation_epoch, monitoring);
`value_and_grad` also returns the gradient with respect to the predicted waveform (the Demucs backward pass that would consume it is
not built).

How: a resolution's STFT is one strided-window GEMM on the fp32 matrix cores (csrc/loss.hip, csrc/demucs.hip):
frames x windowed-DFT matrix, K = the window length (600 / 1200 / 240 of the 1024 / 2048 / 512-point frames).
"""
from __future__ import annotations

import ctypes
import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .._lib import check, lib, ptr, require_gpu, stream
from ..ops_demucs import gemm

_DFT_CACHE: Dict[Tuple[int, int, int, int], Tuple[torch.Tensor, int, int, int]] = {}
_DFT_T_CACHE: Dict[Tuple[int, int, int], torch.Tensor] = {}


def _dev_index(device) -> int:
    """Resolved GPU index: an unindexed 'cuda' device means the CURRENT device, not cuda:0."""
    device = torch.device(device)
    return device.index if device.index is not None else torch.cuda.current_device()


def _dft_matrix_t(fft_size: int, win_length: int, device) -> torch.Tensor:
    """The transposed windowed DFT matrix (rows = window samples, padded to a multiple of 64; K = npad) for the backward GEMM."""
    key = (fft_size, win_length, _dev_index(device))
    if key not in _DFT_T_CACHE:
        W, _, _, kpad = _dft_matrix(fft_size, win_length, device)
        rows = (kpad + 63) // 64 * 64
        Wt = torch.zeros((rows, W.shape[0]), dtype=torch.float32, device=device)
        Wt[:kpad] = W.t()
        _DFT_T_CACHE[key] = Wt.contiguous()
    return _DFT_T_CACHE[key]


def _dft_matrix(fft_size: int, win_length: int, device) -> Tuple[torch.Tensor, int, int, int]:
    """Rows [re bins | zeros | im bins] of hann(win_length, periodic) * exp(-2 pi i k (off + j) / fft_size), K padded to a
    multiple of 16 with zero columns.  Returns (W (npad, Kpad) float32, bins, im_off, Kpad).  Built in float64."""
    key = (fft_size, win_length, _dev_index(device), 0)
    if key not in _DFT_CACHE:
        bins = fft_size // 2 + 1
        off = (fft_size - win_length) // 2
        kpad = (win_length + 31) // 32 * 32           # K of the GEMM: a multiple of 32 so that the bf16x3 kernel can take it
        im_off = (bins + 63) // 64 * 64
        npad = (im_off + bins + 63) // 64 * 64
        j = np.arange(win_length, dtype=np.float64)
        win = 0.5 - 0.5 * np.cos(2.0 * np.pi * j / win_length)                   # torch.hann_window(periodic=True)
        ang = 2.0 * np.pi * np.outer(np.arange(bins, dtype=np.float64), off + j) / fft_size
        W = np.zeros((npad, kpad), dtype=np.float32)
        W[:bins, :win_length] = (win * np.cos(ang)).astype(np.float32)
        W[im_off:im_off + bins, :win_length] = (-win * np.sin(ang)).astype(np.float32)
        _DFT_CACHE[key] = (torch.from_numpy(W).to(device), bins, im_off, kpad)
    return _DFT_CACHE[key]


def _dft_rows(x: torch.Tensor, fft_size: int, hop_size: int, win_length: int, precision: int = 0):
    """(B, T) float32 on the GPU -> GEMM output C (B, frames, npad) float32 with re at [:bins], im at [im_off:im_off+bins]."""
    require_gpu(x, "signal")
    if x.dim() != 2 or x.dtype != torch.float32:
        raise ValueError("expected a float32 (B, T) tensor")
    if win_length > fft_size or (fft_size - win_length) % 8 or hop_size % 2:
        raise NotImplementedError("window offset must be a multiple of 4 samples and the hop even")
    x = x.contiguous()
    B, T = x.shape
    dev = x.device
    W, bins, im_off, kpad = _dft_matrix(fft_size, win_length, dev)
    npad = W.shape[0]
    pad = fft_size // 2
    if pad >= T:
        raise ValueError("reflect padding needs fft_size / 2 < T")               # torch.stft raises as well
    frames = 1 + T // hop_size
    off = (fft_size - win_length) // 2
    Lout = (T + 2 * pad + kpad + 64 + 3) // 4 * 4                                # zero tail: K padding reads stay finite
    L = lib()
    xp = torch.empty((B, Lout), dtype=torch.float32, device=dev)
    check(L.mfpa_reflect_pad(ptr(x), B, T, pad, 0, Lout, ptr(xp), stream()), "mfpa_reflect_pad")
    C = torch.empty((B, frames, npad), dtype=torch.float32, device=dev)
    if hop_size % 4 == 0:
        gemm(ptr(xp) + 4 * off, hop_size, Lout, B, frames, W, None, npad, ptr(C), npad, frames * npad, precision=precision)
    else:
        # odd frames start at off + hop (2 mod 4 floats): read them from a copy shifted by 2 samples so that every row
        # of both GEMMs is 16-byte aligned; even / odd frames interleave in C through the row pitch 2 * npad
        xs = torch.empty((B, Lout), dtype=torch.float32, device=dev)
        check(L.mfpa_reflect_pad(ptr(x), B, T, pad, 2, Lout, ptr(xs), stream()), "mfpa_reflect_pad")
        n_even, n_odd = (frames + 1) // 2, frames // 2
        gemm(ptr(xp) + 4 * off, 2 * hop_size, Lout, B, n_even, W, None, npad, ptr(C), 2 * npad, frames * npad, precision=precision)
        if n_odd:
            gemm(ptr(xs) + 4 * (off + hop_size - 2), 2 * hop_size, Lout, B, n_odd, W, None, npad, ptr(C) + 4 * npad, 2 * npad,
                 frames * npad, precision=precision)
    return C, bins, im_off, frames


def _check_window(window: Optional[torch.Tensor], win_length: int) -> None:
    if window is not None:
        want = torch.hann_window(win_length)
        if window.numel() != win_length or not torch.allclose(window.detach().cpu().float(), want, atol=1e-6):
            raise NotImplementedError("only the reference's hann_window(win_length) is built into the DFT matrix")


def stft(x: torch.Tensor, fft_size: int, hop_size: int, win_length: int, window: Optional[torch.Tensor]) -> torch.Tensor:
    """Magnitude spectrogram (B, #frames, fft_size // 2 + 1), loss.py:10-41."""
    _check_window(window, win_length)
    C, bins, im_off, frames = _dft_rows(x, fft_size, hop_size, win_length)
    B = x.shape[0]
    mag = torch.empty((B, frames, bins), dtype=torch.float32, device=x.device)
    check(lib().mfpa_dft_mag(ptr(C), B * frames, bins, C.shape[2], im_off, ptr(mag), stream()), "mfpa_dft_mag")
    return mag


class STFTLoss(torch.nn.Module):
    """loss.py:86-125."""

    def __init__(self, fft_size: int = 1024, shift_size: int = 120, win_length: int = 600, window: str = "hann_window",
                 precision: int = 0) -> None:
        """`precision` (not in the reference): arithmetic of the DFT GEMMs -- 0 exact fp32 products (default), 1 bf16x3 (relative
        error ~1e-6 on the loss values; what the training step uses)."""
        super().__init__()
        self.precision = precision
        if window != "hann_window":
            raise NotImplementedError("only hann_window (the reference's default and only use)")
        self.fft_size, self.shift_size, self.win_length = fft_size, shift_size, win_length
        self.register_buffer("window", torch.hann_window(win_length))

    @torch.no_grad()
    def forward(self, x: torch.Tensor, y: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        if x.shape != y.shape:
            raise ValueError("predicted and groundtruth signals must have the same shape")
        Cx, bins, im_off, frames = _dft_rows(x, self.fft_size, self.shift_size, self.win_length, self.precision)
        Cy, _, _, _ = _dft_rows(y, self.fft_size, self.shift_size, self.win_length, self.precision)
        rows = x.shape[0] * frames
        L = lib()
        out = torch.empty(3, dtype=torch.float64, device=x.device)
        ws = torch.empty(3 * L.mfpa_loss_blocks(), dtype=torch.float64, device=x.device)
        check(L.mfpa_stft_loss_sums(ptr(Cx), ptr(Cy), rows, bins, Cx.shape[2], im_off, ptr(out), ptr(ws), stream()),
              "mfpa_stft_loss_sums")
        sc = torch.sqrt(out[0]) / torch.sqrt(out[1])                              # ||y - x||_F / ||y||_F   (loss.py:62)
        mag = out[2] / float(rows * bins)                                         # F.l1_loss(log y, log x) (loss.py:83)
        return sc.to(torch.float32), mag.to(torch.float32)

    @torch.no_grad()
    def target_transform(self, y: torch.Tensor) -> torch.Tensor:
        """The DFT rows of the target signal alone (it does not depend on the model: the training step computes it on a side stream
        while the forward pass runs); pass the result to value_and_grad(..., Cy=...)."""
        return _dft_rows(y, self.fft_size, self.shift_size, self.win_length, self.precision)[0]

    def value_and_grad(self, x: torch.Tensor, y: torch.Tensor, w_sc: float, w_mag: float, dx: torch.Tensor, accumulate: bool,
                       Cy: Optional[torch.Tensor] = None):
        """(sc, mag) and d(w_sc * sc + w_mag * mag) / dx written (or added) to dx (B, T): the adjoint chain of the forward --
        loss gradient on (re, im), GEMM with the transposed DFT matrix, overlap-add of the frames, reflect-padding adjoint."""
        if x.shape != y.shape or dx.shape != x.shape or dx.dtype != torch.float32:
            raise ValueError("x, y and dx must be float32 tensors of one shape")
        fs, hop, wl = self.fft_size, self.shift_size, self.win_length
        Cx, bins, im_off, frames = _dft_rows(x, fs, hop, wl, self.precision)
        if Cy is None:
            Cy, _, _, _ = _dft_rows(y, fs, hop, wl, self.precision)
        B, T = x.shape
        rows, npad = B * frames, Cx.shape[2]
        L = lib()
        out = torch.empty(3, dtype=torch.float64, device=x.device)
        ws = torch.empty(3 * L.mfpa_loss_blocks(), dtype=torch.float64, device=x.device)
        check(L.mfpa_stft_loss_sums(ptr(Cx), ptr(Cy), rows, bins, npad, im_off, ptr(out), ptr(ws), stream()), "mfpa_stft_loss_sums")
        check(L.mfpa_stft_loss_grad(ptr(Cx), ptr(Cy), rows, bins, npad, im_off, ptr(out), float(w_sc), float(w_mag), stream()),
              "mfpa_stft_loss_grad")
        Wt = _dft_matrix_t(fs, wl, x.device)
        kp = Wt.shape[0]
        dfr = torch.empty((B, frames, kp), dtype=torch.float32, device=x.device)
        gemm(ptr(Cx), npad, frames * npad, B, frames, Wt, None, kp, ptr(dfr), kp, frames * kp, precision=self.precision)
        pad, off = fs // 2, (fs - wl) // 2
        Lp = T + 2 * pad
        dxp = torch.empty((B, Lp), dtype=torch.float32, device=x.device)
        check(L.mfpa_frames_adjoint(ptr(dfr), B, frames, kp, wl, hop, off, Lp, ptr(dxp), stream()), "mfpa_frames_adjoint")
        check(L.mfpa_reflect_pad_adjoint(ptr(dxp), B, T, pad, Lp, int(accumulate), ptr(dx), stream()), "mfpa_reflect_pad_adjoint")
        sc = torch.sqrt(out[0]) / torch.sqrt(out[1])
        mag = out[2] / float(rows * bins)
        return sc.to(torch.float32), mag.to(torch.float32)


class MultiResolutionSTFTLoss(torch.nn.Module):
    """loss.py:128-186."""

    def __init__(self, fft_sizes: List[int] = [1024, 2048, 512], hop_sizes: List[int] = [120, 240, 50],
                 win_lengths: List[int] = [600, 1200, 240], window: str = "hann_window", factor_sc: float = 0.1,
                 factor_mag: float = 0.1, precision: int = 0) -> None:
        super().__init__()
        assert len(fft_sizes) == len(hop_sizes) == len(win_lengths)
        self.stft_losses = torch.nn.ModuleList([STFTLoss(fs, ss, wl, window, precision) for fs, ss, wl in zip(fft_sizes, hop_sizes, win_lengths)])
        self.factor_sc, self.factor_mag = factor_sc, factor_mag

    @torch.no_grad()
    def forward(self, x: torch.Tensor, y: torch.Tensor):
        sc_loss, mag_loss = 0.0, 0.0
        for f in self.stft_losses:
            sc_l, mag_l = f(x, y)
            sc_loss = sc_loss + sc_l
            mag_loss = mag_loss + mag_l
        sc_loss = sc_loss / len(self.stft_losses)
        mag_loss = mag_loss / len(self.stft_losses)
        return self.factor_sc * sc_loss, self.factor_mag * mag_loss

    @torch.no_grad()
    def target_transforms(self, y: torch.Tensor) -> List[torch.Tensor]:
        return [f.target_transform(y) for f in self.stft_losses]

    def value_and_grad(self, x: torch.Tensor, y: torch.Tensor, dx: Optional[torch.Tensor] = None, accumulate: bool = False,
                       Cys: Optional[List[torch.Tensor]] = None):
        """(sc_loss, mag_loss, d(sc_loss + mag_loss) / dx): the two terms training/train.py:297 adds to the L1 loss, and their
        gradient with respect to the predicted waveform x (what the reference's autograd hands to the Demucs backward pass).
        With `dx` given and `accumulate`, the gradient is ADDED to dx (which then already holds the L1 term's gradient)."""
        n = len(self.stft_losses)
        if dx is None:
            dx, accumulate = torch.empty_like(x, dtype=torch.float32), False
        sc_loss, mag_loss = 0.0, 0.0
        for i, f in enumerate(self.stft_losses):
            sc_l, mag_l = f.value_and_grad(x, y, self.factor_sc / n, self.factor_mag / n, dx, accumulate=accumulate or i > 0,
                                           Cy=None if Cys is None else Cys[i])
            sc_loss = sc_loss + sc_l
            mag_loss = mag_loss + mag_l
        return self.factor_sc * sc_loss / n, self.factor_mag * mag_loss / n, dx
