"""Oracle: Demucs causal waveform denoiser forward, torch-CPU float32.

Functional restatement of training/model.py:22-110,163-326 of the reference driven by a state_dict with the
reference's key names: std normalisation, zero-pad to valid_length, sinc x4 upsampling, 5 x [Conv1d(k8,s4)+ReLU+
Conv1d(1x1)+GLU], 2-layer unidirectional LSTM(768), mirrored decoder with skip additions and ConvTranspose1d(k8,s4),
sinc x4 downsampling, trim, rescale by std."""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

DEPTH, KERNEL, STRIDE, RESAMPLE, FLOOR, ZEROS = 5, 8, 4, 4, 1e-3, 56


def sinc_kernel(zeros: int = ZEROS) -> torch.Tensor:
    """kernel_upsample2 == kernel_downsample2 (model.py:28-38,56-66): windowed sinc at the half-sample offsets."""
    win = torch.hann_window(4 * zeros + 1, periodic=False)
    winodd = win[1::2]
    t = torch.linspace(-zeros + 0.5, zeros - 0.5, 2 * zeros) * math.pi
    return (torch.sin(t) / t * winodd).view(1, 1, -1)


def upsample2(x: torch.Tensor) -> torch.Tensor:
    """model.py:41-53: interleave x with its half-sample sinc interpolation."""
    *other, time = x.shape
    out = F.conv1d(x.reshape(-1, 1, time), sinc_kernel().to(x), padding=ZEROS)[..., 1:].view(*other, time)
    return torch.stack([x, out], dim=-1).view(*other, -1)


def downsample2(x: torch.Tensor) -> torch.Tensor:
    """model.py:69-88."""
    if x.shape[-1] % 2 != 0:
        x = F.pad(x, (0, 1))
    xeven, xodd = x[..., ::2], x[..., 1::2]
    *other, time = xodd.shape
    out = xeven + F.conv1d(xodd.reshape(-1, 1, time), sinc_kernel().to(x), padding=ZEROS)[..., :-1].view(*other, time)
    return out.view(*other, -1).mul(0.5)


def valid_length(length: int) -> int:
    """model.py:269-285."""
    length = math.ceil(length * RESAMPLE)
    for _ in range(DEPTH):
        length = max(math.ceil((length - KERNEL) / STRIDE) + 1, 1)
    for _ in range(DEPTH):
        length = (length - 1) * STRIDE + KERNEL
    return int(math.ceil(length / RESAMPLE))


def lstm(x: torch.Tensor, sd: Dict[str, torch.Tensor]) -> torch.Tensor:
    """nn.LSTM(768, 768, num_layers=2), unidirectional, zero initial state (model.py:91-110 with bi=False).
    x: (T, B, 768).  Gate order i, f, g, o."""
    for layer in range(2):
        wih, whh = sd[f"lstm.lstm.weight_ih_l{layer}"], sd[f"lstm.lstm.weight_hh_l{layer}"]
        b = sd[f"lstm.lstm.bias_ih_l{layer}"] + sd[f"lstm.lstm.bias_hh_l{layer}"]
        xin = x @ wih.t() + b
        h = x.new_zeros(x.shape[1], 768)
        c = x.new_zeros(x.shape[1], 768)
        outs = []
        for t in range(x.shape[0]):
            g = xin[t] + h @ whh.t()
            i, f, gg, o = g.chunk(4, dim=1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
            h = torch.sigmoid(o) * torch.tanh(c)
            outs.append(h)
        x = torch.stack(outs)
    return x


def forward(mix: torch.Tensor, sd: Dict[str, torch.Tensor]) -> torch.Tensor:
    """Demucs.forward, model.py:290-326: (B, T) or (B, 1, T) -> (B, 1, T)."""
    if mix.dim() == 2:
        mix = mix.unsqueeze(1)
    mono = mix.mean(dim=1, keepdim=True)
    std = mono.std(dim=-1, keepdim=True)
    x = mix / (FLOOR + std)
    length = mix.shape[-1]
    x = F.pad(x, (0, valid_length(length) - length))
    x = upsample2(upsample2(x))
    skips = []
    for i in range(DEPTH):
        x = F.relu(F.conv1d(x, sd[f"encoder.{i}.0.weight"], sd[f"encoder.{i}.0.bias"], stride=STRIDE))
        x = F.glu(F.conv1d(x, sd[f"encoder.{i}.2.weight"], sd[f"encoder.{i}.2.bias"]), dim=1)
        skips.append(x)
    x = lstm(x.permute(2, 0, 1), sd).permute(1, 2, 0)
    for d in range(DEPTH):
        skip = skips.pop(-1)
        x = x + skip[..., : x.shape[-1]]
        x = F.glu(F.conv1d(x, sd[f"decoder.{d}.0.weight"], sd[f"decoder.{d}.0.bias"]), dim=1)
        x = F.conv_transpose1d(x, sd[f"decoder.{d}.2.weight"], sd[f"decoder.{d}.2.bias"], stride=STRIDE)
        if d < DEPTH - 1:
            x = F.relu(x)
    x = downsample2(downsample2(x))
    return std * x[..., :length]
