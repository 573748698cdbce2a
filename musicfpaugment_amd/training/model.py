"""Demucs causal waveform denoiser on MI355X -- host-side mirror of the reference's training/model.py:163-326
(next-tier row SURVEY.md §8f-2, BASELINE config 5).

Same constructor defaults, module tree and state_dict keys (encoder.N.0/2, decoder.N.0/2, lstm.lstm.*), same
``forward((B, T) or (B, 1, T)) -> (B, 1, T)``, ``valid_length`` and ``total_stride``.  The torch layers are parameter
containers; forward runs the gfx950 kernels of csrc/demucs.hip (fp32 MFMA GEMMs over time-major activations).
The module has no train/eval-dependent layer, so `forward` is the same in both modes; the training STEP (loss, backward,
Adam -- training/train.py:275-312) is `ops_demucs_train.DemucsTrainEngine`, driven by `training.train.Trainer(input_type=
"audio")`, not torch autograd.  DemucsStreamer is not built.
"""
from __future__ import annotations

import math
from typing import Any

import torch
from torch import nn

from .. import ops_demucs as D
from .._lib import MfpaError, require_gpu


class BLSTM(nn.Module):
    def __init__(self, dim: int, layers: int = 2, bi: bool = True) -> None:
        super().__init__()
        if bi:
            raise NotImplementedError("the reference instantiates Demucs(causal=True): unidirectional LSTM only")
        self.lstm = nn.LSTM(bidirectional=False, num_layers=layers, hidden_size=dim, input_size=dim)
        self.linear = None


class Demucs(nn.Module):
    def __init__(self, chin: int = 1, chout: int = 1, hidden: int = 48, depth: int = 5, kernel_size: int = 8,
                 stride: int = 4, causal: bool = True, resample: int = 4, growth: int = 2, max_hidden: int = 10000,
                 normalize: bool = True, glu: bool = True, rescale: float = 0.1, floor: float = 1e-3,
                 sample_rate: int = 8000) -> None:
        super().__init__()
        if (chin, chout, hidden, depth, kernel_size, stride, causal, resample, growth, normalize, glu) != \
                (1, 1, 48, 5, 8, 4, True, 4, 2, True, True):
            raise NotImplementedError("only the configuration the reference uses (Demucs() defaults) is built")
        self.chin, self.chout, self.hidden, self.depth = chin, chout, hidden, depth
        self.kernel_size, self.stride, self.causal, self.floor = kernel_size, stride, causal, floor
        self.resample, self.normalize, self.sample_rate = resample, normalize, sample_rate
        self.encoder, self.decoder = nn.ModuleList(), nn.ModuleList()
        for index in range(depth):
            self.encoder.append(nn.Sequential(nn.Conv1d(chin, hidden, kernel_size, stride), nn.ReLU(),
                                              nn.Conv1d(hidden, hidden * 2, 1), nn.GLU(1)))
            decode = [nn.Conv1d(hidden, 2 * hidden, 1), nn.GLU(1), nn.ConvTranspose1d(hidden, chout, kernel_size, stride)]
            if index > 0:
                decode.append(nn.ReLU())
            self.decoder.insert(0, nn.Sequential(*decode))
            chout, chin = hidden, hidden
            hidden = min(int(growth * hidden), max_hidden)
        self.lstm = BLSTM(chin, bi=not causal)
        if rescale:                                             # model.py:113-124
            for sub in self.modules():
                if isinstance(sub, (nn.Conv1d, nn.ConvTranspose1d)):
                    std = sub.weight.std().detach()
                    scale = (std / rescale) ** 0.5
                    sub.weight.data /= scale
                    if sub.bias is not None:
                        sub.bias.data /= scale
        self._packed = None
        self._packed_key = None
        self.precision = 1      # GEMM arithmetic: 0 = exact fp32 MFMA products, 1 = bf16x3 (relative L1 ~1e-5 vs fp32)

    def valid_length(self, length: float) -> int:
        return D.valid_length(int(math.ceil(length)) if not isinstance(length, int) else length)

    @property
    def total_stride(self) -> Any:
        return self.stride ** self.depth // self.resample

    def packed_weights(self):
        key = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._packed is None or key != self._packed_key:
            dev = next(self.parameters()).device
            self._packed = D.pack_demucs_weights(self.state_dict(), dev)
            self._packed_key = key
        return self._packed

    @torch.no_grad()
    def forward(self, mix: torch.Tensor) -> torch.Tensor:
        require_gpu(mix, "Demucs input")
        if mix.dim() == 3:
            if mix.shape[1] != 1:
                raise ValueError("expected (B, 1, T)")
            mix = mix[:, 0]
        if mix.dim() != 2 or mix.dtype != torch.float32:
            raise ValueError("expected float32 (B, T) or (B, 1, T)")
        if next(self.parameters()).device != mix.device:
            raise MfpaError("model and input must be on the same GPU")
        return D.demucs_forward(self.packed_weights(), mix.contiguous(), precision=self.precision).unsqueeze(1)
