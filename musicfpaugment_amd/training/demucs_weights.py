"""Demucs parameter inventory and formula weights (reference: training/model.py:163-326, defaults:
hidden 48, depth 5, kernel 8, stride 4, causal LSTM, resample 4, GLU).  18,867,937 parameters; the published
checkpoint is not in the reference tree, so tests/benchmarks use deterministic hash-generated weights whose per-tensor
standard deviations follow the reference's (rescaled) initialisation."""
from __future__ import annotations

from collections import OrderedDict
from typing import Tuple

import numpy as np
import torch

from ..synth import uniform_u32

HIDDEN, DEPTH, KERNEL, STRIDE, RESAMPLE, FLOOR = 48, 5, 8, 4, 4, 1e-3
CHANNELS = [HIDDEN * 2 ** i for i in range(DEPTH)]        # 48, 96, 192, 384, 768


def state_dict_shapes() -> "OrderedDict[str, Tuple[Tuple[int, ...], float]]":
    """Ordered {key: (shape, init std)} exactly as Demucs().state_dict() of the reference."""
    out: "OrderedDict[str, Tuple[Tuple[int, ...], float]]" = OrderedDict()
    enc_std = [(0.1435, 0.0909), (0.0543, 0.0767), (0.0456, 0.0646), (0.0384, 0.0543), (0.0323, 0.0456)]
    chin = 1
    for i, h in enumerate(CHANNELS):
        out[f"encoder.{i}.0.weight"] = ((h, chin, KERNEL), enc_std[i][0]); out[f"encoder.{i}.0.bias"] = ((h,), enc_std[i][0])
        out[f"encoder.{i}.2.weight"] = ((2 * h, h, 1), enc_std[i][1]); out[f"encoder.{i}.2.bias"] = ((2 * h,), enc_std[i][1])
        chin = h
    for d in range(DEPTH):                                  # decoder.0 is the deepest
        h = CHANNELS[DEPTH - 1 - d]
        cout = CHANNELS[DEPTH - 2 - d] if d < DEPTH - 1 else 1
        s1, s0 = enc_std[DEPTH - 1 - d][1], enc_std[DEPTH - 1 - d][0]
        out[f"decoder.{d}.0.weight"] = ((2 * h, h, 1), s1); out[f"decoder.{d}.0.bias"] = ((2 * h,), s1)
        out[f"decoder.{d}.2.weight"] = ((h, cout, KERNEL), s0); out[f"decoder.{d}.2.bias"] = ((cout,), s0)
    for layer in range(2):
        for nm in ("weight_ih", "weight_hh"):
            out[f"lstm.lstm.{nm}_l{layer}"] = ((4 * 768, 768), 0.0208)
        for nm in ("bias_ih", "bias_hh"):
            out[f"lstm.lstm.{nm}_l{layer}"] = ((4 * 768,), 0.0208)
    # reorder the LSTM keys like torch: weight_ih, weight_hh, bias_ih, bias_hh per layer (already so)
    return out


def formula_state_dict(seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for stream, (key, (shape, std)) in enumerate(state_dict_shapes().items()):
        n = int(np.prod(shape))
        u = uniform_u32(seed, 500 + stream, n).astype(np.float64) * (2.0 / 4294967296.0) - 1.0
        sd[key] = torch.from_numpy((u * np.sqrt(3.0) * std).astype(np.float32).reshape(shape))
    return sd
