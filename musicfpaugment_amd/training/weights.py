"""UNet parameter inventory and closed-form ("formula") weights.

The published checkpoints of the reference are not in its tree and there is no
network, so tests and benchmarks use weights generated from an integer seed by a
counter-based hash (synth.uniform_u32): 124 MB of parameters never ship, yet
the reference module, the oracle and the HIP path can all be loaded with the
identical state_dict.  Key names and shapes follow training/unet.py:8-108 of the
reference (118 state_dict entries, 31,036,481 parameters).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Tuple

import numpy as np
import torch

from ..synth import uniform_u32

ENC = [("inc.double_conv", 1, 64), ("down1.maxpool_conv.1.double_conv", 64, 128),
       ("down2.maxpool_conv.1.double_conv", 128, 256), ("down3.maxpool_conv.1.double_conv", 256, 512),
       ("down4.maxpool_conv.1.double_conv", 512, 1024)]
DEC = [("up1", 1024, 512), ("up2", 512, 256), ("up3", 256, 128), ("up4", 128, 64)]


def state_dict_shapes(n_channels: int = 1, n_classes: int = 1) -> "OrderedDict[str, Tuple[int, ...]]":
    """Ordered {key: shape} exactly as UNet(n_channels, n_classes).state_dict() of the reference."""
    out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()

    def dconv(prefix, cin, cout):
        for idx, (ci, co) in ((0, (cin, cout)), (3, (cout, cout))):
            out[f"{prefix}.{idx}.weight"] = (co, ci, 3, 3)
            bn = f"{prefix}.{idx + 1}"
            out[bn + ".weight"] = (co,)
            out[bn + ".bias"] = (co,)
            out[bn + ".running_mean"] = (co,)
            out[bn + ".running_var"] = (co,)
            out[bn + ".num_batches_tracked"] = ()

    for prefix, cin, cout in ENC:
        dconv(prefix, n_channels if prefix.startswith("inc") else cin, cout)
    for name, cin, cout in DEC:
        out[name + ".up.weight"] = (cin, cin // 2, 2, 2)
        out[name + ".up.bias"] = (cin // 2,)
        dconv(name + ".conv.double_conv", cin, cout)
    out["outc.conv.weight"] = (n_classes, 64, 1, 1)
    out["outc.conv.bias"] = (n_classes,)
    return out


def _unit(seed: int, stream: int, n: int) -> np.ndarray:
    """Deterministic values in [-1, 1)."""
    return uniform_u32(seed, stream, n).astype(np.float64) * (2.0 / 4294967296.0) - 1.0


def formula_state_dict(seed: int = 0, n_channels: int = 1, n_classes: int = 1) -> "OrderedDict[str, torch.Tensor]":
    """He-scaled pseudo-random conv weights, near-identity BatchNorm statistics."""
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for stream, (key, shape) in enumerate(state_dict_shapes(n_channels, n_classes).items()):
        n = int(np.prod(shape)) if shape else 1
        u = _unit(seed, 100 + stream, n)
        if key.endswith("num_batches_tracked"):
            sd[key] = torch.tensor(0, dtype=torch.long)
            continue
        if len(shape) == 4:
            if ".up." in key:                      # ConvTranspose2d (Cin, Cout, 2, 2): one tap per output pixel
                fan_in = shape[0]
            else:
                fan_in = shape[1] * shape[2] * shape[3]
            v = u * np.sqrt(3.0) * np.sqrt(2.0 / fan_in)
        elif key.endswith("running_var"):
            v = 1.0 + 0.25 * np.abs(u)
        elif key.endswith("running_mean"):
            v = 0.05 * u
        elif key.endswith(".weight"):              # BN gamma
            v = 1.0 + 0.1 * u
        else:                                      # BN beta, conv biases
            v = 0.05 * u
        sd[key] = torch.from_numpy(v.astype(np.float32).reshape(shape))
    return sd


FAMILIES = ("formula", "bn_spread", "heavy_tail")


def stress_state_dict(family: str, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """Weight families that stress the bf16x3 arithmetic of the convolutions the way a trained checkpoint can (the 1e-4
    forward gate must not rest on one benign family; training/unet.py:97-108 has no output activation, so cancelling sums
    reach the output):
      "formula"     formula_state_dict(seed);
      "bn_spread"   BatchNorm gamma log-uniform over [0.03, 30] with a random sign-free spread per channel, running_var
                    log-uniform over [1e-2, 1e2], running_mean uniform in [-2, 2], beta in [-0.5, 0.5]: per-channel scales
                    gamma / sqrt(var) spread over five decades, so a few channels dominate every sum and ReLU thresholds sit
                    far from zero;
      "heavy_tail"  conv / transposed-conv weights = He scale x |Cauchy| (clipped at 20) x random sign: a handful of taps carry
                    most of each output, the rest cancel.
    (A third stressed family, weights after 50 optimiser steps of UNetTrainEngine, needs the GPU: tests/test_gpu_unet.py.)"""
    if family == "formula":
        return formula_state_dict(seed)
    if family not in FAMILIES:
        raise ValueError(f"unknown weight family {family!r}")
    sd = formula_state_dict(seed)
    for stream, (key, v) in enumerate(sd.items()):
        shape = tuple(v.shape)
        n = v.numel()
        if key.endswith("num_batches_tracked"):
            continue
        u = _unit(seed, 900 + stream, n)
        u2 = _unit(seed, 1900 + stream, n)
        if family == "bn_spread" and len(shape) == 1 and ".up." not in key and not key.startswith("outc"):
            if key.endswith("running_var"):
                w = 10.0 ** (2.0 * u)                                   # log-uniform [1e-2, 1e2]
            elif key.endswith("running_mean"):
                w = 2.0 * u
            elif key.endswith(".weight"):
                w = 10.0 ** (1.5 * u)                                   # log-uniform [0.03, 30] (x the layer's normaliser below)
            else:
                w = 0.5 * u
            sd[key] = torch.from_numpy(w.astype(np.float32).reshape(shape))
        elif family == "heavy_tail" and len(shape) == 4:
            fan_in = shape[0] if ".up." in key else shape[1] * shape[2] * shape[3]
            cauchy = np.minimum(np.abs(np.tan(0.5 * np.pi * np.clip(u, -0.999999, 0.999999))), 20.0)
            sign = np.where(u2 < 0, -1.0, 1.0)
            w = sign * cauchy * np.sqrt(2.0 / fan_in) / np.sqrt(np.mean(cauchy ** 2))      # He variance, heavy-tailed shape
            sd[key] = torch.from_numpy(w.astype(np.float32).reshape(shape))
    if family == "bn_spread":              # keep activations O(1): every BatchNorm layer's rms scale gamma / sqrt(var + eps) is 1
        for key in [k for k in sd if k.endswith("running_var")]:
            g = key[:-len("running_var")] + "weight"
            scale = sd[g].double() / torch.sqrt(sd[key].double() + 1e-5)
            sd[g] = (sd[g].double() / torch.sqrt((scale ** 2).mean())).float()
    return sd


def n_parameters(sd: Dict[str, torch.Tensor]) -> int:
    return sum(v.numel() for k, v in sd.items()
               if not (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked")))
