"""Oracle: UNet spectrogram denoiser, torch-CPU float32 (reference row a11/a12 of SURVEY.md §8a).

A functional restatement of training/unet.py:8-108 driven by a state_dict with
the reference's 118 key names, so the same weights feed the reference module,
this oracle and the HIP path.  Kept in torch because it is a floating-point
kernel oracle (tolerance: relative L1 <= 1e-4, BASELINE.json).
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # nn.BatchNorm2d default, training/unet.py:17,20
BN_MOMENTUM = 0.1


def _bn_relu(x, sd, prefix, training: bool, stats_out=None):
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    if training:
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        if stats_out is not None:
            n = x.numel() // x.shape[1]
            stats_out[prefix] = (mean.detach(), (var * n / max(n - 1, 1)).detach())
    else:
        mean, var = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    y = (x - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + BN_EPS)
    y = y * w[None, :, None, None] + b[None, :, None, None]
    return F.relu(y)


def double_conv(x, sd, prefix, training=False, stats_out=None):
    """training/unet.py:8-25: (conv3x3 pad 1 no bias -> BN -> ReLU) x 2."""
    x = F.conv2d(x, sd[prefix + ".0.weight"], padding=1)
    x = _bn_relu(x, sd, prefix + ".1", training, stats_out)
    x = F.conv2d(x, sd[prefix + ".3.weight"], padding=1)
    return _bn_relu(x, sd, prefix + ".4", training, stats_out)


def down(x, sd, name, training=False, stats_out=None):
    """training/unet.py:28-38: MaxPool2d(2) (floor) then DoubleConv."""
    return double_conv(F.max_pool2d(x, 2), sd, name + ".maxpool_conv.1.double_conv", training, stats_out)


def up(x1, x2, sd, name, training=False, stats_out=None):
    """training/unet.py:41-65: ConvTranspose2d k2 s2 (+bias), zero-pad to the skip size
    (extra row/col goes to the bottom/right), concat [skip, up] on channels, DoubleConv."""
    x1 = F.conv_transpose2d(x1, sd[name + ".up.weight"], sd[name + ".up.bias"], stride=2)
    dy = x2.shape[2] - x1.shape[2]
    dx = x2.shape[3] - x1.shape[3]
    x1 = F.pad(x1, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
    return double_conv(torch.cat([x2, x1], dim=1), sd, name + ".conv.double_conv", training, stats_out)


def forward(x: torch.Tensor, sd: Dict[str, torch.Tensor], training: bool = False, stats_out=None,
            dropout_masks=None) -> torch.Tensor:
    """UNet.forward, training/unet.py:97-108.  nn.Dropout sits on x2..x5 and on up1's output (:99-103); it is the
    identity in eval mode.  In training mode the five multiplicative masks (already scaled by 1/(1-rate)) are passed
    explicitly as dropout_masks = [m2, m3, m4, m5, m_up1] (torch's own Philox stream cannot be reproduced elsewhere)."""
    dm = dropout_masks if dropout_masks is not None else [1.0] * 5
    x1 = double_conv(x, sd, "inc.double_conv", training, stats_out)
    x2 = down(x1, sd, "down1", training, stats_out) * dm[0]
    x3 = down(x2, sd, "down2", training, stats_out) * dm[1]
    x4 = down(x3, sd, "down3", training, stats_out) * dm[2]
    x5 = down(x4, sd, "down4", training, stats_out) * dm[3]
    y = up(x5, x4, sd, "up1", training, stats_out) * dm[4]
    y = up(y, x3, sd, "up2", training, stats_out)
    y = up(y, x2, sd, "up3", training, stats_out)
    y = up(y, x1, sd, "up4", training, stats_out)
    return F.conv2d(y, sd["outc.conv.weight"], sd["outc.conv.bias"])


def relative_l1(y: torch.Tensor, ref: torch.Tensor) -> float:
    """sum|y - ref| / sum|ref| in float64 (BASELINE.json tolerance 1e-4)."""
    y = y.double()
    ref = ref.double()
    return float((y - ref).abs().sum() / ref.abs().sum())


def forward_bf16x3_model(x: torch.Tensor, sd: Dict[str, torch.Tensor]) -> torch.Tensor:
    """Arithmetic MODEL of the device's bf16x3 convolutions (csrc/unet.hip, PREC 1) on the CPU, eval mode: every fp32 operand of
    the MFMA layers is split x = hi + lo into two bfloat16 values and a product is hi*hi + hi*lo + lo*hi (the lo*lo term is
    dropped), accumulated wide; the 1-channel first layer and the 1x1 OutConv stay fp32 (they run on the vector ALUs).  Test
    infrastructure: lets the CPU suite bound the error of the headline arithmetic on any weight family before a GPU is involved
    (tests/test_oracle_golden.py); the device itself is compared with forward() in tests/test_gpu_unet.py."""
    conv2d, conv_t = F.conv2d, F.conv_transpose2d

    def split(t):
        hi = t.to(torch.bfloat16).float()
        return hi.double(), (t - hi).to(torch.bfloat16).double()

    def conv3(inp, w, *a, **k):
        if w.shape[1] == 1 or w.shape[2] == 1:
            return conv2d(inp, w, *a, **k)
        (xh, xl), (wh, wl) = split(inp), split(w)
        return (conv2d(xh, wh, *a, **k) + conv2d(xh, wl, *a, **k) + conv2d(xl, wh, *a, **k)).float()

    def convt3(inp, w, b, **k):
        (xh, xl), (wh, wl) = split(inp), split(w)
        return (conv_t(xh, wh, None, **k) + conv_t(xh, wl, None, **k) + conv_t(xl, wh, None, **k)).float() + b[None, :, None, None]

    F.conv2d, F.conv_transpose2d = conv3, convt3
    try:
        return forward(x, sd)
    finally:
        F.conv2d, F.conv_transpose2d = conv2d, conv_t
