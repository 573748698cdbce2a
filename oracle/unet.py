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


# ---------------------------------------------------------------------------------------------------------------------------------
# Round 6: the algebra behind mfpa_upconv_fused (csrc/unet_up.hip), restated on the CPU.  Up.forward (training/unet.py:58-65) is
#     up = ConvTranspose2d(k 2, s 2)(low) + bt;  pad to the skip's size (diff // 2 in front);  z = conv3x3(cat([skip, up]), W3, padding 1)
# and nothing non-linear sits between the two convolutions, so the `up` half of z is, per output phase (Y & 1, X & 1), a 2 x 2 convolution of
# `low` with COMPOSITE weights plus a bias that depends only on which of the nine taps fall inside the up-sampled extent.  These functions are
# test infrastructure: tests/test_oracle_upconv.py checks them against the reference formulation above in float64 (odd sizes, borders), and
# tests/test_gpu_upconv.py checks the device's mfpa_upconv_pack against them.


def upconv_composite(w3: torch.Tensor, wt: torch.Tensor, bt: torch.Tensor, scale: torch.Tensor = None):
    """w3 (Cout, Cs + Cu, 3, 3) the level's first Conv2d weight, wt (Cl, Cu, 2, 2) / bt (Cu) the ConvTranspose2d's, scale (Cout) an optional
    per-output-channel factor (the folded eval BatchNorm scale).  Returns float64
      wc   (16, Cout, Cl): index ((py * 2 + px) * 2 + ty) * 2 + tx = output phase (py, px), low-resolution tap (ty - 1 + py, tx - 1 + px);
      bias (4, 4, Cout):   by (row class, column class) of the output pixel -- 0: first row / column of the up-sampled extent (tap -1 outside),
                           1: interior, 2: its last (tap +1 outside), 3: the padding row / column of an odd size (only tap -1 inside)."""
    dd = torch.float64
    w3, wt, bt = w3.to(dd), wt.to(dd), bt.to(dd)
    Cout, Cu, Cl = w3.shape[0], wt.shape[1], wt.shape[0]
    w3u = w3[:, w3.shape[1] - Cu:]                                        # (Cout, Cu, 3, 3): the half that multiplies `up`
    wc = torch.zeros(16, Cout, Cl, dtype=dd)
    for py in range(2):
        for px in range(2):
            for a in (-1, 0, 1):
                for b in (-1, 0, 1):
                    ry, rx = (py + a) // 2, (px + b) // 2                 # low-resolution offset of up-sampled pixel (Y + a, X + b) (floor division)
                    ty, tx = ry + 1 - py, rx + 1 - px
                    t16 = ((py * 2 + px) * 2 + ty) * 2 + tx
                    wc[t16] += w3u[:, :, a + 1, b + 1] @ wt[:, :, (py + a) % 2, (px + b) % 2].T       # (Cout, Cu) @ (Cu, Cl)
    inside = {0: (0, 1), 1: (-1, 0, 1), 2: (-1, 0), 3: (-1,)}
    bias = torch.zeros(4, 4, Cout, dtype=dd)
    for rc in range(4):
        for cc in range(4):
            for a in inside[rc]:
                for b in inside[cc]:
                    bias[rc, cc] += w3u[:, :, a + 1, b + 1] @ bt
    if scale is not None:
        wc = wc * scale.to(dd)[None, :, None]
        bias = bias * scale.to(dd)[None, None, :]
    return wc, bias


def upconv_composite_forward(skip: torch.Tensor, low: torch.Tensor, w3: torch.Tensor, wc: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """z of the header comment from the composite operands (float64): conv3x3 of the skip half + the four-phase 2 x 2 convolution of `low` + the
    border-class bias.  skip (B, Cs, H, W), low (B, Cl, Hl, Wl) with H - 2 Hl and W - 2 Wl in {0, 1}."""
    dd = torch.float64
    skip, low, w3 = skip.to(dd), low.to(dd), w3.to(dd)
    B, Cs, H, W = skip.shape
    Hl, Wl = low.shape[2], low.shape[3]
    assert 0 <= H - 2 * Hl <= 1 and 0 <= W - 2 * Wl <= 1
    z = F.conv2d(skip, w3[:, :Cs], padding=1)
    lp = F.pad(low, [1, 1, 1, 1])                                         # zero padding of the low-resolution tensor = that of the up-sampled one
    cls = lambda v, n2: 0 if v == 0 else (1 if v < n2 - 1 else (2 if v == n2 - 1 else 3))
    rcls = torch.tensor([cls(y, 2 * Hl) for y in range(H)])
    ccls = torch.tensor([cls(x, 2 * Wl) for x in range(W)])
    z = z + bias[rcls][:, ccls].permute(2, 0, 1)[None]                    # (H, W, Cout) -> (1, Cout, H, W)
    for py in range(2):
        for px in range(2):
            ys, xs = torch.arange(py, H, 2), torch.arange(px, W, 2)       # output pixels of this phase; low-resolution pixel (Y // 2, X // 2)
            acc = 0
            for ty in range(2):
                for tx in range(2):
                    t16 = ((py * 2 + px) * 2 + ty) * 2 + tx
                    # padded low-resolution index of (Y // 2 + ty - 1 + py, X // 2 + tx - 1 + px): + 1 for the padding (a padding row / column
                    # of the output maps one past the tensor: Y // 2 = Hl -> padded index Hl + ty + py <= Hl + 1 only when ty + py <= 1; beyond: zero)
                    yi = (ys // 2 + ty + py).clamp(max=Hl + 1)
                    xi = (xs // 2 + tx + px).clamp(max=Wl + 1)
                    ok = ((ys // 2 + ty + py) <= Hl + 1)[:, None] & ((xs // 2 + tx + px) <= Wl + 1)[None, :]
                    patch = lp[:, :, yi][:, :, :, xi] * ok[None, None].to(dd)
                    acc = acc + torch.einsum("oc,bcyx->boyx", wc[t16], patch)
            z[:, :, py::2, px::2] += acc
    return z
