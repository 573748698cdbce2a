"""UNet building blocks over the C ABI (csrc/unet.hip): weight packing and the eval forward.

Activations are NHWC float32 tensors (B, H, W, C) -- H = frequency bins, W = frames.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from ._lib import check, lib, ptr, stream

BN_EPS = 1e-5  # nn.BatchNorm2d default (training/unet.py:17,20)

ENC = ["inc.double_conv", "down1.maxpool_conv.1.double_conv", "down2.maxpool_conv.1.double_conv",
       "down3.maxpool_conv.1.double_conv", "down4.maxpool_conv.1.double_conv"]
DEC = ["up1", "up2", "up3", "up4"]


def _fold_bn(sd, prefix):
    scale = sd[prefix + ".weight"].float() / torch.sqrt(sd[prefix + ".running_var"].float() + BN_EPS)
    shift = sd[prefix + ".bias"].float() - sd[prefix + ".running_mean"].float() * scale
    return scale.contiguous(), shift.contiguous()


def pack_conv3x3(w: torch.Tensor) -> torch.Tensor:
    """(Cout, Cin, 3, 3) -> [tap = ky*3+kx][Cout][Cin] (Cin contiguous: both MFMA operands K-contiguous)."""
    co, ci = w.shape[:2]
    return w.detach().float().permute(2, 3, 0, 1).reshape(9, co, ci).contiguous()


def pack_convT2x2(w: torch.Tensor) -> torch.Tensor:
    """(Cin, Cout, 2, 2) -> [tap = dy*2+dx][Cout][Cin]."""
    ci, co = w.shape[:2]
    return w.detach().float().permute(2, 3, 1, 0).reshape(4, co, ci).contiguous()


def split_bf16x3(w: torch.Tensor) -> torch.Tensor:
    """Kernel-layout fp32 weights [taps][Cout][Cin] -> the same-shaped float32 container whose every 32-channel
    chunk (128 bytes) holds [32 bf16 hi | 32 bf16 lo] with w = hi + lo (+ O(2^-17 w)): the LDS row format of the
    bf16x3 convolution (csrc/unet.hip, PREC 1)."""
    t, co, ci = w.shape
    w4 = w.reshape(t, co, ci // 32, 32)
    hi = w4.to(torch.bfloat16)
    lo = (w4 - hi.float()).to(torch.bfloat16)
    return torch.cat([hi, lo], dim=-1).contiguous().view(torch.float32).reshape(t, co, ci)


def pack_unet_weights(sd: Dict[str, torch.Tensor], precision: int = 0) -> Dict[str, torch.Tensor]:
    """precision 0: fp32 MFMA weights; 1: additionally pre-split (bf16x3) copies under '<name>.w3'."""
    pw: Dict[str, torch.Tensor] = {"precision": precision}

    def dconv(prefix, first_layer=False):
        w0 = sd[prefix + ".0.weight"]
        if first_layer:   # (Cout, 1, 3, 3) -> [tap][Cout]
            pw[prefix + ".0.w"] = w0.detach().float().permute(2, 3, 1, 0).reshape(9, w0.shape[0]).contiguous()
        else:
            pw[prefix + ".0.w"] = pack_conv3x3(w0)
        pw[prefix + ".0.scale"], pw[prefix + ".0.shift"] = _fold_bn(sd, prefix + ".1")
        pw[prefix + ".3.w"] = pack_conv3x3(sd[prefix + ".3.weight"])
        pw[prefix + ".3.scale"], pw[prefix + ".3.shift"] = _fold_bn(sd, prefix + ".4")

    for i, p in enumerate(ENC):
        dconv(p, first_layer=(i == 0))
    for name in DEC:
        pw[name + ".up.w"] = pack_convT2x2(sd[name + ".up.weight"])
        pw[name + ".up.b"] = sd[name + ".up.bias"].detach().float().contiguous()
        dconv(name + ".conv.double_conv")
    pw["outc.w"] = sd["outc.conv.weight"].detach().float().reshape(-1).contiguous()
    pw["outc.b"] = sd["outc.conv.bias"].detach().float().reshape(-1).contiguous()
    pw["outc.b_host"] = float(sd["outc.conv.bias"].detach().float().reshape(-1)[0].item())
    if precision == 1:
        for k in [k for k in pw if isinstance(pw[k], torch.Tensor) and pw[k].dim() == 3]:
            pw[k + "3"] = split_bf16x3(pw[k])
    return pw


# ----------------------------------------------------------------------------- kernels
class KernelTimer:
    """Optional HIP-event stopwatch around the MFMA convolution launches (bench.py's roofline leg).
    Events are recorded on the stream the kernels are launched on (torch's current stream)."""

    def __init__(self):
        self.pairs = []

    def start(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def stop(self, e0):
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.pairs.append((e0, e1))

    def total_ms(self) -> float:
        return float(sum(a.elapsed_time(b) for a, b in self.pairs))

    def launches(self) -> int:
        return len(self.pairs)


_TIMER = None


def set_timer(timer):
    global _TIMER
    _TIMER = timer


def conv3x3_bn_relu(x0, w, scale, shift, x1=None, relu=True, precision=0):
    B, H, W, C0 = x0.shape
    Cout = w.shape[1]
    if x1 is not None:
        _, H1, W1, C1 = x1.shape
    else:
        H1 = W1 = C1 = 0
    if w.shape != (9, Cout, C0 + C1):
        raise ValueError(f"weight shape {tuple(w.shape)} does not match input channels {C0}+{C1}")
    y = torch.empty((B, H, W, Cout), dtype=torch.float32, device=x0.device)
    t0 = _TIMER.start() if _TIMER is not None else None
    check(lib().mfpa_conv3x3_bn_relu(ptr(x0), C0, ptr(x1), C1, H1, W1, B, H, W, ptr(w), Cout, ptr(scale), ptr(shift),
                                     int(relu), precision, ptr(y), stream()), "mfpa_conv3x3_bn_relu")
    if t0 is not None:
        _TIMER.stop(t0)
    return y


def conv3x3_c1_bn_relu(w, scale, shift, x32=None, spec64=None, denom=None, per_clip=True, relu=True):
    src = x32 if x32 is not None else spec64
    B, H, W = src.shape
    Cout = w.shape[1]
    y = torch.empty((B, H, W, Cout), dtype=torch.float32, device=src.device)
    check(lib().mfpa_conv3x3_c1_bn_relu(ptr(x32), ptr(spec64), ptr(denom), int(per_clip), B, H, W, ptr(w), Cout,
                                        ptr(scale), ptr(shift), int(relu), ptr(y), stream()), "mfpa_conv3x3_c1_bn_relu")
    return y


def maxpool2(x):
    B, H, W, C = x.shape
    y = torch.empty((B, H // 2, W // 2, C), dtype=torch.float32, device=x.device)
    check(lib().mfpa_maxpool2(ptr(x), B, H, W, C, ptr(y), stream()), "mfpa_maxpool2")
    return y


def convT2x2(x, w, bias, precision=0):
    B, H, W, Cin = x.shape
    Cout = w.shape[1]
    if w.shape != (4, Cout, Cin):
        raise ValueError("transposed-conv weight shape mismatch")
    y = torch.empty((B, 2 * H, 2 * W, Cout), dtype=torch.float32, device=x.device)
    t0 = _TIMER.start() if _TIMER is not None else None
    check(lib().mfpa_convT2x2(ptr(x), B, H, W, Cin, ptr(w), ptr(bias), Cout, precision, ptr(y), stream()),
          "mfpa_convT2x2")
    if t0 is not None:
        _TIMER.stop(t0)
    return y


def conv1x1_out(x, w, bias: float):
    B, H, W, C = x.shape
    y = torch.empty((B, H, W), dtype=torch.float32, device=x.device)
    check(lib().mfpa_conv1x1_out(ptr(x), B * H * W, C, ptr(w), float(bias), ptr(y), stream()), "mfpa_conv1x1_out")
    return y


def unet_forward_eval(pw: Dict[str, torch.Tensor], x32: Optional[torch.Tensor] = None,
                      spec64: Optional[torch.Tensor] = None, denom: Optional[torch.Tensor] = None) -> torch.Tensor:
    """UNet.forward in eval mode (training/unet.py:97-108) on (B, F, T) -> (B, F, T) float32."""
    prec = int(pw.get("precision", 0))
    sfx = "3" if prec == 1 else ""

    def dconv(x, prefix, skip=None):
        if skip is None:
            m = conv3x3_bn_relu(x, pw[prefix + ".0.w" + sfx], pw[prefix + ".0.scale"], pw[prefix + ".0.shift"],
                                precision=prec)
        else:  # decoder: channels = [skip | upsampled], upsampled zero-padded bottom/right to the skip extent
            m = conv3x3_bn_relu(skip, pw[prefix + ".0.w" + sfx], pw[prefix + ".0.scale"], pw[prefix + ".0.shift"], x1=x,
                                precision=prec)
        return conv3x3_bn_relu(m, pw[prefix + ".3.w" + sfx], pw[prefix + ".3.scale"], pw[prefix + ".3.shift"],
                               precision=prec)

    p = ENC[0]
    m = conv3x3_c1_bn_relu(pw[p + ".0.w"], pw[p + ".0.scale"], pw[p + ".0.shift"], x32=x32, spec64=spec64, denom=denom)
    x1 = conv3x3_bn_relu(m, pw[p + ".3.w" + sfx], pw[p + ".3.scale"], pw[p + ".3.shift"], precision=prec)
    del m
    x2 = dconv(maxpool2(x1), ENC[1])
    x3 = dconv(maxpool2(x2), ENC[2])
    x4 = dconv(maxpool2(x3), ENC[3])
    x5 = dconv(maxpool2(x4), ENC[4])
    y = dconv(convT2x2(x5, pw["up1.up.w" + sfx], pw["up1.up.b"], precision=prec), "up1.conv.double_conv", skip=x4)
    del x5, x4
    y = dconv(convT2x2(y, pw["up2.up.w" + sfx], pw["up2.up.b"], precision=prec), "up2.conv.double_conv", skip=x3)
    del x3
    y = dconv(convT2x2(y, pw["up3.up.w" + sfx], pw["up3.up.b"], precision=prec), "up3.conv.double_conv", skip=x2)
    del x2
    y = dconv(convT2x2(y, pw["up4.up.w" + sfx], pw["up4.up.b"], precision=prec), "up4.conv.double_conv", skip=x1)
    del x1
    return conv1x1_out(y, pw["outc.w"], pw["outc.b_host"])
