"""UNet building blocks over the C ABI (csrc/unet.hip): weight packing and the eval forward.

Activations are NHWC float32 tensors (B, H, W, C) -- H = frequency bins, W = frames.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

import ctypes

from ._lib import ConvDesc, UpconvDesc, check, lib, ptr, stream

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
    """Kernel-layout fp32 weights [taps][Cout][Cin] -> the bf16x3 operand image of the convolution kernels (csrc/unet.hip, PREC 1):
    [tap][chunk = Cin / 32][Cout][128 bytes], a row = the 32 channels of a chunk split w = hi + lo (+ O(2^-17 w)) into 8 slots
    of 16 bytes, logical slots 0-3 = 32 bf16 hi, 4-7 = 32 bf16 lo, stored at physical slot (logical ^ ((row >> 1) & 7)) -- a
    (tap, chunk, 128-row) tile is 16 KB contiguous and goes into LDS verbatim by LDS-DMA, the XOR keeps the fragment reads
    bank-conflict-free.  Returned as an opaque float32 tensor of the input's shape (same byte count)."""
    t, co, ci = w.shape
    w4 = w.reshape(t, co, ci // 32, 32)
    hi = w4.to(torch.bfloat16)
    lo = (w4 - hi.float()).to(torch.bfloat16)
    rows = torch.cat([hi, lo], dim=-1).reshape(t, co, ci // 32, 8, 8)             # [t][row][chunk][logical slot][8 bf16]
    swz = (torch.arange(co, device=w.device) >> 1) & 7
    phys = torch.arange(8, device=w.device)[None, :] ^ swz[:, None]                # physical slot p of row r holds logical p ^ swz(r)
    idx = phys[None, :, None, :, None].expand(t, co, ci // 32, 8, 8)
    img = torch.gather(rows, 3, idx).permute(0, 2, 1, 3, 4).contiguous()           # [t][chunk][row][slot][8]
    return img.view(torch.float32).reshape(t, co, ci)


def frag_layout() -> int:
    """Which fragment-ordered image the library's weights-direct kernels read (mfpa_conv_weight_layout on a representative shape):
    2 = the 16 x 16 x 32 form (conv_wd16_kernel), 1 = the 32 x 32 x 16 form (BDIR), 0 = none."""
    return int(lib().mfpa_conv_weight_layout(128, 125, 128, 128, 0, 1))


def split_bf16x3_frag(w: torch.Tensor, layout: int = 2) -> torch.Tensor:
    """Kernel-layout fp32 weights [taps][Cout][Cin] (Cout % 32 == 0, Cin % 32 == 0) -> a FRAGMENT-ORDERED bf16x3 image of the
    "weights direct" convolution kernels (csrc/unet.hip; mfpa_conv_desc.w_layout = `layout`): a wave reads the MFMA weight operand of a
    column tile as 1 KB contiguous pieces, one 16-byte fragment per lane.  Same split w = hi + lo as split_bf16x3.
      layout 2 (v_mfma_f32_16x16x32_bf16, conv_wd16_kernel): [tap][chunk = Cin / 32][Cout / 16][hi | lo][lane 64][8 bf16], lane
               (g = l >> 4, c = l & 15) = output channel 16 t + c, input channels 32 chunk + 8 g .. + 7;
      layout 1 (v_mfma_f32_32x32x16_bf16, BDIR): [tap][chunk][Cout / 32][substep 2][hi | lo][lane 64][8 bf16], lane (lh = l >> 5,
               li = l & 31) = output channel 32 n + li, input channels 32 chunk + 16 substep + 8 lh .. + 7.
    Opaque float32 tensor of w's shape."""
    t, co, ci = w.shape
    if layout == 2:
        w6 = w.reshape(t, co // 16, 16, ci // 32, 4, 8)                            # [t][ct16][c][chunk][g][j]
        hi = w6.to(torch.bfloat16)
        lo = (w6 - hi.float()).to(torch.bfloat16)
        img = torch.stack([hi, lo], dim=0).permute(1, 4, 2, 0, 5, 3, 6).contiguous()    # [t][chunk][ct16][hl][g][c][j]
        return img.view(torch.float32).reshape(t, co, ci)
    if layout != 1:
        raise ValueError("layout must be 1 or 2")
    w6 = w.reshape(t, co // 32, 32, ci // 32, 2, 2, 8)                             # [t][n32][li][chunk][s][lh][j]
    hi = w6.to(torch.bfloat16)
    lo = (w6 - hi.float()).to(torch.bfloat16)
    img = torch.stack([hi, lo], dim=0).permute(1, 4, 2, 5, 0, 6, 3, 7).contiguous()    # [t][chunk][n32][s][hl][lh][li][j]
    return img.view(torch.float32).reshape(t, co, ci)


def frag_f32(w: torch.Tensor) -> torch.Tensor:
    """Kernel-layout fp32 weights [taps][Cout][Cin] (Cout % 16 == 0, Cin % 32 == 0) -> the FP32 fragment image of mfpa_upconv_fused(precision 0)
    (v_mfma_f32_16x16x4_f32): [tap][chunk = Cin / 32][Cout / 16][piece 2][lane 64][4 floats], lane (g = l >> 4, c = l & 15) = output channel
    16 t + c, input channels 32 chunk + 8 g + 4 piece .. + 3.  Same shape and bytes as w."""
    t, co, ci = w.shape
    w7 = w.reshape(t, co // 16, 16, ci // 32, 4, 2, 4)                                # [t][ct16][c][chunk][g][piece][j]
    return w7.permute(0, 3, 1, 5, 4, 2, 6).contiguous().reshape(t, co, ci)            # [t][chunk][ct16][piece][g][c][j]


def pack_unet_weights(sd: Dict[str, torch.Tensor], precision: int = 0) -> Dict[str, torch.Tensor]:
    """precision 0: fp32 MFMA weights; 1: additionally pre-split (bf16x3) copies under '<name>.w3', and for the layers the
    "weights direct" kernels can take (128-channel output tiles, >= 64 input channels) the fragment-ordered image '<name>.wf'."""
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
        lay = frag_layout()
        for k in [k for k in pw if isinstance(pw[k], torch.Tensor) and pw[k].dim() == 3]:
            pw[k + "3"] = split_bf16x3(pw[k])
            if lay and pw[k].shape[0] == 9 and pw[k].shape[1] % 64 == 0 and pw[k].shape[2] >= 64:
                pw[k + "f"] = (lay, split_bf16x3_frag(pw[k], lay))
                # the same image with the folded BatchNorm scale already IN the weights (w * scale[co], then split): where
                # mfpa_conv_scale_folds says so the launch passes it with out_scale = None and the kernel's epilogue is a bare ReLU
                if lay == 2 and k.endswith(".w"):
                    pw[k + "ff"] = (lay, split_bf16x3_frag(pw[k] * pw[k[:-2] + ".scale"][None, :, None], lay))
        if lay == 2 and FOLD_UP:
            for name in FOLD_UP_LEVELS:
                pw.update(pack_upconv(pw, name, 1))
    elif FOLD_UP and FOLD_UP_FP32:
        for name in FOLD_UP_LEVELS:
            pw.update(pack_upconv(pw, name, 0))
    return pw


FOLD_UP = True                # False: never fold a level's transposed convolution into its consumer (A/B runs)
FOLD_UP_LEVELS = ("up1", "up2", "up3", "up4")   # the decoder levels whose Up block runs as ONE launch (mfpa_upconv_fused); same-call pairs against
                                                # the two launches it replaces, 64 clips: +6 / +19 / +32 / +40 % (profiles/r06_upconv_levels.txt)


FOLD_UP_FP32 = True           # the fp32 MFMA path (precision 0) folds too (mfpa_upconv_fused(precision 0)); False: A/B runs


def pack_upconv(pw: Dict[str, torch.Tensor], name: str, precision: int = 1) -> Dict[str, torch.Tensor]:
    """Operands of mfpa_upconv_fused for decoder level `name` from the fp32 kernel-layout weights already in `pw`: the skip half of the
    level's first 3x3 convolution with the folded BatchNorm scale multiplied in (fragment image), the composite weights of its up half
    (ConvTranspose2d folded in: mfpa_upconv_pack on the device, float64 accumulation; fragment image as a 16-tap kernel) and the
    border-class bias table.  Reference: training/unet.py:41-65."""
    w3 = pw[name + ".conv.double_conv.0.w"]                        # [9][Cout][Cs + Cu]
    wt = pw[name + ".up.w"]                                        # [4][Cu][Cl]
    bt = pw[name + ".up.b"]
    scale = pw[name + ".conv.double_conv.0.scale"]
    Cout, Cu, Cl = w3.shape[1], wt.shape[1], wt.shape[2]
    Cs = w3.shape[2] - Cu
    if not w3.is_cuda:
        return {}                                                  # (CPU-side packing: the fused launch is simply not offered)
    wc, tab = upconv_pack_raw(w3, wt, bt, scale)
    wsk = (w3[:, :, :Cs] * scale[None, :, None]).contiguous()
    img = (lambda w_: split_bf16x3_frag(w_, 2)) if precision == 1 else frag_f32
    return {name + ".upc.wsk": img(wsk), name + ".upc.wup": img(wc), name + ".upc.bias": tab, name + ".upc.shape": (Cs, Cl, Cout)}


def upconv_pack_raw(w3: torch.Tensor, wt: torch.Tensor, bt: torch.Tensor, scale: Optional[torch.Tensor]):
    """mfpa_upconv_pack: kernel-layout w3 [9][Cout][Cs + Cu], wt [4][Cu][Cl], bt (Cu), scale (Cout) or None -> (composite weights (16, Cout, Cl),
    border-class bias table (4, 4, Cout)), float32 (float64 accumulation on the device)."""
    Cout, Cu, Cl = w3.shape[1], wt.shape[1], wt.shape[2]
    Cs = w3.shape[2] - Cu
    wc = torch.empty((16, Cout, Cl), dtype=torch.float32, device=w3.device)
    tab = torch.empty((4, 4, Cout), dtype=torch.float32, device=w3.device)
    check(lib().mfpa_upconv_pack(ptr(w3), ptr(wt), ptr(bt), ptr(scale), Cout, Cs, Cu, Cl, ptr(wc), ptr(tab), stream()), "mfpa_upconv_pack")
    return wc, tab


def upconv_fused(skip, low, w_skip, w_up, shift, bias_tab, Cout, relu=True, precision=1):
    """One decoder level's up -> pad -> cat -> conv3x3 + BN + ReLU (training/unet.py:58-65) as one launch; see include/mfpa.h."""
    B, H, W, Cs = skip.shape
    _, Hl, Wl, Cl = low.shape
    y = torch.empty((B, H, W, Cout), dtype=torch.float32, device=skip.device)
    d = UpconvDesc(skip=ptr(skip), low=ptr(low), w_skip=ptr(w_skip), w_up=ptr(w_up), shift=ptr(shift), bias_tab=ptr(bias_tab), y=ptr(y),
                   B=B, H=H, W=W, Cs=Cs, Hl=Hl, Wl=Wl, Cl=Cl, Cout=Cout, relu=int(relu), precision=int(precision))
    t0 = _TIMER.start() if _TIMER is not None else None
    check(lib().mfpa_upconv_fused(ctypes.byref(d), stream()), "mfpa_upconv_fused")
    if t0 is not None:
        _TIMER.stop(t0)
    return y


# ----------------------------------------------------------------------------- kernels
class KernelTimer:
    """Optional HIP-event stopwatch around the MFMA convolution launches (bench.py's roofline leg).
    Events are recorded on the stream the kernels are launched on (torch's current stream)."""

    def __init__(self, every: int = 1):
        """`every` > 1: only every `every`-th step (begin_step() calls) is timed -- an event pair costs 6-10 us of stream time around each launch
        (67 launches per train step: 0.4 ms of a 32 ms step), and the average launch duration does not need every step."""
        self.pairs = []
        self.every = max(1, int(every))
        self.steps = 0           # begin_step() calls
        self.sampled = 0         # ... of which timed
        self._on = True

    def begin_step(self):
        self._on = (self.steps % self.every) == 0
        self.steps += 1
        self.sampled += int(self._on)

    def start(self):
        if not self._on:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def stop(self, e0):
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.pairs.append((e0, e1))

    def total_ms(self) -> float:
        """Milliseconds inside the timed launches, scaled from the sampled steps to all steps (factor 1 without begin_step())."""
        raw = float(sum(a.elapsed_time(b) for a, b in self.pairs))
        return raw * (self.steps / self.sampled) if self.sampled else raw

    def launches(self) -> int:
        return len(self.pairs)


_TIMER = None
_SIDE = {}


def side_stream(dev):
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx not in _SIDE:
        _SIDE[idx] = torch.cuda.Stream(device=dev)
    return _SIDE[idx]


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


USE_WEIGHTS_DIRECT = True     # False: always the row image / LDS-staged weight tiles (A/B runs)
FOLD_SCALE = True             # False: never pass scale-folded weights (A/B runs)
C1_ON_MFMA = True             # False: the fused first layer stays on conv_mfma_kernel<C1SRC> (exact fp32 FMAs in its loader; A/B runs)
SPLIT_EDGES = True            # False: every tensor between two convolutions stays float32 (A/B runs)


def conv3x3_fused(x0, w, scale, shift, *, x1=None, precision=0, pool=False, out1x1=None, store=True, c1=None, wf=None, wff=None,
                  x0_split=False, x1_split=False, y_split=False, pool_split=False):
    """3x3 conv + folded BN + ReLU through mfpa_conv_mfma with optional fused epilogues: `pool` also writes
    MaxPool2d(2) of the output, `out1x1 = (w (64,), bias)` also writes the OutConv result (B,H,W); `store=False`
    skips the full-resolution output.  `w` must already be in the layout of `precision` (pre-split for bf16x3).
    `c1 = dict(x32= | spec64=, denom=, w, scale, shift)` (x0 None): the 64 input channels are the UNet's first layer, computed
    from the 1-channel input while the tile is staged (mfpa_conv_desc.c1_*).
    `wf` = (layout, image): a fragment-ordered image of the same weights (split_bf16x3_frag); used instead of `w` where
    mfpa_conv_weight_layout says the "weights direct" kernel reading that layout serves this shape.  `wff`: the same with `scale`
    folded into the weights, used (with out_scale = None) where mfpa_conv_scale_folds says the serving kernel prefers it.
    `x0_split` / `x1_split` / `y_split` / `pool_split` (mfpa_conv_desc.*_split): that tensor is / leaves in the SPLIT layout -- same shape
    and bytes, [32 bf16 hi | 32 bf16 lo] per 32-channel chunk of a pixel -- which only conv_ws64_kernel launches read and write
    (unet_forward_eval uses it on the edges between two such launches).
    Returns (y | None, y_pool | None, y1x1 | None)."""
    if c1 is not None:
        src = c1.get("x32") if c1.get("x32") is not None else c1["spec64"]
        (B, H, W), C0 = src.shape, 64
    else:
        B, H, W, C0 = x0.shape
    Cout = w.shape[1]
    C1 = 0 if x1 is None else x1.shape[3]
    dev = w.device
    layout = 0
    c1_ok = c1 is None or (C1_ON_MFMA and wf is not None and Cout == 64 and out1x1 is None and lib().mfpa_conv_c1_layout(H, W) == wf[0])
    if (wf is not None and USE_WEIGHTS_DIRECT and precision == 1 and c1_ok and (out1x1 is None or (wf[0] == 2 and Cout == 64))
            and lib().mfpa_conv_weight_layout(H, W, C0 + C1, Cout, 0, 1) == wf[0]):
        layout, w = wf
        if wff is not None and wff[0] == layout and FOLD_SCALE and lib().mfpa_conv_scale_folds(H, W, C0 + C1, Cout) == 1:
            w, scale = wff[1], None
    y = torch.empty((B, H, W, Cout), dtype=torch.float32, device=dev) if store else None
    yp = torch.empty((B, H // 2, W // 2, Cout), dtype=torch.float32, device=dev) if pool else None
    y1 = torch.empty((B, H, W), dtype=torch.float32, device=dev) if out1x1 is not None else None
    d = ConvDesc(x0=ptr(x0), in_scale0=0, in_shift0=0, x1=ptr(x1), w=ptr(w), out_scale=ptr(scale), out_shift=ptr(shift),
                 y=ptr(y), C0=C0, C1=C1, H1=0 if x1 is None else x1.shape[1], W1=0 if x1 is None else x1.shape[2],
                 B=B, H=H, W=W, Cout=Cout, relu=1, yH=H, yW=W, mode=0, drop_seed=0, drop_thresh=0, drop_scale=1.0,
                 precision=precision, y_pool=ptr(yp), w1x1=ptr(out1x1[0]) if out1x1 is not None else 0,
                 b1x1=float(out1x1[1]) if out1x1 is not None else 0.0, y1x1=ptr(y1), w_layout=layout,
                 x0_split=int(x0_split), x1_split=int(x1_split), y_split=int(y_split), y_pool_split=int(pool_split))
    if c1 is not None:
        d.c1_x32, d.c1_spec64, d.c1_denom = ptr(c1.get("x32")), ptr(c1.get("spec64")), ptr(c1.get("denom"))
        d.c1_w, d.c1_scale, d.c1_shift = ptr(c1["w"]), ptr(c1["scale"]), ptr(c1["shift"])
    t0 = _TIMER.start() if _TIMER is not None else None
    check(lib().mfpa_conv_mfma(ctypes.byref(d), stream()), "mfpa_conv_mfma")
    if t0 is not None:
        _TIMER.stop(t0)
    return y, yp, y1


def conv3x3_c1_bn_relu(w, scale, shift, x32=None, spec64=None, denom=None, per_clip=True, relu=True, out_dtype=torch.float32, stats_out=None):
    """`stats_out` (a list, training forward): the kernel's per-row partial BatchNorm statistics of the output, (B * H, 2, Cout) float32, are appended."""
    src = x32 if x32 is not None else spec64
    B, H, W = src.shape
    Cout = w.shape[1]
    y = torch.empty((B, H, W, Cout), dtype=out_dtype, device=src.device)
    part = None
    if stats_out is not None and (Cout <= 256 and 256 % Cout == 0):
        part = torch.empty((B * H, 2, Cout), dtype=torch.float32, device=src.device)
        stats_out.append(part)
    check(lib().mfpa_conv3x3_c1_bn_relu(ptr(x32), ptr(spec64), ptr(denom), int(per_clip), B, H, W, ptr(w), Cout,
                                        ptr(scale), ptr(shift), int(relu), ptr(y), int(out_dtype == torch.bfloat16), ptr(part), stream()), "mfpa_conv3x3_c1_bn_relu")
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


FUSE_FIRST_LAYER = True    # False: run mfpa_conv3x3_c1_bn_relu as its own launch (timing experiments, tiny images)


def unet_forward_eval(pw: Dict[str, torch.Tensor], x32: Optional[torch.Tensor] = None,
                      spec64: Optional[torch.Tensor] = None, denom: Optional[torch.Tensor] = None) -> torch.Tensor:
    """UNet.forward in eval mode (training/unet.py:97-108) on (B, F, T) -> (B, F, T) float32.
    The max-pools (unet.py:34) and the final OutConv (unet.py:71) run inside the epilogues of the convolutions that
    produce their inputs, so neither the pooled tensors' sources are re-read nor is up4's 64-channel output stored."""
    prec = int(pw.get("precision", 0))
    sfx = "3" if prec == 1 else ""

    def c(x, prefix, idx, **kw):
        return conv3x3_fused(x, pw[f"{prefix}.{idx}.w{sfx}"], pw[f"{prefix}.{idx}.scale"], pw[f"{prefix}.{idx}.shift"],
                             precision=prec, wf=pw.get(f"{prefix}.{idx}.wf") if prec == 1 else None,
                             wff=pw.get(f"{prefix}.{idx}.wff") if prec == 1 else None, **kw)

    def on_ws(prefix, idx, H_, W_, cin, cout):
        """Does this launch run on conv_ws64_kernel (the only reader / writer of the split layout)?"""
        wf_ = pw.get(f"{prefix}.{idx}.wf") if prec == 1 else None
        return (SPLIT_EDGES and USE_WEIGHTS_DIRECT and wf_ is not None and wf_[0] == 2
                and lib().mfpa_conv_weight_layout(H_, W_, cin, cout, 0, 1) == 2 and lib().mfpa_conv_scale_folds(H_, W_, cin, cout) == 1)

    def folds(name, H_, W_, Hl_, Wl_):
        """Does decoder level `name` run as mfpa_upconv_fused at this size?"""
        if not (FOLD_UP and name in FOLD_UP_LEVELS and (name + ".upc.wup") in pw):
            return False
        cs, cl, co = pw[name + ".upc.shape"]
        return lib().mfpa_upconv_serves(H_, W_, Hl_, Wl_, cs, cl, co) == 1

    p = ENC[0]
    skips = []
    src = x32 if x32 is not None else spec64
    H0, W0 = src.shape[1], src.shape[2]
    # Edges between two conv_ws64_kernel launches carry their tensor in the SPLIT layout (round 5): the producer splits each value once,
    # the consumer's loader waves only copy.  Every such tensor has exactly one consumer: inc -> up4.0 (the skip), inc's pool -> down1.0,
    # down1.0 -> down1.3, down1.3's pool -> down2.0, up4.0 -> up4.3.  (down1.3's un-pooled output feeds up3.0 on conv_wd16_kernel: float32.)
    fused_inc = FUSE_FIRST_LAYER and W0 > 16 and H0 >= 8
    inc_ws = fused_inc and prec == 1 and C1_ON_MFMA and on_ws(p, 3, H0, W0, 64, 64) and lib().mfpa_conv_c1_layout(H0, W0) == 2
    up4 = DEC[-1] + ".conv.double_conv"
    up4_folds = folds(DEC[-1], H0, W0, H0 // 2, W0 // 2)   # (the folded launch reads / writes plain float32 tensors)
    e_skip0 = inc_ws and on_ws(up4, 0, H0, W0, 128, 64) and not up4_folds
    H1_, W1_ = H0 // 2, W0 // 2
    d1, d2 = ENC[1], ENC[2]
    e_pool0 = inc_ws and on_ws(d1, 0, H1_, W1_, 64, 128)
    e_d10 = on_ws(d1, 0, H1_, W1_, 64, 128) and on_ws(d1, 3, H1_, W1_, 128, 128)
    e_pool1 = on_ws(d1, 3, H1_, W1_, 128, 128) and on_ws(d2, 0, H1_ // 2, W1_ // 2, 128, 256)
    e_up40 = on_ws(up4, 0, H0, W0, 128, 64) and on_ws(up4, 3, H0, W0, 64, 64) and not up4_folds
    if fused_inc:
        # inc.double_conv: the 1 -> 64 layer is evaluated inside the loader of the 64 -> 64 layer (no 64-channel intermediate)
        x, xp, _ = c(None, p, 3, pool=True, y_split=e_skip0, pool_split=e_pool0,
                     c1=dict(x32=x32, spec64=spec64, denom=denom, w=pw[p + ".0.w"], scale=pw[p + ".0.scale"], shift=pw[p + ".0.shift"]))
    else:
        e_skip0 = e_pool0 = False
        m = conv3x3_c1_bn_relu(pw[p + ".0.w"], pw[p + ".0.scale"], pw[p + ".0.shift"], x32=x32, spec64=spec64, denom=denom)
        x, xp, _ = c(m, p, 3, pool=True)
        del m
    skips.append(x)
    for name in ENC[1:]:
        first = name == ENC[1]
        m, _, _ = c(xp, name, 0, x0_split=(e_pool0 if first else (e_pool1 if name == ENC[2] else False)), y_split=(e_d10 and first))
        last = name == ENC[-1]
        x, xp, _ = c(m, name, 3, pool=not last, x0_split=(e_d10 and first), pool_split=(e_pool1 and first))
        del m
        if not last:
            skips.append(x)
    y = x                                                   # x5
    for name in DEC:
        skip = skips.pop()
        final = name == DEC[-1]
        if folds(name, skip.shape[1], skip.shape[2], y.shape[1], y.shape[2]):
            # round 6: the level's transposed convolution folded into its first 3x3 convolution -- one launch, `up` never exists
            m = upconv_fused(skip, y, pw[name + ".upc.wsk"], pw[name + ".upc.wup"], pw[name + ".conv.double_conv.0.shift"],
                             pw[name + ".upc.bias"], pw[name + ".upc.shape"][2], precision=prec)
            del skip
        else:
            u = convT2x2(y, pw[name + ".up.w" + sfx], pw[name + ".up.b"], precision=prec)
            m, _, _ = c(skip, name + ".conv.double_conv", 0, x1=u, x0_split=(e_skip0 and final), y_split=(e_up40 and final))
            del u, skip
        if not final:
            y, _, _ = c(m, name + ".conv.double_conv", 3)
        else:                                               # up4: OutConv fused, the 64-channel tensor is never written
            _, _, y = c(m, name + ".conv.double_conv", 3, out1x1=(pw["outc.w"], pw["outc.b_host"]), store=False, x0_split=e_up40)
        del m
    return y
