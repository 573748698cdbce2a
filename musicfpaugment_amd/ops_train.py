"""UNet training step on MI355X: forward (train mode) + L1 + backward + Adam over the C ABI.

Reference: Trainer.train_epoch, spec branch, training/train.py:257-317 -- spectrogram(clean),
spectrogram(aug), pred = UNet(aug.float()), L1Loss(pred, clean_f64), zero_grad / backward / Adam.step.

torch autograd is not used.  The engine owns ONE flat float32 buffer of all 31,036,481 parameters in
kernel layout (conv weights [tap][Cout][Cin]) laid out in the order the backward pass finishes them
(outc, up4 ... up1, down4 ... inc), with matching flat gradient / Adam-moment buffers: gradients of a
block form a contiguous bucket that can be all-reduced (RCCL) while earlier layers are still in
backward, and Adam is one fused launch over the whole buffer.
"""
from __future__ import annotations

import ctypes
import os
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import torch

from ._lib import ConvDesc, PackJob, WgradDesc, check, lib, ptr, stream
from . import ops_unet as K

BN_EPS = 1e-5
BN_MOMENTUM = 0.1

ENC = K.ENC
DEC = K.DEC
ENC_CH = [(1, 64), (64, 128), (128, 256), (256, 512), (512, 1024)]
DEC_CH = [(1024, 512), (512, 256), (256, 128), (128, 64)]


# ----------------------------------------------------------------------------- thin kernel wrappers
class Stats:
    """Per-channel batch statistics of one BatchNorm layer: mean, invstd and the fused scale/shift."""

    def __init__(self, C, device):
        buf = torch.empty((4, C), dtype=torch.float32, device=device)
        self.mean, self.invstd, self.scale, self.shift = buf[0], buf[1], buf[2], buf[3]
        self.drop = (0, 0, 1.0)      # (seed, thresh = rate * 2^32, 1/(1-rate)) of the nn.Dropout that follows, if any


def dropout_spec(seed: int, rate: float):
    """Stateless dropout parameters of csrc/mfpa_common.h::mfpa_keep."""
    if rate <= 0:
        return (0, 0, 1.0)
    return (seed & 0xFFFFFFFF, min(int(rate * 4294967296.0), 0xFFFFFFFF), 1.0 / (1.0 - rate))


def _drop(st):
    return st.drop if st is not None else (0, 0, 1.0)


DEVICE_PACK = True    # operand images of the weights built by mfpa_pack_conv_weights (False: the torch flip / transpose / split ops)


def pack_weights(w: torch.Tensor, precision: int, flip_transpose: bool = False, row0: int = 0, nrows: Optional[int] = None,
                 layout: int = 0):
    """Master weights [taps][Co][Ci] -> the operand image of one conv launch (see mfpa_pack_conv_weights): forward operand, or with
    flip_transpose the input-gradient operand [taps][ci in row0..row0+nrows][Co] (3x3 kernels flipped).  `layout` 1 (precision 1
    only): the fragment-ordered image of the weights-direct kernels (mfpa_conv_desc.w_layout)."""
    taps, Co, Ci = w.shape
    if nrows is None:
        nrows = (Ci if flip_transpose else Co) - row0
    if not DEVICE_PACK:
        if flip_transpose:
            w = (w.flip(0) if taps == 9 else w).transpose(1, 2)[:, row0:row0 + nrows].contiguous()
        else:
            w = w[:, row0:row0 + nrows].contiguous()
        if precision < 1:
            return w
        return K.split_bf16x3_frag(w, layout) if layout else K.split_bf16x3(w)
    code = (1 + layout) if precision >= 1 else precision
    cache = _PACK_CACHE
    if cache is not None:
        # a training engine's operand images: made by ONE batched launch per step (PackCache.refresh, after the optimiser step); a call that
        # is not in the table yet packs on its own and joins it
        hit = cache.lookup(w, taps, Co, Ci, flip_transpose, row0, nrows, code)
        if hit is not None:
            return hit
    out = torch.empty((taps, nrows, Co if flip_transpose else Ci), dtype=torch.float32, device=w.device)
    check(lib().mfpa_pack_conv_weights(ptr(w), taps, Co, Ci, int(flip_transpose), row0, nrows, code, ptr(out), stream()), "mfpa_pack_conv_weights")
    if cache is not None:
        cache.add(w, taps, Co, Ci, flip_transpose, row0, nrows, code, out)
    return out


class PackCache:
    """The operand images of one training engine (forward and input-gradient forms of every convolution's weights): pointers and shapes never
    change, so after the first step -- which packs them one by one and records the calls -- a step repacks ALL of them with one
    mfpa_pack_conv_weights_batch launch (46 launches of 5-7 us before).  `refresh()` runs it (the engine calls it at the start of a step: the
    weights changed in the optimiser step); `invalidate()` forgets everything (parameters re-loaded from a module)."""

    def __init__(self, owner: Optional[torch.Tensor] = None):
        # `owner`: the persistent buffer whose views may be cached (the engine's flat parameter buffer).  Keys and jobs hold raw data
        # pointers that refresh() re-reads every step: a weight tensor that does not live inside `owner` (a temporary) must never enter
        self.owner_range = None if owner is None else (owner.data_ptr(), owner.data_ptr() + owner.numel() * owner.element_size())
        self.entries = {}            # key -> (out tensor, index into jobs)
        self.jobs = []
        self.table = None            # device copy of the job structs
        self.total_tiles = 0
        self.fresh = set()           # keys whose image matches the current weights

    @staticmethod
    def _key(w, taps, Co, Ci, flip, row0, nrows, code):
        return (w.data_ptr(), taps, Co, Ci, bool(flip), row0, nrows, code)

    def owns(self, w) -> bool:
        if self.owner_range is None:
            return True
        p = w.data_ptr()
        return self.owner_range[0] <= p and p + w.numel() * w.element_size() <= self.owner_range[1]

    def lookup(self, w, taps, Co, Ci, flip, row0, nrows, code):
        if not self.owns(w):
            return None
        k = self._key(w, taps, Co, Ci, flip, row0, nrows, code)
        if k in self.fresh:
            return self.entries[k][0]
        return None

    def add(self, w, taps, Co, Ci, flip, row0, nrows, code, out):
        if not self.owns(w):
            return
        k = self._key(w, taps, Co, Ci, flip, row0, nrows, code)
        K = Co if flip else Ci
        if k in self.entries:                                   # stale image re-made by a single launch into a fresh tensor: point the job at it
            idx = self.entries[k][1]
            self.jobs[idx].out = out.data_ptr()
        else:
            idx = len(self.jobs)
            self.jobs.append(PackJob(w=w.data_ptr(), out=out.data_ptr(), taps=taps, Co=Co, Ci=Ci, flip_transpose=int(bool(flip)), row0=row0,
                                     nrows=nrows, precision=code, pad_=0, tile0=self.total_tiles))
            self.total_tiles += (K // 32) * (nrows // 32) * taps
        self.entries[k] = (out, idx)
        self.fresh.add(k)
        self.table = None

    def refresh(self, device):
        """Re-make every recorded image from the current weights (one launch) and mark them fresh."""
        if not self.jobs:
            return
        if self.table is None:
            raw = b"".join(bytes(j) for j in self.jobs)
            self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
        check(lib().mfpa_pack_conv_weights_batch(ptr(self.table), len(self.jobs), self.total_tiles, stream()), "mfpa_pack_conv_weights_batch")
        self.fresh = set(self.entries.keys())

    def stale(self):
        self.fresh = set()

    def invalidate(self):
        self.__init__()


_PACK_CACHE: Optional[PackCache] = None      # set by UNetTrainEngine around its forward / backward


def weight_layout(H: int, W: int, cin: int, cout: int, precision: int, mode: int = 0) -> int:
    """Which bf16x3 image the fastest kernel for this convolution reads (mfpa_conv_weight_layout): 0 = the row image, 1 / 2 = the
    fragment-ordered images of the weights-direct kernels."""
    if precision < 1 or mode != 0 or not K.USE_WEIGHTS_DIRECT:
        return 0
    return int(lib().mfpa_conv_weight_layout(H, W, cin, cout, 0, 1))


def conv_mfma(x0, w, Cout, *, mode=0, in_affine: Optional[Stats] = None, x1=None, out_scale=None, out_shift=None,
              relu=False, out_hw: Optional[Tuple[int, int]] = None, precision: int = 0, packed: bool = False, w_layout: int = 0,
              x0_bf16_out: Optional[list] = None, stats_out: Optional[list] = None, x1_bf16_out: Optional[list] = None,
              y_bf16_out: Optional[list] = None, bwd_of=None, out_bf16: bool = False):
    """General MFMA convolution (mfpa_conv_mfma).  x0 is NHWC; returns the NHWC output.  precision 1 = bf16x3:
    `w` (fp32, kernel layout) is split into the image the kernel for this shape reads (weight_layout) unless it already is an
    operand image (`packed`, in the layout `w_layout`).  `x0_bf16_out` (a list): when the kernel for this shape can write it
    (w_layout 2), the bf16 copy of the activated source 0 is appended -- the weight gradient's operand, for free.  `stats_out` (a
    list): likewise the kernel's per-wave partial BatchNorm statistics of the output ((rows, 2, Cout) float32, mfpa_conv_stats_reduce).
    `x1_bf16_out` / `y_bf16_out`: likewise bf16 copies of source 1 and of the output.  `bwd_of` = (z, Stats) with `stats_out`: the
    output is dy of relu(bn(z)) and the partials are the BatchNorm backward's two reductions (mfpa_conv_desc.bwd_z).
    bfloat16 sources (x0 and x1 together: the plain-bf16 step's activations kept as bfloat16, Z16_ACTIVATIONS) go to conv_wd16_kernel
    (precision 2, w_layout 2) or the transposed convolution; `out_bf16`: the output exists as bfloat16 only (returned)."""
    B = x0.shape[0]
    if mode == 2:
        H, W = x0.shape[1] // 2, x0.shape[2] // 2
    else:
        H, W = x0.shape[1], x0.shape[2]
    C0 = x0.shape[3]
    C1 = 0 if x1 is None else x1.shape[3]
    # precision 2 = plain bf16 products where conv_wd16_kernel serves the shape (it reads the hi halves of the bf16x3 image); every
    # other kernel of the family (transposed convolutions, the fall-back shapes) stays on bf16x3
    plain, precision = precision == 2, min(precision, 1)
    in16 = x0.dtype == torch.bfloat16                              # the bf16 copy of dz as the input-gradient convolution's source (plain bf16 only)
    if precision == 1 and not packed:
        w_layout = weight_layout(H, W, C0 + C1, Cout, precision, mode)
        if DEVICE_PACK and w.is_contiguous() and w.shape[1] % 32 == 0 and w.shape[2] % 32 == 0:
            w = pack_weights(w, 1, layout=w_layout)
        else:
            w = K.split_bf16x3_frag(w, w_layout) if w_layout else K.split_bf16x3(w)
    if mode == 1:
        oh, ow = 2 * H, 2 * W
    else:
        oh, ow = out_hw if out_hw is not None else (H, W)
    if in16 and x1 is not None and x1.dtype != torch.bfloat16:
        raise ValueError("bfloat16 sources come together (x0 and x1)")
    if out_bf16 and not ((plain and w_layout == 2 and mode == 0) or (mode == 1 and precision == 1)):
        raise ValueError("a bfloat16-only output needs conv_wd16_kernel (precision 2, w_layout 2) or the bf16x3 transposed convolution")
    y = torch.empty((B, oh, ow, Cout), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x0.device)
    xb = None
    if x0_bf16_out is not None and ((w_layout == 2 and mode == 0) or (mode == 1 and precision == 1)):
        if in16 and in_affine is None:
            x0_bf16_out.append(x0)                                 # a bfloat16 source without an on-load transform IS the weight gradient's operand
        else:
            xb = torch.empty(x0.shape, dtype=torch.bfloat16, device=x0.device)
            x0_bf16_out.append(xb)
    x1b = yb16 = None
    if x1_bf16_out is not None and x1 is not None and w_layout == 2 and mode == 0:
        if in16:
            x1_bf16_out.append(x1)                                 # (source 1 carries no on-load transform)
        else:
            x1b = torch.empty(x1.shape, dtype=torch.bfloat16, device=x0.device)
            x1_bf16_out.append(x1b)
    if out_bf16:
        yb16 = y
    elif y_bf16_out is not None and w_layout == 2 and mode == 0:
        yb16 = torch.empty(y.shape, dtype=torch.bfloat16, device=x0.device)
        y_bf16_out.append(yb16)
    part = None
    if stats_out is not None and w_layout == 2 and mode == 0 and out_scale is None and out_shift is None and not relu:
        rows = int(lib().mfpa_conv_stats_rows(B, H, W, C0 + C1, Cout))
        if rows > 0:
            part = torch.empty((rows, 2, Cout), dtype=torch.float32, device=x0.device)
            stats_out.append(part)
    d = ConvDesc(x0=ptr(x0), in_scale0=ptr(in_affine.scale) if in_affine else 0,
                 in_shift0=ptr(in_affine.shift) if in_affine else 0, x1=ptr(x1), w=ptr(w),
                 out_scale=ptr(out_scale), out_shift=ptr(out_shift), y=0 if out_bf16 else ptr(y), C0=C0, C1=C1,
                 H1=0 if x1 is None else x1.shape[1], W1=0 if x1 is None else x1.shape[2],
                 B=B, H=H, W=W, Cout=Cout, relu=int(relu), yH=0 if mode == 1 else oh, yW=0 if mode == 1 else ow,
                 mode=mode, drop_seed=_drop(in_affine)[0], drop_thresh=_drop(in_affine)[1],
                 drop_scale=_drop(in_affine)[2],
                 precision=2 if (plain and ((w_layout == 2 and mode == 0) or (PLAIN_CONVT and (mode == 2 or (mode == 1 and (in16 or out_bf16)))))) else precision,
                 w_layout=w_layout, x0_bf16=ptr(xb), x1_bf16=ptr(x1b), y_bf16=ptr(yb16), stats_part=ptr(part),
                 bwd_z=ptr(bwd_of[0]) if (bwd_of and part is not None) else 0,
                 bwd_scale=ptr(bwd_of[1].scale) if (bwd_of and part is not None) else 0,
                 bwd_shift=ptr(bwd_of[1].shift) if (bwd_of and part is not None) else 0,
                 bwd_mean=ptr(bwd_of[1].mean) if (bwd_of and part is not None) else 0,
                 bwd_invstd=ptr(bwd_of[1].invstd) if (bwd_of and part is not None) else 0, x0_is_bf16=int(in16),
                 bwd_z_is_bf16=int(bool(bwd_of) and part is not None and bwd_of[0].dtype == torch.bfloat16))
    if in16 and not ((plain and w_layout == 2 and mode == 0) or (mode == 1 and precision == 1)):
        raise ValueError("bfloat16 sources need the plain-bf16 conv_wd16_kernel (precision 2, w_layout 2) or the bf16x3 transposed convolution")
    t0 = K._TIMER.start() if K._TIMER is not None else None
    check(lib().mfpa_conv_mfma(ctypes.byref(d), stream()), "mfpa_conv_mfma")
    if t0 is not None:
        K._TIMER.stop(t0)
    return y


PRENORMALISE_INPUT = True   # the train step's float64 spectrogram / clip maximum -> float32 once (mfpa_normalize_f32) instead of inside the first layer's two kernels
C1_STATS = True             # bf16 engines: the first layer's kernel writes its output's BatchNorm row partials (no 200 us statistics pass over 1 GB); the
                            # fp32 engine keeps the float64 pass (see UNetTrainEngine._bn_stats)
C1_WGRAD_BF16 = True        # plain-bf16 step: the 1-channel first layer's weight gradient reads the bf16 copy of dz like every other (False: float32 dz)
POOL_BWD_FUSED = True       # an encoder block's pool backward + last BatchNorm backward without the finished dy in memory (mfpa_maxpool2_bwd_bn_relu_bwd)
SKIP_GRAD_BF16 = True       # plain-bf16 step with bf16 activations: the decoder's gradients w.r.t. the skip tensors wait for the encoder's backward as bfloat16
BATCH_REPACK = True         # the convolutions' operand images (forward + input-gradient forms) re-made by ONE launch per step (PackCache); False: one launch each
FUSED_FINISH = True         # single-GPU BatchNorm statistics / backward sums from row partials: the finish kernels read the block partials directly
                            # (mfpa_conv_stats_bn_finish, mfpa_bn_relu_bwd_from_part: 35 launches fewer per step, bit-identical; False: the separate calls)
PLAIN_CONVT = True          # plain-bf16 step: the transposed convolutions' forward and input gradient with one bf16 MFMA per product too (round 6; False: bf16x3)
DY16_MID = False            # True: a DoubleConv's inner gradient (dy of its first BatchNorm) too leaves its convolution as bfloat16 only -- built, tested, measured
                            # at -0.07 ms of 31.2 (those two kernels are not bound by these bytes): off
Z16_ACTIVATIONS = True      # plain-bf16 step (precision 2 with bf16 weight gradients): the convolutions' raw outputs z, the pooled activations and the
                            # transposed convolutions' outputs live in HBM as bfloat16 only (False: float32, A/B runs) -- see UNetTrainEngine._z16_for
USE_BF16_DZ = True          # plain-bf16 step: input-gradient convolutions read the bf16 copy of dz (False: the float32 dz, A/B runs)
FUSE_POOL_BWD_SUMS = True   # False: mfpa_maxpool2_bwd_add, then the BatchNorm backward's own reduction pass (chan_reduce_kernel<1>)
RANK1_OUTCONV_BWD = True    # False: mfpa_outconv_bwd writes dy = dpred x w (1.06 GB per 64 clips), the BatchNorm backward reduces and reads it
BF16_WGRAD_MIN_CH = 128     # plain-bf16 weight gradients: layers with at least this many channels on both sides read bf16 copies


def act_to_bf16(z: torch.Tensor, in_affine: Optional[Stats] = None) -> torch.Tensor:
    """bf16 copy of an NHWC float32 tensor with the consumer's on-load transform applied (mfpa_act_to_bf16)."""
    out = torch.empty(z.shape, dtype=torch.bfloat16, device=z.device)
    check(lib().mfpa_act_to_bf16(ptr(z), z.numel(), z.shape[-1], ptr(in_affine.scale) if in_affine else 0,
                                 ptr(in_affine.shift) if in_affine else 0, _drop(in_affine)[0], _drop(in_affine)[1],
                                 _drop(in_affine)[2], ptr(out), stream()), "mfpa_act_to_bf16")
    return out


def bf16_wgrad(Cout, cin, precision, have_x0_copy: bool = False) -> bool:
    """Does wgrad_mfma read bf16 copies of its operands for this layer?  From BF16_WGRAD_MIN_CH channels up -- and below that when
    the forward convolution already wrote the copy of the activated input (then no cast pass is paid for it)."""
    return precision == 2 and (have_x0_copy or min(Cout, cin) >= BF16_WGRAD_MIN_CH)


def wgrad_mfma(dz, x0, dw, Cout, *, mode=0, in_affine: Optional[Stats] = None, x1=None, precision=0, dz_bf16=None, x0_bf16=None,
               x1_bf16=None):
    B, H, W, C0 = x0.shape
    cin = C0 + (0 if x1 is None else x1.shape[3])
    if bf16_wgrad(Cout, cin, precision, x0_bf16 is not None):
        # every (co, ci) tile re-reads both operands: bf16 copies (activation applied) halve what the tiles pull through L2; same
        # products as precision 2 (one bf16 MFMA each, fp32 accumulate).  dz's copy comes with the BatchNorm backward (dz_bf16), the
        # activated input's with the forward convolution (x0_bf16); what is missing is cast here.
        dz = dz_bf16 if dz_bf16 is not None else act_to_bf16(dz)
        if x0_bf16 is None and x0.dtype == torch.bfloat16 and in_affine is not None:
            raise ValueError("a bfloat16 activation with a pending on-load transform needs the forward convolution's bf16 copy (x0_bf16)")
        x0, in_affine = (x0_bf16 if x0_bf16 is not None else (x0 if x0.dtype == torch.bfloat16 else act_to_bf16(x0, in_affine))), None
        x1 = None if x1 is None else (x1_bf16 if x1_bf16 is not None else act_to_bf16(x1))
        precision = 3
    d = WgradDesc(dz=ptr(dz), x0=ptr(x0), in_scale0=ptr(in_affine.scale) if in_affine else 0,
                  in_shift0=ptr(in_affine.shift) if in_affine else 0, x1=ptr(x1), dw=ptr(dw), C0=C0,
                  C1=0 if x1 is None else x1.shape[3], H1=0 if x1 is None else x1.shape[1],
                  W1=0 if x1 is None else x1.shape[2], B=B, H=H, W=W, Cout=Cout, mode=mode,
                  drop_seed=_drop(in_affine)[0], drop_thresh=_drop(in_affine)[1], drop_scale=_drop(in_affine)[2],
                  precision=precision)
    t0 = K._TIMER.start() if K._TIMER is not None else None
    check(lib().mfpa_wgrad_mfma(ctypes.byref(d), stream()), "mfpa_wgrad_mfma")
    if t0 is not None:
        K._TIMER.stop(t0)


def _npix(t):
    return t.shape[0] * t.shape[1] * t.shape[2]


def _is16(t) -> int:
    return int(t.dtype == torch.bfloat16)


def flat_layout():
    """Segments {name: (offset, shape)} of the flat parameter buffer, the all-reduce buckets
    [(block, start, end)] and the total element count.  Order = the order the backward pass finishes
    gradients (outc, up4 ... up1, down4 ... inc; inside a DoubleConv the second conv first), so each bucket
    is a contiguous slice that is complete -- and can be all-reduced -- while earlier layers are still in backward."""
    segs: "OrderedDict[str, Tuple[int, Tuple[int, ...]]]" = OrderedDict()
    off = 0

    def add(name, shape):
        nonlocal off
        n = 1
        for s in shape:
            n *= s
        segs[name] = (off, tuple(shape))
        off += n

    def dconv(prefix, cin, cout, first=False):
        add(prefix + ".3.w", (9, cout, cout)); add(prefix + ".4.g", (cout,)); add(prefix + ".4.b", (cout,))
        add(prefix + ".0.w", (9, cout) if first else (9, cout, cin)); add(prefix + ".1.g", (cout,)); add(prefix + ".1.b", (cout,))

    buckets: List[Tuple[str, int, int]] = []
    start = off
    add("outc.wb", (65,))                       # 64 weights + bias
    for name, (cin, cout) in zip(reversed(DEC), reversed(DEC_CH)):      # up4, up3, up2, up1
        dconv(name + ".conv.double_conv", cin, cout)
        add(name + ".up.w", (4, cin // 2, cin)); add(name + ".up.b", (cin // 2,))
        buckets.append((name, start, off)); start = off
    for name, (cin, cout) in zip(reversed(ENC), reversed(ENC_CH)):      # down4 ... inc
        dconv(name, cin, cout, first=(cin == 1))
        buckets.append((name, start, off)); start = off
    return segs, buckets, off


# ----------------------------------------------------------------------------- the engine
class UNetTrainEngine:
    """Owns kernel-layout master parameters, gradients and Adam moments of a UNet(1, 1) and runs train steps."""

    def __init__(self, module, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, process_group=None, precision: int = 0,
                 wgrad_precision: Optional[int] = None, sync_bn: bool = False, collectives_at_world_one: bool = False):
        self.module = module
        # collectives_at_world_one: issue the gradient-bucket all-reduces even when the process group has a single rank (the
        # RCCL path of a one-GPU box: tests/test_gpu_dist.py, `torchrun --nproc-per-node 1 bench.py --mode train`)
        self.collectives_at_world_one = bool(collectives_at_world_one)
        self.comm_calls, self.comm_bytes = 0, 0            # collectives issued by this engine / their payload bytes
        # set to a list (bench.py): a pair of events on the compute stream around the gradient-bucket waits of every step, i.e. the
        # all-reduce time that backward did NOT hide
        self.comm_wait_events = None
        # arithmetic of the convolutions: 0 = fp32 MFMA, 1 = bf16x3 (3 bf16 MFMAs per fp32 product); `wgrad_precision`
        # overrides it for the weight-gradient kernel, which also offers 2 = plain bf16 products (a weight gradient sums over
        # every pixel of the batch: relative L1 ~2e-3).  Reductions, BatchNorm statistics, the loss and Adam are fp32/fp64.
        self.precision = precision
        self.wgrad_precision = precision if wgrad_precision is None else wgrad_precision
        # sync_bn: BatchNorm statistics over the global batch of all ranks (what the single-GPU reference computes) instead of
        # per-GPU statistics (the DDP default): two small float64 all-reduces per BatchNorm layer and step
        import torch.distributed as _dist
        self._global_over_local_batch = 1.0
        self.sync_bn = bool(sync_bn) and _dist.is_available() and _dist.is_initialized() and _dist.get_world_size(process_group) > 1
        self.device = next(module.parameters()).device
        if self.device.type != "cuda":
            raise RuntimeError("the training engine runs on the MI355X only")
        self.rate = float(module.dropout.p)        # nn.Dropout(rate) on x2..x5 and up1's output (unet.py:83,99-103)
        # data-parallel ranks must not apply identical dropout masks to their shards: the rank is mixed into the seed
        self.drop_seed = 0x5EED + 0x9E3779B1 * (_dist.get_rank(process_group) if _dist.is_available() and _dist.is_initialized() else 0) & 0x7FFFFFFF
        self.lr, self.betas, self.eps = lr, betas, eps
        self.step_count = 0
        self.group = process_group
        self._layout()
        self.load_from_module()
        self.workspace = torch.empty(lib().mfpa_red_blocks() * 2 * 1024, dtype=torch.float64, device=self.device)
        self.loss = torch.zeros(1, dtype=torch.float64, device=self.device)
        self._z16, self._z16_cache = False, {}
        self._packs = PackCache(self.flat_p) # the convolutions' operand images, re-made by one launch per step (BATCH_REPACK)

    def _z16_for(self, H: int, W: int) -> bool:
        """Do this step's activations live in HBM as bfloat16?  Z16_ACTIVATIONS, the plain-bf16 arithmetic with bf16 weight gradients
        (precision 2 / 2: every consumer rounds them to bf16 anyway, once, after the BatchNorm affine -- here the raw convolution
        output is rounded too), and EVERY 3x3 convolution of the net on conv_wd16_kernel at this image size (the only kernel that reads
        and writes them: w_layout 2).  Decided per forward from the input's (H, W); BatchNorm statistics still come from the float32
        accumulators (stats_part)."""
        if not (Z16_ACTIVATIONS and self.precision == 2 and self.wgrad_precision == 2):
            return False
        key = (H, W)
        if key not in self._z16_cache:
            ok, h, w = True, H, W
            for cin, cout in ((64, 128), (128, 256), (256, 512), (512, 1024)):
                h, w = h // 2, w // 2
                ok = ok and h >= 1 and w >= 1 and weight_layout(h, w, cin, cout, 2) == 2 and weight_layout(h, w, cout, cout, 2) == 2
            ok = ok and weight_layout(H, W, 64, 64, 2) == 2
            hs = [(H >> k, W >> k) for k in range(4)]               # the Up blocks' resolutions: H/8 .. H (skip sizes)
            for (hh, ww), cout in zip(reversed(hs), (512, 256, 128, 64)):
                ok = ok and weight_layout(hh, ww, 2 * cout, cout, 2) == 2 and weight_layout(hh, ww, cout, cout, 2) == 2
            self._z16_cache[key] = bool(ok)
        return self._z16_cache[key]

    # ------------------------------------------------------------------ flat layout (backward completion order)
    def _layout(self):
        segs, self.buckets, off = flat_layout()
        self.segs, self.n_params = segs, off
        dev = self.device
        self.flat_p = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(off, dtype=torch.float32, device=dev)
        self.P = {k: self.flat_p[o:o + self._n(s)].view(s) for k, (o, s) in segs.items()}
        self.G = {k: self.flat_g[o:o + self._n(s)].view(s) for k, (o, s) in segs.items()}
        self.running: Dict[str, torch.Tensor] = {}

    @staticmethod
    def _n(shape):
        n = 1
        for s in shape:
            n *= s
        return n

    # ------------------------------------------------------------------ module <-> engine
    def _bn_names(self, prefix):
        return [(prefix + ".1", prefix + ".1.g", prefix + ".1.b"), (prefix + ".4", prefix + ".4.g", prefix + ".4.b")]

    def load_from_module(self):
        sd = self.module.state_dict()
        with torch.no_grad():
            for i, p in enumerate(ENC + [d + ".conv.double_conv" for d in DEC]):
                w0 = sd[p + ".0.weight"].float()
                if w0.shape[1] == 1:
                    self.P[p + ".0.w"].copy_(w0.permute(2, 3, 1, 0).reshape(9, w0.shape[0]))
                else:
                    self.P[p + ".0.w"].copy_(K.pack_conv3x3(w0))
                self.P[p + ".3.w"].copy_(K.pack_conv3x3(sd[p + ".3.weight"]))
                for bn, g, b in self._bn_names(p):
                    self.P[g].copy_(sd[bn + ".weight"]); self.P[b].copy_(sd[bn + ".bias"])
                    self.running[bn + ".running_mean"] = sd[bn + ".running_mean"].float().clone()
                    self.running[bn + ".running_var"] = sd[bn + ".running_var"].float().clone()
            for d in DEC:
                self.P[d + ".up.w"].copy_(K.pack_convT2x2(sd[d + ".up.weight"]))
                self.P[d + ".up.b"].copy_(sd[d + ".up.bias"])
            self.P["outc.wb"][:64].copy_(sd["outc.conv.weight"].reshape(-1))
            self.P["outc.wb"][64:].copy_(sd["outc.conv.bias"].reshape(-1))

    def sync_to_module(self):
        """Write the master parameters / running statistics back into the nn.Module (reference key layout)."""
        m = self.module
        sd = m.state_dict()
        with torch.no_grad():
            for p in ENC + [d + ".conv.double_conv" for d in DEC]:
                w0 = self.P[p + ".0.w"]
                if w0.dim() == 2:
                    sd[p + ".0.weight"].copy_(w0.view(3, 3, 1, -1).permute(3, 2, 0, 1))
                else:
                    sd[p + ".0.weight"].copy_(w0.view(3, 3, w0.shape[1], w0.shape[2]).permute(2, 3, 0, 1))
                w3 = self.P[p + ".3.w"]
                sd[p + ".3.weight"].copy_(w3.view(3, 3, w3.shape[1], w3.shape[2]).permute(2, 3, 0, 1))
                for bn, g, b in self._bn_names(p):
                    sd[bn + ".weight"].copy_(self.P[g]); sd[bn + ".bias"].copy_(self.P[b])
                    sd[bn + ".running_mean"].copy_(self.running[bn + ".running_mean"])
                    sd[bn + ".running_var"].copy_(self.running[bn + ".running_var"])
                    sd[bn + ".num_batches_tracked"].fill_(self.step_count)
            for d in DEC:
                wu = self.P[d + ".up.w"]
                sd[d + ".up.weight"].copy_(wu.view(2, 2, wu.shape[1], wu.shape[2]).permute(3, 2, 0, 1))
                sd[d + ".up.bias"].copy_(self.P[d + ".up.b"])
            sd["outc.conv.weight"].copy_(self.P["outc.wb"][:64].view(1, 64, 1, 1))
            sd["outc.conv.bias"].copy_(self.P["outc.wb"][64:])

    def _named(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        """One of the flat buffers (gradients, Adam moments, ...) re-laid-out under the reference's parameter names and shapes."""
        V = {k: flat[o:o + self._n(s)].view(s) for k, (o, s) in self.segs.items()}
        out = {}
        for p in ENC + [d + ".conv.double_conv" for d in DEC]:
            g0 = V[p + ".0.w"]
            out[p + ".0.weight"] = (g0.view(3, 3, 1, -1).permute(3, 2, 0, 1) if g0.dim() == 2
                                    else g0.view(3, 3, g0.shape[1], g0.shape[2]).permute(2, 3, 0, 1)).contiguous()
            g3 = V[p + ".3.w"]
            out[p + ".3.weight"] = g3.view(3, 3, g3.shape[1], g3.shape[2]).permute(2, 3, 0, 1).contiguous()
            for bn, g, b in self._bn_names(p):
                out[bn + ".weight"] = V[g]; out[bn + ".bias"] = V[b]
        for d in DEC:
            gu = V[d + ".up.w"]
            out[d + ".up.weight"] = gu.view(2, 2, gu.shape[1], gu.shape[2]).permute(3, 2, 0, 1).contiguous()
            out[d + ".up.bias"] = V[d + ".up.b"]
        out["outc.conv.weight"] = V["outc.wb"][:64].view(1, 64, 1, 1)
        out["outc.conv.bias"] = V["outc.wb"][64:]
        return out

    def _load_named(self, flat: torch.Tensor, sd: Dict[str, torch.Tensor]) -> None:
        """Inverse of _named: tensors under the reference's parameter names -> a flat buffer in the kernel layouts."""
        V = {k: flat[o:o + self._n(s)].view(s) for k, (o, s) in self.segs.items()}
        with torch.no_grad():
            for p in ENC + [d + ".conv.double_conv" for d in DEC]:
                w0 = sd[p + ".0.weight"].to(flat.device, torch.float32)
                if w0.shape[1] == 1:
                    V[p + ".0.w"].copy_(w0.permute(2, 3, 1, 0).reshape(9, w0.shape[0]))
                else:
                    V[p + ".0.w"].copy_(K.pack_conv3x3(w0))
                V[p + ".3.w"].copy_(K.pack_conv3x3(sd[p + ".3.weight"].to(flat.device, torch.float32)))
                for bn, g, b in self._bn_names(p):
                    V[g].copy_(sd[bn + ".weight"]); V[b].copy_(sd[bn + ".bias"])
            for d in DEC:
                V[d + ".up.w"].copy_(K.pack_convT2x2(sd[d + ".up.weight"].to(flat.device, torch.float32)))
                V[d + ".up.b"].copy_(sd[d + ".up.bias"])
            V["outc.wb"][:64].copy_(sd["outc.conv.weight"].reshape(-1))
            V["outc.wb"][64:].copy_(sd["outc.conv.bias"].reshape(-1))

    def named_grads(self) -> Dict[str, torch.Tensor]:
        """Gradients re-laid-out under the reference's parameter names (tests / inspection)."""
        return self._named(self.flat_g)

    def named_moments(self):
        """Adam's exp_avg / exp_avg_sq under the reference's parameter names (torch-Adam-compatible checkpoints)."""
        return self._named(self.flat_m), self._named(self.flat_v)

    def load_named_moments(self, exp_avg: Dict[str, torch.Tensor], exp_avg_sq: Dict[str, torch.Tensor]) -> None:
        self._load_named(self.flat_m, exp_avg)
        self._load_named(self.flat_v, exp_avg_sq)

    # ------------------------------------------------------------------ kernels with engine state
    def _all_reduce_sums(self, sums_and_count: torch.Tensor) -> torch.Tensor:
        import torch.distributed as dist
        dist.all_reduce(sums_and_count, op=dist.ReduceOp.SUM, group=self.group)
        self.comm_calls += 1; self.comm_bytes += sums_and_count.numel() * sums_and_count.element_size()
        return sums_and_count

    def _bn_stats(self, z, bn, g, b, part=None) -> Stats:
        """Batch statistics of z.  `part`: the producing convolution's per-wave partial sums (conv_mfma(stats_out=...)) -- then z is not
        read again.  Accuracy of that shortcut: the partials are float32 sums over <= 128 pixels, reduced in float64; the variance
        E[z^2] - mean^2 amplifies their error by 1 + mean^2 / var.  Measured (tests/test_gpu_train.py::
        test_batchnorm_statistics_from_conv_partials_large_offset_channel): channels 12 standard deviations off zero (amplification 145)
        read invstd 2e-6 off the float64 pass, channels within 3 standard deviations 1e-8 -- the per-wave rounding errors are ~1e-8 and
        average out, far below the bound a worst-case analysis gives.
        Only the bf16 kernels (conv_wd16_kernel: precision 1 / 2) write partials -- their own operand rounding (2^-17 / 2^-9) is
        amplified by the same factor in z itself -- the fp32 engine (precision 0: the golden step, the autograd default) always takes the
        float64 pass over z below for these FORWARD statistics.  The BatchNorm-backward sums {sum g, sum g * xhat} that ride in
        other passes at every precision (mfpa_maxpool2_bwd_add_sums, mfpa_outconv_bwd_sums) accumulate in float64 inside a workgroup
        and hand float32 row totals to the float64 reduction (tests/test_gpu_train.py::
        test_pool_backward_sums_match_the_float64_reduction_on_a_large_offset_channel)."""
        C = z.shape[-1]
        st = Stats(C, z.device)
        if part is not None and not self.sync_bn and FUSED_FINISH:
            # single-GPU statistics: the finish kernel sums the row blocks' partials itself (two launches instead of three, same sums)
            check(lib().mfpa_conv_stats_bn_finish(ptr(part), part.shape[0], C, float(_npix(z)), ptr(self.P[g]), ptr(self.P[b]), BN_EPS, BN_MOMENTUM,
                                                  ptr(st.mean), ptr(st.invstd), ptr(st.scale), ptr(st.shift),
                                                  ptr(self.running[bn + ".running_mean"]), ptr(self.running[bn + ".running_var"]),
                                                  ptr(self.workspace), stream()), "mfpa_conv_stats_bn_finish")
            st.count_host = float(_npix(z))
            return st
        if self.sync_bn or part is not None:
            # statistics over the GLOBAL batch (the single-GPU reference's semantics): local (sum, sum^2) + pixel count,
            # one small SUM all-reduce, finish from the global sums
            sc = torch.empty(2 * C, dtype=torch.float64, device=z.device)
            if part is not None:
                check(lib().mfpa_conv_stats_reduce(ptr(part), part.shape[0], C, ptr(sc), ptr(self.workspace), stream()),
                      "mfpa_conv_stats_reduce")
            else:
                check(lib().mfpa_bn_stats_sums(ptr(z), _npix(z), C, ptr(sc), ptr(self.workspace), _is16(z), stream()), "mfpa_bn_stats_sums")
            if self.sync_bn:
                self._all_reduce_sums(sc)
            count = float(_npix(z)) * (self._global_over_local_batch if self.sync_bn else 1.0)   # every rank holds clips of the same H x W
            check(lib().mfpa_bn_stats_finish(ptr(sc), count, C, ptr(self.P[g]), ptr(self.P[b]), BN_EPS, BN_MOMENTUM,
                                             ptr(st.mean), ptr(st.invstd), ptr(st.scale), ptr(st.shift),
                                             ptr(self.running[bn + ".running_mean"]), ptr(self.running[bn + ".running_var"]),
                                             stream()), "mfpa_bn_stats_finish")
            st.count_host = count
            return st
        check(lib().mfpa_bn_stats(ptr(z), _npix(z), C, ptr(self.P[g]), ptr(self.P[b]), BN_EPS, BN_MOMENTUM, ptr(st.mean),
                                  ptr(st.invstd), ptr(st.scale), ptr(st.shift), ptr(self.running[bn + ".running_mean"]),
                                  ptr(self.running[bn + ".running_var"]), ptr(self.workspace), _is16(z), stream()), "mfpa_bn_stats")
        return st

    def _bn_relu_bwd(self, dy, z, st: Stats, g, b, bf16_copy: bool = False, part=None, write_f32: bool = True, rank1=None):
        """dy <- gradient w.r.t. z, in place; with bf16_copy also its bf16 copy, written by the same pass (the weight-gradient
        kernel's operand).  `write_f32=False` (needs bf16_copy): ONLY the bf16 copy is written -- every consumer reads it (plain-bf16
        step) -- and None is returned for dz.  Returns (dz, dz_bf16 or None)."""
        write_f32 = write_f32 or not bf16_copy
        C = z.shape[-1]
        coef = torch.empty((3, C), dtype=torch.float32, device=z.device)
        dz16 = torch.empty(z.shape, dtype=torch.bfloat16, device=z.device) if bf16_copy else None
        if rank1 is not None:
            # dy is the OutConv's gradient dpred[p] * w[c] and was never written: `part` holds its BatchNorm-backward partial sums
            # (mfpa_outconv_bwd_sums); the apply pass forms dy again from dpred and w (mfpa_bn_relu_bwd_finish_rank1)
            dpred, w1 = rank1
            loc = torch.empty(2 * C, dtype=torch.float64, device=z.device)
            check(lib().mfpa_conv_stats_reduce(ptr(part), part.shape[0], C, ptr(loc), ptr(self.workspace), stream()), "mfpa_conv_stats_reduce")
            glob = self._all_reduce_sums(loc.clone()) if self.sync_bn else loc
            if not self.sync_bn:
                st.count_host = float(_npix(z))
            dz = torch.empty(z.shape, dtype=torch.float32, device=z.device) if write_f32 else None
            check(lib().mfpa_bn_relu_bwd_finish_rank1(ptr(dpred), ptr(w1), ptr(z), _npix(z), C, ptr(self.P[g]), ptr(st.scale), ptr(st.shift),
                                                      ptr(st.mean), ptr(st.invstd), ptr(loc), ptr(glob), st.count_host, ptr(self.G[g]),
                                                      ptr(self.G[b]), ptr(coef), ptr(dz), ptr(dz16), _is16(z), stream()), "mfpa_bn_relu_bwd_finish_rank1")
            return dz, dz16
        if part is not None and not self.sync_bn and FUSED_FINISH:
            st.count_host = float(_npix(z))
            check(lib().mfpa_bn_relu_bwd_from_part(ptr(dy), ptr(z), _npix(z), C, ptr(self.P[g]), ptr(st.scale), ptr(st.shift), ptr(st.mean),
                                                   ptr(st.invstd), ptr(part), part.shape[0], ptr(self.G[g]), ptr(self.G[b]), ptr(coef),
                                                   ptr(self.workspace), st.drop[0], st.drop[1], st.drop[2], ptr(dz16), int(write_f32),
                                                   _is16(z), _is16(dy), stream()), "mfpa_bn_relu_bwd_from_part")
            return (dy if write_f32 else None), dz16
        if self.sync_bn or part is not None:
            loc = torch.empty(2 * C, dtype=torch.float64, device=z.device)
            if part is not None:             # the convolution that produced dy reduced (sum g, sum g * xhat) in its epilogue (conv_mfma(bwd_of=))
                check(lib().mfpa_conv_stats_reduce(ptr(part), part.shape[0], C, ptr(loc), ptr(self.workspace), stream()),
                      "mfpa_conv_stats_reduce")
            else:
                check(lib().mfpa_bn_relu_bwd_sums(ptr(dy), ptr(z), _npix(z), C, ptr(st.scale), ptr(st.shift), ptr(st.mean),
                                                  ptr(st.invstd), ptr(loc), ptr(self.workspace), st.drop[0], st.drop[1], st.drop[2],
                                                  _is16(z), stream()), "mfpa_bn_relu_bwd_sums")
            glob = self._all_reduce_sums(loc.clone()) if self.sync_bn else loc
            if not self.sync_bn:
                st.count_host = float(_npix(z))
            check(lib().mfpa_bn_relu_bwd_finish(ptr(dy), ptr(z), _npix(z), C, ptr(self.P[g]), ptr(st.scale), ptr(st.shift),
                                                ptr(st.mean), ptr(st.invstd), ptr(loc), ptr(glob), st.count_host, ptr(self.G[g]),
                                                ptr(self.G[b]), ptr(coef), st.drop[0], st.drop[1], st.drop[2], ptr(dz16), int(write_f32), _is16(z), _is16(dy), stream()),
                  "mfpa_bn_relu_bwd_finish")
            return (dy if write_f32 else None), dz16
        check(lib().mfpa_bn_relu_bwd(ptr(dy), ptr(z), _npix(z), C, ptr(self.P[g]), ptr(st.scale), ptr(st.shift),
                                     ptr(st.mean), ptr(st.invstd), ptr(self.G[g]), ptr(self.G[b]), ptr(coef),
                                     ptr(self.workspace), st.drop[0], st.drop[1], st.drop[2], ptr(dz16), int(write_f32), _is16(z), stream()),
              "mfpa_bn_relu_bwd")
        return (dy if write_f32 else None), dz16

    # ------------------------------------------------------------------ forward (train mode)
    def _dconv_fwd(self, prefix, src0, aff0: Optional[Stats], src1=None, first_input=None, drop_id=None):
        cout = self.P[prefix + ".3.w"].shape[1]
        # bf16 copies of the activated inputs, written by the forward convolutions' loaders where their kernel can (the weight
        # gradients' operands; only when those are computed from bf16 operands at all)
        want = self.wgrad_precision == 2
        xb0, xb3, xb1 = ([] if want else None), ([] if want else None), ([] if want else None)
        sp0, sp3 = [], []                    # the convolutions' partial BatchNorm statistics (where their kernel writes them)
        z16 = self._z16
        if first_input is not None:
            x32, spec64, denom = first_input
            z0 = K.conv3x3_c1_bn_relu(self.P[prefix + ".0.w"], None, None, x32=x32, spec64=spec64, denom=denom,
                                      per_clip=True, relu=False, out_dtype=torch.bfloat16 if z16 else torch.float32,
                                      stats_out=sp0 if (C1_STATS and self.precision >= 1) else None)
        else:
            z0 = conv_mfma(src0, self.P[prefix + ".0.w"], cout, in_affine=aff0, x1=src1, precision=self.precision, x0_bf16_out=xb0,
                           stats_out=sp0, x1_bf16_out=xb1, out_bf16=z16)
        st0 = self._bn_stats(z0, prefix + ".1", prefix + ".1.g", prefix + ".1.b", part=sp0[0] if sp0 else None)
        z3 = conv_mfma(z0, self.P[prefix + ".3.w"], cout, in_affine=st0, precision=self.precision, x0_bf16_out=xb3, stats_out=sp3, out_bf16=z16)
        st3 = self._bn_stats(z3, prefix + ".4", prefix + ".4.g", prefix + ".4.b", part=sp3[0] if sp3 else None)
        if drop_id is not None and self.rate > 0:
            st3.drop = dropout_spec(self.drop_seed + 16 * self.step_count + drop_id, self.rate)
        return dict(prefix=prefix, src0=src0, aff0=aff0, src1=src1, first_input=first_input, z0=z0, st0=st0, z3=z3, st3=st3,
                    xb0=xb0[0] if xb0 else None, xb3=xb3[0] if xb3 else None, xb1=xb1[0] if xb1 else None)

    def forward(self, x32=None, spec64=None, denom=None):
        """Train-mode forward.  Input (B,F,T): float32 spectrogram, or raw float64 |STFT| + per-clip denominators
        (the divide + .float() of train.py:264-272 is fused into the first conv).  Returns pred (B,F,T) float32."""
        global _PACK_CACHE
        _PACK_CACHE = self._packs if BATCH_REPACK else None
        if BATCH_REPACK:
            self._packs.refresh(self.device)         # every image recorded so far, from the CURRENT weights, in one launch
        try:
            return self._forward(x32, spec64, denom)
        finally:
            _PACK_CACHE = None

    def _forward(self, x32=None, spec64=None, denom=None):
        recs = {}
        src = x32 if x32 is not None else spec64
        self._z16 = self._z16_for(src.shape[1], src.shape[2])
        if x32 is None and PRENORMALISE_INPUT:
            # the divide + .float() of train.py:264-272 once, as its own 50 MB pass: the same float64 quotient rounded to float32 that the first
            # layer's kernels form on the fly -- but those formed it nine times per pixel (forward) and again in the weight gradient, a float64
            # division each time (conv3x3_c1_kernel 298 -> ... us, wgrad_c1_kernel 459 -> ... us per 64 clips)
            x32 = torch.empty(spec64.shape, dtype=torch.float32, device=spec64.device)
            check(lib().mfpa_normalize_f32(ptr(spec64), spec64.shape[0], spec64.shape[1] * spec64.shape[2], ptr(denom), ptr(x32), stream()),
                  "mfpa_normalize_f32")
            spec64 = denom = None
        r = self._dconv_fwd(ENC[0], None, None, first_input=(x32, spec64, denom))
        recs["inc"] = r
        prev = r
        for i, name in enumerate(ENC[1:]):
            z, st = prev["z3"], prev["st3"]
            B, H, W, C = z.shape
            p = torch.empty((B, H // 2, W // 2, C), dtype=torch.bfloat16 if self._z16 else torch.float32, device=z.device)
            check(lib().mfpa_bn_relu_pool(ptr(z), B, H, W, C, ptr(st.scale), ptr(st.shift), ptr(p), st.drop[0], st.drop[1],
                                          st.drop[2], _is16(z), _is16(p), stream()), "mfpa_bn_relu_pool")
            r = self._dconv_fwd(name, p, None, drop_id=i)                 # x2..x5 = dropout(downN(...))
            recs[name] = r
            prev = r
        skips = [recs[ENC[3]], recs[ENC[2]], recs[ENC[1]], recs["inc"]]
        for name, skip in zip(DEC, skips):
            xbu = [] if self.wgrad_precision == 2 else None          # bf16 copy of the activated input: the weight gradient's operand
            u = conv_mfma(prev["z3"], self.P[name + ".up.w"], self.P[name + ".up.w"].shape[1], mode=1,
                          in_affine=prev["st3"], out_shift=self.P[name + ".up.b"], precision=self.precision, x0_bf16_out=xbu, out_bf16=self._z16)
            if skip["z3"].shape[1] - u.shape[1] > 1 or skip["z3"].shape[2] - u.shape[2] > 1:
                raise NotImplementedError("skip/upsample size difference > 1 (needs top/left padding offsets)")
            r = self._dconv_fwd(name + ".conv.double_conv", skip["z3"], skip["st3"], src1=u,
                                drop_id=4 if name == DEC[0] else None)  # x = dropout(up1(x5, x4))
            r["up_in"], r["up_name"], r["u"] = prev, name, u
            r["up_xb"] = xbu[0] if xbu else None
            recs[name] = r
            prev = r
        z, st = prev["z3"], prev["st3"]
        pred = torch.empty(z.shape[:3], dtype=torch.float32, device=z.device)
        wb = self.P["outc.wb"]
        check(lib().mfpa_outconv_fwd(ptr(z), _npix(z), 64, ptr(st.scale), ptr(st.shift), ptr(wb), ptr(wb[64:]),
                                     ptr(pred), _is16(z), stream()), "mfpa_outconv_fwd")
        self._recs = recs
        return pred

    # ------------------------------------------------------------------ backward
    def _pool_bwd_add(self, r, dy, d_p):
        """dy (the skip path's gradient of an encoder block's output) += route(d_p) through the block's MaxPool2d, in place; returns (dy,
        the BatchNorm-backward row partials the same pass formed, or None)."""
        z, st = r["z3"], r["st3"]
        B, H, W, C = z.shape
        if dy.dtype == torch.bfloat16:
            dy = dy.float()                                            # (a bfloat16 skip gradient on the un-fused path: widened first)
        if FUSE_POOL_BWD_SUMS and 256 % (C // 4) == 0:
            # the pass that finishes dy (skip gradient + routed pool gradient) also forms the partial sums of the BatchNorm
            # backward the block's _dconv_bwd starts with: no separate reduction pass over dy and z
            part = torch.empty((B * (H // 2), 2, C), dtype=torch.float32, device=z.device)
            check(lib().mfpa_maxpool2_bwd_add_sums(ptr(z), B, H, W, C, ptr(st.scale), ptr(st.shift), ptr(st.mean), ptr(st.invstd),
                                                   ptr(d_p), ptr(dy), st.drop[0], st.drop[1], st.drop[2], ptr(part), _is16(z), stream()),
                  "mfpa_maxpool2_bwd_add_sums")
            return dy, part
        check(lib().mfpa_maxpool2_bwd_add(ptr(z), B, H, W, C, ptr(st.scale), ptr(st.shift), ptr(d_p), ptr(dy),
                                          st.drop[0], st.drop[1], st.drop[2], _is16(z), stream()), "mfpa_maxpool2_bwd_add")
        return dy, None

    def _dconv_bwd(self, r, dy, need_input_grad=True, dy_part=None, dy_rank1=None, pool_dp=None):
        """dy: gradient w.r.t. the DoubleConv's (lazy BN+ReLU) output.  Returns gradients w.r.t. (src0, src1).
        dy_part: the partial sums of this block's last BatchNorm backward when the pass that finished dy already formed them
        (mfpa_maxpool2_bwd_add_sums, mfpa_outconv_bwd_sums); dy_rank1 = (dpred, w) with dy None: dy = dpred x w, never written.
        pool_dp (an encoder block): dy is only the SKIP path's part; the gradient that comes back through the block's MaxPool2d (pool_dp,
        w.r.t. the pooled tensor) still has to be routed into it -- here, so that the fused form can skip the finished dy altogether."""
        prefix = r["prefix"]
        cout = r["z3"].shape[-1]
        H_, W_ = r["z3"].shape[1], r["z3"].shape[2]
        lay = weight_layout(H_, W_, cout, cout, self.precision)
        wg16 = bf16_wgrad(cout, cout, self.wgrad_precision, r["xb3"] is not None)
        # plain bf16 step: when both consumers of dz (input-gradient convolution on conv_wd16_kernel, weight gradient) read its bf16
        # copy, the float32 dz is never written (mfpa_bn_relu_bwd(write_f32 = 0)) and the convolution's loader moves half the bytes
        only16 = USE_BF16_DZ and self.precision == 2 and lay == 2 and wg16
        fused_pool = False
        if pool_dp is not None:
            z3, st3 = r["z3"], r["st3"]
            fused_pool = bool(POOL_BWD_FUSED and only16 and not self.sync_bn and FUSED_FINISH and 256 % (cout // 4) == 0
                              and (cout & (cout - 1)) == 0 and cout <= 1024)
            if fused_pool:
                # mfpa_maxpool2_bwd_bn_relu_bwd: the finished dy (skip + routed pool gradient) is neither written nor read back
                Bq = z3.shape[0]
                part = torch.empty((Bq * (H_ // 2), 2, cout), dtype=torch.float32, device=z3.device)
                coef = torch.empty((3, cout), dtype=torch.float32, device=z3.device)
                dz16 = torch.empty(z3.shape, dtype=torch.bfloat16, device=z3.device)
                st3.count_host = float(_npix(z3))
                check(lib().mfpa_maxpool2_bwd_bn_relu_bwd(ptr(z3), Bq, H_, W_, cout, ptr(self.P[prefix + ".4.g"]), ptr(st3.scale), ptr(st3.shift),
                                                          ptr(st3.mean), ptr(st3.invstd), ptr(pool_dp), ptr(dy), _is16(dy), st3.drop[0], st3.drop[1],
                                                          st3.drop[2], ptr(part), ptr(self.G[prefix + ".4.g"]), ptr(self.G[prefix + ".4.b"]),
                                                          ptr(coef), ptr(self.workspace), ptr(dz16), _is16(z3), stream()),
                      "mfpa_maxpool2_bwd_bn_relu_bwd")
                dz3 = None
            else:
                dy, dy_part = self._pool_bwd_add(r, dy, pool_dp)
        if not fused_pool:
            dz3, dz16 = self._bn_relu_bwd(dy, r["z3"], r["st3"], prefix + ".4.g", prefix + ".4.b", bf16_copy=wg16, write_f32=not only16,
                                          part=dy_part, rank1=dy_rank1)
        wgrad_mfma(dz3, r["z0"], self.G[prefix + ".3.w"], cout, in_affine=r["st0"], precision=self.wgrad_precision, dz_bf16=dz16,
                   x0_bf16=r["xb3"])
        r["xb3"] = None
        wt3 = pack_weights(self.P[prefix + ".3.w"], self.precision, flip_transpose=True, layout=lay)   # [tap'][ci][co]
        spm = [] if r["st0"].drop[1] == 0 else None      # dmid is dy of relu(bn(z0)): the BatchNorm backward's reductions in the epilogue
        cin0 = 0 if r["first_input"] is not None else r["src0"].shape[-1] + (0 if r["src1"] is None else r["src1"].shape[-1])
        wg16_0 = cin0 > 0 and bf16_wgrad(cout, cin0, self.wgrad_precision, r["xb0"] is not None)
        c0_ = 0 if r["first_input"] is not None else r["src0"].shape[-1]
        c1_ = 0 if (r["first_input"] is not None or r["src1"] is None) else r["src1"].shape[-1]
        only16_0 = (USE_BF16_DZ and self.precision == 2 and wg16_0 and need_input_grad and c0_ > 0
                    and weight_layout(H_, W_, cout, c0_, self.precision) == 2 and (c1_ == 0 or weight_layout(H_, W_, cout, c1_, self.precision) == 2))
        # activations as bfloat16 (Z16_ACTIVATIONS): dmid too leaves its convolution as bfloat16 only when the BatchNorm backward that reads
        # it writes nothing but the bf16 dz (and takes its two reductions from that convolution's epilogue)
        dmid16 = bool(DY16_MID and only16 and only16_0 and spm is not None and r["z0"].dtype == torch.bfloat16)
        dmid = conv_mfma(dz16 if only16 else dz3, wt3, cout, precision=self.precision, packed=True, w_layout=lay, stats_out=spm,
                         bwd_of=(r["z0"], r["st0"]), out_bf16=dmid16)
        del dz3, dz16
        wg16, only16 = wg16_0, only16_0
        if r["first_input"] is not None and C1_WGRAD_BF16 and self.precision == 2 and self.wgrad_precision == 2:
            wg16 = only16 = True             # the first layer's weight gradient reads the bf16 dz too (plain-bf16 step): no float32 dz anywhere
        if dmid16 and not spm:
            raise RuntimeError("a bfloat16 dmid needs the convolution's BatchNorm-backward partial sums")
        dz0, dz16 = self._bn_relu_bwd(dmid, r["z0"], r["st0"], prefix + ".1.g", prefix + ".1.b", bf16_copy=wg16,
                                      part=spm[0] if spm else None, write_f32=not only16)
        if r["first_input"] is not None:
            x32, spec64, denom = r["first_input"]
            dsrc = dz0 if dz0 is not None else dz16
            B, H, W, C = dsrc.shape
            check(lib().mfpa_wgrad_c1(ptr(dsrc), ptr(x32), ptr(spec64), ptr(denom), B, H, W, C, ptr(self.G[prefix + ".0.w"]),
                                      _is16(dsrc), stream()), "mfpa_wgrad_c1")
            return None, None
        wgrad_mfma(dz0, r["src0"], self.G[prefix + ".0.w"], cout, in_affine=r["aff0"], x1=r["src1"],
                   precision=self.wgrad_precision, dz_bf16=dz16, x0_bf16=r["xb0"], x1_bf16=r["xb1"])
        r["xb0"] = r["xb1"] = None
        if not need_input_grad:
            return None, None
        w0 = self.P[prefix + ".0.w"]                                                # (9, cout, cin)
        c0 = r["src0"].shape[-1]
        lay = weight_layout(H_, W_, cout, c0, self.precision)
        dsrc = dz16 if only16 else dz0
        # a decoder block's gradient w.r.t. its skip source waits for the encoder's backward: as bfloat16 (SKIP_GRAD_BF16) when the kernel can
        # write that (bf16 dz in, conv_wd16_kernel) -- the pool-backward passes read it as it is
        d0_16 = bool(SKIP_GRAD_BF16 and self._z16 and only16 and r["src1"] is not None and lay == 2)
        d0 = conv_mfma(dsrc, pack_weights(w0, self.precision, True, 0, c0, layout=lay), c0, precision=self.precision, packed=True, w_layout=lay,
                       out_bf16=d0_16)
        d1 = None
        if r["src1"] is not None:
            c1 = r["src1"].shape[-1]
            lay = weight_layout(H_, W_, cout, c1, self.precision)
            sp = []                          # per-channel sums of d1 (the transposed convolution's bias gradient) from the kernel's epilogue
            yb = [] if self.wgrad_precision == 2 else None       # and its bf16 copy (the transposed convolution's weight gradient reads it)
            d1 = conv_mfma(dsrc, pack_weights(w0, self.precision, True, c0, c1, layout=lay), c1, out_hw=(r["src1"].shape[1], r["src1"].shape[2]),
                           precision=self.precision, packed=True, w_layout=lay, stats_out=sp, y_bf16_out=yb)
            r["d1_sums"] = sp[0] if sp else None
            r["d1_bf16"] = yb[0] if yb else None
        return d0, d1

    def backward(self, dpred):
        global _PACK_CACHE
        _PACK_CACHE = self._packs if BATCH_REPACK else None      # (images made at the start of this step's forward: the weights have not changed since)
        try:
            self._backward(dpred)
        finally:
            _PACK_CACHE = None

    def _backward(self, dpred):
        recs = self._recs
        self.flat_g.zero_()
        last = recs[DEC[-1]]
        z, st = last["z3"], last["st3"]
        wb = self.P["outc.wb"]
        dy, dy_part, dy_rank1 = None, None, None
        if RANK1_OUTCONV_BWD and st.drop[1] == 0:
            rows = ctypes.c_int(0)
            check(lib().mfpa_outconv_bwd_rows(_npix(z), 64, ctypes.byref(rows)), "mfpa_outconv_bwd_rows")
            dy_part = torch.empty((rows.value, 2, 64), dtype=torch.float32, device=z.device)
            check(lib().mfpa_outconv_bwd_sums(ptr(z), ptr(dpred), _npix(z), 64, ptr(st.scale), ptr(st.shift), ptr(st.mean), ptr(st.invstd),
                                              ptr(wb), ptr(self.G["outc.wb"]), ptr(self.workspace), ptr(dy_part), _is16(z), stream()),
                  "mfpa_outconv_bwd_sums")
            dy_rank1 = (dpred, wb)
        else:
            dy = torch.empty(z.shape, dtype=torch.float32, device=z.device)
            check(lib().mfpa_outconv_bwd(ptr(z), ptr(dpred), _npix(z), 64, ptr(st.scale), ptr(st.shift), ptr(wb), ptr(dy),
                                         ptr(self.G["outc.wb"]), ptr(self.workspace), _is16(z), stream()), "mfpa_outconv_bwd")
        handles = []
        dskip = {}
        enc_of_dec = {DEC[0]: ENC[3], DEC[1]: ENC[2], DEC[2]: ENC[1], DEC[3]: ENC[0]}
        for name in reversed(DEC):                                                  # up4 ... up1
            r = recs[name]
            d_skip, d_u = self._dconv_bwd(r, dy, dy_part=dy_part, dy_rank1=dy_rank1)
            dy_part = dy_rank1 = None
            dskip[enc_of_dec[name]] = d_skip
            # transposed conv: bias, weight and input gradients
            cout = d_u.shape[-1]
            if r.get("d1_sums") is not None:                              # sum over pixels of d_u: the convolution that produced it already has it
                sums = torch.empty(2 * cout, dtype=torch.float64, device=d_u.device)
                check(lib().mfpa_conv_stats_reduce(ptr(r["d1_sums"]), r["d1_sums"].shape[0], cout, ptr(sums), ptr(self.workspace),
                                                   stream()), "mfpa_conv_stats_reduce")
                self.G[name + ".up.b"].copy_(sums.view(cout, 2)[:, 0])
                r["d1_sums"] = None
            else:
                check(lib().mfpa_colsum(ptr(d_u), _npix(d_u), cout, ptr(self.G[name + ".up.b"]), ptr(self.workspace),
                                        stream()), "mfpa_colsum")
            prev = r["up_in"]
            wgrad_mfma(d_u, prev["z3"], self.G[name + ".up.w"], cout, mode=1, in_affine=prev["st3"],
                       precision=self.wgrad_precision, dz_bf16=r.get("d1_bf16"), x0_bf16=r.get("up_xb"))
            r["d1_bf16"] = r["up_xb"] = None
            wt = pack_weights(self.P[name + ".up.w"], self.precision, flip_transpose=True)   # (4, cin, cout), taps kept
            dy = conv_mfma(d_u, wt, wt.shape[1], mode=2, precision=self.precision, packed=True)
            handles.append(self._reduce_bucket(name))
        pool_dp = None
        for i in range(len(ENC) - 1, -1, -1):                                       # down4 ... inc
            name = ENC[i]
            r = recs[name if i else "inc"]
            d_p, _ = self._dconv_bwd(r, dy, pool_dp=pool_dp)
            if i:
                dy, pool_dp = dskip[ENC[i - 1]], d_p                   # the block below: its skip gradient, and what comes back through its pool
            handles.append(self._reduce_bucket(name))
        ev = None
        if self.comm_wait_events is not None and any(h is not None for h in handles):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
        for h in handles:
            if h is not None:
                h.wait()
        if ev is not None:
            ev1 = torch.cuda.Event(enable_timing=True)
            ev1.record()
            self.comm_wait_events.append((ev, ev1))
        self._recs = None

    def _reduce_bucket(self, name):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return None
        if dist.get_world_size(self.group) == 1 and not self.collectives_at_world_one:
            return None
        for bname, s, e in self.buckets:
            if bname == name:
                self.comm_calls += 1; self.comm_bytes += 4 * (e - s)
                return dist.all_reduce(self.flat_g[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return None

    # ------------------------------------------------------------------ loss / optimiser / step
    def l1_loss(self, pred, target64, want_grad=True):
        n = pred.numel()
        dpred = torch.empty_like(pred) if want_grad else None
        check(lib().mfpa_l1_loss(ptr(pred), ptr(target64), n, ptr(dpred), ptr(self.loss), ptr(self.workspace), stream()),
              "mfpa_l1_loss")
        return self.loss, dpred

    def optimizer_step(self):
        import torch.distributed as dist
        world = dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1
        self.step_count += 1
        self._packs.stale()
        check(lib().mfpa_adam_step(ptr(self.flat_p), ptr(self.flat_g), ptr(self.flat_m), ptr(self.flat_v), self.n_params,
                                   self.lr, self.betas[0], self.betas[1], self.eps, self.step_count, 1.0 / world,
                                   stream()), "mfpa_adam_step")

    def train_step(self, aug_spec64, aug_denom, clean_spec64):
        """One optimisation step on spectrograms: aug_spec64 raw float64 |STFT| (B,F,T) with its normaliser
        aug_denom (B,), clean_spec64 the normalised float64 target.  Returns the loss (device float64 scalar)."""
        if self.sync_bn:                      # one host round trip per step: global batch size / local batch size
            import torch.distributed as dist
            nb = torch.tensor([float(aug_spec64.shape[0])], dtype=torch.float64, device=aug_spec64.device)
            dist.all_reduce(nb, op=dist.ReduceOp.SUM, group=self.group)
            self.comm_calls += 1; self.comm_bytes += 8
            self._global_over_local_batch = float(nb.item()) / float(aug_spec64.shape[0])
        pred = self.forward(spec64=aug_spec64, denom=aug_denom)
        loss, dpred = self.l1_loss(pred, clean_spec64)
        self.backward(dpred)
        self.optimizer_step()
        return loss
