"""UNet spectrogram denoiser on MI355X -- host-side mirror of the reference's training/unet.py.

Same constructor, same module tree (hence the same 118 state_dict keys: published
checkpoints ``{"model_state_dict": ...}`` load unchanged), same ``forward((B,1,F,T)) ->
(B,1,F,T)`` contract.  The torch ``nn`` layers below are parameter containers only: forward
never calls them.  It runs the hand-written gfx950 kernels of libmfpa.so (csrc/unet.hip):
NHWC activations, float32 MFMA implicit-GEMM convolutions with the eval-mode BatchNorm
folded into a per-channel affine + ReLU epilogue, pad + concat folded into the decoder
convolutions' loaders.  There is no PyTorch/CPU fallback.

Reference: training/unet.py:8-25 DoubleConv, :28-38 Down, :41-65 Up, :68-74 OutConv,
:77-108 UNet.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn as nn

from .. import ops_unet as K
from .._lib import MfpaError, require_gpu


class _UNetTrainFn(torch.autograd.Function):
    """Autograd node around UNetTrainEngine.forward / .backward (no autograd inside: the engine keeps its own activations)."""

    @staticmethod
    def run_forward(module, eng, x):
        B, _, F_, T_ = x.shape
        eng.step_count = eng.fwd_count                 # the stateless dropout masks are keyed by (seed, forward count, layer)
        pred = eng.forward(x32=x.contiguous().view(B, F_, T_))
        eng.fwd_count += 1
        # BatchNorm side effects of a train-mode forward (running statistics, num_batches_tracked) land in the module's buffers
        names = [k for k in eng.running]
        bufs = dict(module.named_buffers())
        torch._foreach_copy_([bufs[k] for k in names], [eng.running[k] for k in names])
        torch._foreach_add_([b for k, b in bufs.items() if k.endswith("num_batches_tracked")], 1)
        return pred.view(B, 1, F_, T_)

    @staticmethod
    def forward(ctx, module, x, *params):
        eng = module.train_engine()
        pred = _UNetTrainFn.run_forward(module, eng, x)
        ctx.module, ctx.eng, ctx.token = module, eng, eng.fwd_count
        ctx.names = [k for k, _ in module.named_parameters()]
        return pred

    @staticmethod
    def backward(ctx, dpred):
        eng = ctx.eng
        if eng.fwd_count != ctx.token or eng._recs is None:
            raise RuntimeError("UNet backward: the activations of this forward are gone (another train-mode forward ran, or backward "
                               "was already called); the engine keeps ONE forward's activations")
        B = dpred.shape[0]
        eng.backward(dpred.contiguous().view(B, dpred.shape[2], dpred.shape[3]).float())
        grads = eng.named_grads()
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(eng.group) > 1:
            # the engine all-reduced (SUM) its gradient buckets over the data-parallel group: hand autograd the mean, like DDP
            inv = 1.0 / dist.get_world_size(eng.group)
            grads = {k: v * inv for k, v in grads.items()}             # (fresh tensors)
        else:
            # OWNED tensors: the BatchNorm / bias / OutConv entries of named_grads() are views into the engine's persistent flat
            # gradient buffer, which the next backward zeroes and rewrites.  AccumulateGrad steals what it is handed when
            # param.grad is None, so a view would make param.grad alias that buffer (accumulation over two backward passes would
            # read 2 * g2 instead of g1 + g2).
            grads = {k: v.clone() for k, v in grads.items()}
        return (None, None) + tuple(grads[k] for k in ctx.names)      # the 1-channel input gets no gradient (its layer's is skipped)


class DoubleConv(nn.Module):
    """(convolution => [BN] => ReLU) * 2 -- parameter container (training/unet.py:8-25)."""

    def __init__(self, in_channels, out_channels, mid_channels=None):
        super().__init__()
        if not mid_channels:
            mid_channels = out_channels
        self.double_conv = nn.Sequential(
            nn.Conv2d(in_channels, mid_channels, kernel_size=3, padding=1, bias=False),
            nn.BatchNorm2d(mid_channels),
            nn.ReLU(inplace=True),
            nn.Conv2d(mid_channels, out_channels, kernel_size=3, padding=1, bias=False),
            nn.BatchNorm2d(out_channels),
            nn.ReLU(inplace=True),
        )


class Down(nn.Module):
    """MaxPool2d(2) then DoubleConv (training/unet.py:28-38)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.maxpool_conv = nn.Sequential(nn.MaxPool2d(2), DoubleConv(in_channels, out_channels))


class Up(nn.Module):
    """ConvTranspose2d(k2, s2) then pad + concat + DoubleConv (training/unet.py:41-65)."""

    def __init__(self, in_channels, out_channels, bilinear=False):
        super().__init__()
        if bilinear:
            raise NotImplementedError("bilinear=True is never used by the reference's experiments "
                                      "(training/train.py:646 builds UNet(1, 1, rate=0.05)); not built")
        self.up = nn.ConvTranspose2d(in_channels, in_channels // 2, kernel_size=2, stride=2)
        self.conv = DoubleConv(in_channels, out_channels)


class OutConv(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=1)


class UNet(nn.Module):
    def __init__(self, n_channels, n_classes, rate=0, bilinear=False):
        super().__init__()
        if n_channels != 1 or n_classes != 1:
            raise NotImplementedError("the hot path is the 1-channel spectrogram denoiser UNet(1, 1, ...)")
        self.n_channels = n_channels
        self.n_classes = n_classes
        self.bilinear = bilinear
        self.dropout = nn.Dropout(rate)

        self.inc = DoubleConv(n_channels, 64)
        self.down1 = Down(64, 128)
        self.down2 = Down(128, 256)
        self.down3 = Down(256, 512)
        self.down4 = Down(512, 1024)
        self.up1 = Up(1024, 512, bilinear)
        self.up2 = Up(512, 256, bilinear)
        self.up3 = Up(256, 128, bilinear)
        self.up4 = Up(128, 64, bilinear)
        self.outc = OutConv(64, n_classes)

        self.max_clips_per_pass = 128      # activations of one pass: ~0.12 GB per 8 s clip (round 5: 64 -> 128, +0.8 % on the headline; 256 is no faster)
        self.two_streams = False           # experiment: alternate the passes of a batch on two streams
        # inference arithmetic of the MFMA convolutions: 0 = fp32 MFMA (exact fp32 products, rel. L1 ~1e-6 vs the
        # reference), 1 = bf16x3 split (3 bf16 MFMAs per product, rel. L1 ~2e-5; tolerance is 1e-4)
        self.precision = 0
        self._packed: Optional[Dict[str, torch.Tensor]] = None
        self._packed_key = None

    # ------------------------------------------------------------------ packed weights
    def _weights_key(self):
        return tuple((p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers()))

    def packed_weights(self) -> Dict[str, torch.Tensor]:
        """Kernel-layout weights ([tap][Cout][Cin]) + folded eval BatchNorm; rebuilt when parameters change."""
        key = (self._weights_key(), self.precision)
        if self._packed is None or key != self._packed_key:
            self._packed = K.pack_unet_weights(self.state_dict(), self.precision)
            self._packed_key = key
        return self._packed

    # ------------------------------------------------------------------ forward
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        require_gpu(x, "UNet input")
        if x.dim() != 4 or x.shape[1] != 1:
            raise ValueError("expected (B, 1, F, T)")
        if x.dtype != torch.float32:
            raise TypeError("UNet input must be float32 (the reference casts with .float(), training/train.py:272)")
        if self.training:
            return self._forward_train(x)
        pw = self.packed_weights()
        B = x.shape[0]
        outs: List[torch.Tensor] = []
        for s in range(0, B, self.max_clips_per_pass):
            xs = x[s:s + self.max_clips_per_pass].contiguous()
            outs.append(K.unet_forward_eval(pw, x32=xs.view(xs.shape[0], xs.shape[2], xs.shape[3])))
        y = outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)
        return y.view(B, 1, x.shape[2], x.shape[3])

    # ------------------------------------------------------------------ train-mode forward under torch.autograd
    def train_engine(self):
        """The hand-written training engine (ops_train.UNetTrainEngine) bound to this module's parameters; created on first use,
        re-created when the module moved to another device."""
        from ..ops_train import UNetTrainEngine
        dev = next(self.parameters()).device
        eng = self.__dict__.get("_engine")
        if eng is None or eng.device != dev:
            eng = UNetTrainEngine(self, lr=1e-3, precision=self.train_precision, wgrad_precision=self.train_wgrad_precision)
            eng.fwd_count = 0
            self.__dict__["_engine"] = eng          # not a sub-module: no parameters of its own, nothing for state_dict()
            self.__dict__["_engine_key"] = None
        return eng

    train_precision = 0            # arithmetic of the training convolutions (0 = fp32 MFMA like the reference, 1 = bf16x3)
    train_wgrad_precision = 0      # arithmetic of the weight-gradient kernels (0 = fp32, 1 = bf16x3, 2 = bf16)

    def _forward_train(self, x: torch.Tensor) -> torch.Tensor:
        """The reference's training lines run unchanged on this module (training/train.py:273-316):

            predicted = model(x); loss = criterion(predicted, clean); optimizer.zero_grad(); loss.backward(); optimizer.step()

        Train-mode forward (BatchNorm batch statistics + running-stat update, nn.Dropout(rate) on x2..x5 and up1's output,
        training/unet.py:97-108) runs the engine's HIP kernels; the returned tensor carries a grad_fn whose backward runs the
        hand-written backward pass and hands the gradients of all 60 parameters (reference shapes) to autograd, which accumulates
        them into `param.grad` like any other op -- so torch.optim.Adam(model.parameters()) works as in the reference.  The fused
        Adam of training.train.Trainer stays the fast path (no per-step re-layout of the weights)."""
        eng = self.train_engine()
        params = [p for _, p in self.named_parameters()]
        key = tuple((p.data_ptr(), p._version) for p in params)
        if key != self.__dict__.get("_engine_key"):     # an optimiser step / load_state_dict changed the weights: re-lay them out
            eng.load_from_module()
            self.__dict__["_engine_key"] = key
        if not torch.is_grad_enabled():
            return _UNetTrainFn.run_forward(self, eng, x)
        return _UNetTrainFn.apply(self, x, *params)

    def denoise_spectrogram(self, spec64: torch.Tensor, clip_max: torch.Tensor, per_clip: bool) -> torch.Tensor:
        """Fused entry for the pipeline: raw float64 |STFT| (B,F,T) + maxima -> denoised (B,F,T) float32.
        The normalise + .float() step (peak_extractor.py:263-265 / train.py:272) runs inside the first conv."""
        require_gpu(spec64, "spectrogram")
        if self.training:
            raise MfpaError("denoise_spectrogram is the inference path; call .eval()")
        pw = self.packed_weights()
        B = spec64.shape[0]
        if not per_clip:
            clip_max = clip_max.max().expand(B).contiguous()
        outs = []
        ranges = [(s, min(B, s + self.max_clips_per_pass)) for s in range(0, B, self.max_clips_per_pass)]
        if self.two_streams and len(ranges) > 1:
            # alternate passes on two streams: the tail of one pass's launches (the last wave of workgroups, the 16 x 15 levels)
            # overlaps the other pass's kernels
            main = torch.cuda.current_stream(spec64.device)
            side = K.side_stream(spec64.device)
            side.wait_stream(main)
            for i, (s, e) in enumerate(ranges):
                if i % 2:
                    with torch.cuda.stream(side):
                        outs.append(K.unet_forward_eval(pw, spec64=spec64[s:e], denom=clip_max[s:e].contiguous()))
                else:
                    outs.append(K.unet_forward_eval(pw, spec64=spec64[s:e], denom=clip_max[s:e].contiguous()))
            main.wait_stream(side)
        else:
            for s, e in ranges:
                outs.append(K.unet_forward_eval(pw, spec64=spec64[s:e], denom=clip_max[s:e].contiguous()))
        return outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)
