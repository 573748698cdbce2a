#!/usr/bin/env python3
"""Per-layer timing of the UNet's 3x3 weight-gradient launches (64 clips, random operands, HIP events on the launch stream).
Every layer takes bf16 operands (precision 3) -- the copies the forward convolution and the BatchNorm backward write in the engine.
usage: exp_wgrad.py [--lib PATH] [--reps N] [--clips B]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--clips", type=int, default=64)
ap.add_argument("--only", default=None, help="substring of the layer names to run")
ap.add_argument("--fp32-operands", action="store_true", help="layers below ops_train.BF16_WGRAD_MIN_CH channels on the fp32-operand kernel (the engine before the forward convolutions wrote the bf16 copies)")
args = ap.parse_args()
if args.lib:
    from musicfpaugment_amd import _lib
    _lib.set_library_path(args.lib)
from musicfpaugment_amd import ops_train as T
B = args.clips
layers = [("inc.3   64->64   @257x251", 257, 251, 64, 0, 64), ("up4.0  64+64->64 @257x251", 257, 251, 64, 64, 64),
          ("up4.3   64->64   @257x251", 257, 251, 64, 0, 64),
          ("d1.0    64->128  @128x125", 128, 125, 64, 0, 128), ("d1.3   128->128  @128x125", 128, 125, 128, 0, 128),
          ("up3.0 128+128->128 @128x125", 128, 125, 128, 128, 128), ("up3.3  128->128  @128x125", 128, 125, 128, 0, 128),
          ("d2.0   128->256  @64x62", 64, 62, 128, 0, 256), ("d2.3   256->256  @64x62", 64, 62, 256, 0, 256),
          ("up2.0 256+256->256 @64x62", 64, 62, 256, 256, 256), ("up2.3  256->256  @64x62", 64, 62, 256, 0, 256),
          ("d3.0   256->512  @32x31", 32, 31, 256, 0, 512), ("d3.3   512->512  @32x31", 32, 31, 512, 0, 512),
          ("up1.0 512+512->512 @32x31", 32, 31, 512, 512, 512), ("up1.3  512->512  @32x31", 32, 31, 512, 0, 512),
          ("d4.0   512->1024 @16x15", 16, 15, 512, 0, 1024), ("d4.3  1024->1024 @16x15", 16, 15, 1024, 0, 1024)]
def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.reps * 1e-3
tot = {2: 0.0, 3: 0.0}
for name, H, W, C0, C1, Co in layers:
    if args.only and args.only not in name: continue
    dz = torch.randn(B, H, W, Co, device="cuda") * 0.1
    x0 = torch.randn(B, H, W, C0, device="cuda")
    x1 = torch.randn(B, H - 1 if C1 else H, W - 1 if C1 else W, C1, device="cuda") if C1 else None
    dw = torch.zeros(9, Co, C0 + C1, device="cuda")
    aff = T.Stats(C0, "cuda")
    aff.scale.copy_(torch.rand(C0, device="cuda") + 0.5); aff.shift.copy_(torch.randn(C0, device="cuda") * 0.1)
    prec = 2 if args.fp32_operands and min(Co, C0 + C1) < T.BF16_WGRAD_MIN_CH else 3
    if prec == 3:                                # as the engine issues them: both operands as the bf16 copies its other kernels wrote
        dzb, x0b = T.act_to_bf16(dz), T.act_to_bf16(x0, aff)
        x1b = None if x1 is None else T.act_to_bf16(x1)
        fn = lambda: T.wgrad_mfma(dzb, x0b, dw, Co, x1=x1b, precision=3)
    else:
        fn = lambda: T.wgrad_mfma(dz, x0, dw, Co, in_affine=aff, x1=x1, precision=2)
    t = timed(fn); tot[prec] += t
    fl = 2.0 * B * H * W * (C0 + C1) * Co * 9
    print(f"{name:30s} precision {prec} {t*1e6:8.1f} us {fl/t/1e12:6.1f} TF/s", flush=True)
print(f"sum fp32 operands {tot[2]*1e3:.3f} ms  bf16 operands {tot[3]*1e3:.3f} ms")
