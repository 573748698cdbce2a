#!/usr/bin/env python3
"""Per-layer timing of the UNet's 3x3 convolutions (64 clips, random operands, HIP events on the launch stream).
usage: exp_conv.py [--lib PATH] [--precision 0|1] [--reps N]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--precision", type=int, default=1)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--clips", type=int, default=64)
ap.add_argument("--c64", action="store_true", help="the three 64-channel full-resolution launches exactly as the UNet issues them (fused first layer + pool, "
                "two-source concat, fused OutConv without store)")
ap.add_argument("--both", action="store_true", help="precision 1: time the LDS-staged kernel and the weights-direct kernel side by side")
args = ap.parse_args()
if args.lib:
    from musicfpaugment_amd import _lib
    _lib.set_library_path(args.lib)
from musicfpaugment_amd import ops_unet as K
prec, B = args.precision, args.clips
layers = [("inc.3   64->64   @257x251", 257, 251, 64, 64), ("up4.0  128->64   @257x251", 257, 251, 128, 64),
          ("d1.0    64->128  @128x125", 128, 125, 64, 128), ("d1.3   128->128  @128x125", 128, 125, 128, 128),
          ("up3.0  256->128  @128x125", 128, 125, 256, 128), ("d2.0   128->256  @64x62", 64, 62, 128, 256),
          ("d2.3   256->256  @64x62", 64, 62, 256, 256), ("up2.0  512->256  @64x62", 64, 62, 512, 256),
          ("d3.0   256->512  @32x31", 32, 31, 256, 512), ("d3.3   512->512  @32x31", 32, 31, 512, 512),
          ("up1.0 1024->512  @32x31", 32, 31, 1024, 512), ("d4.0   512->1024 @16x15", 16, 15, 512, 1024),
          ("d4.3  1024->1024 @16x15", 16, 15, 1024, 1024)]
def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.reps * 1e-3

if args.c64:
    H, W = 257, 251
    g = torch.Generator(device="cuda").manual_seed(0)
    sc = torch.ones(64, device="cuda"); sh = torch.zeros(64, device="cuda")
    spec = torch.rand(B, H, W, device="cuda", dtype=torch.float64); den = torch.ones(B, device="cuda", dtype=torch.float64)
    w1 = torch.randn(9, 64, device="cuda") * 0.1
    w64 = K.split_bf16x3(torch.randn(9, 64, 64, device="cuda") * 0.05)
    w128 = K.split_bf16x3(torch.randn(9, 64, 128, device="cuda") * 0.05)
    x = torch.randn(B, H, W, 64, device="cuda"); u = torch.randn(B, H - 1, W - 1, 64, device="cuda")
    wo = torch.randn(64, device="cuda")
    runs = [("inc.3  c1src + pool", 64, lambda: K.conv3x3_fused(None, w64, sc, sh, precision=1, pool=True, c1=dict(spec64=spec, denom=den, w=w1, scale=sc, shift=sh))),
            ("up4.0  skip 64 + up 64 (concat)", 128, lambda: K.conv3x3_fused(x, w128, sc, sh, x1=u, precision=1)),
            ("up4.3  + OutConv, no store", 64, lambda: K.conv3x3_fused(x, w64, sc, sh, precision=1, out1x1=(wo, 0.1), store=False)),
            ("plain 64->64 with store", 64, lambda: K.conv3x3_fused(x, w64, sc, sh, precision=1))]
    tot = 0.0
    for name, ci, fn in runs:
        t = timed(fn); tot += t
        fl = 2.0 * B * H * W * ci * 64 * 9
        print(f"{name:34s} {t*1e6:8.1f} us {fl/t/1e12:6.1f} TF/s", flush=True)
    print(f"sum of the first three {sum(0 for _ in ())+0:.0f}", end="")
    print(f" {tot*1e3:.3f} ms (all four)")
    sys.exit(0)

if args.both:
    from musicfpaugment_amd._lib import lib
    tl = td = 0.0
    for name, H, W, Ci, Co in layers:
        x = torch.randn(B, H, W, Ci, device="cuda")
        w = torch.randn(9, Co, Ci, device="cuda") * 0.05
        w3 = K.split_bf16x3(w)
        sc = torch.ones(Co, device="cuda"); sh = torch.zeros(Co, device="cuda")
        fl = 2.0 * B * H * W * Ci * Co * 9
        a = timed(lambda: K.conv3x3_fused(x, w3, sc, sh, precision=1))
        if lib().mfpa_conv_weight_layout(H, W, Ci, Co, 0, 1) == K.frag_layout() and K.frag_layout():
            wf = (K.frag_layout(), K.split_bf16x3_frag(w, K.frag_layout()))
            d = timed(lambda: K.conv3x3_fused(x, w3, sc, sh, precision=1, wf=wf))
        else:
            d = a
        tl += a; td += d
        ya = K.conv3x3_fused(x, w3, sc, sh, precision=1)[0]
        yd = K.conv3x3_fused(x, w3, sc, sh, precision=1, wf=wf)[0] if d is not a else ya
        diff = float((ya - yd).abs().max() / ya.abs().max())
        print(f"{name:28s} lds {a*1e6:8.1f} us {fl/a/1e12:6.1f} TF/s | direct {d*1e6:8.1f} us {fl/d/1e12:6.1f} TF/s  ({(a/d-1)*100:+.1f} %)  max diff {diff:.1e}", flush=True)
        del ya, yd
    print(f"sum lds {tl*1e3:.3f} ms direct {td*1e3:.3f} ms")
    sys.exit(0)

tot = 0.0
for name, H, W, Ci, Co in layers:
    x = torch.randn(B, H, W, Ci, device="cuda")
    w = torch.randn(9, Co, Ci, device="cuda") * 0.05
    wp = K.split_bf16x3(w) if prec else w
    sc = torch.ones(Co, device="cuda"); sh = torch.zeros(Co, device="cuda")
    K.conv3x3_bn_relu(x, wp, sc, sh, precision=prec); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps): K.conv3x3_bn_relu(x, wp, sc, sh, precision=prec)
    e1.record(); torch.cuda.synchronize()
    dt = e0.elapsed_time(e1) / args.reps * 1e-3
    fl = 2.0 * B * H * W * Ci * Co * 9
    tot += dt
    print(f"{name:28s} {dt*1e6:9.1f} us  {fl/dt/1e12:7.1f} TF/s algorithmic", flush=True)
print(f"sum {tot*1e3:.3f} ms")
