#!/usr/bin/env python3
"""Per decoder level, same call: the Up block's first half as the two launches of rounds 1-5 (mfpa_convT2x2 + mfpa_conv_mfma on the product
routing) against ONE mfpa_upconv_fused launch (the transposed convolution folded into its consumer, csrc/unet_up.hip).  Random operands,
HIP events on the launch stream.  usage: exp_upconv.py [--clips 64|128] [--reps N] [--levels up4,up3,up2,up1]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--clips", type=int, default=64)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--levels", default="up4,up3,up2,up1")
ap.add_argument("--precision", type=int, default=1, help="1 bf16x3 (default), 0 exact fp32 products")
args = ap.parse_args()
from musicfpaugment_amd import ops_unet as K
from musicfpaugment_amd._lib import lib
B = args.clips
LEVELS = {"up4": (257, 251, 128, 125, 64), "up3": (128, 125, 64, 62, 128), "up2": (64, 62, 32, 31, 256), "up1": (32, 31, 16, 15, 512)}


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.reps * 1e-3


print(f"# {B} clips, {args.reps} repetitions each; us per launch; 'two' = convT + conv (product routing of round 5), 'fused' = mfpa_upconv_fused")
tot2 = totf = 0.0
for name in args.levels.split(","):
    H, W, Hl, Wl, C = LEVELS[name]
    Cs = Cu = Cout = C
    Cl = 2 * C
    g = torch.Generator(device="cuda").manual_seed(1)
    skip = torch.randn(B, H, W, Cs, device="cuda", generator=g)
    low = torch.randn(B, Hl, Wl, Cl, device="cuda", generator=g)
    w3 = torch.randn(9, Cout, Cs + Cu, device="cuda", generator=g) / (9 * (Cs + Cu)) ** 0.5
    wt = torch.randn(4, Cu, Cl, device="cuda", generator=g) / Cl ** 0.5
    bt = torch.randn(Cu, device="cuda", generator=g)
    sc = torch.rand(Cout, device="cuda", generator=g) + 0.5
    sh = torch.randn(Cout, device="cuda", generator=g) * 0.1
    pw = {"L.conv.double_conv.0.w": w3, "L.up.w": wt, "L.up.b": bt, "L.conv.double_conv.0.scale": sc, "L.conv.double_conv.0.shift": sh}
    P = args.precision
    w3s, wts = (K.split_bf16x3(w3), K.split_bf16x3(wt)) if P == 1 else (w3, wt)
    wf = (2, K.split_bf16x3_frag(w3, 2)) if P == 1 else None
    wff = (2, K.split_bf16x3_frag(w3 * sc[None, :, None], 2)) if P == 1 else None
    t_ct = timed(lambda: K.convT2x2(low, wts, bt, precision=P))
    u = K.convT2x2(low, wts, bt, precision=P)
    t_cv = timed(lambda: K.conv3x3_fused(skip, w3s, sc, sh, x1=u, precision=P, wf=wf, wff=wff))
    two = K.conv3x3_fused(skip, w3s, sc, sh, x1=u, precision=P, wf=wf, wff=wff)[0]
    del u
    line = f"{name}  {Cl:4d}->{Cu:3d} up, {Cs + Cu:4d}->{Cout:3d} @{H}x{W}   two {t_ct * 1e6:7.1f} + {t_cv * 1e6:7.1f} = {(t_ct + t_cv) * 1e6:7.1f}"
    tot2 += t_ct + t_cv
    if lib().mfpa_upconv_serves(H, W, Hl, Wl, Cs, Cl, Cout) == 1:
        pk = K.pack_upconv(pw, "L", P)
        fn = lambda: K.upconv_fused(skip, low, pk["L.upc.wsk"], pk["L.upc.wup"], sh, pk["L.upc.bias"], Cout, precision=P)
        t_f = timed(fn)
        y = fn()
        diff = float((y - two).abs().max() / two.abs().max())
        line += f" | fused {t_f * 1e6:7.1f}  ({(t_ct + t_cv) / t_f - 1:+.1%})  max diff / max {diff:.1e}"
        totf += t_f
    else:
        line += " | fused: not served"
        totf += t_ct + t_cv
    print(line, flush=True)
print(f"sum two {tot2 * 1e3:.3f} ms, fused {totf * 1e3:.3f} ms")
