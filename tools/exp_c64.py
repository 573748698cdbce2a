#!/usr/bin/env python3
"""The 64-channel full-resolution launches of the eval chain exactly as ops_unet.unet_forward_eval issues them (64 clips, 257 x 251):
inc.3 (first layer in the loader + pool), up4.0 (skip 64 + up 64 -> 64, weights-direct persistent form), up4.3 (+ OutConv, no store),
a plain 64 -> 64 with store, and the transposed convolution up4.up.  usage: exp_c64.py [--lib PATH] [--reps N]
(experiments build + MFPA_CONV_DBG: skip experiments on conv_wd16_kernel<.., WMW = 4>)."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--clips", type=int, default=64)
args = ap.parse_args()
if args.lib:
    from musicfpaugment_amd import _lib
    _lib.set_library_path(args.lib)
from musicfpaugment_amd import ops_unet as K
B, H, W = args.clips, 257, 251
def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.reps * 1e-3
lay = K.frag_layout()
sc = torch.ones(64, device="cuda"); sh = torch.zeros(64, device="cuda")
spec = torch.rand(B, H, W, device="cuda", dtype=torch.float64); den = torch.ones(B, device="cuda", dtype=torch.float64)
w1 = torch.randn(9, 64, device="cuda") * 0.1
def packs(co, ci):
    w = torch.randn(9, co, ci, device="cuda") * 0.05
    return K.split_bf16x3(w), (lay, K.split_bf16x3_frag(w, lay))
w64, wf64 = packs(64, 64)
w128, wf128 = packs(64, 128)
x = torch.relu(torch.randn(B, H, W, 64, device="cuda")); u = torch.randn(B, H - 1, W - 1, 64, device="cuda")
y2 = torch.relu(torch.randn(B, H // 2, W // 2, 128, device="cuda"))
wt = K.split_bf16x3(torch.randn(4, 64, 128, device="cuda") * 0.05); bt = torch.zeros(64, device="cuda")
wo = torch.randn(64, device="cuda")
runs = [("inc.3  c1src + pool", 64, lambda: K.conv3x3_fused(None, w64, sc, sh, precision=1, pool=True, wf=wf64, c1=dict(spec64=spec, denom=den, w=w1, scale=sc, shift=sh))),
        ("up4.0  skip 64 + up 64 (concat)", 128, lambda: K.conv3x3_fused(x, w128, sc, sh, x1=u, precision=1, wf=wf128)),
        ("up4.3  + OutConv, no store", 64, lambda: K.conv3x3_fused(x, w64, sc, sh, precision=1, out1x1=(wo, 0.1), store=False, wf=wf64)),
        ("plain 64->64 with store", 64, lambda: K.conv3x3_fused(x, w64, sc, sh, precision=1, wf=wf64)),
        ("plain 64->64 + pool", 64, lambda: K.conv3x3_fused(x, w64, sc, sh, precision=1, pool=True, wf=wf64))]
print("MFPA_CONV_DBG =", os.environ.get("MFPA_CONV_DBG", "0"))
for name, ci, fn in runs:
    t = timed(fn)
    fl = 2.0 * B * H * W * ci * 64 * 9
    print(f"{name:34s} {t*1e6:8.1f} us {fl/t/1e12:6.1f} TF/s", flush=True)
t = timed(lambda: K.convT2x2(y2, wt, bt, precision=1))
print(f"{'up4.up convT 128->64 @128x125':34s} {t*1e6:8.1f} us {2.0*B*(H//2)*(W//2)*4*128*64/t/1e12:6.1f} TF/s")
