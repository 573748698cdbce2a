#!/usr/bin/env python3
"""STFT A/B: product library vs a variant (--lib), 256 and 8192 clips; prints us per call and the largest difference of the variant's
magnitudes from the product's in ulps.  usage: ab_stft.py --lib PATH"""
import argparse, os, subprocess, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser(); ap.add_argument("--lib", default=None); ap.add_argument("--dump", default=None)
args = ap.parse_args()
if args.lib:
    from musicfpaugment_amd import _lib
    _lib.set_library_path(args.lib)
from musicfpaugment_amd import ops, synth
base = synth.batch(32, seed=59)
for B in (256, 8192):
    wav = torch.from_numpy(np.concatenate([base] * (B // 32)).copy()).cuda()
    for _ in range(3): ops.stft_mag(wav, torch.float64)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20 if B == 256 else 5
    e0.record()
    for _ in range(reps): mag, _ = ops.stft_mag(wav, torch.float64)
    e1.record(); torch.cuda.synchronize()
    print(f"{args.lib or 'product':40s} B={B:5d} stft_mag f64 {e0.elapsed_time(e1) * 1e3 / reps:8.1f} us", flush=True)
    if B == 256 and args.dump:
        np.save(args.dump, mag[:32].cpu().numpy())
