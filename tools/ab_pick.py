#!/usr/bin/env python3
"""Audfprint pick stage timing (mfpa_audfprint_pick: prep_sum + fused pruner), 256 clips.  usage: ab_pick.py [--lib PATH]"""
import argparse, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser(); ap.add_argument("--lib", default=None); ap.add_argument("--clips", type=int, default=256)
args = ap.parse_args()
if args.lib:
    from musicfpaugment_amd import _lib
    _lib.set_library_path(args.lib)
from musicfpaugment_amd import ops, synth
B = args.clips
base = synth.batch(32, seed=59)
wav = torch.from_numpy(np.concatenate([base] * (B // 32)).copy()).cuda()
mag, cmax = ops.stft_mag(wav, torch.float64)
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
print(f"{args.lib or 'product':40s} B={B} stft {t(lambda: ops.stft_mag(wav, torch.float64)):7.1f} us  pick {t(lambda: ops.audfprint_pick(mag, cmax)):7.1f} us", flush=True)
# floor of the scan: a constant spectrogram has (almost) no candidate frames -- what the frame loops cost with every frame quiet
flat = torch.ones_like(mag); fmax = torch.ones_like(cmax)
print(f"{'':40s} constant spectrogram: pick {t(lambda: ops.audfprint_pick(flat, fmax)):7.1f} us   (npeaks {int(ops.audfprint_pick(flat, fmax)[1].sum())})", flush=True)
print(f"{'':40s} real: npeaks per clip {float(ops.audfprint_pick(mag, cmax)[1].float().mean()):.1f}", flush=True)
