#!/usr/bin/env python3
"""Per-op timing of the bandwidth / latency-bound kernels at a BASELINE batch (default 256 clips): STFT, Audfprint
prepare + prune, Dejavu specgram + prepare + local-max, peak metrics, PSNR stats, landmarks / hashes.
Prints microseconds per call and the algorithmic GB/s (bytes in + out of each op's own operands)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicfpaugment_amd import ops, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
base = synth.batch(32, seed=59)
wav = torch.from_numpy(np.concatenate([base] * ((B + 31) // 32))[:B].copy()).cuda()

def nbytes(*ts):
    return sum(t.numel() * t.element_size() for t in ts if isinstance(t, torch.Tensor))

def timeit(name, fn, ins, reps=20):
    out = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    outs = out if isinstance(out, (tuple, list)) else (out,)
    by = nbytes(*ins) + nbytes(*outs)
    print(f"{name:34s} {us:9.1f} us  {by / 1e6:9.1f} MB  {by / us / 1e3:8.1f} GB/s  ({100 * by / us / 1e3 / 8000:.1f} % of 8 TB/s)", flush=True)
    return out

mag, cmax = timeit("stft_mag f64", lambda: ops.stft_mag(wav, torch.float64), (wav,))
mag32, _ = timeit("stft_mag f32", lambda: ops.stft_mag(wav, torch.float32), (wav,))
filt = timeit("audfprint_prepare (f64 |stft|)", lambda: ops.audfprint_prepare(mag, denom=cmax, mean_order=1), (mag, cmax))
filt = timeit("audfprint_prepare (denom = max)", lambda: ops.audfprint_prepare(mag, denom=cmax, mean_order=1, denom_is_clip_max=True), (mag, cmax))
mask, npk = timeit("audfprint_prune", lambda: ops.audfprint_prune(filt), (filt,))
psd, pmax = timeit("specgram_psd (dejavu)", lambda: ops.specgram_psd(wav, scale_in=32767.0), (wav,))
arr = timeit("dejavu_prepare", lambda: ops.dejavu_prepare(psd, pmax, 10.0, mean_order=1), (psd, pmax))
dmask, dn = timeit("localmax2d 21x21", lambda: ops.localmax2d(arr, 10, 50.0), (arr,))
mt = mask.transpose(1, 2).contiguous()
cnt = timeit("peak_metrics_counts", lambda: ops.peak_metrics_counts(mt, mt), (mt, mt))
spec = ops.normalize_(mag.clone(), cmax, per_clip=True)
st = timeit("psnr_stats", lambda: ops.psnr_stats(spec, spec), (spec, spec))
lm = timeit("audfprint_landmarks", lambda: ops.audfprint_landmarks(mask), (mask,))
dh = timeit("dejavu_hashes (SHA-1)", lambda: ops.dejavu_hashes(dmask), (dmask,))
