#!/usr/bin/env python3
"""Dejavu chain stage timing at 256 clips: specgram_psd, dejavu_prepare + localmax2d (two calls), dejavu_pick (fused pair)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    from musicfpaugment_amd import _lib
    _lib.set_library_path(sys.argv[1])
from musicfpaugment_amd import ops, synth
B = 256
base = synth.batch(32, seed=59)
wav = torch.from_numpy(np.concatenate([base] * (B // 32)).copy()).cuda()
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
psd, cmax = ops.specgram_psd(wav, scale_in=32767.0)
arr = ops.dejavu_prepare(psd, cmax, 10.0, mean_order=1)
print(f"specgram_psd {t(lambda: ops.specgram_psd(wav, scale_in=32767.0)):7.1f} us  dejavu_prepare {t(lambda: ops.dejavu_prepare(psd, cmax, 10.0, mean_order=1)):7.1f} us  "
      f"localmax2d {t(lambda: ops.localmax2d(arr, 10, 50.0)):7.1f} us  dejavu_pick {t(lambda: ops.dejavu_pick(psd, cmax, 10.0, 1, 10, 50.0)):7.1f} us  "
      f"normalize_ {t(lambda: ops.normalize_(psd.clone(), cmax, per_clip=True)):7.1f} us (incl. clone)", flush=True)
