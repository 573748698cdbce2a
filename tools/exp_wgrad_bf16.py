#!/usr/bin/env python3
"""A/B of the plain-bf16 weight gradient with fp32 operands (precision 2) and with bf16 copies (precision 3 incl. the cast passes), per layer shape."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicfpaugment_amd import ops_train as T
B = 64
shapes = [("128->128 @128x125", 128, 125, 128, 128), ("256->256 @64x62", 64, 62, 256, 256), ("512->256 @64x62", 64, 62, 512, 256),
          ("512->512 @32x31", 32, 31, 512, 512), ("1024->512 @32x31", 32, 31, 1024, 512), ("1024->1024 @16x15", 16, 15, 1024, 1024)]
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for name, H, W, Ci, Co in shapes:
    x = torch.randn(B, H, W, Ci, device="cuda"); dz = torch.randn(B, H, W, Co, device="cuda") * 0.01
    dw = torch.zeros(9, Co, Ci, device="cuda")
    res = {}
    for thr, label in ((1 << 30, "fp32 operands"), (64, "bf16 copies")):
        T.BF16_WGRAD_MIN_CH = thr
        dw.zero_(); T.wgrad_mfma(dz, x, dw, Co, precision=2); res[label + "_dw"] = dw.clone()
        res[label] = timeit(lambda: T.wgrad_mfma(dz, x, dw, Co, precision=2))
    rel = float((res["bf16 copies_dw"] - res["fp32 operands_dw"]).abs().sum() / res["fp32 operands_dw"].abs().sum())
    fl = 2.0 * B * H * W * Ci * Co * 9
    print(f"{name:20s} fp32 operands {res['fp32 operands']:8.1f} us ({fl/res['fp32 operands']/1e6:6.1f} TF/s)   bf16 copies {res['bf16 copies']:8.1f} us ({fl/res['bf16 copies']/1e6:6.1f} TF/s)   rel diff {rel:.2e}", flush=True)
