#!/usr/bin/env python3
"""Timing of the weight-gradient kernel per layer shape, fp32 MFMA vs bf16x3 vs bf16: python tools/exp_wgrad.py [--lib PATH]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[1] == "--lib":
    from musicfpaugment_amd import _lib
    _lib.set_library_path(sys.argv[2])
from musicfpaugment_amd import ops_train as T
layers = [("inc.3 64->64 @257x251", 64, 257, 251, 64, 64), ("d1.3 128->128 @128x125", 64, 128, 125, 128, 128),
          ("d2.3 256->256 @64x62", 64, 64, 62, 256, 256), ("d3.3 512->512 @32x31", 64, 32, 31, 512, 512),
          ("d4.3 1024->1024 @16x15", 64, 16, 15, 1024, 1024), ("up4.0 128->64 @257x251", 64, 257, 251, 128, 64)]
for name, B, H, W, Ci, Co in layers:
    x = torch.randn(B, H, W, Ci, device="cuda")
    dz = torch.randn(B, H, W, Co, device="cuda")
    dw = torch.zeros(9, Co, Ci, device="cuda")
    out = []
    for prec in (0, 1, 2):
        T.wgrad_mfma(dz, x, dw, Co, precision=prec); torch.cuda.synchronize()
        t = time.time()
        for _ in range(3): T.wgrad_mfma(dz, x, dw, Co, precision=prec)
        torch.cuda.synchronize(); dt = (time.time() - t) / 3
        fl = 2.0 * B * H * W * Ci * Co * 9
        out.append(f"{dt*1e6:9.1f} us {fl/dt/1e12:6.1f} TF/s-eq")
    print(f"{name:26s} fp32 {out[0]}   bf16x3 {out[1]}   bf16 {out[2]}", flush=True)
