#!/usr/bin/env python3
"""Timing experiments on single conv layers (MFPA_CONV_DBG flags): usage exp_conv.py <precision>"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicfpaugment_amd import ops_unet as K
prec = int(sys.argv[1]) if len(sys.argv) > 1 else 1
layers = [("inc.3 64->64 @257x251", 64, 257, 251, 64, 64), ("d2.3 256->256 @64x62", 64, 64, 62, 256, 256),
          ("up1.0 1024->512 @32x31", 64, 32, 31, 1024, 512), ("up4.0 128->64 @257x251", 64, 257, 251, 128, 64),
          ("d1.3 128->128 @128x125", 64, 128, 125, 128, 128), ("d4.3 1024->1024 @16x15", 64, 16, 15, 1024, 1024)]
for name, B, H, W, Ci, Co in layers:
    x = torch.randn(B, H, W, Ci, device="cuda")
    w = torch.randn(9, Co, Ci, device="cuda") * 0.05
    wp = K.split_bf16x3(w) if prec else w
    sc = torch.ones(Co, device="cuda"); sh = torch.zeros(Co, device="cuda")
    K.conv3x3_bn_relu(x, wp, sc, sh, precision=prec); torch.cuda.synchronize()
    t = time.time()
    for _ in range(5): K.conv3x3_bn_relu(x, wp, sc, sh, precision=prec)
    torch.cuda.synchronize(); dt = (time.time() - t) / 5
    fl = 2.0 * B * H * W * Ci * Co * 9
    print(f"{name:28s} {dt*1e6:9.1f} us  {fl/dt/1e12:7.1f} TF/s-eq", flush=True)
