#!/usr/bin/env python3
"""Small driver for rocprofv3: a few UNet eval forwards at a given precision (0 = fp32 MFMA, 1 = bf16x3)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicfpaugment_amd.training.unet import UNet  # noqa: E402
from musicfpaugment_amd.training.weights import formula_state_dict  # noqa: E402

prec = int(sys.argv[1]) if len(sys.argv) > 1 else 0
clips = int(sys.argv[2]) if len(sys.argv) > 2 else 64
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 2
net = UNet(1, 1)
net.load_state_dict(formula_state_dict(0))
net = net.cuda().eval()
net.precision = prec
x = torch.rand(clips, 1, 257, 251, device="cuda")
for _ in range(iters):
    net(x)
torch.cuda.synchronize()
