#!/usr/bin/env python3
"""The AugmentFP chain alone (64 clips of 8 s, the bench's banks and seeds), a few calls: for a rocprofv3 --kernel-trace + tools/launch_trace.py."""
import os, random, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicfpaugment_amd import synth
from musicfpaugment_amd.augmentation import AugmentFP, synthetic_banks
dev = torch.device("cuda")
B = 64
irs, noises = synthetic_banks(0)
aug = AugmentFP(None, 8000, ir_bank=irs, noise_bank=noises, device=dev)
random.seed(1); torch.manual_seed(1)
wav = torch.from_numpy(synth.batch(B, seed=9000)).to(dev).unsqueeze(1)
for _ in range(3):
    y = aug.batch_augment(wav)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    y = aug.batch_augment(wav)
torch.cuda.synchronize()
print(f"AugmentFP.batch_augment, {B} clips: {(time.perf_counter() - t0) * 100:.3f} ms per call")
