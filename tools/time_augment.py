"""Where does AugmentFP.batch_augment spend its time?  (host draws vs device kernels)  python tools/time_augment.py [B]"""
import random
import sys
import time

import torch

from musicfpaugment_amd import synth
from musicfpaugment_amd.augmentation import AugmentFP, synthetic_banks

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
irs, noises = synthetic_banks(0)
af = AugmentFP(None, 8000, ir_bank=irs, noise_bank=noises)
wav = torch.from_numpy(synth.batch(B, seed=1)).cuda()[:, None, :]
random.seed(0); torch.manual_seed(0)
for _ in range(2):
    af.batch_augment(wav)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    af.batch_augment(wav)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"B={B}: host-side enqueue {1e3*(t1-t0)/5:.1f} ms/call, total {1e3*(t2-t0)/5:.1f} ms/call")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    af.batch_augment(wav)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=12, max_name_column_width=60))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); af.batch_augment(wav); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
