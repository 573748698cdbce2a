#!/usr/bin/env python3
"""Time the two LSTM layers of the Demucs forward alone (B clips, Tn steps, H units; random weights), HIP events on the launch stream.
usage: exp_lstm.py [--lib PATH] [--clips B] [--steps Tn] [--persistent 0|1]"""
import argparse, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--clips", type=int, default=256)
ap.add_argument("--steps", type=int, default=248)
ap.add_argument("--hidden", type=int, default=768)
ap.add_argument("--persistent", type=int, default=1)
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()
if args.lib:
    from musicfpaugment_amd import _lib
    _lib.set_library_path(args.lib)
from musicfpaugment_amd import ops_demucs as D
D.PERSISTENT_LSTM = bool(args.persistent)
B, Tn, H = args.clips, args.steps, args.hidden
g = torch.Generator().manual_seed(0)
x = (torch.randn(B, Tn, H, generator=g) * 0.5).cuda()
skip = torch.randn(B, Tn, H, generator=g).cuda()
wih = [(torch.randn(4 * H, H, generator=g) / np.sqrt(H)).cuda() for _ in range(2)]
whh = [torch.randn(4 * H, H, generator=g) / np.sqrt(H) for _ in range(2)]
bias = [(torch.randn(4 * H, generator=g) * 0.1).cuda() for _ in range(2)]
grouped = [w.reshape(4, H // 16, 16, H).permute(1, 0, 2, 3).reshape(4 * H, H).contiguous().cuda() for w in whh]
for w in wih:
    D.attach_split(w)
D.lstm_two_layers(x, skip, wih, bias, grouped, 1, False); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(args.reps):
    D.lstm_two_layers(x, skip, wih, bias, grouped, 1, False)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / args.reps
# the projections alone
e0.record()
for _ in range(args.reps):
    for k in range(2):
        xp = torch.empty(B, Tn, 4 * H, device="cuda")
        D.gemm(D._p(x), H, 0, 1, B * Tn, wih[k], bias[k], 4 * H, D._p(xp), 4 * H, 0, precision=1)
e1.record(); torch.cuda.synchronize()
pj = e0.elapsed_time(e1) / args.reps
print(f"B {B} Tn {Tn} H {H} persistent {args.persistent}: both layers {ms:.3f} ms, of which projections {pj:.3f} ms -> "
      f"{(ms - pj) * 1e3 / (2 * Tn):.2f} us per recurrent step; error word {D.lstm_seq_error() if args.persistent else '-'}", flush=True)
