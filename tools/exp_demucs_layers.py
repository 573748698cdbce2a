#!/usr/bin/env python3
"""Per-launch table of the Demucs forward's GEMMs at 256 clips: shape, time (HIP events around each call, synchronised), the
bytes its operands occupy (A window + output [+ addend]) and its algorithmic FLOP rate.
usage: exp_demucs_layers.py [--lib PATH] [--clips B]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--clips", type=int, default=256)
args = ap.parse_args()
if args.lib:
    from musicfpaugment_amd import _lib
    _lib.set_library_path(args.lib)
from musicfpaugment_amd import ops_demucs as D, synth
from musicfpaugment_amd.training.model import Demucs
from musicfpaugment_amd.training.demucs_weights import formula_state_dict
net = Demucs(); net.load_state_dict(formula_state_dict(0)); net = net.cuda().eval()
wav = torch.from_numpy(synth.batch(args.clips, seed=1)).cuda()
net(wav); torch.cuda.synchronize()
rows = []
orig = D.gemm
def timed(A, lda, strideA, batch, M, W, bias, N, C, ldc, strideC, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    orig(A, lda, strideA, batch, M, W, bias, N, C, ldc, strideC, **kw)
    e1.record(); torch.cuda.synchronize()
    K = W.shape[1]
    rows_ = batch * M
    a_bytes = rows_ * min(lda, K) * 4                  # a strided window re-reads rows: count each input row once
    c_bytes = rows_ * (N // 2 if kw.get("mode", 0) == 1 else N) * 4
    add_bytes = rows_ * N * 4 if kw.get("addend", 0) else 0
    c2 = rows_ * kw.get("ldc2", 0) * 4 if kw.get("C2", 0) else 0
    rows.append((rows_, N, K, W.shape[0], kw.get("mode", 0), bool(kw.get("c1")), e0.elapsed_time(e1), a_bytes + c_bytes + add_bytes + c2,
                 2.0 * rows_ * N * K))
D.gemm = timed
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record(); net(wav); t1.record(); torch.cuda.synchronize()
D.gemm = orig
tot = 0.0
print(f"{'rows':>10} {'N':>5} {'K':>5} {'npad':>5} mode c1 {'ms':>7} {'GB':>6} {'TB/s':>6} {'TF/s alg':>9}")
for r in rows:
    tot += r[6]
    print(f"{r[0]:10d} {r[1]:5d} {r[2]:5d} {r[3]:5d} {r[4]:4d} {int(r[5]):2d} {r[6]:7.3f} {r[7]/1e9:6.2f} {r[7]/r[6]/1e9:6.2f} {r[8]/r[6]/1e9:9.1f}")
print(f"GEMM launches {len(rows)}: {tot:.2f} ms of {t0.elapsed_time(t1):.2f} ms (synchronised run)")
