#!/usr/bin/env python3
"""Demucs forward of 256 clips: one batch vs two (four) concurrent sub-batches on separate streams (the LSTM's persistent launches use
96 (48) CUs each and are fabric / latency-bound; the other sub-batch's GEMMs can run beside them).  usage: exp_demucs_split.py [--clips B]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser(); ap.add_argument("--clips", type=int, default=256); ap.add_argument("--reps", type=int, default=10)
args = ap.parse_args()
from musicfpaugment_amd import ops_demucs as D, synth
from musicfpaugment_amd.training.model import Demucs
from musicfpaugment_amd.training.demucs_weights import formula_state_dict
net = Demucs(); net.load_state_dict(formula_state_dict(0)); net = net.cuda().eval()
wav = torch.from_numpy(synth.batch(args.clips, seed=1)).cuda()
def whole(): return net(wav)
streams = [torch.cuda.Stream() for _ in range(4)]
def split(n):
    cur = torch.cuda.current_stream()
    outs = []
    parts = wav.chunk(n)
    for s, p in zip(streams, parts):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs.append(net(p.contiguous()))
    for s in streams[:n]: cur.wait_stream(s)
    return torch.cat(outs)
def t(fn):
    with torch.no_grad():
        fn(); fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps): out = fn()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.reps, out
a, o1 = t(whole)
b, o2 = t(lambda: split(2))
c, o4 = t(lambda: split(4))
a2, _ = t(whole)
ok = D.lstm_results_ok
D.lstm_results_ok = lambda dev: True        # timing only: no host wait between the sub-batches' enqueues
b2, _ = t(lambda: split(2)); c2, _ = t(lambda: split(4)); a3, _ = t(whole)
D.lstm_results_ok = ok
print(f'without the host-side LSTM check: one batch {a3:.2f} ms; halves {b2:.2f} ms; quarters {c2:.2f} ms', flush=True)
print(f"clips {args.clips}: one batch {a:.2f} / {a2:.2f} ms; two concurrent halves {b:.2f} ms; four quarters {c:.2f} ms; "
      f"max |diff| vs one batch {float((o1 - o2).abs().max()):.2e} / {float((o1 - o4).abs().max()):.2e}; lstm fallbacks {D.persistent_lstm_fallbacks}", flush=True)
