#!/usr/bin/env python3
"""hipGraph capture of the inference paths through torch.cuda.CUDAGraph: the C ABI never synchronises and allocates nothing,
so STFT -> UNet -> peak-pick and the Demucs forward (2 x 248 LSTM step launches) replay as one graph.  Prints eager vs replay time."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicfpaugment_amd import synth
from musicfpaugment_amd.pipeline import HotPath
from musicfpaugment_amd.training.demucs_weights import formula_state_dict as demucs_formula
from musicfpaugment_amd.training.model import Demucs
from musicfpaugment_amd.training.unet import UNet
from musicfpaugment_amd.training.weights import formula_state_dict


def capture(fn, static_in):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            fn(static_in)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn(static_in)
    return g, out


def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
net = UNet(1, 1); net.load_state_dict(formula_state_dict(0)); net = net.cuda().eval(); net.precision = 1
hp = HotPath(net)
wav = torch.from_numpy(synth.batch(B, seed=59)).cuda()
g, out = capture(hp, wav)
print(f"UNet hot path, {B} clips: eager {timeit(lambda: hp(wav)):.2f} ms, graph replay {timeit(g.replay):.2f} ms", flush=True)
dm = Demucs(); dm.load_state_dict(demucs_formula(0)); dm = dm.cuda().eval()
g2, out2 = capture(dm, wav)
want = dm(wav)
g2.replay(); torch.cuda.synchronize()
print("Demucs graph replay equals eager:", bool(torch.equal(out2, want)))
print(f"Demucs forward, {B} clips: eager {timeit(lambda: dm(wav)):.2f} ms, graph replay {timeit(g2.replay):.2f} ms", flush=True)
