"""Demucs forward with the two N = 192 layers on the pipelined 256 x 128 tile (weight rows padded to 256) vs on the 128 x 64 tile."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicfpaugment_amd import ops_demucs as D, synth
from musicfpaugment_amd.training.demucs_weights import formula_state_dict
from musicfpaugment_amd.training.model import Demucs
wav = torch.from_numpy(np.concatenate([synth.batch(32, seed=1)] * 8)).cuda()
outs = {}
for flag in (True, False, True, False):
    D.PAD_N_TO_WIDE_TILE = flag
    net = Demucs(); net.load_state_dict(formula_state_dict(0)); net = net.cuda().eval()
    for _ in range(2): y = net(wav)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(6): y = net(wav)
    torch.cuda.synchronize(); print("pad to 128" if flag else "pad to 64 ", "%.3f ms" % ((time.perf_counter() - t) / 6 * 1e3), flush=True)
    outs[flag] = y
print("max abs difference", (outs[True] - outs[False]).abs().max().item(), "of", outs[False].abs().max().item())
