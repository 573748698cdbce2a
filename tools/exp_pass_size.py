import sys, time, torch
sys.path.insert(0, '.')
from musicfpaugment_amd import synth
from musicfpaugment_amd.pipeline import HotPath
from musicfpaugment_amd.training.unet import UNet
from musicfpaugment_amd.training.weights import formula_state_dict
net = UNet(1, 1); net.load_state_dict(formula_state_dict(0)); net = net.cuda().eval(); net.precision = 1
hp = HotPath(net)
base = synth.batch(16, seed=59)
import numpy as np
wav = torch.from_numpy(np.concatenate([base] * 16)).cuda()
for m in (32, 64, 128, 256):
    net.max_clips_per_pass = m
    hp(wav); torch.cuda.synchronize()
    t = time.time()
    for _ in range(3): hp(wav)
    torch.cuda.synchronize()
    print(m, f"{(time.time() - t) / 3 * 1e3:.2f} ms per 256 clips", flush=True)
