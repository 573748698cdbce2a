import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicfpaugment_amd.training.model import Demucs
from musicfpaugment_amd.training.demucs_weights import formula_state_dict
net = Demucs(); net.load_state_dict(formula_state_dict(0)); net = net.cuda().eval()
for B in (64, 256):
    x = torch.randn(B, 64000, device="cuda") * 0.1
    net(x); torch.cuda.synchronize()
    t = time.time(); n = 2
    for _ in range(n): net(x)
    torch.cuda.synchronize(); dt = (time.time() - t) / n
    print(f"B={B}: {dt*1e3:.1f} ms  -> {B/dt:.0f} clips/s, {20.13e9*B/dt/1e12:.1f} TFLOP/s")
