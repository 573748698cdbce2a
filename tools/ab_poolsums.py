#!/usr/bin/env python3
"""UNet train step with / without the rank-1 OutConv backward (ops_train.RANK1_OUTCONV_BWD; FUSE_POOL_BWD_SUMS likewise, edit `run`):
time per step and the largest relative difference of the parameter gradients.  64 clips x 8 s, plain bf16."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicfpaugment_amd import ops_train as T, synth
from musicfpaugment_amd.training.weights import formula_state_dict
from musicfpaugment_amd.training.unet import UNet
import numpy as np
B = 64
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.random((B, 257, 251))).cuda()
y = torch.from_numpy(rng.random((B, 257, 251))).cuda()
den = torch.ones(B, dtype=torch.float64, device="cuda")
def run(flag, precision):
    T.FUSE_POOL_BWD_SUMS = True
    T.RANK1_OUTCONV_BWD = flag
    net = UNet(1, 1); net.load_state_dict(formula_state_dict(0)); net = net.cuda().train()
    eng = T.UNetTrainEngine(net, lr=1e-4, precision=precision, wgrad_precision=2 if precision == 2 else None)
    def step():
        pred = eng.forward(spec64=x, denom=den)
        loss, dpred = eng.l1_loss(pred, y)
        eng.backward(dpred)
    for _ in range(3): step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): step()
    e1.record(); torch.cuda.synchronize()
    g = {k: v.clone() for k, v in eng.named_grads().items()}
    return e0.elapsed_time(e1) / 10, g
for prec in (2, 1):
    t0, g0 = run(False, prec); t1, g1 = run(True, prec); t0b, _ = run(False, prec); t1b, _ = run(True, prec)
    worst = max(float((g0[k] - g1[k]).abs().sum() / (g0[k].abs().sum() + 1e-30)) for k in g0)
    print(f"precision {prec}: fwd+bwd with dy written {t0:.2f} / {t0b:.2f} ms, rank-1 (dy never written) {t1:.2f} / {t1b:.2f} ms; "
          f"largest relative L1 difference of a parameter gradient {worst:.2e}", flush=True)
