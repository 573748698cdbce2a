"""Self-checks of the training arithmetic that both tests/test_gpu_train.py and bench.py's config 4 entry run (device code only: no
oracle, no CPU fallback).  Reference loop: training/train.py:257-317 (L1 loss, Adam lr 1e-3, Dropout 0.05)."""
from __future__ import annotations

import numpy as np
import torch

from .weights import formula_state_dict


def convergence_data(n_batches=4, B=16, nsamp=24000, seed0=4000):
    """Fixed synthetic clean / noisy pairs (3 s clips, the reference's training length): `n_batches` training batches + one held-out."""
    from musicfpaugment_amd import ops, synth
    out = []
    for k in range(n_batches + 1):
        clean = synth.batch(B, seed=seed0 + B * k, n=nsamp)
        noisy = (0.7 * clean + 0.3 * synth.batch(B, seed=seed0 + 5000 + B * k, n=nsamp, tonal=False)).astype(np.float32)
        cm, cmax = ops.stft_mag(torch.from_numpy(clean).cuda(), torch.float64)
        am, amax = ops.stft_mag(torch.from_numpy(noisy).cuda(), torch.float64)
        ops.normalize_(cm, cmax.max().expand(B).contiguous(), per_clip=True)
        out.append((am, amax.max().expand(B).contiguous(), cm))
    return out[:-1], out[-1]


def run_convergence(precisions, steps=200, B=16, lr=1e-3, verbose=True):
    """Train UNet(1,1,rate=0.05) from the same weights, on the same batches, with the same (stateless, step-keyed) dropout masks, in each
    arithmetic of `precisions` = {name: (precision, wgrad_precision[, activations kept as bfloat16 in HBM: ops_train.Z16_ACTIVATIONS])}.  Returns {name: (mean training loss of the last 8 steps, held-out
    L1 of the trained weights evaluated by the fp32 inference kernels)}.  Shared with bench.py's config 4 entry."""
    from musicfpaugment_amd.ops_train import UNetTrainEngine
    from musicfpaugment_amd.training.unet import UNet
    train, held = convergence_data(B=B)
    res = {}
    from musicfpaugment_amd import ops_train
    for name, spec in precisions.items():
        prec, wprec = spec[0], spec[1]
        m = UNet(1, 1, rate=0.05)
        m.load_state_dict(formula_state_dict(0))
        m = m.cuda().train()
        keep_z16 = ops_train.Z16_ACTIVATIONS
        if len(spec) > 2:                                                # (precision, wgrad_precision, activations kept as bfloat16?)
            ops_train.Z16_ACTIVATIONS = bool(spec[2])
        try:
            eng = UNetTrainEngine(m, lr=lr, precision=prec, wgrad_precision=wprec)
            losses = []
            for k in range(steps):
                am, aden, cm = train[k % len(train)]
                losses.append(eng.train_step(am, aden, cm).clone())      # (the engine returns its persistent loss scalar)
            losses = [float(l) for l in losses]
        finally:
            ops_train.Z16_ACTIVATIONS = keep_z16                         # the module-level switch goes back whatever happened in between
        if len(spec) > 2 and bool(spec[2]) != bool(eng._z16):
            raise RuntimeError(f"{name}: asked for bfloat16 activations = {bool(spec[2])}, the engine ran with {bool(eng._z16)}")
        eng.sync_to_module()
        m.eval()
        m.precision = 0
        am, aden, cm = held
        with torch.no_grad():
            pred = m((am / aden[:, None, None]).float().unsqueeze(1))[:, 0]
        held_l1 = float((pred.double() - cm).abs().mean())
        res[name] = (float(np.mean(losses[-8:])), held_l1, losses[0])
        if verbose:
            print(f"[convergence, lr {lr:g}] {name:8s} loss {losses[0]:.5f} -> {res[name][0]:.5f} (last 8 of {steps}), held-out L1 {held_l1:.5f}")
        del eng, m
        torch.cuda.empty_cache()
    return res
