import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)

    return load


def _trained_state_dict(steps=50, seed=0):
    """The third stressed family: the formula weights after `steps` optimiser steps of UNetTrainEngine (lr 1e-3, Dropout 0.05,
    bf16x3 training arithmetic like bench.py's config 4) on synthetic clean / noisy pairs -- BatchNorm statistics, scales and the
    output layer are what training made them, not what a generator drew."""
    import numpy as np
    import torch
    from musicfpaugment_amd import ops, synth
    from musicfpaugment_amd.training.weights import formula_state_dict
    from musicfpaugment_amd.ops_train import UNetTrainEngine
    from musicfpaugment_amd.training.unet import UNet
    m = UNet(1, 1, rate=0.05)
    m.load_state_dict(formula_state_dict(seed))
    m = m.cuda().train()
    eng = UNetTrainEngine(m, lr=1e-3, precision=1, wgrad_precision=2)
    losses = []
    for k in range(steps):
        clean = synth.batch(8, seed=4000 + 8 * (k % 6), n=24000)
        noisy = (0.7 * clean + 0.3 * synth.batch(8, seed=9000 + 8 * (k % 6), n=24000, tonal=False)).astype(np.float32)
        cm, cmax = ops.stft_mag(torch.from_numpy(clean).cuda(), torch.float64)
        am, amax = ops.stft_mag(torch.from_numpy(noisy).cuda(), torch.float64)
        ops.normalize_(cm, cmax.max().expand(8).contiguous(), per_clip=True)
        losses.append(float(eng.train_step(am, amax.max().expand(8).contiguous(), cm)))
    eng.sync_to_module()
    assert losses[-1] < 0.5 * losses[0], (losses[0], losses[-1])             # it really trained
    return {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}


@pytest.fixture(scope="session")
def trained_sd():
    """() -> state_dict of the "trained" weight family (trained once per session on the GPU)."""
    cache = {}

    def get():
        if "sd" not in cache:
            cache["sd"] = _trained_state_dict()
        return {k: v.clone() for k, v in cache["sd"].items()}

    return get
