"""Worker for tests/test_gpu_dist.py: one train step with SYNCHRONISED BatchNorm on two ranks (cuda:0, gloo): rank r holds clips
[2r, 2r+2) of the 4-clip batch.  The result must equal a single-process step on all four clips."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicfpaugment_amd import ops, synth  # noqa: E402
from musicfpaugment_amd.ops_train import UNetTrainEngine  # noqa: E402
from musicfpaugment_amd.training.train import _global_max  # noqa: E402
from musicfpaugment_amd.training.unet import UNet  # noqa: E402
from musicfpaugment_amd.training.weights import formula_state_dict  # noqa: E402


def step(lo, hi, sync_bn):
    clean = synth.batch(4, seed=900, n=8000)
    aug = (0.7 * clean + 0.3 * synth.batch(4, seed=901, n=8000, tonal=False)).astype(np.float32)
    cm, cmax = ops.stft_mag(torch.from_numpy(clean[lo:hi]).cuda(), torch.float64)
    am, amax = ops.stft_mag(torch.from_numpy(aug[lo:hi]).cuda(), torch.float64)
    gc, ga = _global_max(cmax), _global_max(amax)
    target = ops.normalize_(cm, gc.expand(hi - lo).contiguous(), per_clip=True)
    net = UNet(1, 1, rate=0.0)
    net.load_state_dict(formula_state_dict(0))
    eng = UNetTrainEngine(net.cuda().train(), lr=1e-3, precision=0, sync_bn=sync_bn)
    loss = eng.train_step(am, ga.expand(hi - lo).contiguous(), target)
    torch.cuda.synchronize()
    return eng, float(loss)


if __name__ == "__main__":
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    eng, loss = step(rank * 4 // world, (rank + 1) * 4 // world, True)
    assert eng.sync_bn
    np.savez(os.path.join(sys.argv[1], f"rank{rank}.npz"), params=eng.flat_p.cpu().numpy(), loss=loss,
             rm=eng.running["inc.double_conv.1.running_mean"].cpu().numpy(),
             rv=eng.running["down4.maxpool_conv.1.double_conv.4.running_var"].cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()
