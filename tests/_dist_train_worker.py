"""Worker for tests/test_gpu_dist.py: one data-parallel UNet train step per rank (both ranks on cuda:0, gloo backend so that two
processes can share one GPU).  Rank r trains on clips [2r, 2r+2) of a 4-clip batch and dumps its updated flat parameters."""
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


def main():
    out_dir = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    clean = synth.batch(4, seed=900, n=8000)
    aug = (0.7 * clean + 0.3 * synth.batch(4, seed=901, n=8000, tonal=False)).astype(np.float32)
    lo, hi = rank * 4 // world, (rank + 1) * 4 // world
    cm, cmax = ops.stft_mag(torch.from_numpy(clean[lo:hi]).cuda(), torch.float64)
    am, amax = ops.stft_mag(torch.from_numpy(aug[lo:hi]).cuda(), torch.float64)
    gc, ga = _global_max(cmax), _global_max(amax)                  # scalar MAX all-reduce: spectrogram()'s whole-batch max
    target = ops.normalize_(cm, gc.expand(hi - lo).contiguous(), per_clip=True)
    net = UNet(1, 1, rate=0.0)
    net.load_state_dict(formula_state_dict(0))
    eng = UNetTrainEngine(net.cuda().train(), lr=1e-3, precision=0)
    loss = eng.train_step(am, ga.expand(hi - lo).contiguous(), target)
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), params=eng.flat_p.cpu().numpy(), loss=float(loss),
             gmax=np.array([float(gc), float(ga)]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
