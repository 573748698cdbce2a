"""Worker for tests/test_gpu_dist.py: one data-parallel Demucs train step per rank (both ranks on cuda:0, gloo backend so that two
processes can share one GPU).  Rank r trains on clips [2r, 2r+2) of a 4-clip batch and dumps its gradients and parameters."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicfpaugment_amd import synth  # noqa: E402
from musicfpaugment_amd.ops_demucs_train import DemucsTrainEngine  # noqa: E402
from musicfpaugment_amd.training.demucs_weights import formula_state_dict  # noqa: E402


def main():
    out_dir = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    clean = synth.batch(4, seed=910, n=4000)
    aug = (clean + 0.05 * synth.batch(4, seed=911, n=4000)).astype(np.float32)
    lo, hi = rank * 4 // world, (rank + 1) * 4 // world
    eng = DemucsTrainEngine(formula_state_dict(0), "cuda", lr=1e-3, precision=0)
    loss = eng.train_step(torch.from_numpy(clean[lo:hi]).cuda(), torch.from_numpy(aug[lo:hi]).cuda())
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), params=eng.flat_p.cpu().numpy(), grads=eng.flat_g.cpu().numpy(),
             loss=float(loss))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
