"""Worker for tests/test_gpu_dist.py: Trainer.training_loop on two ranks (both on cuda:0, gloo).  Each rank has its OWN training and
validation shards, so the per-rank validation losses differ: the schedule / early-stopping / checkpoint decisions must still be
identical on both ranks (the losses are averaged over the engine's process group), the replicas must stay in sync, and only rank 0
writes the checkpoints."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicfpaugment_amd import synth  # noqa: E402
from musicfpaugment_amd.training.train import Trainer  # noqa: E402
from musicfpaugment_amd.training.unet import UNet  # noqa: E402
from musicfpaugment_amd.training.weights import formula_state_dict  # noqa: E402


def loader(seed):
    k = 0
    while True:
        clean = synth.batch(2, seed=seed + 2 * (k % 3), n=8000)
        noise = synth.batch(2, seed=seed + 100 + 2 * (k % 3), n=8000, tonal=False)
        yield torch.from_numpy(clean)[:, :, None], torch.from_numpy((0.7 * clean + 0.3 * noise).astype(np.float32))[:, :, None]
        k += 1


def main():
    out_dir = sys.argv[1]
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    torch.cuda.set_device(0)
    net = UNet(1, 1, rate=0.0)
    net.load_state_dict(formula_state_dict(2))
    # scheduler patience 0: the learning rate drops as soon as the (rank-averaged) validation loss fails to improve
    tr = Trainer(net, loader(500 + 40 * rank), loader(700 + 40 * rank), learning_rate=1e-3, train_steps=3, val_steps=2,
                 ckpt_path=os.path.join(out_dir, "ckpt"), scheduler_patience=0, early_stop_patience=50)
    tr.training_loop(nb_epochs=4)                       # epochs 1, 2, 3
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), params=tr.engine.flat_p.cpu().numpy(), lr=tr.engine.lr, epoch=tr.epoch,
             val=np.array(tr.losses["val"]), best=tr.best_val_loss, sched=np.array([tr.scheduler.best, tr.scheduler.num_bad]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
