"""Worker for tests/test_gpu_dist.py: the sharded peak-metrics experiment (testing/audfprint_exps.compute_peaks_metrics) with
two ranks on cuda:0 over gloo.  Every rank holds all queries and processes its shard; rank 0 dumps the result dictionary."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicfpaugment_amd import synth  # noqa: E402
from musicfpaugment_amd.afp.audfprint.peak_extractor import Audfprint_peaks  # noqa: E402
from musicfpaugment_amd.testing.audfprint_exps import compute_peaks_metrics  # noqa: E402
from musicfpaugment_amd.training.unet import UNet  # noqa: E402
from musicfpaugment_amd.training.weights import formula_state_dict  # noqa: E402


def make_inputs():
    clean = synth.batch(7, seed=950, n=16000)                       # 7 queries: ragged shards (4 + 3)
    aug = (0.8 * clean + 0.2 * synth.batch(7, seed=951, n=16000, tonal=False)).astype(np.float32)
    return torch.from_numpy(clean), torch.from_numpy(aug)


def run():
    net = UNet(1, 1)
    net.load_state_dict(formula_state_dict(0))
    net = net.cuda().eval()
    an_no = Audfprint_peaks(None)
    an_den = Audfprint_peaks(None, denoising=True, denoising_model="unet", unet=net)
    clean, aug = make_inputs()
    return compute_peaks_metrics(clean, aug, an_no, an_den, batch=3)


if __name__ == "__main__":
    dist.init_process_group("gloo")
    torch.cuda.set_device(0)
    res = run()
    if dist.get_rank() == 0:
        with open(os.path.join(sys.argv[1], "metrics.json"), "w") as fh:
            json.dump(res, fh)
    dist.barrier()
    dist.destroy_process_group()
