"""The hot path as one batched, sync-free device pipeline (what bench.py times).

    waveform (B, 64000) f32  ->  fused window+rFFT+|.| (f64)  ->  [per-clip /max fused into the first conv]
      ->  UNet eval forward (fp32 MFMA)  ->  log/mean/high-pass  ->  forward+backward pruning
      ->  peak mask (B, 256, 251) u8 + peak counts

Mirrors Audfprint_peaks(denoising=True, "unet").find_peaks applied to every clip of a batch
(afp/audfprint/peak_extractor.py:236-311), which is the reference's own end-to-end chain
(testing/audfprint_exps.py:105-117).  Clips are independent, so a multi-GPU job shards the batch
across ranks with no data-path collective.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import ops
from .afp.audfprint.peak_extractor import Audfprint_peaks

# Algorithmic work per 8 s clip (SURVEY.md §8d / BASELINE.md §2)
UNET_FWD_GFLOP_PER_CLIP = 93.398
UNET_MFMA_GFLOP_PER_CLIP = 93.398 - 0.074 - 0.008   # minus inc.0 (1 input channel) and outc (1x1 to 1 class): VALU kernels
STFT_BYTES_PER_CLIP = 514028
PRUNER_BYTES_PER_CLIP = 322284


class HotPath:
    def __init__(self, unet, device="cuda"):
        self.extractor = Audfprint_peaks(None, denoising=unet is not None, denoising_model="unet" if unet else None,
                                         unet=unet, device=device)

    @torch.no_grad()
    def __call__(self, wav: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        mask, npeaks, _ = self.extractor.find_peaks_batch(wav)
        return mask, npeaks


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [start, end) slice of n_items owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def reduce_metric_counts(local_counts: torch.Tensor, group=None) -> torch.Tensor:
    """Sum the (4,) [hits_p, n_p, hits_r, n_r] integer counts over ranks: the only exchange the sharded
    peak-metrics evaluation needs (testing/audfprint_exps.py:127-134 averages per query; integer sums keep it exact)."""
    import torch.distributed as dist
    out = local_counts.clone()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
    return out
