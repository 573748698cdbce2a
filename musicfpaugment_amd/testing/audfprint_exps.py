"""Peak-metrics experiment on MI355X -- mirror of compute_peaks_metrics in the reference's
testing/audfprint_exps.py:86-157 (the harness row SURVEY.md §8f-4), batched and sharded.

The reference loops over 10 000 query FILES, runs find_peaks three times per query (clean, augmented,
augmented + denoiser), and averages per-query Precision / Recall / F1 (masks transposed to (1, 251, 256),
:119-121) and spectrogram PSNR.  File I/O is out of scope, so this version takes the clean and
augmented waveforms as tensors; everything else -- three peak extractions, per-query metrics, means -- runs as
batched device kernels.  With torch.distributed initialised the queries are split over ranks
(pipeline.shard_range) and the per-query results are gathered: the means are identical to a single-GPU run.
"""
from __future__ import annotations

import math
from typing import Dict

import torch

from .. import ops
from ..afp.audfprint.peak_extractor import Audfprint_peaks
from ..pipeline import shard_range


def _prf(counts: torch.Tensor):
    """(B, 4) int64 [hits_p, n_p, hits_r, n_r] -> per-query precision, recall, F1 as the reference computes them
    (testing/metrics.py: 0.0 for an empty mask; F1 = 0 when P + R is ~0)."""
    c = counts.to(torch.float64)
    p = torch.where(c[:, 1] > 0, c[:, 0] / c[:, 1].clamp_min(1), torch.zeros_like(c[:, 0]))
    r = torch.where(c[:, 3] > 0, c[:, 2] / c[:, 3].clamp_min(1), torch.zeros_like(c[:, 0]))
    s = p + r
    f1 = torch.where(s.abs() <= 1e-9, torch.zeros_like(s), 2.0 * p * r / s.clamp_min(1e-300))   # math.isclose(p + r, 0.0)
    return p, r, f1


def _psnr(pred: torch.Tensor, target64: torch.Tensor) -> torch.Tensor:
    st = ops.psnr_stats(pred, target64)
    n = pred[0].numel()
    rng = st[:, 2] - st[:, 1]
    return 10.0 * torch.log10(rng * rng / (st[:, 0] / n))


@torch.no_grad()
def compute_peaks_metrics(clean_wav: torch.Tensor, augmented_wav: torch.Tensor, analyzer_no_den: Audfprint_peaks,
                          analyzer_den: Audfprint_peaks, batch: int = 256, per_query: bool = False):
    """clean_wav, augmented_wav: (N, T) float32 (any device).  Returns the reference's result dictionary; with
    `per_query=True` also the (N, 8) float64 tensor of per-query values behind the means (columns in the dictionary's key order)."""
    import torch.distributed as dist
    N = clean_wav.shape[0]
    ddp = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    rank, world = (dist.get_rank(), dist.get_world_size()) if ddp else (0, 1)
    lo, hi = shard_range(N, rank, world)
    dev = analyzer_no_den.device
    rows = []
    for s in range(lo, hi, batch):
        e = min(hi, s + batch)
        clean = clean_wav[s:e].to(dev, torch.float32).contiguous()
        aug = augmented_wav[s:e].to(dev, torch.float32).contiguous()
        m_clean, _, sg_clean = analyzer_no_den.wav2peaks_batch(clean)
        m_aug, _, sg_aug = analyzer_no_den.wav2peaks_batch(aug)
        m_den, _, sg_den = analyzer_den.wav2peaks_batch(aug)          # UNet inside find_peaks, Demucs on the waveform
        tr = lambda m: m.transpose(1, 2).contiguous()               # (B, 251, 256): audfprint_exps.py:119-121
        c_aug = ops.peak_metrics_counts(tr(m_aug), tr(m_clean))
        c_den = ops.peak_metrics_counts(tr(m_den), tr(m_clean))
        p, r, f1 = _prf(c_aug)
        pd, rd, f1d = _prf(c_den)
        rows.append(torch.stack([p, r, f1, _psnr(sg_aug, sg_clean), pd, rd, f1d, _psnr(sg_den, sg_clean)], dim=1))
    local = torch.cat(rows) if rows else torch.zeros((0, 8), dtype=torch.float64, device=dev)
    if ddp:
        sizes = [shard_range(N, r, world)[1] - shard_range(N, r, world)[0] for r in range(world)]
        pad = torch.zeros((max(sizes), 8), dtype=torch.float64, device=dev)
        pad[: local.shape[0]] = local
        gathered = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(gathered, pad)
        local = torch.cat([g[:n] for g, n in zip(gathered, sizes)])
    mean = (local.sum(dim=0) / max(N, 1)).cpu().tolist()             # sums in query order: identical on every rank
    keys = ["precision_no_den", "recall_no_den", "f1_score_no_den", "psnr_no_den_spec", "prec_den", "rec_den", "f1_den",
            "psnr_den_spec"]
    return (dict(zip(keys, mean)), local) if per_query else dict(zip(keys, mean))


def compute_peaks_metrics_files(queries_augmented, clean_dir: str, analyzer_no_den: Audfprint_peaks, analyzer_den: Audfprint_peaks,
                                batch: int = 256) -> Dict[str, float]:
    """The reference's signature (audfprint_exps.py:86-157): a list of augmented query files (.pkl / .wav) whose clean
    counterparts carry the same file name under `clean_dir` (queries_paths["cleans"], :106-107).  Files are read on the host
    and handed to the batched device path; queries must share one length."""
    import os
    aug = [Audfprint_peaks._read_waveform(q, analyzer_no_den.target_sr) for q in queries_augmented]
    clean = [Audfprint_peaks._read_waveform(os.path.join(clean_dir, os.path.basename(q)), analyzer_no_den.target_sr)
             for q in queries_augmented]
    if len({len(a) for a in aug} | {len(c) for c in clean}) > 1:
        raise ValueError("queries of different lengths: group them by length before calling the batched harness")
    return compute_peaks_metrics(torch.stack(clean), torch.stack(aug), analyzer_no_den, analyzer_den, batch=batch)
