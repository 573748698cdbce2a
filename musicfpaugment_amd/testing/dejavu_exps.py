"""Dejavu peak-metrics experiment on MI355X -- mirror of compute_peaks_metrics in the reference's
testing/dejavu_exps.py:82-167 (harness row SURVEY.md §8f-4), batched and sharded like testing/audfprint_exps.py.

The reference walks the augmented query files, calls Dejavu.generate_fingerprints(get_masks=True) three times per query
(clean, augmented, augmented through the denoising Dejavu instance -- afp/dejavu/dejavu.py:255-289), transposes the
(257, 249) peak masks (:118-120) and averages per-query Precision / Recall / F1 and the PSNR of the normalised
spectrograms.  The MySQL-backed Dejavu class (database, matching) is out of scope; `DejavuPeaks` carries the
two attributes of it this path reads -- `denoising`, `denoising_model` -- plus the networks, which the reference loads at
import time.  Queries are tensors (or files through `compute_peaks_metrics_files`); with torch.distributed initialised
they are split over the ranks and the per-query rows gathered, so the means equal a single-GPU run's.

The reference's result dictionary is kept key for key, including its "psnr_*_wav" entries, which it fills from the
spectrogram PSNR (:139-140,158).
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch

from .. import ops
from ..afp.dejavu.fingerprint import fingerprint_peaks_batch
from ..constants import afp_settings
from ..pipeline import shard_range
from .audfprint_exps import _prf, _psnr


class DejavuPeaks:
    """The peak-extraction face of afp/dejavu/dejavu.py's Dejavu (constructor :121-134, generate_fingerprints :255-289)."""

    def __init__(self, settings: Optional[dict] = None, denoising: bool = False, denoising_model: Optional[str] = None,
                 unet=None, demucs=None, device="cuda") -> None:
        self.settings = dict(settings or afp_settings["dejavu"])
        self.denoising = denoising
        self.denoising_model = denoising_model
        if self.denoising is True:
            assert self.denoising_model in ["unet", "demucs"]
            if (unet if denoising_model == "unet" else demucs) is None:
                raise ValueError(f"denoising_model={denoising_model!r} needs the {denoising_model} module")
        self.unet, self.demucs = unet, demucs
        self.device = torch.device(device)

    @torch.no_grad()
    def generate_fingerprints_batch(self, wav: torch.Tensor):
        """(B, T) float32 waveforms in [-1, 1] -> (peak_mask (B, 257, nF) uint8, specgram (B, 257, nF)), the get_masks=True
        return of generate_fingerprints for every clip (read() scales by 32767, dejavu.py:106)."""
        wav = wav.to(self.device, torch.float32).contiguous()
        mask, _, spec = fingerprint_peaks_batch(wav, amp_min=self.settings["amp_min"], scale_in=32767.0,
                                                denoising=bool(self.denoising), denoising_model=self.denoising_model or "unet",
                                                unet=self.unet, demucs=self.demucs)
        return mask, spec


KEYS = ["precision_no_den", "recall_no_den", "f1_score_no_den", "psnr_no_den_spec", "psnr_no_den_wav", "prec_den", "rec_den",
        "f1_den", "psnr_den_spec", "psnr_den_wav"]


@torch.no_grad()
def compute_peaks_metrics(clean_wav: torch.Tensor, augmented_wav: torch.Tensor, djv_no_den: DejavuPeaks, djv_den: DejavuPeaks,
                          batch: int = 256) -> Dict[str, float]:
    """clean_wav, augmented_wav: (N, T) float32 (any device).  Returns the reference's result dictionary."""
    import torch.distributed as dist
    N = clean_wav.shape[0]
    ddp = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    rank, world = (dist.get_rank(), dist.get_world_size()) if ddp else (0, 1)
    lo, hi = shard_range(N, rank, world)
    dev = djv_no_den.device
    rows = []
    for s in range(lo, hi, batch):
        e = min(hi, s + batch)
        m_clean, sg_clean = djv_no_den.generate_fingerprints_batch(clean_wav[s:e])
        m_aug, sg_aug = djv_no_den.generate_fingerprints_batch(augmented_wav[s:e])
        m_den, sg_den = djv_den.generate_fingerprints_batch(augmented_wav[s:e])
        tr = lambda m: m.transpose(1, 2).contiguous()               # (B, nF, 257): dejavu_exps.py:118-120
        p, r, f1 = _prf(ops.peak_metrics_counts(tr(m_aug), tr(m_clean)))
        pd, rd, f1d = _prf(ops.peak_metrics_counts(tr(m_den), tr(m_clean)))
        ps, psd = _psnr(sg_aug, sg_clean), _psnr(sg_den, sg_clean)
        rows.append(torch.stack([p, r, f1, ps, ps, pd, rd, f1d, psd, psd], dim=1))
    local = torch.cat(rows) if rows else torch.zeros((0, len(KEYS)), dtype=torch.float64, device=dev)
    if ddp:
        sizes = [shard_range(N, r, world)[1] - shard_range(N, r, world)[0] for r in range(world)]
        pad = torch.zeros((max(sizes), len(KEYS)), dtype=torch.float64, device=dev)
        pad[: local.shape[0]] = local
        gathered = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(gathered, pad)
        local = torch.cat([g[:n] for g, n in zip(gathered, sizes)])
    mean = (local.sum(dim=0) / max(N, 1)).cpu().tolist()             # sums in query order: identical on every rank
    return dict(zip(KEYS, mean))


def compute_peaks_metrics_files(queries_augmented, clean_dir: str, djv_no_den: DejavuPeaks, djv_den: DejavuPeaks,
                                batch: int = 256) -> Dict[str, float]:
    """The reference's signature (dejavu_exps.py:82-86): augmented query files (.pkl / .wav) whose clean counterparts carry the
    same file name under `clean_dir` (queries_paths["cleans"], :104-105).  Files are read on the host and handed to the
    batched device path; queries must share one length."""
    from ..afp.audfprint.peak_extractor import Audfprint_peaks
    sr = djv_no_den.settings["samplerate"]
    aug = [Audfprint_peaks._read_waveform(q, sr) for q in queries_augmented]
    clean = [Audfprint_peaks._read_waveform(os.path.join(clean_dir, os.path.basename(q)), sr) for q in queries_augmented]
    if len({len(a) for a in aug} | {len(c) for c in clean}) > 1:
        raise ValueError("queries of different lengths: group them by length before calling the batched harness")
    return compute_peaks_metrics(torch.stack(clean), torch.stack(aug), djv_no_den, djv_den, batch=batch)
