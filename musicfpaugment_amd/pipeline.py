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


def unet_mfma_gflop(H: int, W: int) -> float:
    """Algorithmic GFLOP of the UNet's MFMA launches (every 3x3 conv but inc.0, the four transposed convs; training/unet.py:77-108)
    for one (H, W) spectrogram: 93.316 at 257 x 251, 92.986 at Dejavu's 257 x 249."""
    hs = [(H, W)]
    for _ in range(4):
        hs.append((hs[-1][0] // 2, hs[-1][1] // 2))
    ch = [64, 128, 256, 512, 1024]
    macs = hs[0][0] * hs[0][1] * 9 * 64 * 64
    for lvl in range(1, 5):
        macs += hs[lvl][0] * hs[lvl][1] * 9 * (ch[lvl - 1] * ch[lvl] + ch[lvl] * ch[lvl])
    for lvl in range(3, -1, -1):
        macs += hs[lvl + 1][0] * hs[lvl + 1][1] * 4 * ch[lvl + 1] * ch[lvl]
        macs += hs[lvl][0] * hs[lvl][1] * 9 * (2 * ch[lvl] * ch[lvl] + ch[lvl] * ch[lvl])
    return 2.0 * macs / 1e9


def unet_mfma_gflop_executed(H: int, W: int, folded_levels=("up1", "up2", "up3", "up4")) -> float:
    """GFLOP the MFMA launches actually EXECUTE per (H, W) spectrogram with the given decoder levels folded (mfpa_upconv_fused): a folded
    level replaces its transposed convolution (4 C_low C_up MACs per low-resolution pixel) and the up half of its first 3x3 convolution
    (9 C_up C_out per pixel) by four composite taps on the low-resolution tensor (4 C_low C_out per pixel): 87.1 against 93.3 at 257 x 251.
    `roofline.achieved` stays priced in unet_mfma_gflop()'s ALGORITHMIC work; this is only what the issue-rate figure uses."""
    hs = [(H, W)]
    for _ in range(4):
        hs.append((hs[-1][0] // 2, hs[-1][1] // 2))
    ch = [64, 128, 256, 512, 1024]
    saved = 0
    for lvl, name in zip(range(3, -1, -1), ("up1", "up2", "up3", "up4")):
        if name in folded_levels:
            saved += hs[lvl + 1][0] * hs[lvl + 1][1] * 4 * ch[lvl + 1] * ch[lvl]             # the transposed convolution
            saved += hs[lvl][0] * hs[lvl][1] * (9 * ch[lvl] * ch[lvl] - 4 * ch[lvl + 1] * ch[lvl])    # 9 C_up -> 4 C_low products per output
    return unet_mfma_gflop(H, W) - 2.0 * saved / 1e9


class HotPath:
    """picker "audfprint": STFT -> /max -> [UNet] -> log / mean / high-pass -> decaying-threshold prune (peak_extractor.py:236-311);
    picker "dejavu": specgram PSD -> /max -> [UNet, squared] -> 10 ln / mean -> 21 x 21 local maxima (fingerprint.py:56-171)."""

    def __init__(self, unet, device="cuda", picker: str = "audfprint", streams: int = 1):
        """`streams` > 1: consecutive calls run on `streams` HIP streams in turn, so that batch k + 1's STFT overlaps batch k's pruner
        (one wavefront per clip: at 256 clips it leaves the card almost empty for 263 us of a 449 us batch).  Every call still waits
        for the work the caller has queued on the current stream (its input); the RESULTS of such calls are ordered behind the current
        stream only by `join()` -- call it before reading them.  The kernels and their results are the serial path's (per-call buffers)."""
        if picker not in ("audfprint", "dejavu"):
            raise ValueError("picker must be 'audfprint' or 'dejavu'")
        if streams < 1:
            raise ValueError("streams must be >= 1")
        self.picker, self.unet = picker, unet
        self.extractor = Audfprint_peaks(None, denoising=unet is not None, denoising_model="unet" if unet else None,
                                         unet=unet, device=device)
        self._side = [torch.cuda.Stream(device=device) for _ in range(streams)] if streams > 1 else []
        self._turn = 0

    @torch.no_grad()
    def _run(self, wav: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        if self.picker == "dejavu":
            from .afp.dejavu.fingerprint import fingerprint_peaks_batch
            mask, npeaks, _ = fingerprint_peaks_batch(wav, denoising=self.unet is not None, denoising_model="unet", unet=self.unet,
                                                      want_spec=False)
            return mask, npeaks
        mask, npeaks, _ = self.extractor.find_peaks_batch(wav, want_spec=False)
        return mask, npeaks

    def __call__(self, wav: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        if not self._side:
            return self._run(wav)
        s = self._side[self._turn % len(self._side)]
        self._turn += 1
        cur = torch.cuda.current_stream(wav.device)
        s.wait_stream(cur)                                     # the input was produced on the caller's stream
        with torch.cuda.stream(s):
            mask, npeaks = self._run(wav)
        for t in (wav, mask, npeaks):                          # the allocator must not hand these blocks out while either stream uses them
            t.record_stream(s)
            t.record_stream(cur)
        return mask, npeaks

    def join(self) -> None:
        """Order the current stream behind every call made so far (no host wait).  No-op with one stream."""
        for s in self._side:
            torch.cuda.current_stream(s.device).wait_stream(s)


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
