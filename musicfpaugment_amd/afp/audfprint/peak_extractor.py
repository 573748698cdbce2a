"""Audfprint peak extraction on MI355X -- mirror of afp/audfprint/peak_extractor.py:76-311.

``Audfprint_peaks(params, denoising, denoising_model)`` keeps the reference's constructor and
``find_peaks(d) -> (pklist, peaks_mask, spec)`` contract (empty input -> ``([], np.array([]))``,
peak_extractor.py:253-254).  Unlike the reference, importing this module loads no checkpoint and
needs no NVIDIA GPU: the denoiser is passed in (``unet=``).  ``find_peaks_batch`` is the batched,
sync-free form the benchmark uses; ``find_peaks`` wraps it for one clip.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

from ... import ops
from ...constants import afp_settings


def landmarks2hashes(landmarks_list) -> np.ndarray:
    """[(time, bin1, bin2, dtime)] -> (n, 2) int32 (time, hash).  afp/audfprint/peak_extractor.py:40-58.
    Pure bit packing of a host list (no device work needed); the batched device path packs hashes inside
    mfpa_audfprint_landmarks."""
    lm = np.array(landmarks_list, dtype=np.int64).reshape(-1, 4)
    if lm.shape[0] == 0:
        return np.zeros((0, 2), dtype=np.int32)
    out = np.zeros((lm.shape[0], 2), dtype=np.int32)
    out[:, 0] = lm[:, 0]
    out[:, 1] = ((lm[:, 1] & 255) << 12) | (((lm[:, 2] - lm[:, 1]) & 63) << 6) | (lm[:, 3] & 63)
    return out


class Audfprint_peaks(object):
    def __init__(self, params: Optional[Dict[str, Any]] = None, denoising: bool = False, denoising_model=None,
                 unet=None, device="cuda", demucs=None) -> None:
        params = afp_settings["audfprint"] if params is None else params
        self.density = params["density"]
        self.target_sr = params["samplerate"]
        self.n_fft = params["n_fft"]
        self.n_hop = params["n_hop"]
        self.shifts = params["shifts"]
        self.f_sd = params["freq-sd"]
        self.maxpksperframe = params["pks-per-frame"]
        self.maxpairsperpeak = 3
        self.mindt = 2
        self.targetdt = 63
        self.targetdf = 31
        self.denoising = denoising
        self.denoising_model = denoising_model
        self.device = torch.device(device)
        if self.n_fft != 512 or self.n_hop != 256:
            raise NotImplementedError("the HIP STFT is built for n_fft 512 / hop 256 (testing/parameters.py:24-25)")
        if self.denoising:
            assert self.denoising_model in ["demucs", "unet"]
            if self.denoising_model == "demucs":
                if demucs is None:
                    raise ValueError("denoising_model='demucs' needs the Demucs instance (demucs=...): this module loads no checkpoint")
                self.unet, self.demucs = None, demucs.to(self.device).eval()
            else:
                if unet is None:
                    raise ValueError("denoising=True needs the UNet instance (unet=...): this module loads no checkpoint")
                self.unet, self.demucs = unet.to(self.device).eval(), None
        else:
            self.unet, self.demucs = None, None

    # ------------------------------------------------------------------ batched device path
    def find_peaks_batch(self, wav: torch.Tensor, want_spec: bool = True) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """(B, T) float32 on the GPU -> (mask (B,256,nF) uint8, npeaks (B,) int32, spec (B,257,nF)).

        STFT -> per-clip /max -> [UNet] -> log/mean/high-pass -> forward + backward pruning; no host sync.
        spec is float64 without denoising and float32 with it, like the reference's third return value.  `want_spec=False`
        (peak masks only, no denoiser): the normalised spectrogram is not materialised -- the division by the clip maximum
        happens inside the log / high-pass kernel, same float64 quotient -- and None is returned in its place.
        """
        mag, cmax = ops.stft_mag(wav, torch.float64)
        a_dec = ops.audfprint_a_dec(self.density, self.n_hop)
        if self.unet is not None:
            spec = self.unet.denoise_spectrogram(mag, cmax, per_clip=True)        # float32 (B,257,nF)
            filtered = ops.audfprint_prepare(spec, None, mean_order=0)            # C-contiguous in the reference
        elif want_spec:
            spec = ops.normalize_(mag, cmax, per_clip=True)
            filtered = ops.audfprint_prepare(spec, None, mean_order=1)            # |stft| is a transposed view there
        else:
            if mag.shape[2] <= 512 and (256 * mag.shape[2]) % 16 == 0:               # stages 1 + 2 fused: no filtered spectrogram in memory
                mask, npeaks = ops.audfprint_pick(mag, cmax, a_dec, self.maxpksperframe, float(self.f_sd))
                return mask, npeaks, None
            spec = None
            filtered = ops.audfprint_prepare(mag, cmax, mean_order=1, denom_is_clip_max=True)
        mask, npeaks = ops.audfprint_prune(filtered, a_dec, self.maxpksperframe, float(self.f_sd))
        return mask, npeaks, spec

    def wav2peaks_batch(self, wav: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """wavfile2peaks (peak_extractor.py:347-430) for waveforms already in memory: with denoising_model == "demucs" the
        WAVEFORM goes through the Demucs denoiser first (:369-376), then find_peaks; the UNet acts inside find_peaks."""
        if self.demucs is not None:
            wav = self.demucs(wav)[:, 0].contiguous()
        return self.find_peaks_batch(wav)

    # ------------------------------------------------------------------ reference call surface
    def find_peaks(self, d) -> Tuple[List[Tuple[int, int]], np.ndarray, np.ndarray]:
        if len(d) == 0:
            return [], np.array([])
        x = torch.as_tensor(np.asarray(d) if not isinstance(d, torch.Tensor) else d).to(self.device, torch.float32)
        mask, _, spec = self.find_peaks_batch(x.reshape(1, -1))
        m = mask[0]
        cols, bins = torch.nonzero(m.t(), as_tuple=True)                          # column-major ascending (:305-309)
        pklist = list(zip(cols.tolist(), bins.tolist()))
        return pklist, m.to(torch.float32).cpu().numpy(), spec[0].cpu().numpy()

    # ------------------------------------------------------------------ file-level entry points of the reference
    @staticmethod
    def _read_waveform(filename: str, target_sr: int) -> torch.Tensor:
        """.pkl: the pickled 8 kHz waveform of the reference's query sets (peak_extractor.py:361-368); .wav: PCM / float at the
        target rate.  mp3 decoding and resampling (torchaudio) are file I/O outside the hot path."""
        ext = filename.rsplit(".", 1)[-1].lower()
        if ext == "pkl":
            import pickle
            with open(filename, "rb") as fh:
                return torch.as_tensor(np.asarray(pickle.load(fh)), dtype=torch.float32).reshape(-1)
        if ext == "wav":
            from scipy.io import wavfile
            sr, data = wavfile.read(filename)
            if int(sr) != int(target_sr):
                raise NotImplementedError(f"{filename}: {sr} Hz, expected {target_sr} Hz -- resample offline")
            x = np.asarray(data)
            x = x.astype(np.float32) / float(2 ** (8 * x.dtype.itemsize - 1)) if x.dtype.kind == "i" else x.astype(np.float32)
            return torch.from_numpy(np.ascontiguousarray(x.mean(axis=1) if x.ndim == 2 else x, dtype=np.float32))
        raise NotImplementedError(f"{filename}: only .pkl and .wav inputs (decode other formats offline)")

    def wavfile2peaks(self, filename: str, shifts: Optional[int] = None, get_masks_waveforms: bool = False):
        """peak_extractor.py:347-424: [(time, bin)] of one file, or the list of `shifts` such lists, or
        (peaks_mask, waveform, sgram) with get_masks_waveforms."""
        d = self._read_waveform(filename, self.target_sr)
        if self.demucs is not None:
            d = self.demucs(d.reshape(1, -1).to(self.device))[0, 0].cpu()           # :369-376
        self.soundfiledur = len(d) / self.target_sr
        if shifts is None or shifts < 2:
            peaks, peaks_mask, sgram = self.find_peaks(d)
        else:
            peaks = [self.find_peaks(d[int(s / self.shifts * self.n_hop):])[0] for s in range(shifts)]
            peaks_mask = sgram = None
        if get_masks_waveforms:
            return peaks_mask, d, sgram
        return peaks

    def wavfile2hashes(self, filename: str) -> np.ndarray:
        """peak_extractor.py:426-460: unique sorted (time, hash) rows of one file (the instance's `shifts`)."""
        d = self._read_waveform(filename, self.target_sr).reshape(1, -1).to(self.device)
        if self.demucs is not None:
            d = self.demucs(d)[:, 0].contiguous()
        uq, n = self.hashes_batch(d)
        n0 = int(n[0])
        if n0 < 0:                                           # never a slice bound: hashes_batch raises on overflow, this guards the contract
            raise ValueError("landmark capacity exceeded")
        return uq[0, :n0].cpu().numpy().astype(np.int32)

    # ------------------------------------------------------------------ landmarks / hashes (next-tier row §8f-1)
    def hashes_batch(self, wav: torch.Tensor, cap: int = 4096, shifts: Optional[int] = None):
        """(B, T) float32 on the GPU -> (unique sorted (time, hash) rows (B, cap', 2) int32, counts (B,) int32):
        wavfile2hashes (peak_extractor.py:426-460) for a whole batch, peaks never leaving the device.  `shifts` (default:
        the instance's, 1 in testing/parameters.py): hashes of the waveform advanced by s / shifts frames, s = 0..shifts-1,
        are merged before the duplicate removal (:406-424, :437-444); cap' = cap * shifts."""
        shifts = self.shifts if shifts is None else shifts
        if shifts is None or shifts < 2:
            mask, _, _ = self.find_peaks_batch(wav)
            while True:
                _, _, uniq, counts = ops.audfprint_landmarks(mask, cap, self.mindt, self.targetdt, self.targetdf,
                                                             self.maxpairsperpeak)
                if not bool((counts < 0).any()):            # the kernel flags an overflowing clip with counts [-1, -1]
                    return uniq, counts[:, 1].contiguous()
                if cap >= 8192:
                    raise ValueError("more than 8 peaks in one frame or more than 8192 landmarks in one clip: outside the device "
                                     "kernel's limits")
                cap = 8192                                   # retry once with the kernel's largest capacity
        B = wav.shape[0]
        keys = []
        for s in range(shifts):
            shiftsamps = int(s / self.shifts * self.n_hop) if self.shifts and self.shifts > 1 else int(s / shifts * self.n_hop)
            mask, _, _ = self.find_peaks_batch(wav[:, shiftsamps:].contiguous())
            _, hs, _, counts = ops.audfprint_landmarks(mask, cap, self.mindt, self.targetdt, self.targetdf, self.maxpairsperpeak)
            if bool((counts < 0).any()):
                raise ValueError("landmark capacity exceeded: raise `cap` (the kernel's limit is 8192 landmarks per clip)")
            k = (hs[:, :, 0].to(torch.int64) << 32) + (hs[:, :, 1].to(torch.int64) & 0xFFFFFFFF)     # :447-449
            valid = torch.arange(cap, device=wav.device)[None, :] < counts[:, :1]
            keys.append(torch.where(valid, k, torch.full_like(k, torch.iinfo(torch.int64).max)))
        k = torch.sort(torch.cat(keys, dim=1), dim=1).values                                           # np.sort(np.unique(.))
        first = torch.ones_like(k, dtype=torch.bool)
        first[:, 1:] = k[:, 1:] != k[:, :-1]
        first &= k != torch.iinfo(torch.int64).max
        n = first.sum(dim=1).to(torch.int32)
        pos = torch.cumsum(first.to(torch.int64), dim=1) - 1
        out = torch.zeros((B, cap * shifts, 2), dtype=torch.int32, device=wav.device)
        b_idx = torch.arange(B, device=wav.device)[:, None].expand_as(k)
        out[b_idx[first], pos[first], 0] = (k[first] >> 32).to(torch.int32)
        out[b_idx[first], pos[first], 1] = (k[first] & 0xFFFFFFFF).to(torch.int32)
        return out, n

    def peaks2landmarks(self, pklist: List[Tuple[int, int]]) -> List[Tuple[int, int, int, int]]:
        """[(col, bin)] -> [(col, bin1, bin2, dcol)], peak_extractor.py:313-346, through the device kernel."""
        if len(pklist) == 0:
            return []
        pk = np.asarray(pklist, dtype=np.int64).reshape(-1, 2)
        T, R = int(pk[:, 0].max()) + 1, 256
        if pk[:, 1].max() >= R:
            raise ValueError("bins must be < 256 (they are packed into 8 bits, peak_extractor.py:54)")
        mask = torch.zeros((1, R, T), dtype=torch.uint8)
        mask[0, pk[:, 1], pk[:, 0]] = 1
        cap = min(8192, max(16, 3 * len(pk)))
        lm, _, _, counts = ops.audfprint_landmarks(mask.to(self.device), cap, self.mindt, self.targetdt, self.targetdf,
                                                   self.maxpairsperpeak)
        n = int(counts[0, 0])
        if n < 0:
            raise ValueError("more than 8 peaks in one frame or more than 8192 landmarks: outside the device kernel's limits")
        return [tuple(r) for r in lm[0, :n].cpu().tolist()]
