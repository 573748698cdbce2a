"""AugmentFP on MI355X -- mirror of the reference's augmentation/__init__.py:16-101 (next-tier row SURVEY.md §8f-3).

Same fixed chain and defaults (HighPass -> impulse response -> background noise @SNR -> Gain -> Clipping -> LowPass ->
HighPass -> PeakNormalization, each Bernoulli-gated per example), same `__call__((1,T)) -> (1,T)` and
`batch_augment((B,1,T)) -> (B,1,T)`, `.augmentation_pipeline.transforms[i].transform_parameters` with the reference's keys.

Differences, on purpose:
  * impulse responses and background noises live in banks resident in HBM, given in memory (`ir_bank`, `noise_bank`;
    `synthetic_banks()` builds deterministic ones) or read ONCE at construction from the reference's arguments
    (`impulse_response_dir`, `background_paths`) when they hold PCM / float .wav files already at `sample_rate` -- the
    reference decodes and resamples a file per example with torchaudio (file I/O is out of scope);
  * the random draws use the same distributions (torch.distributions / random.choice) but are made for the whole batch
    on the host; the per-sample arithmetic runs in csrc/augment.hip;
  * by default `batch_augment` treats every example like `__call__` does, which is how the training data is made
    (training/dataset.py:143); the reference's own batch_augment lets Clipping take its quantiles over the flattened
    selected sub-batch (torch.quantile without a dim, clipping.py:77-93) -- `clipping_scope="batch"` reproduces that;
  * the windowed-sinc filters restate julius 0.2.7 (not in the reference tree): parity unpinned, property-tested.
"""
from __future__ import annotations

import ctypes
import os
import random
from typing import Any, Dict, List, Optional

import numpy as np
import torch

from .._lib import MfpaError, check, lib, ptr, stream
from .constants import DEFAULT_PARAMETERS, IMPULSE_RESPONSE_DIR  # noqa: F401

ZEROS = 8



def _up(t: torch.Tensor, dev) -> torch.Tensor:
    """Host tensor -> device WITHOUT a stream synchronisation: pinned staging buffer + asynchronous copy.  A pageable
    `.to(device)` waits for everything queued on the stream; with ~25 small parameter uploads per batch that stalled the launch
    queue of the training step that follows (Demucs step: 1500 launches; 70 -> 58 ms per 64-clip step with this)."""
    if t.is_cuda:
        return t
    return t.contiguous().pin_memory().to(dev, non_blocking=True)


def _mels(f: torch.Tensor) -> torch.Tensor:          # augmentation/utils.py:36-42
    return 2595.0 * torch.log10(1.0 + f / 700.0)


def _hz(m: torch.Tensor) -> torch.Tensor:            # augmentation/utils.py:45-51
    return 700.0 * (10 ** (m / 2595.0) - 1.0)


def rms_normalize(x: torch.Tensor) -> torch.Tensor:  # augmentation/utils.py:190-205
    return x / (x.square().mean(dim=-1, keepdim=True).sqrt() + 1e-8)


def _find_wavs(paths: List[str]) -> List[str]:
    """augmentation/utils.py:83-137: .wav files given directly or found (recursively, sorted per directory) under directories."""
    out: List[str] = []
    for p in paths:
        if str(p).lower().endswith(".wav"):
            out.append(os.path.abspath(p))
        elif os.path.isdir(p):
            for root, _, names in os.walk(p, followlinks=True):
                out += [os.path.join(root, n) for n in sorted(names) if n.lower().endswith(".wav")]
    return out


def _read_wav(path: str, sample_rate: int) -> torch.Tensor:
    """A PCM / float .wav as a mono float32 tensor in [-1, 1] (the reference decodes with torchaudio and resamples on the fly,
    augmentation/utils.py:140-330; here the file must already be at `sample_rate`: resampling and other codecs are file I/O,
    outside the hot path)."""
    from scipy.io import wavfile
    sr, data = wavfile.read(path)
    if int(sr) != int(sample_rate):
        raise NotImplementedError(f"{path}: {sr} Hz, expected {sample_rate} Hz -- resample the banks offline")
    x = np.asarray(data)
    if x.dtype.kind == "i":
        x = x.astype(np.float32) / float(2 ** (8 * x.dtype.itemsize - 1))
    elif x.dtype.kind == "u":                                            # 8-bit PCM is unsigned
        x = (x.astype(np.float32) - 128.0) / 128.0
    else:
        x = x.astype(np.float32)
    if x.ndim == 2:
        x = x.mean(axis=1)                                               # mono=True
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))


def synthetic_banks(seed: int = 0, sample_rate: int = 8000, n_ir: int = 8, n_noise: int = 8, noise_seconds: float = 10.0):
    """Deterministic stand-ins for the MIT IR survey / DCASE scenes: decaying-noise impulse responses and coloured noises."""
    from .. import synth
    rng = np.random.default_rng(seed)
    irs = []
    for i in range(n_ir):
        L = int(sample_rate * (0.05 + 0.1 * i))
        e = np.exp(-np.arange(L) / (0.2 * L)) * synth.noise(seed * 100 + i, L)
        e[int(rng.integers(0, 20))] += 4.0
        irs.append(torch.from_numpy(e.astype(np.float32)))
    noises = {}
    for s in range(2):
        noises[f"scene{s}"] = []
        for i in range(n_noise // 2):
            n = synth.noise(seed * 100 + 50 + 10 * s + i, int(sample_rate * noise_seconds))
            n = np.convolve(n, np.ones(1 + 4 * s + i) / (1 + 4 * s + i), mode="same")
            noises[f"scene{s}"].append(torch.from_numpy(n.astype(np.float32)))
    return irs, noises


class _Transform:
    """One Bernoulli-gated stage: holds `p` and the `transform_parameters` of the last call (reference key names)."""

    def __init__(self, name: str, p: float):
        self.name, self.p = name, p
        self.transform_parameters: Dict[str, Any] = {}
        self.draws: Dict[str, Any] = {}          # the un-gated per-example draws of the last call (for replaying a batch)

    def gate(self, B: int) -> torch.Tensor:
        should = torch.distributions.Bernoulli(self.p).sample((B,)).to(torch.bool)     # transform.py:101-105
        self.transform_parameters = {"should_apply": should}
        return should


class _Pipeline:
    def __init__(self, transforms):
        self.transforms = transforms
        self.are_parameters_frozen = False

    def freeze_parameters(self, seed: int = 0) -> None:       # transform.py:158-165: reseeds, nothing else
        self.are_parameters_frozen = True
        random.seed(seed)
        torch.manual_seed(seed)

    def unfreeze_parameters(self) -> None:
        self.are_parameters_frozen = False

    def to(self, device):
        return self


class AugmentFP(object):
    def __init__(self, background_paths: Optional[Dict[str, List[str]]] = None, sample_rate: int = 8000,
                 parameters: Dict[str, float] = DEFAULT_PARAMETERS, impulse_response_dir: Optional[str] = None, *,
                 ir_bank: Optional[List[torch.Tensor]] = None, noise_bank: Optional[Dict[str, List[torch.Tensor]]] = None,
                 device="cuda", clipping_scope: str = "example") -> None:
        if clipping_scope not in ("example", "batch"):
            raise ValueError("clipping_scope must be 'example' (like __call__) or 'batch' (the reference's batch_augment)")
        self.clipping_scope = clipping_scope
        if ir_bank is None:
            if impulse_response_dir is None:
                raise ValueError("pass `impulse_response_dir` (a directory of .wav files) or an in-memory `ir_bank`")
            ir_bank = [_read_wav(os.path.join(impulse_response_dir, f), int(sample_rate))
                       for f in sorted(os.listdir(impulse_response_dir)) if f.endswith(".wav")]      # __init__.py:41-45
        if noise_bank is None:
            if background_paths is None:
                raise ValueError("pass `background_paths` ({scene: [files or directories]}) or an in-memory `noise_bank`")
            noise_bank = {scene: [_read_wav(f, int(sample_rate)) for f in _find_wavs(paths)]
                          for scene, paths in background_paths.items()}
            noise_bank = {k: v for k, v in noise_bank.items() if len(v) > 0}
        if len(ir_bank) == 0 or len(noise_bank) == 0:
            raise ValueError("There are no supported audio files found.")     # EmptyPathException in the reference
        self.sample_rate = int(sample_rate)
        self.parameters = dict(parameters)
        self.device = torch.device(device)
        self.ir_bank = [torch.as_tensor(i, dtype=torch.float32).reshape(-1) for i in ir_bank]
        self.noise_bank = {k: [torch.as_tensor(n, dtype=torch.float32).reshape(-1) for n in v] for k, v in noise_bank.items()}
        # both banks live in HBM: impulse responses time-reversed (FIR taps), noises back to back
        self._ir_off = np.concatenate([[0], np.cumsum([len(i) for i in self.ir_bank])]).astype(np.int64)
        self._ir_dev = torch.cat([i.flip(0) for i in self.ir_bank]).to(self.device)
        self._noise_files = [(scene, k) for scene, v in self.noise_bank.items() for k in range(len(v))]
        flat = [self.noise_bank[sc][k] for sc, k in self._noise_files]
        self._noise_off = dict(zip(self._noise_files, np.concatenate([[0], np.cumsum([len(n) for n in flat])]).tolist()))
        self._noise_dev = torch.cat(flat).to(self.device)
        p = self.parameters
        self.t_hp1 = _Transform("HighPassFilter", p["proba_cutoff_freq1"])
        self.t_ir = _Transform("ApplyImpulseResponse", p["proba_ir_response"])
        self.t_bg = _Transform("AddBackgroundNoise", p["proba_snr_in_db"])
        self.t_gain = _Transform("Gain", p["proba_gain_in_db"])
        self.t_clip = _Transform("Clipping", p["proba_percentile_threshold"])
        self.t_lp = _Transform("LowPassFilter", p["proba_cutoff_freq2"])
        self.t_hp3 = _Transform("HighPassFilter", p["proba_cutoff_freq3"])
        self.t_peak = _Transform("PeakNormalization", 1.0)
        self.augmentation_pipeline = _Pipeline([self.t_hp1, self.t_ir, self.t_bg, self.t_gain, self.t_clip, self.t_lp,
                                                self.t_hp3, self.t_peak])

    # ------------------------------------------------------------------ host-side draws
    def _cutoffs(self, t: _Transform, lo: float, hi: float, B: int) -> torch.Tensor:
        """pass_filters.py:59-82: uniform in mel space between ceil(mel(lo)) and floor(mel(hi))."""
        dist = torch.distributions.Uniform(low=torch.ceil(_mels(torch.tensor(lo, dtype=torch.float32))),
                                           high=torch.floor(_mels(torch.tensor(hi, dtype=torch.float32))), validate_args=True)
        cut = _hz(dist.sample((B,)))
        t.transform_parameters["cutoff_freq"] = cut[t.transform_parameters["should_apply"]]
        t.draws = {"cutoff_freq": cut}
        return cut

    def _random_background(self, T: int):
        """background_noise.py:64-141 (non-mixup branch): random scene, random file, random offset; the slices are
        RMS-normalised, concatenated until T samples, and RMS-normalised again (on the device: mfpa_gather_background).
        Returns [(scene, file index, offset, length)]."""
        pieces, missing = [], T
        while missing > 0:
            scene = random.choice(list(self.noise_bank.keys()))
            k = random.randrange(len(self.noise_bank[scene]))       # random.choice(list) draws the same index
            n = len(self.noise_bank[scene][k])
            if n >= missing:
                pieces.append((scene, k, random.randint(0, n - missing), missing))
                missing = 0
            else:
                pieces.append((scene, k, 0, n))
                missing -= n
        return pieces

    # ------------------------------------------------------------------ device chain
    def _filter(self, x, t: _Transform, cut_hz: torch.Tensor, highpass: bool):
        B, T = x.shape
        should = t.transform_parameters["should_apply"]
        frac = (cut_hz / self.sample_rate).tolist()
        half = []
        for b in range(B):
            if should[b]:
                c = frac[b]
                if not (0.0 < c <= 0.5):
                    raise ValueError(f"Buggy cutoff freq. {c}")            # pass_filters.py:103-110
                if ZEROS / c / 2 >= 2 ** 30:
                    raise ValueError(f"Buggy cutoff freq. {c}: more than 2^31 taps")
                half.append(int(ZEROS / c / 2))
            else:
                half.append(1)
        dev = x.device
        ntaps = [2 * h + 1 for h in half]
        tap_off = np.concatenate([[0], np.cumsum(ntaps)]).astype(np.int64)          # ragged: a 0.5 Hz cut-off has 128 001 taps
        cutoff_d = _up(torch.tensor([f if s else 0.25 for f, s in zip(frac, should.tolist())], dtype=torch.float32), dev)
        half_d = _up(torch.tensor(half, dtype=torch.int32), dev)
        ntaps_d = _up(torch.tensor(ntaps, dtype=torch.int32), dev)
        off_d = _up(torch.from_numpy(tap_off[:-1].copy()), dev)
        taps = torch.empty((int(tap_off[-1]),), dtype=torch.float32, device=dev)
        check(lib().mfpa_lowpass_taps(ptr(cutoff_d), ptr(half_d), ptr(off_d), B, ptr(taps), stream()), "mfpa_lowpass_taps")
        y = torch.empty_like(x)
        apply_d = _up(should.to(torch.uint8), dev)
        check(lib().mfpa_fir(ptr(x), B, T, T, ptr(taps), ptr(off_d), ptr(ntaps_d), ptr(half_d), ptr(apply_d), 0,
                             1 if highpass else 0, ptr(y), 0, stream()), "mfpa_fir")
        return y

    @torch.no_grad()
    def batch_augment(self, waveforms: torch.Tensor) -> Any:
        if waveforms.dim() != 3 or waveforms.shape[1] != 1:
            raise RuntimeError("expects three-dimensional input tensors [batch_size, 1, num_samples]")    # transform.py:67-73
        x = waveforms[:, 0].to(self.device, torch.float32).contiguous()
        B, T = x.shape
        if B * T == 0:
            return waveforms
        dev, L, p = x.device, lib(), self.parameters
        u8 = lambda m: _up(m.to(torch.uint8), dev)
        # 1 HighPass(0-150 Hz)
        self.t_hp1.gate(B)
        x = self._filter(x, self.t_hp1, self._cutoffs(self.t_hp1, p["min_cutoff_freq1"], p["max_cutoff_freq1"], B), True)
        # 2 impulse response: full convolution, / peak of the full result, first T samples (impulse_response.py:73-117)
        should = self.t_ir.gate(B)
        pick = [random.randrange(len(self.ir_bank)) for _ in range(B)]
        irs = [self.ir_bank[i] for i in pick]
        nlen = [len(i) for i in irs]
        nmax = max(nlen)
        self.t_ir.transform_parameters["ir"] = irs
        self.t_ir.draws = {"ir": irs}
        n_d = _up(torch.tensor(nlen, dtype=torch.int32), dev)
        off_d = n_d - 1                                              # y[t] = sum_k ir[n-1-k] x[t + k - (n-1)]
        toff_d = _up(torch.from_numpy(self._ir_off[pick]), dev)
        y, peak, on_d = torch.empty_like(x), torch.empty((B,), dtype=torch.float32, device=dev), u8(should)
        check(L.mfpa_fir(ptr(x), B, T, T + nmax - 1, ptr(self._ir_dev), ptr(toff_d), ptr(n_d), ptr(off_d), ptr(on_d), 1, 2, ptr(y),
                         ptr(peak), stream()), "mfpa_fir")
        check(L.mfpa_scale_rows(ptr(y), B, T, ptr(peak), ptr(on_d), 1, ptr(y), stream()), "mfpa_scale_rows")
        x = y
        # 3 background noise at a random SNR (background_noise.py:143-215)
        should = self.t_bg.gate(B)
        pieces = [self._random_background(T) for _ in range(B)]
        P = max(len(pc) for pc in pieces)
        src, ln = np.zeros((B, P), dtype=np.int64), np.zeros((B, P), dtype=np.int32)
        for b, pc in enumerate(pieces):
            for j, (scene, k, off, n) in enumerate(pc):
                src[b, j], ln[b, j] = self._noise_off[(scene, k)] + off, n
        src_d, ln_d = _up(torch.from_numpy(src), dev), _up(torch.from_numpy(ln), dev)
        noise = torch.empty((B, T), dtype=torch.float32, device=dev)
        check(L.mfpa_gather_background(ptr(self._noise_dev), ptr(src_d), ptr(ln_d), B, P, T, ptr(noise), stream()),
              "mfpa_gather_background")
        snr = torch.distributions.Uniform(torch.tensor(float(p["min_snr_in_db"])), torch.tensor(float(p["max_snr_in_db"]))).sample((B,))
        sel = _up(torch.nonzero(should).flatten(), dev)          # host-known indices: no device-side nonzero, no sync
        self.t_bg.transform_parameters.update(background=noise.index_select(0, sel), snr_in_db=snr[should])
        self.t_bg.draws = {"background": noise, "snr_in_db": snr, "pieces": pieces}
        y, snr_d, on_d = torch.empty_like(x), _up(snr, dev), u8(should)
        check(L.mfpa_mix_background(ptr(x), B, T, ptr(noise), ptr(snr_d), ptr(on_d), ptr(y), stream()), "mfpa_mix_background")
        x = y
        # 4 gain (gain.py:44-70)
        should = self.t_gain.gate(B)
        gdb = torch.distributions.Uniform(torch.tensor(float(p["min_gain_in_db"])), torch.tensor(float(p["max_gain_in_db"]))).sample((B,))
        fac = 10 ** (gdb / 20)
        self.t_gain.transform_parameters["gain_factors"] = fac[should].unsqueeze(1).unsqueeze(1)
        self.t_gain.draws = {"gain_in_db": gdb}
        y = torch.empty_like(x)
        fac_d, on_d = _up(fac, dev), u8(should)
        check(L.mfpa_scale_rows(ptr(x), B, T, ptr(fac_d), ptr(on_d), 0, ptr(y), stream()), "mfpa_scale_rows")
        x = y
        # 5 clipping at per-example quantiles (clipping.py:40-100)
        should = self.t_clip.gate(B)
        pct = torch.distributions.Uniform(torch.tensor(0.0), torch.tensor(float(p["max_percentile_threshold"]))).sample((B,))
        self.t_clip.transform_parameters["percentile_threshold"] = pct[should].unsqueeze(1)
        self.t_clip.draws = {"percentile_threshold": pct}
        y = torch.empty_like(x)
        pct_d, on_d = _up(pct, dev), u8(should)
        if self.clipping_scope == "batch" and B > 1:
            check(L.mfpa_clip_quantile_flat(ptr(x), B, T, ptr(pct_d), ptr(on_d), int(should.sum()), ptr(y), stream()),
                  "mfpa_clip_quantile_flat")
        else:
            check(L.mfpa_clip_quantile(ptr(x), B, T, ptr(pct_d), ptr(on_d), ptr(y), stream()), "mfpa_clip_quantile")
        x = y
        # 6 LowPass(3000-3999 Hz), 7 HighPass(30-150 Hz)
        self.t_lp.gate(B)
        x = self._filter(x, self.t_lp, self._cutoffs(self.t_lp, p["min_cutoff_freq2"], p["max_cutoff_freq2"], B), False)
        self.t_hp3.gate(B)
        x = self._filter(x, self.t_hp3, self._cutoffs(self.t_hp3, p["min_cutoff_freq3"], p["max_cutoff_freq3"], B), True)
        # 8 peak normalisation, p = 1 (peak_normalization.py:38-67)
        self.t_peak.gate(B)
        y = torch.empty_like(x)
        check(L.mfpa_mix_background(ptr(x), B, T, 0, 0, 0, ptr(y), stream()), "mfpa_mix_background")
        return y.unsqueeze(1)

    def __call__(self, waveform: torch.Tensor) -> Any:
        return self.batch_augment(waveform.unsqueeze(0)).squeeze(0)
