"""Deterministic synthetic 8 s / 8 kHz clips.

The reference ships no audio (its datasets are not in the tree), so every test
and benchmark input is generated here from an integer seed.  The generator is
counter based (a 32-bit integer mix of ``seed`` and the sample index), so a clip
never has to be shipped: fixtures store the seed plus a digest of the samples.

Clip ``k`` of seed ``s``:
  0.1 * N^(0,1) noise (sum of four uniforms, centred and scaled)  [noise clips stop here]
  + 6 Gaussian-enveloped tone bursts with integer-derived frequency / onset
  then peak-normalised to 1 and cast to float32.
"""
from __future__ import annotations

import hashlib

import numpy as np

SAMPLE_RATE = 8000
CLIP_SECONDS = 8
CLIP_SAMPLES = SAMPLE_RATE * CLIP_SECONDS  # 64000
BASE_SEED = 59  # the reference's own seed, training/utils.py:65


def _mix32(x: np.ndarray) -> np.ndarray:
    """murmur3 finaliser on uint32 lanes (wraps mod 2^32)."""
    x = x.astype(np.uint32, copy=True)
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x85EBCA6B)
    x ^= x >> np.uint32(13)
    x *= np.uint32(0xC2B2AE35)
    x ^= x >> np.uint32(16)
    return x


def uniform_u32(seed: int, stream: int, n: int) -> np.ndarray:
    """n uint32 words of stream ``stream`` of ``seed`` (pure integer arithmetic)."""
    with np.errstate(over="ignore"):
        base = np.uint32((seed * 0x9E3779B9 + stream * 0x7F4A7C15 + 0x1B873593) & 0xFFFFFFFF)
        idx = np.arange(n, dtype=np.uint32)
        return _mix32(_mix32(idx + base) ^ np.uint32((seed ^ (stream << 16)) & 0xFFFFFFFF))


def noise(seed: int, n: int = CLIP_SAMPLES) -> np.ndarray:
    """Approximately N(0,1) float64 noise: centred Irwin-Hall sum of 4 uniforms."""
    acc = np.zeros(n, dtype=np.float64)
    for stream in range(4):
        acc += uniform_u32(seed, stream, n).astype(np.float64) * (1.0 / 4294967296.0)
    return (acc - 2.0) * np.sqrt(3.0)


def clip(seed: int, n: int = CLIP_SAMPLES, tonal: bool = True) -> np.ndarray:
    """One float32 clip in [-1, 1]."""
    x = 0.1 * noise(seed, n)
    if tonal:
        par = uniform_u32(seed, 7, 12).astype(np.float64) * (1.0 / 4294967296.0)
        t = np.arange(n, dtype=np.float64) / SAMPLE_RATE
        dur = n / SAMPLE_RATE
        for b in range(6):
            f = 100.0 + 3800.0 * par[2 * b]
            t0 = dur * (0.0625 + 0.875 * par[2 * b + 1])
            env = np.exp(-0.5 * ((t - t0) / (0.0375 * dur)) ** 2)
            x = x + 0.5 * env * np.sin(2.0 * np.pi * f * t)
    peak = np.max(np.abs(x))
    if peak > 0:
        x = x / peak
    return x.astype(np.float32)


def batch(n_clips: int, seed: int = BASE_SEED, n: int = CLIP_SAMPLES, tonal: bool = True) -> np.ndarray:
    """(n_clips, n) float32; clip k uses seed ``seed + k``."""
    return np.stack([clip(seed + k, n, tonal) for k in range(n_clips)])


def digest(a: np.ndarray) -> str:
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()
