"""Oracle: landmark pairing and hashing (the step right after the peak pickers).

Audfprint: peaks2landmarks (afp/audfprint/peak_extractor.py:313-346), landmarks2hashes (:40-58) and the
duplicate removal of wavfile2hashes (:443-460).  Dejavu: generate_hashes (afp/dejavu/fingerprint.py:174-213).
Integer-only: the device results must be identical.
"""
from __future__ import annotations

import hashlib
from typing import List, Tuple

import numpy as np

# peak_extractor.py:99-107
MAXPAIRSPERPEAK = 3
MINDT = 2
TARGETDT = 63
TARGETDF = 31
# afp/dejavu/variables.py:20-22, testing/parameters.py:31
MIN_HASH_TIME_DELTA = 0
MAX_HASH_TIME_DELTA = 200
FINGERPRINT_REDUCTION = 20
FAN_VALUE = 3


def peaks2landmarks(pklist: List[Tuple[int, int]]) -> List[Tuple[int, int, int, int]]:
    """[(col, bin)] column-sorted -> [(col, bin1, bin2, dcol)]: each peak pairs with at most 3 later peaks,
    2 <= dcol < 63 (and col2 < last peak column + 1), |dbin| < 31, scanning col2 then bin2 ascending."""
    landmarks = []
    if len(pklist) == 0:
        return landmarks
    scols = pklist[-1][0] + 1
    peaks_at = [[] for _ in range(scols)]
    for col, b in pklist:
        peaks_at[col].append(b)
    for col in range(scols):
        for peak in peaks_at[col]:
            pairs = 0
            for col2 in range(col + MINDT, min(scols, col + TARGETDT)):
                if pairs >= MAXPAIRSPERPEAK:
                    break
                for peak2 in peaks_at[col2]:
                    if abs(peak2 - peak) < TARGETDF and pairs < MAXPAIRSPERPEAK:
                        landmarks.append((col, peak, peak2, col2 - col))
                        pairs += 1
    return landmarks


def landmarks2hashes(landmarks) -> np.ndarray:
    """(time, hash) int32 rows; hash = (bin1 & 255) << 12 | ((bin2 - bin1) & 63) << 6 | (dt & 63)."""
    lm = np.array(landmarks, dtype=np.int64).reshape(-1, 4)
    out = np.zeros((lm.shape[0], 2), dtype=np.int32)
    out[:, 0] = lm[:, 0]
    out[:, 1] = ((lm[:, 1] & 255) << 12) | (((lm[:, 2] - lm[:, 1]) & 63) << 6) | (lm[:, 3] & 63)
    return out


def unique_sorted_hashes(hashes: np.ndarray) -> np.ndarray:
    """wavfile2hashes' duplicate removal: merge (time, hash) into time << 32 + hash, unique, sort, split."""
    if len(hashes) == 0:
        return np.zeros((0, 2), dtype=np.int32)
    merged = (hashes[:, 0].astype(np.uint64) << np.uint64(32)) + hashes[:, 1].astype(np.uint64)
    u = np.sort(np.unique(merged))
    return np.stack([(u >> np.uint64(32)), (u & np.uint64((1 << 32) - 1))], axis=1).astype(np.int32)


def audfprint_hashes_from_mask(mask: np.ndarray) -> np.ndarray:
    """peaks_mask (bins, frames) -> the unique sorted (time, hash) rows of wavfile2hashes (shifts = 1)."""
    cols, bins = np.nonzero(mask.T)
    return unique_sorted_hashes(landmarks2hashes(peaks2landmarks(list(zip(cols.tolist(), bins.tolist())))))


def dejavu_generate_hashes(peaks: List[Tuple[int, int]], fan_value: int = FAN_VALUE) -> List[Tuple[str, int]]:
    """[(freq, time)] -> [(sha1("f1|f2|dt")[:20], t1)]: stable sort by time, each peak with its next fan_value-1."""
    peaks = sorted(peaks, key=lambda p: p[1])
    out = []
    for i in range(len(peaks)):
        for j in range(1, fan_value):
            if i + j < len(peaks):
                f1, t1 = peaks[i]
                f2, t2 = peaks[i + j]
                dt = t2 - t1
                if MIN_HASH_TIME_DELTA <= dt <= MAX_HASH_TIME_DELTA:
                    h = hashlib.sha1(f"{f1}|{f2}|{dt}".encode("utf-8"))
                    out.append((h.hexdigest()[:FINGERPRINT_REDUCTION], t1))
    return out


def dejavu_hashes_from_mask(mask: np.ndarray) -> List[Tuple[str, int]]:
    freqs, times = np.nonzero(mask)
    return dejavu_generate_hashes(list(zip(freqs.tolist(), times.tolist())))
