"""Oracle: Audfprint spectral-peak picker (reference rows a3-a8 of SURVEY.md §8a).

Follows afp/audfprint/peak_extractor.py of the reference.  All comparisons are
IEEE float64 (or float32 for the log/mean step when the spectrogram is float32,
exactly as numpy promotes in the reference), so given the same spectrogram the
peak set is a pure function of it.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

from . import stft as ostft

# testing/parameters.py:17-26 (afp_settings["audfprint"])
DENSITY = 20
MAX_PKS_PER_FRAME = 5
F_SD = 30.0
N_FFT = 512
N_HOP = 256
HPF_POLE = 0.98  # peak_extractor.py:287


def a_dec(density: float = DENSITY, n_hop: int = N_HOP) -> float:
    """Masking-envelope decay per frame.  peak_extractor.py:295."""
    return float(1 - 0.01 * (density * np.sqrt(n_hop / 352.8) / 35))


def gauss_table(npoints: int = 256, width: float = F_SD) -> np.ndarray:
    """exp(-0.5 (j/width)^2), j = -npoints..npoints.  peak_extractor.py:163-165."""
    return np.exp(-0.5 * ((np.arange(-npoints, npoints + 1) / width) ** 2))


def locmax(vec: np.ndarray) -> np.ndarray:
    """Boolean local-maximum mask of a 1-D vector.  peak_extractor.py:61-73.

    i is a peak iff (i == 0 or v[i] >= v[i-1]) and not (i < n-1 and v[i+1] >= v[i]).
    """
    n = len(vec)
    ge_prev = np.ones(n + 1, dtype=bool)
    ge_prev[1:n] = vec[1:] >= vec[:-1]
    ge_prev[n] = False
    return ge_prev[:-1] & ~ge_prev[1:]


def spread(vec: np.ndarray, table: np.ndarray, base=None) -> np.ndarray:
    """max over local maxima p of vec[p] * G[k - p] (over ``base`` or zeros).

    peak_extractor.py:115-171 (spreadpeaksinvector + spreadpeaks).
    """
    n = len(vec)
    out = np.zeros(n, dtype=np.float32) if base is None else np.copy(base)
    for p in np.nonzero(locmax(vec))[0]:
        out = np.maximum(out, vec[p] * table[n - p : 2 * n - p])
    return out


def fwd_prune(sgram: np.ndarray, adec: float, table: np.ndarray, maxpks: int = MAX_PKS_PER_FRAME) -> np.ndarray:
    """Forward decaying-threshold pass.  peak_extractor.py:173-204.

    Candidates of a column are decided against the threshold as it stood BEFORE
    any update from the same column; the best ``maxpks`` by (value, bin)
    descending are kept and each raises the threshold by its Gaussian skirt.
    """
    rows, cols = sgram.shape
    sthresh = spread(np.max(sgram[:, : min(10, cols)], axis=1), table)
    peaks = np.zeros((rows, cols), dtype=np.float32)
    for c in range(cols):
        col = sgram[:, c]
        cand = np.nonzero(locmax(col) & (col > sthresh))[0]
        order = sorted(zip(col[cand], cand), reverse=True)[:maxpks]
        for val, p in order:
            sthresh = np.maximum(sthresh, val * table[rows - p : 2 * rows - p])
            peaks[p, c] = 1
        sthresh = sthresh * adec
    return peaks


def bwd_prune(sgram: np.ndarray, peaks: np.ndarray, adec: float, table: np.ndarray) -> np.ndarray:
    """Backward pruning pass (in place on ``peaks``).  peak_extractor.py:206-234.

    Unlike the forward pass the threshold is updated BETWEEN peaks of one column;
    a surviving peak also clears the same-bin peak of the following column.
    """
    rows, cols = sgram.shape
    sthresh = spread(sgram[:, -1], table)
    for c in range(cols - 1, -1, -1):
        pk = np.nonzero(peaks[:, c])[0]
        for val, p in sorted(zip(sgram[pk, c], pk), reverse=True):
            if val >= sthresh[p]:
                sthresh = np.maximum(sthresh, val * table[rows - p : 2 * rows - p])
                if c + 1 < cols:
                    peaks[p, c + 1] = 0
            else:
                peaks[p, c] = 0
        sthresh = adec * sthresh
    return peaks


NPY_BUFSIZE = 8192  # numpy's default ufunc buffer size, in elements


def pairwise_sum(a: np.ndarray):
    """numpy's pairwise summation of one inner-loop block, restated.

    numpy/_core/src/umath/loops_utils.h.src: PW_BLOCKSIZE 128, eight strided
    accumulators per leaf, split at (n/2 rounded down to a multiple of 8).
    Accumulates in a.dtype.
    """
    n = len(a)
    if n < 8:
        s = a.dtype.type(0)
        for v in a:
            s = s + v
        return s
    if n <= 128:
        m = n - (n % 8)
        r = a[:m].reshape(-1, 8)
        acc = r[0].copy()
        for row in r[1:]:
            acc = acc + row
        s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]))
        for v in a[m:]:
            s = s + v
        return s
    n2 = n // 2
    n2 -= n2 % 8
    return pairwise_sum(a[:n2]) + pairwise_sum(a[n2:])


def numpy_sum(flat: np.ndarray):
    """np.add.reduce of a contiguous array given in MEMORY order, restated.

    A full reduction walks memory order in buffer-sized chunks of 8192 elements:
    acc = 0; acc += pairwise_sum(chunk) for each chunk.  np.mean(sgram) in
    peak_extractor.py:276 therefore depends on the array's layout: the
    un-denoised spectrogram is the transpose of a (frames, bins) array
    (afp/audfprint/stft.py:62) and is summed frame-major; the UNet output
    (peak_extractor.py:269) is C-contiguous and is summed bin-major.  The device
    kernel reproduces both orders so the mean is bit-identical.
    """
    acc = flat.dtype.type(0)
    for s in range(0, len(flat), NPY_BUFSIZE):
        acc = acc + pairwise_sum(flat[s : s + NPY_BUFSIZE])
    return acc


def numpy_mean(a: np.ndarray):
    """np.mean of a C- or F-contiguous 2-D array (sum in memory order, one divide in a.dtype)."""
    flat = a.reshape(-1) if a.flags["C_CONTIGUOUS"] else a.T.reshape(-1)
    return a.dtype.type(numpy_sum(flat) / a.dtype.type(a.size))


def highpass(rows: np.ndarray, pole: float = HPF_POLE) -> np.ndarray:
    """scipy.signal.lfilter([1, -1], [1, -pole], row) per row, restated.

    peak_extractor.py:286-290.  Direct-form-II-transposed, zero initial state,
    float64, no fused multiply-add:  y[n] = x[n] + z ;  z = -x[n] + pole * y[n].
    Float32 input is promoted to float64 first (scipy's result type).
    """
    x = np.asarray(rows, dtype=np.float64)
    y = np.empty_like(x)
    z = np.zeros(x.shape[0], dtype=np.float64)
    for n in range(x.shape[1]):
        xn = x[:, n]
        yn = xn + z
        z = -xn - (-pole) * yn
        y[:, n] = yn
    return y


def log_mean_normalise(sgram: np.ndarray, order: str = "memory") -> np.ndarray:
    """log(max(s, max/1e6)) - mean, skipped when max <= 0.  peak_extractor.py:272-280.

    Keeps the input dtype (float32 after the UNet, float64 otherwise).  np.mean sums in
    the array's memory order (see numpy_sum).  ``order``: "memory" = whatever layout
    ``sgram`` has (what the reference does); "F" = frame-major (the reference's
    un-denoised path); "C" = bin-major (the reference's UNet path) regardless of layout.
    """
    smax = np.max(sgram)
    if smax > 0.0:
        sgram = np.log(np.maximum(sgram, smax / 1e6))
        if order == "memory":
            mean = np.mean(sgram)
        else:
            flat = np.ascontiguousarray(sgram if order == "C" else sgram.T).reshape(-1)
            mean = sgram.dtype.type(numpy_sum(flat) / sgram.dtype.type(sgram.size))
        sgram = sgram - mean
    return sgram


def preprocess(sgram: np.ndarray, order: str = "memory") -> np.ndarray:
    """Spectrogram (257, T) -> onset-emphasised log spectrogram (256, T) float64.

    peak_extractor.py:271-290: log/mean step, per-row high-pass, drop the Nyquist row.
    """
    return highpass(log_mean_normalise(sgram, order))[:-1]


def peaks_from_filtered(filtered: np.ndarray, density: float = DENSITY, maxpks: int = MAX_PKS_PER_FRAME,
                        f_sd: float = F_SD, n_hop: int = N_HOP) -> np.ndarray:
    """Forward + backward pruning of an already filtered (256, T) float64 array -> mask float32."""
    table = gauss_table(filtered.shape[0], f_sd)
    adec = a_dec(density, n_hop)
    peaks = fwd_prune(filtered, adec, table, maxpks)
    return bwd_prune(filtered, peaks, adec, table)


def pklist_from_mask(mask: np.ndarray) -> List[Tuple[int, int]]:
    """[(col, bin)] column-major ascending.  peak_extractor.py:303-309."""
    cols, bins = np.nonzero(mask.T)
    return list(zip(cols.tolist(), bins.tolist()))


def find_peaks_from_sgram(sgram: np.ndarray, order: str = "memory"):
    """Everything of find_peaks after the (optional) UNet: returns (pklist, mask, spec)."""
    spec = sgram.copy()
    mask = peaks_from_filtered(preprocess(sgram, order))
    return pklist_from_mask(mask), mask, spec


def find_peaks(d: np.ndarray):
    """Audfprint_peaks.find_peaks(d) without denoising.  peak_extractor.py:236-311."""
    if len(d) == 0:
        return [], np.array([])
    sgram = ostft.magnitude(d)
    sgram = sgram / np.max(sgram)
    return find_peaks_from_sgram(sgram)


def mask_from_sgram_c(sgram: np.ndarray) -> np.ndarray:
    """(256, T) uint8 peak mask of a C-contiguous spectrogram as find_peaks leaves the UNet branch (peak_extractor.py:265-311): a
    picklable one-argument worker for process pools (tests compare whole batches of device masks with it)."""
    return (np.asarray(find_peaks_from_sgram(np.ascontiguousarray(sgram), order="C")[1]) != 0).astype(np.uint8)


def float32_log_variants(sgram32: np.ndarray):
    """The two logarithms the denoised branch can see (peak_extractor.py:275 takes np.log of a FLOAT32 array after the UNet):
    numpy's own float32 log (SIMD, not correctly rounded: differs from the correctly rounded value on a few per cent of the arguments,
    by up to a few ulp, CPU-dependent) and the float64 log rounded once to float32 (what csrc/audfprint.hip computes).  Returns
    (mask with numpy's log, mask with the rounded float64 log, number of log cells that differ, largest difference in float32 ulp)."""
    assert sgram32.dtype == np.float32
    smax = np.max(sgram32)
    floored = np.maximum(sgram32, smax / 1e6)
    if not smax > 0.0:
        m = mask_from_sgram_c(sgram32)
        return m, m, 0, 0.0
    la = np.log(floored)                                              # float32 in, float32 out: numpy's SIMD kernel
    lb = np.log(floored.astype(np.float64)).astype(np.float32)        # correctly rounded (glibc's double log is < 1 ulp of double)
    ulp = np.abs(la.view(np.int32).astype(np.int64) - lb.view(np.int32).astype(np.int64))

    def rest(lg):                                                     # log_mean_normalise's mean (bin-major order) and everything after it
        flat = np.ascontiguousarray(lg).reshape(-1)
        mean = lg.dtype.type(numpy_sum(flat) / lg.dtype.type(lg.size))
        return (np.asarray(peaks_from_filtered(highpass(lg - mean)[:-1])) != 0).astype(np.uint8)
    return rest(la), rest(lb), int(np.count_nonzero(ulp)), float(ulp.max())


def masks_both_logs(sgram32: np.ndarray):
    """Picklable worker: (mask with numpy's float32 log -- the reference's arithmetic --, mask with the float64 log rounded once -- the
    device's) of one float32 spectrogram of the denoised branch."""
    a, b, _, _ = float32_log_variants(np.ascontiguousarray(sgram32, dtype=np.float32))
    return a, b

