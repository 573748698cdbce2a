"""Oracle: STFT magnitude spectrograms (reference rows a1, a2, a9 of SURVEY.md §8a)."""
from __future__ import annotations

import numpy as np

N_FFT = 512
N_HOP = 256
N_BINS = N_FFT // 2 + 1  # 257


def hann_window(n_fft: int = N_FFT) -> np.ndarray:
    """np.hanning(n_fft + 2)[1:-1]: symmetric Hann with the zero end points dropped.

    training/visualisation.py:18 and afp/audfprint/peak_extractor.py:257.
    """
    return np.hanning(n_fft + 2)[1:-1]


def n_frames(n_samples: int, n_fft: int = N_FFT, hop: int = N_HOP) -> int:
    """afp/audfprint/stft.py:53-54 with centre padding: 1 + n_samples // hop."""
    return 1 + (n_samples + 2 * (n_fft // 2) - n_fft) // hop


def stft_audfprint(signal: np.ndarray, n_fft: int = N_FFT, hop: int = N_HOP, window=None) -> np.ndarray:
    """Complex STFT, (n_fft/2+1, frames) complex128.  afp/audfprint/stft.py:15-62.

    Reflect-pad by n_fft//2 on both sides, cut hop-spaced frames, multiply by the
    window (float32 samples x float64 window -> float64), real FFT, transpose.
    """
    if window is None:
        window = hann_window(n_fft)
    x = np.pad(np.asarray(signal), n_fft // 2, mode="reflect")
    nfr = 1 + (x.shape[0] - n_fft) // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(nfr)[:, None]
    frames = x[idx] * window
    return np.fft.rfft(frames, n_fft).T


def magnitude(signal: np.ndarray) -> np.ndarray:
    """|STFT| before any normalisation, (257, frames) float64."""
    return np.abs(stft_audfprint(signal))


def spectrogram(waveform: np.ndarray) -> np.ndarray:
    """training/visualisation.py:13-36 with amplitude=False.

    torch.stft(n_fft 512, hop 256, window hann(514)[1:-1] float64, center, reflect,
    onesided, unnormalised) -> abs -> divide by ONE max over the whole tensor
    (all clips of the batch share it, visualisation.py:29).  Output float64,
    shape (..., 257, 1 + T // 256).
    """
    w = np.asarray(waveform)
    lead = w.shape[:-1]
    flat = w.reshape(-1, w.shape[-1])
    out = np.stack([magnitude(row) for row in flat])
    out = out / np.max(out)
    return out.reshape(*lead, *out.shape[-2:])


def specgram_psd(samples: np.ndarray, n_fft: int = N_FFT, fs: float = 8000.0, noverlap: int = 256) -> np.ndarray:
    """matplotlib.mlab.specgram(..., window=mlab.window_hanning, noverlap)[0] restated.

    afp/dejavu/fingerprint.py:60-66.  No centring/padding: frames start at
    multiples of (n_fft - noverlap) while a full frame fits -> (len - noverlap)
    // step frames (249 for 64000 samples).  Window np.hanning(n_fft) (symmetric,
    zero end points), no detrend, one-sided PSD: |X|^2 / (fs * sum(w^2)), doubled
    for every bin except DC and Nyquist.
    """
    x = np.asarray(samples, dtype=np.float64)
    step = n_fft - noverlap
    nfr = (x.shape[0] - noverlap) // step
    idx = np.arange(n_fft)[:, None] + step * np.arange(nfr)[None, :]
    win = np.hanning(n_fft)
    frames = x[idx] * win[:, None]
    spec = np.fft.fft(frames, n=n_fft, axis=0)[: n_fft // 2 + 1]
    psd = np.conj(spec) * spec
    psd[1:-1] *= 2.0
    psd /= fs
    psd /= (np.abs(win) ** 2).sum()
    return psd.real
