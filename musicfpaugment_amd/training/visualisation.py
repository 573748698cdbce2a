"""`spectrogram()` on MI355X -- mirror of the reference's training/visualisation.py:13-36.

STFT (n_fft 512, hop 256, window np.hanning(514)[1:-1], centre/reflect, one-sided), magnitude,
division by ONE maximum over the whole tensor (all clips share it, visualisation.py:29).  Returns
float64 like the reference (its float64 window promotes the STFT).  Runs the fused HIP kernel
(csrc/stft.hip); there is no CPU path.
"""
from __future__ import annotations

import torch

from .. import ops
from ..constants import WAVEFORM_SAMPLING_RATE  # noqa: F401  (re-exported like training/parameters.py:14)


def spectrogram(waveform: torch.Tensor, amplitude=False, device="cuda") -> torch.Tensor:
    if amplitude:
        raise NotImplementedError("amplitude=True only feeds librosa/matplotlib plotting in the reference "
                                  "(visualisation.py:31-34, :39-63); plotting is outside the hot path")
    if not isinstance(waveform, torch.Tensor):
        waveform = torch.as_tensor(waveform)
    dev = torch.device("cuda" if str(device) in ("cpu", "cuda") and not waveform.is_cuda else
                       (waveform.device if waveform.is_cuda else device))
    w = waveform.to(dev, dtype=torch.float32)
    lead = w.shape[:-1]
    flat = w.reshape(-1, w.shape[-1])
    mag, cmax = ops.stft_mag(flat, torch.float64)
    ops.normalize_(mag, cmax, per_clip=False)
    return mag.reshape(*lead, mag.shape[-2], mag.shape[-1])
