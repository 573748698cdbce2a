"""GPU parity: AugmentFP kernels vs the oracle (torch-CPU restatement of augmentation/transformations/*.py) and the
golden outputs of the real reference transforms (g10).  The julius-style filters are parity-unpinned (property tests)."""
import math

import numpy as np
import pytest
import torch

from musicfpaugment_amd import synth

pytestmark = pytest.mark.gpu


def _lib():
    from musicfpaugment_amd._lib import check, lib, ptr, stream
    return check, lib(), ptr, stream


def _ones(B):
    return torch.ones(B, dtype=torch.uint8, device="cuda")


def test_ir_noise_gain_clip_peak_against_reference_golden(golden):
    check, L, ptr, stream = _lib()
    g = golden("g10_augment")
    x = torch.from_numpy(synth.batch(3, seed=int(g["seed_x"]), n=int(g["n"]))).cuda()
    B, T = x.shape
    # impulse response
    ir = torch.from_numpy(g["ir"])[:, 0]
    n = ir.shape[1]
    taps = ir.flip(1).contiguous().cuda()
    nd = torch.full((B,), n, dtype=torch.int32, device="cuda")
    od = torch.full((B,), n - 1, dtype=torch.int32, device="cuda")
    y, peak = torch.empty_like(x), torch.empty(B, device="cuda")
    on = _ones(B)
    check(L.mfpa_fir(ptr(x), B, T, T + n - 1, ptr(taps), n, ptr(nd), ptr(od), ptr(on), 1, 2, ptr(y), ptr(peak), stream()), "fir")
    check(L.mfpa_scale_rows(ptr(y), B, T, ptr(peak), ptr(on), 1, ptr(y), stream()), "scale")
    np.testing.assert_allclose(y.cpu().numpy(), g["y_ir"][:, 0], rtol=0, atol=5e-6)
    # background noise
    from oracle import augment as oau
    noise = oau.rms_normalize(torch.from_numpy(synth.batch(3, seed=int(g["seed_noise"]), n=int(g["n"]), tonal=False))).cuda()
    y, snr, on = torch.empty_like(x), torch.from_numpy(g["snr"]).cuda(), _ones(B)      # keep every operand alive across the call
    check(L.mfpa_mix_background(ptr(x), B, T, ptr(noise), ptr(snr), ptr(on), ptr(y), stream()), "mix")
    np.testing.assert_allclose(y.cpu().numpy(), g["y_bg"][:, 0], rtol=0, atol=2e-6)
    # gain
    fac = (10 ** (torch.from_numpy(g["gain_db"]) / 20)).cuda()
    check(L.mfpa_scale_rows(ptr(x), B, T, ptr(fac), ptr(on), 0, ptr(y), stream()), "gain")
    np.testing.assert_array_equal(y.cpu().numpy(), g["y_gain"][:, 0])
    # clipping: exact quantiles (order statistics + torch's lerp)
    pct = torch.from_numpy(g["percentile"]).cuda()
    check(L.mfpa_clip_quantile(ptr(x), B, T, ptr(pct), ptr(on), ptr(y), stream()), "clip")
    np.testing.assert_array_equal(y.cpu().numpy(), g["y_clip"][:, 0])
    # peak normalisation (a silent clip is left alone)
    xs = x * torch.from_numpy(g["peak_scale"]).cuda()[:, None]
    check(L.mfpa_mix_background(ptr(xs), B, T, 0, 0, 0, ptr(y), stream()), "peak")
    np.testing.assert_array_equal(y.cpu().numpy(), g["y_peak"][:, 0])
    # gates off: copy through
    off = torch.zeros(B, dtype=torch.uint8, device="cuda")
    check(L.mfpa_clip_quantile(ptr(x), B, T, ptr(pct), ptr(off), ptr(y), stream()), "clip")
    assert torch.equal(y, x)


def test_clip_quantiles_many_sizes():
    check, L, ptr, stream = _lib()
    g = torch.Generator().manual_seed(0)
    for T in (2, 17, 1000, 64000):
        x = torch.randn(4, T, generator=g)
        x[1, : T // 2] = x[1, 0]                                       # ties
        pct = torch.tensor([0.0, 0.003, 0.01, 0.5])
        want = torch.stack([torch.clip(x[b], min=torch.quantile(x[b], pct[b] / 2), max=torch.quantile(x[b], 1 - pct[b] / 2)) for b in range(4)])
        y, xd, pd, on = torch.empty(4, T, device="cuda"), x.cuda(), pct.cuda(), _ones(4)
        check(L.mfpa_clip_quantile(ptr(xd), 4, T, ptr(pd), ptr(on), ptr(y), stream()), "clip")
        np.testing.assert_array_equal(y.cpu().numpy(), want.numpy())


def test_windowed_sinc_filters_vs_oracle_restatement():
    check, L, ptr, stream = _lib()
    from oracle import augment as oau
    x = torch.from_numpy(synth.batch(3, seed=1500, n=16000))
    cut = [150.0 / 8000, 3500.0 / 8000, 31.0 / 8000]
    half = [int(8 / c / 2) for c in cut]
    mt = 2 * max(half) + 1
    taps = torch.empty(3, mt, device="cuda")
    hd = torch.tensor(half, dtype=torch.int32, device="cuda")
    cd, on = torch.tensor(cut, device="cuda"), _ones(3)
    check(L.mfpa_lowpass_taps(ptr(cd), ptr(hd), 3, mt, ptr(taps), stream()), "taps")
    for b in range(3):
        want = oau.lowpass_taps(cut[b])
        np.testing.assert_allclose(taps[b, : 2 * half[b] + 1].cpu().numpy(), want.numpy(), rtol=0, atol=2e-7)
    nd = (2 * hd + 1).to(torch.int32)
    for mode, fn in ((0, oau.lowpass), (1, oau.highpass)):
        y, xd = torch.empty(3, 16000, device="cuda"), x.cuda()
        check(L.mfpa_fir(ptr(xd), 3, 16000, 16000, ptr(taps), mt, ptr(nd), ptr(hd), ptr(on), 0, mode, ptr(y), 0, stream()), "fir")
        want = torch.stack([fn(x[b:b + 1], cut[b])[0] for b in range(3)])
        np.testing.assert_allclose(y.cpu().numpy(), want.numpy(), rtol=0, atol=2e-5)


def test_augmentfp_call_surface_and_statistics():
    from musicfpaugment_amd.augmentation import AugmentFP, synthetic_banks
    irs, noises = synthetic_banks(0)
    torch.manual_seed(0)
    import random
    random.seed(0)
    af = AugmentFP(None, 8000, ir_bank=irs, noise_bank=noises)
    wav = torch.from_numpy(synth.batch(16, seed=1700, n=24000))[:, None, :]
    out = af.batch_augment(wav)
    assert out.shape == wav.shape and out.is_cuda
    assert torch.isfinite(out).all()
    np.testing.assert_allclose(out.abs().amax(dim=2).cpu().numpy(), 1.0, rtol=1e-6)       # PeakNormalization p = 1
    names = [t.name for t in af.augmentation_pipeline.transforms]
    assert names == ["HighPassFilter", "ApplyImpulseResponse", "AddBackgroundNoise", "Gain", "Clipping", "LowPassFilter",
                     "HighPassFilter", "PeakNormalization"]
    keys = [set(t.transform_parameters) for t in af.augmentation_pipeline.transforms]
    assert "cutoff_freq" in keys[0] and "ir" in keys[1] and {"background", "snr_in_db"} <= keys[2]
    assert "gain_factors" in keys[3] and "percentile_threshold" in keys[4] and all("should_apply" in k for k in keys)
    one = af(wav[0])
    assert one.shape == (1, 24000)
    # with every probability 0 only the final peak normalisation acts
    from musicfpaugment_amd.augmentation.constants import DEFAULT_PARAMETERS
    p0 = {k: (0.0 if k.startswith("proba") else v) for k, v in DEFAULT_PARAMETERS.items()}
    af0 = AugmentFP(None, 8000, parameters=p0, ir_bank=irs, noise_bank=noises)
    out0 = af0.batch_augment(wav)
    np.testing.assert_allclose(out0.cpu().numpy(), (wav / wav.abs().amax(dim=2, keepdim=True)).numpy(), rtol=0, atol=1e-7)
    with pytest.raises(NotImplementedError):
        AugmentFP({"scene": ["a.wav"]}, 8000)
