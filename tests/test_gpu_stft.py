"""GPU parity: fused STFT-magnitude kernel vs the oracle (float64; tolerance = FFT rounding)."""
import numpy as np
import pytest
import torch

from musicfpaugment_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from musicfpaugment_amd import ops as _ops
    return _ops


def _wav(n_clips, n, seed=10):
    return synth.batch(n_clips, seed=seed, n=n)


@pytest.mark.parametrize("n", [8000, 64000, 4099, 257, 24000])
def test_stft_mag_matches_oracle(ops, n):
    from oracle import stft as ostft
    wav = _wav(3, n)
    mag, cmax = ops.stft_mag(torch.from_numpy(wav).cuda(), torch.float64)
    want = np.stack([ostft.magnitude(w) for w in wav])
    assert mag.shape == want.shape == (3, 257, 1 + n // 256)
    got = mag.cpu().numpy()
    scale = want.max()
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-13 * scale)
    np.testing.assert_array_equal(cmax.cpu().numpy(), got.reshape(3, -1).max(axis=1))


def test_stft_golden_and_float32_output(ops, golden):
    g = golden("g1_spectrogram")
    wav = synth.batch(2, seed=int(g["seed"]), n=int(g["n"]))
    x = torch.from_numpy(wav).cuda()
    mag, cmax = ops.stft_mag(x, torch.float64)
    spec = ops.normalize_(mag.clone(), cmax, per_clip=False)
    np.testing.assert_allclose(spec.cpu().numpy(), g["spectrogram"], rtol=0, atol=1e-12)
    assert float(spec.max()) == 1.0
    mag32, _ = ops.stft_mag(x, torch.float32)
    assert mag32.dtype == torch.float32
    np.testing.assert_array_equal(mag32.cpu().numpy(), mag.cpu().numpy().astype(np.float32))
    per = ops.normalize_(mag.clone(), cmax, per_clip=True)
    assert per.reshape(2, -1).max(dim=1).values.tolist() == [1.0, 1.0]
    np.testing.assert_array_equal(ops.f64_to_f32(spec).cpu().numpy(), spec.cpu().numpy().astype(np.float32))


def test_stft_rejects_bad_shapes(ops):
    with pytest.raises(ValueError):
        ops.stft_mag(torch.zeros(2, 256, device="cuda"))
    with pytest.raises(ValueError):
        ops.stft_mag(torch.zeros(2, 3, 8000, device="cuda"))
    from musicfpaugment_amd._lib import MfpaError
    with pytest.raises(MfpaError):
        ops.stft_mag(torch.zeros(2, 8000))          # CPU tensor: no fallback
    mag, cmax = ops.stft_mag(torch.zeros(0, 8000, device="cuda"))
    assert mag.shape == (0, 257, 32)


def test_specgram_psd_matches_oracle(ops):
    from oracle import stft as ostft
    wav = _wav(2, 64000, seed=20)
    psd, cmax = ops.specgram_psd(torch.from_numpy(wav).cuda(), scale_in=32767.0)
    want = np.stack([ostft.specgram_psd(w.astype(np.float64) * 32767.0) for w in wav])
    assert psd.shape == want.shape == (2, 257, 249)
    got = psd.cpu().numpy()
    # the library leaves out the constant 1/(Fs*sum(w^2)); compare after the reference's own /max
    got_n = got / got.reshape(2, -1).max(axis=1)[:, None, None]
    want_n = want / want.reshape(2, -1).max(axis=1)[:, None, None]
    np.testing.assert_allclose(got_n, want_n, rtol=0, atol=1e-13)
    np.testing.assert_array_equal(cmax.cpu().numpy(), got.reshape(2, -1).max(axis=1))
