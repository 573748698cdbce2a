"""GPU parity: MultiResolutionSTFTLoss (forward values) vs the oracle and the golden values of the real reference (g11)."""
import numpy as np
import pytest
import torch

from musicfpaugment_amd import synth

pytestmark = pytest.mark.gpu


def _signals(g):
    n = int(g["n"])
    x = synth.batch(3, seed=int(g["seed_x"]), n=n)
    y = (0.8 * x + 0.2 * synth.batch(3, seed=int(g["seed_noise"]), n=n, tonal=False)).astype(np.float32)
    return torch.from_numpy(x), torch.from_numpy(y)


def test_stft_magnitudes_every_resolution_vs_oracle(golden):
    from musicfpaugment_amd.training.loss import stft
    from oracle import loss as ol
    g = golden("g11_mrstft_loss")
    x, _ = _signals(g)
    for fs, hop, wl in ((1024, 120, 600), (2048, 240, 1200), (512, 50, 240)):        # hop 50: the even / odd frame split
        got = stft(x.cuda(), fs, hop, wl, torch.hann_window(wl)).cpu()
        want = ol.stft_mag(x, fs, hop, wl)
        assert got.shape == want.shape == (3, 1 + x.shape[1] // hop, fs // 2 + 1)
        err = float((got - want).abs().max() / want.abs().max())
        assert err < 2e-6, (fs, err)
    m0 = stft(x[:1].cuda(), 1024, 120, 600, torch.hann_window(600)).cpu().numpy()
    np.testing.assert_allclose(m0[0, ::7, ::9], g["mag0_sub"], rtol=0, atol=2e-6 * float(g["mag0_sub"].max()))
    with pytest.raises(NotImplementedError):
        stft(x.cuda(), 1024, 120, 600, torch.ones(600))


@pytest.mark.parametrize("precision,rtol", [(0, 2e-5), (1, 1e-4)])
def test_multi_resolution_loss_vs_reference_golden_and_oracle(golden, precision, rtol):
    """precision 1 = the DFT GEMMs with bf16x3 products (the Demucs training step's setting)."""
    from musicfpaugment_amd.training.loss import MultiResolutionSTFTLoss
    from oracle import loss as ol
    g = golden("g11_mrstft_loss")
    x, y = _signals(g)
    crit = MultiResolutionSTFTLoss(factor_sc=float(g["factor_sc"]), factor_mag=float(g["factor_mag"]), precision=precision).cuda()
    sc, mag = crit(x.cuda(), y.cuda())
    np.testing.assert_allclose([float(sc), float(mag)], [float(g["sc"]), float(g["mag"])], rtol=rtol)
    per = np.array([[float(a), float(b)] for a, b in (f(x.cuda(), y.cuda()) for f in crit.stft_losses)])
    np.testing.assert_allclose(per, g["per_resolution"], rtol=rtol)
    # a silent prediction: every magnitude sits on the 1e-7 clamp
    zs, zm = crit(torch.zeros(2, 8000, device="cuda"), y[:2, :8000].cuda())
    np.testing.assert_allclose([float(zs), float(zm)], [float(g["sc_silent"]), float(g["mag_silent"])], rtol=rtol)
    # full 8 s clips, the reference's training factors (training/parameters.py:29-30), against the oracle
    x8 = torch.from_numpy(synth.batch(4, seed=1500))
    y8 = torch.from_numpy((0.7 * synth.batch(4, seed=1500) + 0.3 * synth.batch(4, seed=1501, tonal=False)).astype(np.float32))
    crit5 = MultiResolutionSTFTLoss(factor_sc=0.5, factor_mag=0.5, precision=precision).cuda()
    sc8, mag8 = crit5(x8.cuda(), y8.cuda())
    wsc, wmag, _ = ol.multi_resolution_stft_loss(x8, y8, factor_sc=0.5, factor_mag=0.5)
    np.testing.assert_allclose([float(sc8), float(mag8)], [float(wsc), float(wmag)], rtol=rtol)
    # identical signals: both terms vanish
    s0, m0 = crit(x.cuda(), x.cuda())
    assert float(s0) == 0.0 and float(m0) == 0.0


def test_multi_resolution_loss_gradient_vs_oracle_autograd(golden):
    """d(sc + mag) / d(predicted waveform): device adjoint chain vs torch autograd through the oracle (CPU, float32)."""
    from musicfpaugment_amd.training.loss import MultiResolutionSTFTLoss
    from oracle import loss as ol
    g = golden("g11_mrstft_loss")
    x, y = _signals(g)
    for fsc, fmag in ((0.5, 0.5), (0.1, 0.0), (0.0, 0.1)):                            # training factors; each term alone
        crit = MultiResolutionSTFTLoss(factor_sc=fsc, factor_mag=fmag).cuda()
        sc, mag, dx = crit.value_and_grad(x.cuda(), y.cuda())
        xr = x.clone().requires_grad_()
        wsc, wmag, _ = ol.multi_resolution_stft_loss(xr, y, factor_sc=fsc, factor_mag=fmag)
        (wsc + wmag).backward()
        np.testing.assert_allclose([float(sc), float(mag)], [float(wsc), float(wmag)], rtol=2e-5, atol=1e-9)
        err = float((dx.cpu() - xr.grad).abs().sum() / xr.grad.abs().sum())
        assert err < 2e-4, (fsc, fmag, err)
    # hop 50 / odd frame counts / a short clip
    xs, ys = x[:2, :9000].contiguous(), y[:2, :9000].contiguous()
    crit = MultiResolutionSTFTLoss(factor_sc=0.5, factor_mag=0.5).cuda()
    _, _, dx = crit.value_and_grad(xs.cuda(), ys.cuda())
    xr = xs.clone().requires_grad_()
    a, b, _ = ol.multi_resolution_stft_loss(xr, ys, factor_sc=0.5, factor_mag=0.5)
    (a + b).backward()
    assert float((dx.cpu() - xr.grad).abs().sum() / xr.grad.abs().sum()) < 2e-4
