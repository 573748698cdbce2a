"""GPU parity: peak pickers and peak-mask metrics vs the oracle -- bit-exact index sets."""
import numpy as np
import pytest
import torch

from musicfpaugment_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from musicfpaugment_amd import ops as _ops
    return _ops


def unpack(bits, shape):
    return np.unpackbits(bits)[: int(np.prod(shape))].reshape(shape).astype(np.uint8)


def test_prune_strict_golden(ops, golden):
    """Reference's own filtered log-spectrogram in, reference's own mask out (known answer)."""
    g = golden("g3_audfprint_peaks")
    filt = g["short_filtered"]                                   # (256, 32) float64
    fm = torch.from_numpy(np.ascontiguousarray(filt.T))[None].cuda()   # frame-major (1, T, 256)
    mask, npk = ops.audfprint_prune(fm)
    want = unpack(g["short_mask"], filt.shape)
    np.testing.assert_array_equal(mask[0].cpu().numpy(), want)
    assert int(npk[0]) == int(want.sum()) == len(g["short_pklist"])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("order", ["C", "F"])
def test_prepare_strict_log_input_bit_exact(ops, dtype, order):
    """log_input=1: mean + high-pass on the device must equal numpy/scipy bit for bit."""
    from oracle import audfprint as oa
    rng = np.random.default_rng(5)
    lg = np.log(np.maximum(rng.random((3, 257, 251)), 1e-6)).astype(dtype)
    got = ops.audfprint_prepare(torch.from_numpy(lg).cuda(), mean_order=int(order == "F"), log_input=True)
    for b in range(3):
        flat = np.ascontiguousarray(lg[b] if order == "C" else lg[b].T).reshape(-1)
        mean = dtype(oa.numpy_sum(flat) / dtype(flat.size))
        want = oa.highpass(lg[b] - mean)[:-1]
        np.testing.assert_array_equal(got[b].cpu().numpy().T, want)


def _oracle_masks(sgrams, order):
    from oracle import audfprint as oa
    return np.stack([oa.peaks_from_filtered(oa.preprocess(s, order)) for s in sgrams]).astype(np.uint8)


def test_audfprint_full_clips_float64(ops, golden):
    """|STFT| -> normalise -> log/mean/high-pass -> prune, all on the device, vs the oracle and the goldens."""
    from oracle import stft as ostft
    g = golden("g3_audfprint_peaks")
    seeds = [int(s) for s in g["seeds"]] + list(range(700, 713))
    tonal = [bool(t) for t in g["tonal"]] + [k % 2 == 0 for k in range(13)]
    wav = np.stack([synth.clip(s, tonal=t) for s, t in zip(seeds, tonal)])
    mag, cmax = ops.stft_mag(torch.from_numpy(wav).cuda(), torch.float64)
    filtered = ops.audfprint_prepare(mag, denom=cmax, mean_order=1)
    mask, npk = ops.audfprint_prune(filtered)
    got = mask.cpu().numpy()
    for i in range(3):      # golden masks from the real reference
        np.testing.assert_array_equal(got[i], unpack(g[f"mask{i}"], (256, 251)))
    # same spectrogram (the device's) through the oracle: bit-exact index sets
    m = mag.cpu().numpy()
    sg = m / m.reshape(len(seeds), -1).max(axis=1)[:, None, None]
    want = _oracle_masks(sg, "F")
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(npk.cpu().numpy(), want.reshape(len(seeds), -1).sum(axis=1))
    # and against the oracle's own STFT
    want2 = np.stack([(lambda r: r[1])(__import__("oracle.audfprint", fromlist=["x"]).find_peaks(w)) for w in wav[:4]])
    np.testing.assert_array_equal(got[:4], want2.astype(np.uint8))


def test_audfprint_float32_spectrogram_path(ops, golden):
    """float32 (UNet-output-like) spectrograms: log and mean are float32 in the reference."""
    g = golden("g3b_audfprint_peaks_unet")
    spec = g["spec"]                                              # (257, 32) float32 from the real reference UNet
    rng = np.random.default_rng(11)
    extra = (rng.random((6, 257, 251)) ** 4).astype(np.float32)
    extra[1] -= 0.05                                              # UNet outputs can be negative (floored at max/1e6)
    filt = ops.audfprint_prepare(torch.from_numpy(spec[None]).cuda(), mean_order=0)
    mask, _ = ops.audfprint_prune(filt)
    np.testing.assert_array_equal(mask[0].cpu().numpy(), unpack(g["mask"], tuple(g["mask_shape"])))
    filt = ops.audfprint_prepare(torch.from_numpy(extra).cuda(), mean_order=0)
    mask, _ = ops.audfprint_prune(filt)
    want = _oracle_masks(extra, "C")
    mism = [(mask[b].cpu().numpy() != want[b]).sum() for b in range(len(extra))]
    assert sum(mism) == 0, f"float32 path peak flips per clip: {mism}"


def test_audfprint_edge_cases(ops):
    from oracle import audfprint as oa
    # all-zero clip: max <= 0 -> log/mean skipped (peak_extractor.py:272-280), ragged T, tiny T
    z = torch.zeros((1, 257, 40), dtype=torch.float64, device="cuda")
    mask, npk = ops.audfprint_prune(ops.audfprint_prepare(z))
    want = oa.peaks_from_filtered(oa.preprocess(np.zeros((257, 40))))
    np.testing.assert_array_equal(mask[0].cpu().numpy(), want.astype(np.uint8))
    rng = np.random.default_rng(2)
    for T in (1, 3, 9, 17, 130):
        s = rng.random((2, 257, T))
        mask, _ = ops.audfprint_prune(ops.audfprint_prepare(torch.from_numpy(s).cuda(), mean_order=0))
        np.testing.assert_array_equal(mask.cpu().numpy(), _oracle_masks(s, "C"))
    with pytest.raises(ValueError):
        ops.audfprint_prune(torch.zeros((1, 8, 255), dtype=torch.float64, device="cuda"))
    m, n = ops.audfprint_prune(torch.zeros((0, 8, 256), dtype=torch.float64, device="cuda"))
    assert m.shape == (0, 256, 8)


def test_dejavu_golden_and_oracle(ops, golden):
    from oracle import dejavu as od
    g = golden("g4_dejavu_peaks")
    arr = torch.from_numpy(g["arr"])[None].cuda()
    mask, npk = ops.localmax2d(arr, 10, 50.0)
    np.testing.assert_array_equal(mask[0].cpu().numpy(), g["mask"])
    assert int(npk[0]) == len(g["coords"])
    z = torch.zeros((1, 30, 30), dtype=torch.float64, device="cuda")
    mask, npk = ops.localmax2d(z, 10, -1.0)
    np.testing.assert_array_equal(mask[0].cpu().numpy(), g["zeros_mask"])
    # random arrays with ties / zeros, several shapes (tile edges, smaller than the window)
    rng = np.random.default_rng(3)
    for shape in [(257, 249), (64, 48), (33, 65), (7, 5), (21, 130)]:
        a = np.round(rng.normal(size=(2,) + shape) * 40)
        a[rng.random(a.shape) < 0.2] = 0.0
        mask, npk = ops.localmax2d(torch.from_numpy(a).cuda(), 10, 50.0)
        for b in range(2):
            coords, want = od.get_2d_peaks(a[b], 50)
            np.testing.assert_array_equal(mask[b].cpu().numpy(), want.astype(np.uint8))
            assert int(npk[b]) == len(coords)
        # other neighbourhood sizes run the kernel's run-time-radius instantiation (10 is compiled in)
        for radius in (3, 7, 16):
            mask, npk = ops.localmax2d(torch.from_numpy(a).cuda(), radius, 20.0)
            for b in range(2):
                coords, want = od.get_2d_peaks(a[b], 20, radius=radius)
                np.testing.assert_array_equal(mask[b].cpu().numpy(), want.astype(np.uint8))
                assert int(npk[b]) == len(coords)


def test_dejavu_full_pipeline(ops, golden):
    from oracle import dejavu as od
    g = golden("g4_dejavu_peaks")
    seeds = [int(g["full_seed"]), 62, 63, 64]
    wav = np.stack([synth.clip(s) for s in seeds])
    psd, cmax = ops.specgram_psd(torch.from_numpy(wav).cuda(), scale_in=32767.0)
    arr = ops.dejavu_prepare(psd, cmax, 10.0, mean_order=1)
    mask, npk = ops.localmax2d(arr, 10, 50.0)
    f_idx, t_idx = np.nonzero(mask[0].cpu().numpy())
    np.testing.assert_array_equal(np.stack([f_idx, t_idx], axis=1), g["full_coords"])
    for b, w in enumerate(wav):
        coords, want, _ = od.fingerprint_peaks(w.astype(np.float64) * 32767.0)
        np.testing.assert_array_equal(mask[b].cpu().numpy(), want.astype(np.uint8))


def test_dejavu_prepare_float32_branch_bit_exact(ops):
    """The denoised branch's arithmetic (fingerprint.py:74-79) on float32 network-like outputs, negatives and an all-equal
    clip included: x**2 -> float32 floor / log / mean -> picker; masks identical to numpy's, values within a float32 ulp."""
    from oracle import dejavu as od
    rng = np.random.default_rng(13)
    x = (rng.random((5, 257, 249)) ** 3).astype(np.float32)
    x[1] -= 0.3
    x[2] *= 1e-3
    x[3, 100:, :] = 0.0
    x[4] = 0.25
    xd = torch.from_numpy(x).cuda()
    arr = ops.dejavu_prepare_f32(xd, square=True, scale=10.0, mean_order=0)
    for amp in (50.0, 5.0):
        mask, npk = ops.localmax2d(arr, 10, amp)
        for b in range(len(x)):
            want_arr, spec = od.preprocess_denoised(x[b])
            assert want_arr.dtype == np.float32
            coords, want = od.get_2d_peaks(want_arr, amp)
            np.testing.assert_array_equal(mask[b].cpu().numpy(), want.astype(np.uint8))
            assert int(npk[b]) == len(coords)
    got = arr.cpu().numpy()
    for b in range(len(x)):
        want_arr, _ = od.preprocess_denoised(x[b])
        assert np.array_equal(got[b].astype(np.float32).astype(np.float64), got[b])          # float32 values, widened
        np.testing.assert_allclose(got[b], want_arr.astype(np.float64), rtol=0, atol=1e-4)     # 10 x a 1-ulp float32 log difference
    # odd shape, several 8192-element chunks plus a tail in the float32 pairwise mean
    y = (rng.random((2, 37, 531)) ** 2).astype(np.float32)
    arr = ops.dejavu_prepare_f32(torch.from_numpy(y).cuda())
    mask, _ = ops.localmax2d(arr, 10, 3.0)
    for b in range(2):
        _, want = od.get_2d_peaks(od.preprocess_denoised(y[b])[0], 3.0)
        np.testing.assert_array_equal(mask[b].cpu().numpy(), want.astype(np.uint8))
    with pytest.raises(ValueError):
        ops.dejavu_prepare_f32(xd.double())


@pytest.mark.parametrize("precision", [0, 1])
def test_dejavu_denoised_fingerprint_vs_reference(ops, golden, precision):
    """fingerprint(denoising=True, denoising_model="unet") against the golden run of the real reference (g13): specgram
    within the UNet tolerance (1e-4 relative L1), the amp_min=50 peak set identical, the dense amp_min=5 set to F1 >= 0.97
    (a peak within 1e-5 of a neighbour may flip under MFMA summation order), and -- given the device's own network
    output -- the picker bit-exact against numpy."""
    from oracle import dejavu as od
    from musicfpaugment_amd.afp.dejavu.fingerprint import fingerprint_peaks_batch
    from musicfpaugment_amd.training.unet import UNet
    from musicfpaugment_amd.training.weights import formula_state_dict
    g = golden("g13_dejavu_denoised")
    net = UNet(1, 1, rate=0.05)
    net.load_state_dict(formula_state_dict(0))
    net = net.cuda().eval()
    net.precision = precision
    seeds = [int(s) for s in g["seeds"]]
    wav = torch.from_numpy(np.stack([synth.clip(s, tonal=True) for s in seeds])).cuda()
    mask, npk, spec = fingerprint_peaks_batch(wav, denoising=True, denoising_model="unet", unet=net)
    assert spec.dtype == torch.float32 and tuple(mask.shape[1:]) == tuple(g["shape"])
    amp_low = float(g["amp_min_low"])
    mask_lo, _, _ = fingerprint_peaks_batch(wav, amp_min=amp_low, denoising=True, unet=net)
    y = net.denoise_spectrogram(*ops.specgram_psd(wav, scale_in=32767.0), per_clip=True).cpu().numpy()
    for i in range(len(seeds)):
        sub, want_sub = spec[i].cpu().numpy()[::8, ::8], g[f"spec_sub{i}"]
        assert np.abs(sub - want_sub).sum() / np.abs(want_sub).sum() <= 1e-4
        f_idx, t_idx = np.nonzero(mask[i].cpu().numpy())
        np.testing.assert_array_equal(np.stack([f_idx, t_idx], axis=1), g[f"coords{i}"])
        assert int(npk[i]) == len(g[f"coords{i}"])
        got = set(map(tuple, np.argwhere(mask_lo[i].cpu().numpy())))
        want = set(map(tuple, g[f"coords_low{i}"]))
        f1 = 2 * len(got & want) / (len(got) + len(want))
        assert f1 >= 0.97, f"clip {seeds[i]}: F1 {f1:.4f} ({len(got)} vs {len(want)} peaks)"
        _, want_own = od.get_2d_peaks(od.preprocess_denoised(y[i])[0], amp_low)
        np.testing.assert_array_equal(mask_lo[i].cpu().numpy(), want_own.astype(np.uint8))


def test_dejavu_demucs_branch_and_argument_checks(ops):
    """denoising_model="demucs" (dejavu.py:100-106): the waveform goes through Demucs before the x 32767 scaling."""
    from musicfpaugment_amd.afp.dejavu.fingerprint import fingerprint_batch, fingerprint_peaks_batch
    from musicfpaugment_amd.training.demucs_weights import formula_state_dict as demucs_formula
    from musicfpaugment_amd.training.model import Demucs
    dm = Demucs()
    dm.load_state_dict(demucs_formula(0))
    dm = dm.cuda().eval()
    wav = torch.from_numpy(synth.batch(3, seed=70, n=64000)).cuda()
    mask, npk, spec = fingerprint_peaks_batch(wav, denoising=True, denoising_model="demucs", demucs=dm)
    den = dm(wav)[:, 0].contiguous()
    mask2, npk2, spec2 = fingerprint_peaks_batch(den)
    assert torch.equal(mask, mask2) and torch.equal(npk, npk2) and torch.equal(spec, spec2) and spec.dtype == torch.float64
    dig, t1, counts, mask3, _ = fingerprint_batch(wav, denoising=True, denoising_model="demucs", demucs=dm)
    assert torch.equal(mask3, mask) and int(counts.min()) >= 0
    with pytest.raises(AssertionError):
        fingerprint_peaks_batch(wav, denoising=True, denoising_model="wavenet")
    with pytest.raises(ValueError):
        fingerprint_peaks_batch(wav, denoising=True, denoising_model="unet")


def test_peak_metrics(ops, golden):
    from oracle import metrics as om
    g = golden("g5_metrics")
    pred, gt = torch.from_numpy(g["pred"]).cuda(), torch.from_numpy(g["gt"]).cuda()
    c = ops.peak_metrics_counts(pred, gt).cpu().numpy()
    np.testing.assert_array_equal(c, om.counts(g["pred"], g["gt"]))
    for k in range(3):
        p = c[k, 0] / c[k, 1] if c[k, 1] else 0.0
        r = c[k, 2] / c[k, 3] if c[k, 3] else 0.0
        assert [p, r] == list(g["prf"][k][:2])
    rng = np.random.default_rng(9)
    a = (rng.random((5, 251, 256)) < 0.02).astype(np.uint8)
    b = (rng.random((5, 251, 256)) < 0.02).astype(np.uint8)
    b[:2] |= a[:2] & (rng.random((2, 251, 256)) < 0.7)
    a[:, 0, :] |= rng.random((5, 256)) < 0.1
    a[:, :, 0] |= rng.random((5, 251)) < 0.1
    c = ops.peak_metrics_counts(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()).cpu().numpy()
    np.testing.assert_array_equal(c, om.counts(a, b))


def test_peaks_metrics_harness_vs_oracle():
    """compute_peaks_metrics (testing/audfprint_exps.py:86-157) on synthetic queries: per-query P/R/F1 means and PSNR."""
    from musicfpaugment_amd.afp.audfprint.peak_extractor import Audfprint_peaks
    from musicfpaugment_amd.testing.audfprint_exps import compute_peaks_metrics
    from musicfpaugment_amd.training.unet import UNet
    from musicfpaugment_amd.training.weights import formula_state_dict
    from oracle import audfprint as oa
    from oracle import metrics as om
    n = 5
    clean = synth.batch(n, seed=800)
    aug = (0.8 * clean + 0.2 * synth.batch(n, seed=900, tonal=False)).astype(np.float32)
    net = UNet(1, 1, rate=0.05)
    net.load_state_dict(formula_state_dict(0))
    a0 = Audfprint_peaks()
    a1 = Audfprint_peaks(None, denoising=True, denoising_model="unet", unet=net)
    res = compute_peaks_metrics(torch.from_numpy(clean), torch.from_numpy(aug), a0, a1, batch=2)
    # oracle for the un-denoised half (exact), sanity for the denoised half
    P = R = F = 0.0
    for k in range(n):
        mc = oa.find_peaks(clean[k])[1].T[None]
        ma = oa.find_peaks(aug[k])[1].T[None]
        P += om.precision(ma, mc); R += om.recall(ma, mc); F += om.f1score(ma, mc)
    assert abs(res["precision_no_den"] - P / n) < 1e-12 and abs(res["recall_no_den"] - R / n) < 1e-12
    assert abs(res["f1_score_no_den"] - F / n) < 1e-12
    assert 0.0 <= res["prec_den"] <= 1.0 and 0.0 <= res["rec_den"] <= 1.0 and np.isfinite(res["psnr_den_spec"])
    from oracle import stft as ostft
    sc = np.stack([(lambda m: m / m.max())(ostft.magnitude(c)) for c in clean])
    sa = np.stack([(lambda m: m / m.max())(ostft.magnitude(c)) for c in aug])
    want_psnr = np.mean([10 * np.log10((sc[k].max() - sc[k].min()) ** 2 / np.mean((sa[k] - sc[k]) ** 2)) for k in range(n)])
    assert abs(res["psnr_no_den_spec"] - want_psnr) < 1e-9


def test_dejavu_peaks_metrics_harness_vs_oracle():
    """compute_peaks_metrics of testing/dejavu_exps.py:82-167 on synthetic queries: the un-denoised half and the denoised
    half given the device's network output are exact against the oracle; the reference's key set (with its psnr_*_wav
    entries filled from the spectrogram PSNR) is kept."""
    from musicfpaugment_amd import ops
    from musicfpaugment_amd.testing.dejavu_exps import DejavuPeaks, KEYS, compute_peaks_metrics
    from musicfpaugment_amd.training.unet import UNet
    from musicfpaugment_amd.training.weights import formula_state_dict
    from oracle import dejavu as od
    from oracle import metrics as om
    n = 5
    clean = synth.batch(n, seed=810)
    aug = (0.8 * clean + 0.2 * synth.batch(n, seed=910, tonal=False)).astype(np.float32)
    net = UNet(1, 1, rate=0.05)
    net.load_state_dict(formula_state_dict(0))
    net = net.cuda().eval()
    d0 = DejavuPeaks()
    d0.settings["amp_min"] = 5                   # the formula-weight network leaves few bins 50 above the mean
    d1 = DejavuPeaks(d0.settings, denoising=True, denoising_model="unet", unet=net)
    res = compute_peaks_metrics(torch.from_numpy(clean), torch.from_numpy(aug), d0, d1, batch=2)
    assert list(res.keys()) == KEYS
    y = net.denoise_spectrogram(*ops.specgram_psd(torch.from_numpy(aug).cuda(), scale_in=32767.0), per_clip=True).cpu().numpy()
    acc = np.zeros(8)
    for k in range(n):
        _, mc, sc = od.fingerprint_peaks(clean[k].astype(np.float64) * 32767.0, amp_min=5)
        _, ma, sa = od.fingerprint_peaks(aug[k].astype(np.float64) * 32767.0, amp_min=5)
        arr, sd_ = od.preprocess_denoised(y[k])
        _, md = od.get_2d_peaks(arr, 5)
        mc, ma, md = mc.T[None], ma.T[None], md.T[None]
        psnr = lambda a, b: 10 * np.log10((b.max() - b.min()) ** 2 / np.mean((a.astype(np.float64) - b) ** 2))
        acc += [om.precision(ma, mc), om.recall(ma, mc), om.f1score(ma, mc), psnr(sa, sc),
                om.precision(md, mc), om.recall(md, mc), om.f1score(md, mc), psnr(sd_, sc)]
    acc /= n
    for key, want in zip(["precision_no_den", "recall_no_den", "f1_score_no_den"], acc[:3]):
        assert abs(res[key] - want) < 1e-12, key
    for key, want in zip(["prec_den", "rec_den", "f1_den"], acc[4:7]):
        assert abs(res[key] - want) < 1e-12, key
    assert abs(res["psnr_no_den_spec"] - acc[3]) < 1e-9 and abs(res["psnr_den_spec"] - acc[7]) < 1e-5
    assert res["psnr_no_den_wav"] == res["psnr_no_den_spec"] and res["psnr_den_wav"] == res["psnr_den_spec"]
    with pytest.raises(AssertionError):
        DejavuPeaks(denoising=True, denoising_model="wavenet")


def test_file_level_entry_points(tmp_path):
    """wavfile2peaks / wavfile2hashes / compute_peaks_metrics on .pkl and .wav files (reference: peak_extractor.py:347-460,
    audfprint_exps.py:86-157) agree with the tensor paths."""
    import pickle
    from scipy.io import wavfile
    from musicfpaugment_amd.afp.audfprint.peak_extractor import Audfprint_peaks
    from musicfpaugment_amd.testing.audfprint_exps import compute_peaks_metrics, compute_peaks_metrics_files
    clean = synth.batch(3, seed=2300, n=16000)
    aug = (0.8 * clean + 0.2 * synth.batch(3, seed=2301, n=16000, tonal=False)).astype(np.float32)
    (tmp_path / "cleans").mkdir(); (tmp_path / "aug").mkdir()
    files = []
    for i in range(3):
        with open(tmp_path / "cleans" / f"q{i}.pkl", "wb") as fh:
            pickle.dump(clean[i], fh)
        with open(tmp_path / "aug" / f"q{i}.pkl", "wb") as fh:
            pickle.dump(aug[i], fh)
        files.append(str(tmp_path / "aug" / f"q{i}.pkl"))
    ext = Audfprint_peaks(None)
    pk = ext.wavfile2peaks(files[0])
    assert pk == ext.find_peaks(aug[0])[0] and len(pk) > 0
    mask, wav, sg = ext.wavfile2peaks(files[1], get_masks_waveforms=True)
    assert mask.shape == (256, 63) and sg.shape == (257, 63) and torch.equal(wav, torch.from_numpy(aug[1]))
    h = ext.wavfile2hashes(files[2])
    uq, n = ext.hashes_batch(torch.from_numpy(aug[2:3]).cuda())
    np.testing.assert_array_equal(h, uq[0, : int(n[0])].cpu().numpy())
    wavfile.write(tmp_path / "a.wav", 8000, (aug[0] * 32767).astype(np.int16))
    assert len(ext.wavfile2peaks(str(tmp_path / "a.wav"))) > 0
    assert len(ext.wavfile2peaks(files[0], shifts=4)) == 4
    got = compute_peaks_metrics_files(files, str(tmp_path / "cleans"), ext, ext)
    want = compute_peaks_metrics(torch.from_numpy(clean), torch.from_numpy(aug), ext, ext)
    assert got == want
    with pytest.raises(NotImplementedError):
        ext.wavfile2peaks("x.mp3")


def test_masks_only_path_equals_the_full_path_bit_for_bit(ops):
    """find_peaks_batch(want_spec=False) -- what the bench pipeline runs: the division by the clip maximum inside the log / high-pass
    kernel and the skipped max pass (the maximum of spec / max is 1, or NaN for a silent clip, without looking) -- gives the masks of the
    path that materialises the normalised spectrogram, including a silent clip, a one-sample clip and clips of several lengths."""
    from musicfpaugment_amd import synth
    from musicfpaugment_amd.afp.audfprint.peak_extractor import Audfprint_peaks
    ext = Audfprint_peaks(None, device="cuda")
    for n in (64000, 24000, 8000, 2049):
        wav = synth.batch(6, seed=4100 + n, n=n)
        wav[2] = 0.0                                           # silent: 0 / 0 = NaN spectrogram, no log step, no peaks
        wav[4] = 0.0
        wav[4, n // 2] = 0.75                                  # a single click
        x = torch.from_numpy(wav).cuda()
        m0, n0, spec = ext.find_peaks_batch(x)
        m1, n1, none = ext.find_peaks_batch(x, want_spec=False)
        assert none is None and spec is not None
        assert torch.equal(m0, m1) and torch.equal(n0, n1)
        assert int(n0[2]) == 0


def test_fused_pick_equals_prepare_plus_prune_bit_for_bit():
    """mfpa_audfprint_pick (log values frame-major + np.mean's node sums, then a pruner that applies "- mean, lfilter" to the frames as it
    walks them) against the two-stage path it replaces, mfpa_audfprint_prepare(denom = clip_max, mean_order = 1, log_input = 2) +
    mfpa_audfprint_prune: identical masks and counts, for full clips, short / ragged frame counts (partial pairwise-tree chunks, fewer
    than ten frames), tonal and noise clips, a silent and a half-silent clip; and against the oracle's find_peaks."""
    from musicfpaugment_amd import ops
    from oracle import audfprint as oa
    for n, B in ((64000, 12), (24000, 5), (2304, 3), (8000 + 77, 4), (130560, 2)):
        wav = np.stack([synth.clip(700 + i, n=n, tonal=(i % 3 != 1)) for i in range(B)])
        if B > 3:
            wav[1] = 0.0
            wav[2, : n // 2] = 0.0
        x = torch.from_numpy(wav).cuda()
        mag, cmax = ops.stft_mag(x, torch.float64)
        a_dec = ops.audfprint_a_dec()
        filt = ops.audfprint_prepare(mag, cmax, mean_order=1, denom_is_clip_max=True)
        want_mask, want_n = ops.audfprint_prune(filt, a_dec)
        got_mask, got_n = ops.audfprint_pick(mag, cmax, a_dec)
        assert torch.equal(got_mask, want_mask) and torch.equal(got_n, want_n), (n, B)
        for i in (0, B - 1):
            ref = oa.find_peaks(wav[i])[1]
            np.testing.assert_array_equal(got_mask[i].cpu().numpy(), np.asarray(ref).astype(np.uint8) if np.size(ref) else 0)


def test_dejavu_pick_equals_prepare_plus_localmax_bit_for_bit(ops):
    """mfpa_dejavu_pick (log values + np.mean's node sums by many workgroups per clip, the mean subtracted inside the local-maximum
    kernel) against the two stand-alone calls it replaces: identical masks and counts, on real clips, a silent clip, both summation
    orders and a second radius."""
    from musicfpaugment_amd import synth
    wav = torch.from_numpy(synth.batch(12, seed=77)).cuda()
    wav[5] = 0.0                                                     # a silent clip: NaN all the way, no peaks
    psd, cmax = ops.specgram_psd(wav, scale_in=32767.0)
    for mean_order in (1, 0):
        for radius, amp in ((10, 50.0), (10, 10.0), (6, 30.0)):
            arr = ops.dejavu_prepare(psd, cmax, 10.0, mean_order=mean_order)
            want_mask, want_n = ops.localmax2d(arr, radius, amp)
            mask, n = ops.dejavu_pick(psd, cmax, 10.0, mean_order, radius, amp)
            assert torch.equal(mask, want_mask)
            assert torch.equal(n, want_n)
    assert int(want_n[5]) == 0 and int(want_n.sum()) > 0
    # a shorter clip (a partial last tile, fewer chunks)
    psd, cmax = ops.specgram_psd(wav[:3, :20000].contiguous(), scale_in=32767.0)
    arr = ops.dejavu_prepare(psd, cmax, 10.0, mean_order=1)
    want_mask, want_n = ops.localmax2d(arr, 10, 20.0)
    mask, n = ops.dejavu_pick(psd, cmax, 10.0, 1, 10, 20.0)
    assert torch.equal(mask, want_mask) and torch.equal(n, want_n)


def test_division_by_a_correctly_rounded_reciprocal_is_the_ieee_division():
    """prep_div_fast (csrc/mfpa_prepsum.h): the pick kernels divide every spectrogram cell by the clip maximum as q0 = v * RN(1 / max) plus two
    residual / correction multiply-adds.  Against numpy's float64 division (IEEE, correctly rounded) on 2^24 random pairs over 600 binades and
    2^22 constructed NEAR-TIES (v = RN(den * midpoint of two adjacent doubles): the quotient lies within 2^-53 relative of a rounding boundary):
    bit-identical, every one."""
    from musicfpaugment_amd._lib import lib, check, ptr, stream
    rng = np.random.default_rng(20260105)
    n1, n2 = 1 << 24, 1 << 22
    den = np.ldexp(1.0 + rng.random(n1 + n2), rng.integers(-300, 300, n1 + n2))
    v = np.empty(n1 + n2)
    v[:n1] = den[:n1] * rng.random(n1) * np.ldexp(1.0, rng.integers(-20, 1, n1))          # 0 <= v <= den, down to 1e-6 of it
    m = rng.integers(1 << 52, 1 << 53, n2).astype(np.float64)                               # a 53-bit significand ...
    # ... and the midpoint above it, (m + 0.5) 2^-53 (54 bits: not a double), times den, rounded once more: v = RN(den m 2^-53 + den 2^-54)
    v[n1:] = np.ldexp(den[n1:] * m, -53) + np.ldexp(den[n1:], -54)
    want = v / den
    vd, dd = torch.from_numpy(v).cuda(), torch.from_numpy(den).cuda()
    out = torch.empty_like(vd)
    check(lib().mfpa_div_by_reciprocal(ptr(vd), ptr(dd), v.size, ptr(out), stream()), "mfpa_div_by_reciprocal")
    got = out.cpu().numpy()
    bad = np.nonzero(got.view(np.int64) != want.view(np.int64))[0]
    assert bad.size == 0, (bad.size, v[bad[:4]], den[bad[:4]], got[bad[:4]], want[bad[:4]])


def test_pick_log_values_equal_the_ieee_division_build_bit_for_bit(ops):
    """The masks-only Audfprint pick on 64 synthetic clips: its per-cell quotients go through prep_div_fast; the (separate) prepare path of the
    same spectrograms divides with the IEEE division in mfpa_normalize first.  Same masks, bit for bit (the log values themselves are compared
    by test_fused_pick_equals_prepare_plus_prune_bit_for_bit through the pruner's decisions on every frame)."""
    wav = torch.from_numpy(np.stack([synth.clip(8800 + i, tonal=(i % 2 == 0)) for i in range(64)])).cuda()
    mag, cmax = ops.stft_mag(wav, torch.float64)
    a_dec = ops.audfprint_a_dec()
    m1, n1_ = ops.audfprint_pick(mag, cmax, a_dec)
    spec = ops.normalize_(mag.clone(), cmax, per_clip=True)                                # IEEE division per element
    filt = ops.audfprint_prepare(spec, None, mean_order=1)
    m2, n2_ = ops.audfprint_prune(filt, a_dec)
    assert torch.equal(m1, m2) and torch.equal(n1_, n2_)
