"""GPU parity: landmark pairing + hashing -- integer work, identical to the oracle / reference."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def unpack(bits, shape):
    return np.unpackbits(bits)[: int(np.prod(shape))].reshape(shape).astype(np.uint8)


def test_audfprint_landmarks_golden_and_random(golden):
    from musicfpaugment_amd import ops
    from oracle import hashes as oh
    g = golden("g8_hashes")
    masks = np.stack([unpack(g[f"aud_mask{i}"], (256, 251)) for i in range(2)])
    rng = np.random.default_rng(0)
    rnd = np.zeros((6, 256, 251), dtype=np.uint8)
    for b in range(6):                                   # up to 5 peaks per frame, like the pruner's output
        for c in range(251):
            k = rng.integers(0, 6)
            rnd[b, rng.choice(256, size=k, replace=False), c] = 1
    rnd[4] = 0                                           # empty clip
    rnd[5, :, 100:] = 0                                  # scols < T (peak_extractor.py:327)
    allm = np.concatenate([masks, rnd])
    lm, hs, uq, counts = ops.audfprint_landmarks(torch.from_numpy(allm).cuda(), cap=4096)
    lm, hs, uq, counts = lm.cpu().numpy(), hs.cpu().numpy(), uq.cpu().numpy(), counts.cpu().numpy()
    for b in range(len(allm)):
        cols, bins = np.nonzero(allm[b].T)
        want_lm = np.array(oh.peaks2landmarks(list(zip(cols.tolist(), bins.tolist()))), dtype=np.int32).reshape(-1, 4)
        want_hs = oh.landmarks2hashes(want_lm)
        want_uq = oh.unique_sorted_hashes(want_hs)
        assert counts[b].tolist() == [len(want_lm), len(want_uq)], b
        np.testing.assert_array_equal(lm[b, :len(want_lm)], want_lm)
        np.testing.assert_array_equal(hs[b, :len(want_lm)], want_hs)
        np.testing.assert_array_equal(uq[b, :len(want_uq)], want_uq)
    for i in range(2):                                   # the reference's own lists
        np.testing.assert_array_equal(lm[i, :counts[i, 0]], g[f"aud_landmarks{i}"])
        np.testing.assert_array_equal(uq[i, :counts[i, 1]], g[f"aud_unique{i}"])
    # capacity overflow is reported, not silently truncated
    _, _, _, c2 = ops.audfprint_landmarks(torch.from_numpy(allm[2:3]).cuda(), cap=16)
    assert c2.cpu().numpy().tolist() == [[-1, -1]]


def test_dejavu_hashes_golden_and_random(golden):
    from musicfpaugment_amd import ops
    from oracle import hashes as oh
    g = golden("g8_hashes")
    m0 = unpack(g["dej_mask"], (257, 249))
    rng = np.random.default_rng(1)
    m1 = (rng.random((257, 249)) < 0.002).astype(np.uint8)
    m2 = np.zeros((257, 249), dtype=np.uint8)
    m2[[0, 256, 9, 10, 100], [0, 0, 5, 248, 248]] = 1   # border peaks, one-digit and three-digit fields
    allm = np.stack([m0, m1, m2, np.zeros_like(m0)])
    dig, t1, counts = ops.dejavu_hashes(torch.from_numpy(allm).cuda())
    dig, t1, counts = dig.cpu().numpy(), t1.cpu().numpy(), counts.cpu().numpy()
    for b in range(len(allm)):
        want = oh.dejavu_hashes_from_mask(allm[b])
        assert counts[b] == len(want)
        got_hex = [bytes(dig[b, i]).hex() for i in range(counts[b])]
        assert got_hex == [h for h, _ in want]
        assert t1[b, :counts[b]].tolist() == [t for _, t in want]
    assert [bytes(dig[0, i]).hex() for i in range(counts[0])] == [str(x) for x in g["dej_hex"]]


def test_reference_call_surface_and_batch_paths(golden):
    """peaks2landmarks / landmarks2hashes / generate_hashes with the reference's list arguments, and the batched
    wav -> hashes paths against the oracle chain."""
    from musicfpaugment_amd import synth
    from musicfpaugment_amd.afp.audfprint.peak_extractor import Audfprint_peaks, landmarks2hashes
    from musicfpaugment_amd.afp.dejavu.fingerprint import fingerprint_batch, generate_hashes
    from oracle import audfprint as oa
    from oracle import dejavu as od
    from oracle import hashes as oh
    g = golden("g8_hashes")
    ext = Audfprint_peaks()
    mask0 = unpack(g["aud_mask0"], (256, 251))
    cols, bins = np.nonzero(mask0.T)
    lms = ext.peaks2landmarks(list(zip(cols.tolist(), bins.tolist())))
    assert np.array_equal(np.array(lms, dtype=np.int32), g["aud_landmarks0"])
    assert np.array_equal(landmarks2hashes(lms), g["aud_hashes0"])
    assert ext.peaks2landmarks([]) == [] and landmarks2hashes([]).shape == (0, 2)
    dmask = unpack(g["dej_mask"], (257, 249))
    f_idx, t_idx = np.nonzero(dmask)
    dh = generate_hashes(list(zip(f_idx.tolist(), t_idx.tolist())), fan_value=3)
    assert [h for h, _ in dh] == [str(x) for x in g["dej_hex"]] and [t for _, t in dh] == g["dej_t1"].tolist()
    wav = synth.batch(3, seed=59)
    uniq, counts = ext.hashes_batch(torch.from_numpy(wav).cuda())
    for b in range(3):
        want = oh.audfprint_hashes_from_mask(oa.find_peaks(wav[b])[1])
        assert int(counts[b]) == len(want)
        np.testing.assert_array_equal(uniq[b, :len(want)].cpu().numpy(), want)
    np.testing.assert_array_equal(uniq[0, :int(counts[0])].cpu().numpy(), g["aud_unique0"])    # clip seed 59 = golden clip 0
    dig, t1, cnt, dmask_b, _ = fingerprint_batch(torch.from_numpy(wav).cuda())
    for b in range(3):
        _, m, _ = od.fingerprint_peaks(wav[b].astype(np.float64) * 32767.0)
        want = oh.dejavu_hashes_from_mask(m)
        assert int(cnt[b]) == len(want)
        assert [bytes(dig[b, i].cpu().numpy()).hex() for i in range(len(want))] == [h for h, _ in want]


def test_hashes_with_part_frame_shifts_vs_oracle():
    """wavfile2hashes with shifts = 4 (peak_extractor.py:406-424, 437-460): the union over four part-frame shifts."""
    from musicfpaugment_amd import synth
    from musicfpaugment_amd.afp.audfprint.peak_extractor import Audfprint_peaks
    from musicfpaugment_amd.constants import afp_settings
    from oracle import audfprint as oa
    from oracle import hashes as oh
    params = dict(afp_settings["audfprint"], shifts=4)
    ext = Audfprint_peaks(params)
    wav = synth.batch(3, seed=2100, n=32000)
    uq, n = ext.hashes_batch(torch.from_numpy(wav).cuda())
    uq, n = uq.cpu().numpy(), n.cpu().numpy()
    for b in range(3):
        hs = []
        for s in range(4):
            mask = np.asarray(oa.find_peaks(wav[b, int(s / 4 * 256):])[1]).astype(np.uint8)
            cols, bins = np.nonzero(mask.T)
            hs.append(oh.landmarks2hashes(oh.peaks2landmarks(list(zip(cols.tolist(), bins.tolist())))))
        want = oh.unique_sorted_hashes(np.concatenate(hs))
        assert n[b] == len(want)
        np.testing.assert_array_equal(uq[b, :n[b]], want)
    # shifts = 1 keeps the single-pass path
    u1, n1 = Audfprint_peaks(None).hashes_batch(torch.from_numpy(wav).cuda())
    assert int(n1.min()) > 0 and bool((n1.cpu().numpy() <= n).all())


def test_hashes_batch_never_returns_an_overflowed_count():
    """The landmark kernel flags a clip that exceeds `cap` (or has more than 8 peaks in a frame) with counts [-1, -1]; the
    single-shift path of hashes_batch (the default, what wavfile2hashes uses) retries once with the kernel's largest capacity and
    otherwise raises -- it never hands a negative count on as a row count."""
    from musicfpaugment_amd import ops
    from musicfpaugment_amd.afp.audfprint.peak_extractor import Audfprint_peaks
    rng = np.random.default_rng(5)
    mask = np.zeros((2, 256, 251), dtype=np.uint8)
    for b in range(2):
        for t in range(251):
            mask[b, rng.choice(256, size=5, replace=False), t] = 1          # 5 peaks in every frame: ~3700 landmarks per clip
    dmask = torch.from_numpy(mask).cuda()
    _, _, uq_full, c_full = ops.audfprint_landmarks(dmask, 8192)
    assert int(c_full.min()) > 64
    _, _, _, c_small = ops.audfprint_landmarks(dmask, 64)
    assert c_small.cpu().tolist() == [[-1, -1], [-1, -1]]
    ext = Audfprint_peaks(None, device="cuda")
    ext.find_peaks_batch = lambda wav: (dmask, None, None)
    uq, n = ext.hashes_batch(torch.zeros((2, 64000), device="cuda"), cap=64)
    assert torch.equal(n.cpu(), c_full[:, 1].cpu()) and int(n.min()) > 0
    for b in range(2):
        assert torch.equal(uq[b, :int(n[b])].cpu(), uq_full[b, :int(n[b])].cpu())
    mask[0, :16, 100] = 1                                                     # 16+ peaks in one frame: beyond the kernel's limits
    dmask = torch.from_numpy(mask).cuda()
    ext.find_peaks_batch = lambda wav: (dmask, None, None)
    with pytest.raises(ValueError):
        ext.hashes_batch(torch.zeros((2, 64000), device="cuda"))


def test_dejavu_fingerprint_reference_signature_and_return_forms(golden):
    """fingerprint(channel_samples, Fs, wsize, n_hop, fan_value, amp_min, denoising, denoising_model, get_masks) --
    afp/dejavu/fingerprint.py:34-91: the default get_masks is the STRING "False" (only the hash list comes back), `get_masks=True`
    adds (peak_mask float64, specgram); peak set and normalised PSD against golden g4 = the real reference's own
    fingerprint(d * 32767, get_masks=True) on clip seed 61; the hash list against the oracle's pairing + SHA-1 of that mask."""
    from musicfpaugment_amd import synth
    from musicfpaugment_amd.afp.dejavu.fingerprint import fingerprint, set_denoisers
    from musicfpaugment_amd.training.unet import UNet
    from musicfpaugment_amd.training.weights import formula_state_dict
    from oracle import dejavu as od
    from oracle import hashes as oh
    g = golden("g4_dejavu_peaks")
    d = synth.clip(int(g["full_seed"]), tonal=True)
    samples = d.astype(np.float64) * 32767.0                     # what the golden run fed the reference (the device takes float32 samples)
    only = fingerprint(samples)                                   # positional default: a list of (hex, t1), no masks
    assert isinstance(only, list) and isinstance(only[0], tuple) and isinstance(only[0][0], str) and len(only[0][0]) == 20
    assert fingerprint(samples, get_masks="True") == only        # a truthy string is NOT `True` (reference quirk)
    hashes, mask, spec = fingerprint(list(samples), 8000, 512, 256, 3, 50, False, "unet", True)
    assert hashes == only and mask.dtype == np.float64 and tuple(mask.shape) == tuple(g["full_shape"])
    f_idx, t_idx = np.nonzero(mask)
    np.testing.assert_array_equal(np.stack([f_idx, t_idx], axis=1), g["full_coords"])
    np.testing.assert_allclose(spec[::8, ::8], g["full_spec_sub"], rtol=0, atol=1e-6)      # float32 rounding of d * 32767
    _, m32, s32 = od.fingerprint_peaks(samples.astype(np.float32).astype(np.float64))        # the oracle on exactly the device's samples
    np.testing.assert_array_equal(mask, m32.astype(np.float64))
    np.testing.assert_allclose(spec, s32, rtol=0, atol=1e-12)
    want = oh.dejavu_hashes_from_mask(mask.astype(np.uint8))
    assert [h for h, _ in hashes] == [h for h, _ in want] and [t for _, t in hashes] == [t for _, t in want]
    with pytest.raises(NotImplementedError):
        fingerprint(samples, wsize=1024)
    with pytest.raises(ValueError):
        fingerprint(samples, denoising=True, denoising_model="unet")          # no module handed over yet
    net = UNet(1, 1)
    net.load_state_dict(formula_state_dict(0))
    set_denoisers(unet=net.cuda().eval())
    h_dn, m_dn, s_dn = fingerprint(samples, denoising=True, denoising_model="unet", get_masks=True)
    assert s_dn.dtype == np.float32 and m_dn.shape == mask.shape and isinstance(h_dn, list)
    assert fingerprint(samples, denoising=True, denoising_model="demucs") == only   # denoised upstream by Dejavu, not here
