#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (deezer/musicFPaugment).

Build-container only: needs /root/reference (read-only) and is never run on the
GPU box.  The reference is imported unmodified with import-time stubs for the
packages it imports but this path never calls (librosa, torchaudio, julius,
nnAudio, torchmetrics, GPUtil/tensorflow via training.utils) and with
``torch.load`` patched during import so the module-level checkpoint loads in
afp/audfprint/peak_extractor.py:24-37 and afp/dejavu/fingerprint.py:27-31 get a
formula state_dict instead of the unpublished checkpoint.  Nothing of the
reference is copied: only numeric inputs/outputs are written.

Usage:  python tools/make_goldens.py            (writes tests/golden/)
"""
from __future__ import annotations

import os
import sys
import types
import warnings

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

from musicfpaugment_amd import synth  # noqa: E402
from musicfpaugment_amd.training.weights import formula_state_dict  # noqa: E402


def import_reference():
    import torch

    sys.path.insert(0, REF)
    sys.path.insert(1, os.path.join(REF, "afp"))

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    stub("librosa")
    stub("librosa.display")
    sys.modules["librosa"].display = sys.modules["librosa.display"]

    class _Resample:
        def __init__(self, *a, **k):
            pass

        def __call__(self, x):
            return x

    ta = stub("torchaudio")
    ta.transforms = stub("torchaudio.transforms", Resample=_Resample)
    stub("julius")
    stub("nnAudio")
    stub("nnAudio.features")

    class _PSNR:
        def __init__(self, **k):
            pass

    stub("torchmetrics", PeakSignalNoiseRatio=_PSNR)
    import training  # noqa: F401  (reference namespace package)

    stub("training.utils", set_gpus=lambda *a, **k: "cpu")
    from training.model import Demucs
    from training.unet import UNet

    unet_sd = formula_state_dict(0)
    demucs_sd = Demucs().state_dict()
    orig = torch.load
    torch.load = lambda path, *a, **k: {"model_state_dict": unet_sd if "unet" in path else demucs_sd}
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            import afp.audfprint.peak_extractor as pe
            import dejavu.fingerprint as fp
    finally:
        torch.load = orig
    import testing.metrics as tm
    from testing.parameters import afp_settings
    from training.visualisation import spectrogram

    return dict(pe=pe, fp=fp, tm=tm, spectrogram=spectrogram, UNet=UNet, afp_settings=afp_settings)


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def g12_demucs_train_step(versions):
    """G12: one training step of the audio branch (training/train.py:275-312) with the REAL reference: Demucs (train mode,
    formula weights) -> L1 + MultiResolutionSTFTLoss -> backward -> Adam(lr 1e-3, betas (0.9, 0.999)) (train.py:661)."""
    import torch
    import training.loss as rloss
    from training.model import Demucs
    from musicfpaugment_amd.training.demucs_weights import formula_state_dict as demucs_formula

    real_stft = torch.stft

    def stft_shim(x, n_fft, hop_length=None, win_length=None, window=None, **kw):
        return torch.view_as_real(real_stft(x, n_fft, hop_length, win_length, window, return_complex=True, **kw))

    n, seed_clean, seed_noise = 4000, 1500, 1501
    clean = torch.from_numpy(synth.batch(2, seed=seed_clean, n=n))
    aug = (clean + 0.05 * torch.from_numpy(synth.batch(2, seed=seed_noise, n=n))).float()
    dm = Demucs()
    dm.load_state_dict(demucs_formula(0))
    dm.train()
    opt = torch.optim.Adam(dm.parameters(), lr=1e-3, betas=(0.9, 0.999))
    torch.stft = stft_shim
    try:
        crit_l1 = torch.nn.L1Loss()
        crit = rloss.MultiResolutionSTFTLoss(factor_sc=0.1, factor_mag=0.1)
        pred = dm(aug.unsqueeze(1)).squeeze(1)                                     # train.py:262,276
        l1 = crit_l1(pred, clean)
        sc, mag = crit(pred, clean)
        loss = l1 + sc + mag
        opt.zero_grad()
        loss.backward()
    finally:
        torch.stft = real_stft
    def head(t):                                       # the first 8 entries (zero-padded: the last bias has one)
        out = np.zeros(8, dtype=np.float32)
        v = t.detach().reshape(-1)[:8].numpy()
        out[:v.size] = v
        return out

    names = [k for k, _ in dm.named_parameters()]
    gnorm = np.array([float(p.grad.double().norm()) for _, p in dm.named_parameters()])
    ghead = np.stack([head(p.grad) for _, p in dm.named_parameters()])
    before = np.stack([head(p) for _, p in dm.named_parameters()])
    opt.step()
    after = np.stack([head(p) for _, p in dm.named_parameters()])
    save("g12_demucs_train_step", weight_seed=0, n=n, seed_clean=seed_clean, seed_noise=seed_noise, noise_gain=0.05, lr=1e-3,
         l1=float(l1), sc=float(sc), mag=float(mag), pred_sub=pred.detach().numpy()[:, ::8].copy(), names=np.array(names),
         grad_norm=gnorm, grad_head=ghead, param_head_before=before, param_head_after=after, versions=versions)


AMP_MIN_LOW = 5


def g13_dejavu_denoised(fp, versions):
    """G13: Dejavu's fingerprint() with denoising=True, denoising_model="unet" (afp/dejavu/fingerprint.py:58-84) on full
    8-second clips; fp.unet holds the formula weights (seed 0) injected at import."""
    out = {}
    seeds = [61, 2061]
    for i, s_ in enumerate(seeds):
        d = synth.clip(s_, tonal=True)
        _, mask, spec = fp.fingerprint((d.astype(np.float64) * 32767.0), denoising=True, denoising_model="unet",
                                       get_masks=True)
        assert spec.dtype == np.float32
        f_idx, t_idx = np.nonzero(mask)
        out[f"wav_digest{i}"] = synth.digest(d)
        out[f"coords{i}"] = np.stack([f_idx, t_idx], axis=1).astype(np.int32)
        out[f"spec_sub{i}"] = spec[::8, ::8].copy()
        # the formula weights leave few bins 50 above the mean: a second run with a low threshold exercises the picker
        _, mask_lo, _ = fp.fingerprint((d.astype(np.float64) * 32767.0), denoising=True, denoising_model="unet",
                                       amp_min=AMP_MIN_LOW, get_masks=True)
        fl, tl = np.nonzero(mask_lo)
        out[f"coords_low{i}"] = np.stack([fl, tl], axis=1).astype(np.int32)
        print(f"  dejavu + unet, clip {s_}: {len(f_idx)} peaks (amp_min 50), {len(fl)} (amp_min {AMP_MIN_LOW}), "
              f"specgram max {spec.max():.4g}")
    save("g13_dejavu_denoised", amp_min_low=AMP_MIN_LOW, seeds=np.array(seeds), shape=np.array(mask.shape), versions=versions, **out)


def main():
    import scipy.signal
    import torch

    os.makedirs(OUT, exist_ok=True)
    ref = import_reference()
    pe, fp, tm = ref["pe"], ref["fp"], ref["tm"]
    torch.manual_seed(0)
    torch.set_num_threads(4)
    versions = np.array([f"torch {torch.__version__}", f"numpy {np.__version__}",
                         f"scipy {__import__('scipy').__version__}",
                         f"matplotlib {__import__('matplotlib').__version__}"])
    if "--only-g12" in sys.argv:
        g12_demucs_train_step(versions)
        return
    if "--only-g13" in sys.argv:
        g13_dejavu_denoised(fp, versions)
        return

    # ---- G1 / G2: spectrogram() and audfprint stft on two 1-second clips -------------------
    wav = synth.batch(2, seed=1000, n=8000, tonal=True)
    spec = ref["spectrogram"](torch.from_numpy(wav)).numpy()
    assert spec.dtype == np.float64 and spec.shape == (2, 257, 32)
    win = np.hanning(514)[1:-1]
    import afp.audfprint.stft as rstft

    cplx = rstft.stft(wav[0], n_fft=512, hop_length=256, window=win)
    save("g1_spectrogram", seed=1000, n=8000, wav_digest=synth.digest(wav), spectrogram=spec,
         stft0=cplx, versions=versions)

    # ---- G3: find_peaks without denoising on full 8-second clips ---------------------------
    analyzer = pe.Audfprint_peaks(ref["afp_settings"]["audfprint"])
    seeds = [59, 60, 2059]
    tonal = [True, True, False]
    g3 = {}
    for i, (s, t) in enumerate(zip(seeds, tonal)):
        d = synth.clip(s, tonal=t)
        pklist, mask, spec_i = analyzer.find_peaks(d)
        # intermediate stages with the reference's own calls (peak_extractor.py:259-301)
        sg = np.abs(rstft.stft(d, n_fft=512, hop_length=256, window=win))
        sg /= np.max(sg)
        lg = np.log(np.maximum(sg, np.max(sg) / 1e6))
        lg = lg - np.mean(lg)
        filt = np.array([scipy.signal.lfilter([1, -1], [1, -(0.98 ** 1)], r) for r in lg])[:-1, ]
        adec = 1 - 0.01 * (analyzer.density * np.sqrt(analyzer.n_hop / 352.8) / 35)
        fwd = analyzer._decaying_threshold_fwd_prune(filt, adec)
        g3[f"pklist{i}"] = np.array(pklist, dtype=np.int32).reshape(-1, 2)
        g3[f"mask{i}"] = np.packbits(mask.astype(bool))
        g3[f"fwdmask{i}"] = np.packbits(fwd.astype(bool))
        g3[f"spec_sub{i}"] = spec_i[::8, ::8].copy()
        g3[f"filt_sub{i}"] = filt[::8, ::8].copy()
        g3[f"wav_digest{i}"] = synth.digest(d)
        g3[f"a_dec{i}"] = adec
        print(f"  clip seed {s} tonal {t}: {len(pklist)} peaks (fwd {int(fwd.sum())})")
    # a short clip whose filtered log-spectrogram ships in full (strict-entry known answer)
    d = synth.clip(77, n=8000)
    pklist, mask, _ = analyzer.find_peaks(d)
    sg = np.abs(rstft.stft(d, n_fft=512, hop_length=256, window=win))
    sg /= np.max(sg)
    lg = np.log(np.maximum(sg, np.max(sg) / 1e6))
    lg = lg - np.mean(lg)
    filt = np.array([scipy.signal.lfilter([1, -1], [1, -(0.98 ** 1)], r) for r in lg])[:-1, ]
    g3.update(short_sgram=sg, short_filtered=filt, short_pklist=np.array(pklist, dtype=np.int32).reshape(-1, 2),
              short_mask=np.packbits(mask.astype(bool)), short_seed=77)
    e_pk, e_mask = analyzer.find_peaks(np.zeros(0, dtype=np.float32))
    assert e_pk == [] and e_mask.size == 0
    save("g3_audfprint_peaks", seeds=np.array(seeds), tonal=np.array(tonal), mask_shape=np.array([256, 251]),
         versions=versions, **g3)

    # ---- G3b: find_peaks WITH the UNet (float32 spectrogram path) on a 1-second clip ------
    analyzer_dn = pe.Audfprint_peaks(ref["afp_settings"]["audfprint"], denoising=True, denoising_model="unet")
    d = synth.clip(4242, n=8000)
    pklist, mask, spec_dn = analyzer_dn.find_peaks(d)
    assert spec_dn.dtype == np.float32
    save("g3b_audfprint_peaks_unet", seed=4242, n=8000, wav_digest=synth.digest(d), unet_seed=0,
         spec=spec_dn, pklist=np.array(pklist, dtype=np.int32).reshape(-1, 2),
         mask=np.packbits(mask.astype(bool)), mask_shape=np.array(mask.shape), versions=versions)

    # ---- G4: get_2D_peaks ---------------------------------------------------------------------
    rng = np.random.default_rng(4)
    a = rng.normal(0.0, 30.0, size=(64, 48))
    a[5:9, 7:12] = 80.0                      # plateau: every cell of it ties with the max
    a[0, 0] = 120.0                          # corner peak
    a[63, 20] = 95.0                         # border peak
    a[30, 47] = 90.0
    a[40:64, 24:48] = 0.0                    # exact-zero background (erosion term)
    a[50, 36] = 70.0
    coords, mask = fp.get_2D_peaks(a.copy(), plot=False, amp_min=50)
    b = np.zeros((30, 30))                   # all-zero array: eroded background cancels local maxima
    coords_b, mask_b = fp.get_2D_peaks(b.copy(), plot=False, amp_min=-1)
    d = synth.clip(61, tonal=True)
    _, mask_full, specgram_full = fp.fingerprint((d.astype(np.float64) * 32767.0), get_masks=True)
    f_idx, t_idx = np.nonzero(mask_full)
    save("g4_dejavu_peaks", arr=a, coords=np.array(coords, dtype=np.int32).reshape(-1, 2), mask=mask.astype(np.uint8),
         zeros_coords=np.array(coords_b, dtype=np.int32).reshape(-1, 2), zeros_mask=mask_b.astype(np.uint8),
         full_seed=61, full_wav_digest=synth.digest(d), full_shape=np.array(mask_full.shape),
         full_coords=np.stack([f_idx, t_idx], axis=1).astype(np.int32), full_spec_sub=specgram_full[::8, ::8].copy(),
         versions=versions)
    print(f"  dejavu: {len(coords)} peaks on the 64x48 case, {len(f_idx)} on the full clip")

    # ---- G5: Precision / Recall / F1 -----------------------------------------------------------
    rng = np.random.default_rng(5)
    pred = (rng.random((3, 40, 30)) < 0.06).astype(np.float32)
    gt = (rng.random((3, 40, 30)) < 0.06).astype(np.float32)
    gt[0] = np.where(rng.random((40, 30)) < 0.5, pred[0], gt[0])
    for m in (pred, gt):
        m[0, 0, 0] = 1
        m[0, 0, 7] = 1
        m[0, 9, 0] = 1
        m[0, 39, 29] = 1
        m[0, 39, 3] = 1
        m[0, 12, 29] = 1
    pred[0, 1, 1] = 1
    gt[0, 1, 8] = 1
    pred[2] = 0                               # empty prediction -> precision 0, recall 0
    res = []
    for k in range(3):
        p_, g_ = torch.from_numpy(pred[k:k + 1]), torch.from_numpy(gt[k:k + 1])
        res.append([tm.Precision()(p_, g_), tm.Recall()(p_, g_), tm.F1score()(p_, g_)])
    p_, g_ = torch.from_numpy(pred), torch.from_numpy(gt)
    res.append([tm.Precision()(p_, g_), tm.Recall()(p_, g_), tm.F1score()(p_, g_)])
    save("g5_metrics", pred=pred.astype(np.uint8), gt=gt.astype(np.uint8), prf=np.array(res, dtype=np.float64),
         versions=versions)

    # ---- G6: UNet eval forward with formula weights -------------------------------------------
    net = ref["UNet"](1, 1, rate=0.05)
    net.load_state_dict(formula_state_dict(0))
    net.eval()
    wav = synth.batch(2, seed=300, n=8000)
    x = ref["spectrogram"](torch.from_numpy(wav)).float().unsqueeze(1)       # (2,1,257,32)
    with torch.no_grad():
        y = net(x)
    wav8 = synth.batch(1, seed=301)
    x8 = ref["spectrogram"](torch.from_numpy(wav8)).float().unsqueeze(1)      # (1,1,257,251)
    with torch.no_grad():
        y8 = net(x8)
    save("g6_unet_forward", weight_seed=0, x=x.numpy(), y=y.numpy(), seed8=301, x8_digest=synth.digest(wav8),
         y8_sub=y8.numpy()[0, 0, ::4, ::4].copy(), y8_abs_sum=float(y8.double().abs().sum()),
         y8_sum=float(y8.double().sum()), versions=versions)

    # ---- G7: one UNet training step (train.py:257-317, spec branch), dropout 0 ---------------
    net = ref["UNet"](1, 1, rate=0.0)
    net.load_state_dict(formula_state_dict(1))
    net.train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, betas=(0.9, 0.999))
    clean = synth.batch(2, seed=500, n=8000)
    aug = (0.7 * clean + 0.3 * synth.batch(2, seed=600, n=8000, tonal=False)).astype(np.float32)
    clean_spec = ref["spectrogram"](torch.from_numpy(clean))
    aug_spec = ref["spectrogram"](torch.from_numpy(aug))
    pred = net(aug_spec.unsqueeze(1).float()).squeeze(1)
    loss = torch.nn.L1Loss()(pred, clean_spec)
    opt.zero_grad()
    loss.backward()
    names = [n for n, _ in net.named_parameters()]
    gnorm = np.array([float(p.grad.double().norm()) for _, p in net.named_parameters()])
    ghead = np.stack([p.grad.flatten()[:4].double().numpy() for _, p in net.named_parameters()
                      if p.numel() >= 4])
    opt.step()
    whead = np.stack([p.detach().flatten()[:4].double().numpy() for _, p in net.named_parameters() if p.numel() >= 4])
    sd = net.state_dict()
    rm = np.concatenate([sd[k].numpy()[:4] for k in sd if k.endswith("running_mean")])
    rv = np.concatenate([sd[k].numpy()[:4] for k in sd if k.endswith("running_var")])
    save("g7_unet_train_step", weight_seed=1, clean_seed=500, noise_seed=600, n=8000, loss=float(loss),
         loss_dtype=str(loss.dtype), names=np.array(names), grad_norm=gnorm, grad_head=ghead, weight_head=whead,
         running_mean_head=rm, running_var_head=rv, pred_sub=pred.detach().numpy()[:, ::4, ::4].copy(),
         versions=versions)
    # ---- G8: landmarks / hashes ---------------------------------------------------
    g8 = {}
    for i, (s_, t_) in enumerate([(59, True), (2059, False)]):
        d = synth.clip(s_, tonal=t_)
        pklist, mask, _ = analyzer.find_peaks(d)
        lms = analyzer.peaks2landmarks(pklist)
        hashes = pe.landmarks2hashes(lms)
        merged = (hashes[:, 0].astype(np.uint64) << np.uint64(32)) + hashes[:, 1].astype(np.uint64)   # peak_extractor.py:447-459
        u = np.sort(np.unique(merged))
        uniq = np.hstack([(u >> np.uint64(32))[:, np.newaxis], (u & np.uint64((1 << 32) - 1))[:, np.newaxis]]).astype(np.int32)
        g8[f"aud_mask{i}"] = np.packbits(mask.astype(bool))
        g8[f"aud_landmarks{i}"] = np.array(lms, dtype=np.int32).reshape(-1, 4)
        g8[f"aud_hashes{i}"] = hashes
        g8[f"aud_unique{i}"] = uniq
        print(f"  audfprint clip {s_}: {len(pklist)} peaks -> {len(lms)} landmarks -> {len(uniq)} unique hashes")
    rng = np.random.default_rng(8)
    dmask = (rng.random((257, 249)) < 0.004)
    f_idx, t_idx = np.nonzero(dmask)
    dh = fp.generate_hashes(list(zip(f_idx.tolist(), t_idx.tolist())), fan_value=3)
    g8["dej_mask"] = np.packbits(dmask)
    g8["dej_hex"] = np.array([h for h, _ in dh])
    g8["dej_t1"] = np.array([int(t) for _, t in dh], dtype=np.int32)
    print(f"  dejavu: {int(dmask.sum())} peaks -> {len(dh)} hashes")
    save("g8_hashes", versions=versions, **g8)

    # ---- G9: Demucs forward (training/model.py:290-326) with formula weights ----------------------
    from training.model import Demucs
    from musicfpaugment_amd.training.demucs_weights import formula_state_dict as demucs_formula
    dm = Demucs()
    dm.load_state_dict(demucs_formula(0))
    dm.eval()
    w1 = synth.batch(2, seed=70, n=8000)
    w8 = synth.batch(1, seed=71)
    with torch.no_grad():
        y1 = dm(torch.from_numpy(w1)).numpy()
        y8 = dm(torch.from_numpy(w8)).numpy()
    save("g9_demucs_forward", weight_seed=0, seed1=70, n1=8000, wav1_digest=synth.digest(w1), y1=y1, seed8=71,
         wav8_digest=synth.digest(w8), y8_sub=y8[0, 0, ::16].copy(), y8_abs_sum=float(np.abs(y8.astype(np.float64)).sum()),
         valid_length_64000=dm.valid_length(64000), valid_length_8000=dm.valid_length(8000), versions=versions)

    # ---- G10: AugmentFP transforms with explicit parameters (augmentation/transformations/*.py) --------------
    # The transform classes are created without __init__ (which needs audio files) and fed their parameters directly.
    from augmentation.transformations.background_noise import AddBackgroundNoise
    from augmentation.transformations.clipping import Clipping
    from augmentation.transformations.gain import Gain
    from augmentation.transformations.impulse_response import ApplyImpulseResponse, convolve
    from augmentation.transformations.peak_normalization import PeakNormalization
    from augmentation.utils import Audio

    def bare(cls, **attrs):
        obj = object.__new__(cls)
        torch.nn.Module.__init__(obj)
        obj.__dict__.update(attrs)
        return obj

    x = torch.from_numpy(synth.batch(3, seed=1200, n=8000))[:, None, :]                       # (3,1,8000)
    rng = np.random.default_rng(10)
    ir = torch.zeros(3, 1, 700)
    for b, L_ in enumerate((700, 431, 96)):
        e = np.exp(-np.arange(L_) / (0.15 * L_)) * rng.normal(size=L_)
        e[3 * b] += 2.0
        ir[b, 0, :L_] = torch.from_numpy(e.astype(np.float32))
    t_ir = bare(ApplyImpulseResponse, convolve_mode="full", compensate_for_propagation_delay=False,
                transform_parameters={"ir": ir})
    y_ir = t_ir.apply_transform(x.clone(), 8000).samples
    noise = Audio.rms_normalize(torch.from_numpy(synth.batch(3, seed=1300, n=8000, tonal=False)))
    snr = torch.tensor([-10.0, 0.0, 7.5])
    t_bg = bare(AddBackgroundNoise, transform_parameters={"background": noise, "snr_in_db": snr})
    y_bg = t_bg.apply_transform(x.clone(), 8000).samples
    gdb = torch.tensor([-5.0, 0.3, 5.0])
    t_g = bare(Gain, transform_parameters={"gain_factors": (10 ** (gdb / 20)).unsqueeze(1).unsqueeze(1)})
    y_g = t_g.apply_transform(x.clone(), 8000).samples
    pct = torch.tensor([0.0, 0.004, 0.01])
    # one example at a time, as AugmentFP.__call__ is used by the data pipeline (training/dataset.py:143-154,
    # testing/generate_queries.py:72-92).  (With B > 1 the reference's torch.quantile call flattens the whole batch:
    # clipping.py:72-86 -- a batch_augment-only quirk, see y_clip_batchquirk.)
    y_c = torch.cat([bare(Clipping, transform_parameters={"percentile_threshold": pct[b:b + 1].unsqueeze(1)})
                     .apply_transform(x[b:b + 1].clone(), 8000).samples for b in range(3)])
    y_cq = bare(Clipping, transform_parameters={"percentile_threshold": pct.unsqueeze(1)}).apply_transform(x.clone(), 8000).samples
    t_p = bare(PeakNormalization)
    xs = x.clone() * torch.tensor([0.3, 0.0, 2.0]).view(3, 1, 1)
    t_p.transform_parameters = {}
    t_p.randomize_parameters(xs)
    y_p = t_p.apply_transform(xs.clone(), 8000).samples
    save("g10_augment", seed_x=1200, seed_noise=1300, n=8000, ir=ir.numpy(), y_ir=y_ir.numpy(), snr=snr.numpy(),
         y_bg=y_bg.numpy(), gain_db=gdb.numpy(), y_gain=y_g.numpy(), percentile=pct.numpy(), y_clip=y_c.numpy(), y_clip_batchquirk=y_cq.numpy(),
         peak_scale=np.array([0.3, 0.0, 2.0], dtype=np.float32), y_peak=y_p.numpy(),
         conv_full=convolve(x[:1], ir[:1]).numpy(), versions=versions)
    # ---- G11: MultiResolutionSTFTLoss (training/loss.py:10-186) --------------------------------------------
    # loss.py calls torch.stft without return_complex (torch 1.11 API); the shim only adds that keyword and the
    # view_as_real the old API implied -- the reference's arithmetic runs unmodified.
    import training.loss as rloss
    real_stft = torch.stft

    def stft_shim(x, n_fft, hop_length=None, win_length=None, window=None, **kw):
        return torch.view_as_real(real_stft(x, n_fft, hop_length, win_length, window, return_complex=True, **kw))

    torch.stft = stft_shim
    try:
        crit = rloss.MultiResolutionSTFTLoss(factor_sc=0.1, factor_mag=0.1)       # training/train.py:116-119
        xs = torch.from_numpy(synth.batch(3, seed=1400, n=24000))
        ys = torch.from_numpy((0.8 * synth.batch(3, seed=1400, n=24000) + 0.2 * synth.batch(3, seed=1401, n=24000, tonal=False)).astype(np.float32))
        sc, mag = crit(xs, ys)
        per = [[float(v) for v in f(xs, ys)] for f in crit.stft_losses]
        m0 = rloss.stft(xs[:1], 1024, 120, 600, torch.hann_window(600)).numpy()
        zs, zm = crit(torch.zeros(2, 8000), ys[:2, :8000])                          # a silent prediction (clamp 1e-7 path)
    finally:
        torch.stft = real_stft
    save("g11_mrstft_loss", seed_x=1400, seed_noise=1401, n=24000, factor_sc=0.1, factor_mag=0.1, sc=float(sc), mag=float(mag), per_resolution=np.array(per),
         mag0_sub=m0[0, ::7, ::9].copy(), mag0_shape=np.array(m0.shape), sc_silent=float(zs), mag_silent=float(zm), versions=versions)
    g12_demucs_train_step(versions)
    g13_dejavu_denoised(fp, versions)
    print("done")


if __name__ == "__main__":
    main()
