"""CPU-side checks of the drop-in boundary: libmfpa.so loads, exports every symbol include/mfpa.h declares,
host-only entry points work, and the product package refuses to run without a GPU (no CPU fallback)."""
import ctypes
import hashlib
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from musicfpaugment_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from musicfpaugment_amd.csrc.build import build
        build(verbose=False)
    return _lib


def test_header_and_library_agree(lib):
    header = open(os.path.join(ROOT, "include", "mfpa.h")).read()
    declared = sorted(set(re.findall(r"^int\s+(mfpa_\w+)\s*\(", header, flags=re.M)))
    assert declared == lib.exported_symbols(), "include/mfpa.h and the ctypes signature table diverged"
    handle = lib.lib()
    for name in declared:
        assert hasattr(handle, name), name
    assert handle.mfpa_version() == lib.ABI_VERSION
    m = re.search(r"#define MFPA_STFT_TABLE_LEN (\d+)", header)
    assert int(m.group(1)) == lib.STFT_TABLE_LEN


def test_host_only_entry_points(lib):
    h = lib.lib()
    assert h.mfpa_stft_frames(64000) == 251 and h.mfpa_stft_frames(24000) == 94 and h.mfpa_specgram_frames(64000) == 249
    win = np.hanning(514)[1:-1]
    out = np.empty(lib.STFT_TABLE_LEN)
    assert h.mfpa_stft_tables(win.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p)) == 0
    np.testing.assert_array_equal(out[:512], win)
    m = np.arange(256)
    np.testing.assert_allclose(out[512:1024:2], np.cos(2 * np.pi * m / 256), atol=3e-16)
    np.testing.assert_allclose(out[513:1024:2], -np.sin(2 * np.pi * m / 256), atol=3e-16)
    assert h.mfpa_stft_tables(None, None) == lib.EINVAL
    # resident workgroups of a persistent LSTM launch: host arithmetic against the current device's CU count (none here -> 0,
    # i.e. the per-step path); bad arguments are rejected
    n = ctypes.c_int(-1)
    assert h.mfpa_lstm_seq_workgroups(64, 768, 0, ctypes.addressof(n)) == 0 and n.value >= 0
    assert h.mfpa_lstm_bwd_seq_workgroups(64, 768, 0, ctypes.addressof(n)) == 0 and n.value >= 0
    assert h.mfpa_lstm_seq_workgroups(64, 768, 0, None) == lib.EINVAL and h.mfpa_lstm_seq_workgroups(64, 100, 0, ctypes.addressof(n)) == lib.EINVAL


def test_argument_errors_do_not_touch_the_gpu(lib):
    h = lib.lib()
    # invalid shapes are rejected on the host before any launch (works without a GPU)
    assert h.mfpa_stft_mag(1, 2, 100, 1, 1, 0, None, None) == lib.EINVAL        # T_w <= 256
    assert h.mfpa_stft_mag(None, 2, 8000, None, None, 0, None, None) == lib.EINVAL
    assert h.mfpa_audfprint_prune(1, 1, 255, 10, 1, 0.99, 5, 1, 1, None) == lib.EINVAL   # R % 4
    assert h.mfpa_audfprint_prune(1, 1, 256, 10, 1, 0.99, 9, 1, 1, None) == lib.EINVAL   # maxpks > 8
    assert h.mfpa_peak_metrics(1, 1, 1, 1, 5, 1, None) == lib.EINVAL                      # N1 < 2
    assert h.mfpa_conv3x3_bn_relu(1, 48, None, 0, 0, 0, 1, 8, 8, 1, 64, None, None, 1, 0, 1, None) == lib.EINVAL
    assert h.mfpa_stft_mag(None, 0, 8000, None, None, 0, None, None) == 0      # empty batch is a no-op
    # descriptor entry points: null descriptors, bad modes / precisions / channel counts
    assert h.mfpa_conv_mfma(None, None) == lib.EINVAL and h.mfpa_wgrad_mfma(None, None) == lib.EINVAL
    assert h.mfpa_gemm_mfma(None, None) == lib.EINVAL
    d = lib.ConvDesc(x0=1, w=1, y=1, C0=64, C1=0, B=1, H=8, W=8, Cout=64, mode=3)
    assert h.mfpa_conv_mfma(ctypes.byref(d), None) == lib.EINVAL                          # mode
    d = lib.ConvDesc(x0=1, w=1, y=1, C0=64, C1=0, B=1, H=8, W=8, Cout=64, mode=0, precision=2)
    assert h.mfpa_conv_mfma(ctypes.byref(d), None) == lib.EINVAL                          # precision
    d = lib.ConvDesc(x0=1, w=1, y=1, C0=48, C1=0, B=1, H=8, W=8, Cout=64)
    assert h.mfpa_conv_mfma(ctypes.byref(d), None) == lib.EINVAL                          # C0 % 32
    d = lib.ConvDesc(x0=1, w=1, y=1, C0=64, C1=0, B=0, H=8, W=8, Cout=64)
    assert h.mfpa_conv_mfma(ctypes.byref(d), None) == 0                                   # empty batch
    g = lib.WgradDesc(dz=1, x0=1, dw=1, C0=96, C1=0, B=1, H=8, W=8, Cout=64)
    assert h.mfpa_wgrad_mfma(ctypes.byref(g), None) == lib.EINVAL                         # C0 % 64
    g = lib.WgradDesc(dz=1, x0=1, dw=1, C0=64, C1=0, B=1, H=8, W=8, Cout=64, precision=7)
    assert h.mfpa_wgrad_mfma(ctypes.byref(g), None) == lib.EINVAL
    m = lib.GemmDesc(A=1, lda=6, strideA=0, W=1, C=1, ldc=64, strideC=0, batch=1, M=8, N=64, K=16, npad=64)
    assert h.mfpa_gemm_mfma(ctypes.byref(m), None) == lib.EINVAL                          # lda % 4: float4 rows
    m = lib.GemmDesc(A=1, lda=8, strideA=0, W=1, C=1, ldc=64, strideC=0, batch=1, M=8, N=64, K=24, npad=64)
    assert h.mfpa_gemm_mfma(ctypes.byref(m), None) == lib.EINVAL                          # K % 16
    # AugmentFP, Demucs and loss entry points
    assert h.mfpa_fir(1, 1, 100, 99, 1, 1, 1, 1, 1, 0, 0, 1, None, None) == lib.EINVAL    # Tout < T
    assert h.mfpa_fir(1, 1, 100, 100, 1, 1, 1, 1, 1, 2, 0, 1, None, None) == lib.EINVAL   # pad_mode
    assert h.mfpa_fir(1, 1, 100, 100, 1, 1, 1, 1, 1, 0, 2, 1, None, None) == lib.EINVAL   # impulse-response mode needs `peak`
    assert h.mfpa_clip_quantile(None, 1, 100, 1, 1, 1, None) == lib.EINVAL
    assert h.mfpa_clip_quantile_flat(1, 4, 100, 1, 1, 5, 1, None) == lib.EINVAL          # more selected than examples
    assert h.mfpa_clip_quantile_flat(1, 300, 64000, 1, 1, 300, 1, None) == lib.EINVAL     # beyond torch.quantile's input limit
    assert h.mfpa_gather_background(1, 1, 1, 1, 0, 100, 1, None) == lib.EINVAL            # P < 1
    assert h.mfpa_lstm_step(None, 0, 1, 1, 3072, 1, 4, 760, 1, 768, None, None, 0, None) == lib.EINVAL   # H % 128
    assert h.mfpa_lstm_step(None, 0, 1, 1, 3072, 1, 0, 768, 1, 768, None, None, 0, None) == 0           # empty batch
    assert h.mfpa_localmax2d(1, 1, 257, 249, 1, 50.0, 1, 1, None) == lib.EINVAL           # radius < 2
    assert h.mfpa_reflect_pad(1, 1, 100, 100, 0, 400, 1, None) == lib.EINVAL              # pad >= T
    assert h.mfpa_stft_loss_sums(1, 1, 10, 513, 1000, 576, 1, 1, None) == lib.EINVAL      # ldc < im_off + bins
    assert h.mfpa_loss_blocks() > 0
    # Demucs training entry points
    assert h.mfpa_gemm_tn(None, None) == lib.EINVAL
    t = lib.GemmTnDesc(A=1, lda=48, strideA=0, Bm=1, ldb=48, strideB=0, C=1, ldc=48, batch=1, R=10, M=46, N=48, precision=0)
    assert h.mfpa_gemm_tn(ctypes.byref(t), None) == lib.EINVAL                            # M % 4
    t = lib.GemmTnDesc(A=1, lda=48, strideA=0, Bm=1, ldb=48, strideB=0, C=1, ldc=48, batch=1, R=10, M=48, N=48, precision=3)
    assert h.mfpa_gemm_tn(ctypes.byref(t), None) == lib.EINVAL                            # precision
    t = lib.GemmTnDesc(A=1, lda=48, strideA=0, Bm=1, ldb=48, strideB=0, C=1, ldc=48, batch=1, R=0, M=48, N=48)
    assert h.mfpa_gemm_tn(ctypes.byref(t), None) == 0                                     # no rows: a no-op
    assert h.mfpa_glu_bwd(1, 10, 100, 48, 1, 48, None) == lib.EINVAL                      # npad % 64
    assert h.mfpa_glu_bwd(1, 10, 128, 96, 1, 96, None) == lib.EINVAL                      # N > npad / 2
    assert h.mfpa_colsum_any(1, 10, 46, 48, 1, None) == lib.EINVAL                        # C % 4
    assert h.mfpa_c1_wgrad(1, 100, 1, 48, 4800, 1, 100, 48, 1, None) == lib.EINVAL        # ldx too short for L windows
    assert h.mfpa_downsample2_adjoint(1, 1, 10, 600, 1, None, 1001, 1, None) == lib.EINVAL   # nout > ceil(T / 2) / ldy < nout
    assert h.mfpa_lstm_step_bwd(None, 0, 1, 1, 3072, 1, 768, None, 0, 1, 768, 1, 4, 760, None) == lib.EINVAL   # H % 128
    assert h.mfpa_lstm_layer(1, 1, 1, None, None, 4, 10, 768, None, None, 0, None) == lib.EINVAL    # inference needs cstate
    assert h.mfpa_lstm_layer(1, 1, 1, None, None, 4, 10, 768, None, None, 1, None) == lib.EINVAL    # training needs cseq
    assert h.mfpa_lstm_layer_range(1, 1, 1, 1, None, 4, 10, 768, None, None, 1, 5, 11, None) == lib.EINVAL   # t1 > Tn
    assert h.mfpa_lstm_layer_bwd(1, 1, 1, 1, None, 4, 10, 768, None) == lib.EINVAL        # dcstate
    m = lib.GemmDesc(A=1, lda=48, strideA=0, W=1, C=1, ldc=1 << 29, strideC=0, batch=1, M=16, N=64, K=48, npad=64)
    assert h.mfpa_gemm_mfma(ctypes.byref(m), None) == lib.EINVAL                          # a clip's output beyond 32-bit byte offsets
    m = lib.GemmDesc(A=1, lda=48, strideA=0, W=1, C=1, ldc=64, strideC=0, batch=1, M=16, N=64, K=48, npad=64, mode=3)
    assert h.mfpa_gemm_mfma(ctypes.byref(m), None) == lib.EINVAL                          # mode 3 needs the addend (the ReLU output)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_no_cpu_fallback():
    from musicfpaugment_amd import ops
    from musicfpaugment_amd._lib import MfpaError
    from musicfpaugment_amd.training.unet import UNet
    with pytest.raises(MfpaError):
        ops.stft_mag(torch.zeros(1, 8000))
    with pytest.raises(MfpaError):
        UNet(1, 1).eval()(torch.zeros(1, 1, 257, 32))
    with pytest.raises(MfpaError):
        ops.peak_metrics_counts(torch.zeros(1, 4, 4, dtype=torch.uint8), torch.zeros(1, 4, 4, dtype=torch.uint8))
    from musicfpaugment_amd.training.loss import MultiResolutionSTFTLoss
    from musicfpaugment_amd.training.model import Demucs
    with pytest.raises(MfpaError):
        MultiResolutionSTFTLoss()(torch.zeros(1, 8000), torch.zeros(1, 8000))
    with pytest.raises(MfpaError):
        Demucs().eval()(torch.zeros(1, 8000))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "musicfpaugment_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dirpath, f)


def test_committed_build_record_matches_the_sources_in_the_tree():
    """profiles/BUILD_rNN.txt (written by __graft_entry__.build()) must describe THESE sources: `sources_sha256` covers every .hip / .h of
    csrc/ and include/mfpa.h.  A record committed before the round's last source change fails here (round 5's was one commit stale)."""
    import json
    import __graft_entry__ as ge
    from musicfpaugment_amd.csrc import build as b
    path = os.path.join(ROOT, "profiles", ge.BUILD_RECORD)
    assert os.path.exists(path), f"{path}: run __graft_entry__.build() and commit the record"
    rec = json.load(open(path))
    h = hashlib.sha256()
    for f in b._sources() + sorted(x for x in os.listdir(b.HERE) if x.endswith(".h")) + ["../../include/mfpa.h"]:
        with open(os.path.join(b.HERE, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    assert rec["sources_sha256"] == h.hexdigest(), "profiles/%s is stale: rebuild (python -c 'import __graft_entry__ as g; g.build()') and commit it" % ge.BUILD_RECORD
    if os.path.exists(b.OUT):                                        # the library in the tree (git-ignored, travels to the GPU box) is the recorded one
        lib_h = hashlib.sha256(open(b.OUT, "rb").read()).hexdigest()
        assert rec["sha256"] == lib_h, "libmfpa.so in the tree is not the build the record describes"
