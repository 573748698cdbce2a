"""GPU parity: Demucs forward (fp32 MFMA GEMMs) vs the torch-CPU oracle and the golden output of the real reference."""
import numpy as np
import pytest
import torch

from musicfpaugment_amd import synth
from musicfpaugment_amd.training.demucs_weights import formula_state_dict, state_dict_shapes

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def net():
    from musicfpaugment_amd.training.model import Demucs
    m = Demucs()
    m.load_state_dict(formula_state_dict(0))
    return m.cuda().eval()


def test_state_dict_and_lengths(net):
    shapes = state_dict_shapes()
    sd = net.state_dict()
    assert list(sd.keys()) == list(shapes.keys())
    assert all(tuple(sd[k].shape) == shapes[k][0] for k in shapes)
    assert sum(v.numel() for v in sd.values()) == 18_867_937
    assert net.valid_length(64000) == 64085 and net.total_stride == 256


def test_building_blocks():
    import torch.nn.functional as F
    from musicfpaugment_amd import ops_demucs as D
    from musicfpaugment_amd._lib import check, lib, ptr, stream
    from oracle import demucs as od
    from oracle.unet import relative_l1
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 1001, generator=g)
    ker = D.sinc_kernel("cuda")
    y = torch.empty(3, 2002, device="cuda")
    xd = x.cuda()
    check(lib().mfpa_upsample2(ptr(xd), 3, 1001, ptr(ker), ptr(y), stream()), "up")
    assert relative_l1(y.cpu(), od.upsample2(x)) < 1e-6
    z = torch.empty(3, 501, device="cuda")
    check(lib().mfpa_downsample2(ptr(xd), 3, 1001, ptr(ker), ptr(z), 501, 0, 0, stream()), "down")
    assert relative_l1(z.cpu(), od.downsample2(x)) < 1e-6
    # strided-window GEMM = Conv1d(k8, s4) + ReLU on (B, L, C)
    B, Lin, Cin, Cout = 2, 404, 48, 96
    h = torch.randn(B, Cin, Lin, generator=g)
    w = torch.randn(Cout, Cin, 8, generator=g) / np.sqrt(8 * Cin)
    bias = torch.randn(Cout, generator=g)
    want = F.relu(F.conv1d(h, w, bias, stride=4))
    Lout = want.shape[-1]
    hn = h.permute(0, 2, 1).contiguous().cuda()
    out = torch.empty(B, Lout, Cout, device="cuda")
    D.gemm(D._p(hn), 4 * Cin, Lin * Cin, B, Lout, D._pad_rows(w.permute(0, 2, 1).reshape(Cout, -1)).cuda(),
           D._pad_rows(bias).cuda(), Cout, D._p(out), Cout, Lout * Cout, relu=1)
    assert relative_l1(out.cpu().permute(0, 2, 1), want) < 1e-5
    # 1x1 + GLU
    wg = torch.randn(2 * Cout, Cout, generator=g) / np.sqrt(Cout)
    bg = torch.randn(2 * Cout, generator=g)
    want_g = F.glu(F.conv1d(want, wg[:, :, None], bg), dim=1)
    wgp, bgp = D._pack_glu(wg, bg)
    outg = torch.empty(B, Lout, Cout, device="cuda")
    D.gemm(D._p(out), Cout, Lout * Cout, B, Lout, wgp.cuda(), bgp.cuda(), Cout, D._p(outg), Cout, Lout * Cout, mode=1)
    assert relative_l1(outg.cpu().permute(0, 2, 1), want_g) < 1e-5


def test_forward_golden_and_oracle(net, golden):
    from oracle import demucs as od
    from oracle.unet import relative_l1
    g = golden("g9_demucs_forward")
    w1 = synth.batch(2, seed=int(g["seed1"]), n=int(g["n1"]))
    y1 = net(torch.from_numpy(w1).cuda()).cpu()
    assert y1.shape == (2, 1, 8000)
    assert relative_l1(y1, torch.from_numpy(g["y1"])) <= TOL
    w8 = synth.batch(2, seed=int(g["seed8"]))                       # 8 s clips: clip 0 is the golden one
    y8 = net(torch.from_numpy(w8).cuda()).cpu()
    assert relative_l1(y8[0, 0, ::16], torch.from_numpy(g["y8_sub"])) <= TOL
    assert abs(float(y8[0].double().abs().sum()) - float(g["y8_abs_sum"])) <= TOL * float(g["y8_abs_sum"])
    with torch.no_grad():
        want = od.forward(torch.from_numpy(w8[1:2]), formula_state_dict(0))
    assert relative_l1(y8[1:2], want) <= TOL


def test_errors(net):
    from musicfpaugment_amd._lib import MfpaError
    with pytest.raises(MfpaError):
        net(torch.zeros(1, 8000))
    with pytest.raises(ValueError):
        net(torch.zeros(1, 2, 8000, device="cuda"))


def test_presplit_weights_give_the_same_bits(net):
    """mfpa_gemm_mfma precision 2 (W split once on the host, ops_demucs.split_rows) = precision 1 (split in every workgroup):
    the same bf16 hi / lo values reach the same MFMAs, so the outputs are identical, GEMM by GEMM and for the whole network."""
    from musicfpaugment_amd import ops_demucs as D
    g = torch.Generator().manual_seed(3)
    for (M, N, K) in [(1000, 384, 768), (257, 128, 128), (64, 1536, 3072)]:
        A = torch.randn(M, K, generator=g).cuda()
        W = (torch.randn(N, K, generator=g) / np.sqrt(K)).cuda()
        bias = torch.randn(N, generator=g).cuda()
        c1, c2 = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
        D.gemm(D._p(A), K, 0, 1, M, W, bias, N, D._p(c1), N, 0, precision=1)
        D.attach_split(W)
        assert W._mfpa_split[1].shape == W.shape
        D.gemm(D._p(A), K, 0, 1, M, W, bias, N, D._p(c2), N, 0, precision=1)
        assert torch.equal(c1, c2)
        assert (c1 - (A.double() @ W.double().t() + bias.double()).float()).abs().max() < 1e-4
        W.mul_(2.0)                                        # an in-place update retires the split copy
        D.gemm(D._p(A), K, 0, 1, M, W, bias, N, D._p(c2), N, 0, precision=1)
        torch.testing.assert_close(c2 - bias, 2 * (c1 - bias), rtol=1e-5, atol=1e-5)
    wav = torch.from_numpy(synth.noise_batch(2, 16000, seed=5)).cuda() if hasattr(synth, "noise_batch") else torch.randn(2, 16000, generator=g).cuda()
    y_split = net(wav)
    old = D.PRESPLIT_WEIGHTS
    try:
        D.PRESPLIT_WEIGHTS = False
        net._packed = None
        y_fly = net(wav)
    finally:
        D.PRESPLIT_WEIGHTS = old
        net._packed = None
    assert torch.equal(y_split, y_fly)


def _lstm_reference(x, skip, wih, bias, whh):
    """torch.nn.LSTM's recurrence (model.py:91-110) in float64 on the CPU: two layers, then + skip."""
    B, Tn, H = x.shape
    inp = x.double()
    for k in range(2):
        h = torch.zeros(B, H, dtype=torch.float64)
        c = torch.zeros(B, H, dtype=torch.float64)
        out = []
        for t in range(Tn):
            g = inp[:, t] @ wih[k].double().t() + bias[k].double() + h @ whh[k].double().t()
            i, f, gg, o = g.split(H, dim=1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
            h = torch.sigmoid(o) * torch.tanh(c)
            out.append(h)
        inp = torch.stack(out, dim=1)
    return inp + skip.double()


@pytest.mark.parametrize("B,Tn,H,train", [(70, 9, 256, False), (256, 7, 768, False), (33, 6, 768, True), (130, 20, 512, False), (65, 5, 1024, True)])
def test_persistent_lstm_layer(B, Tn, H, train):
    """mfpa_lstm_layer_seq (one persistent launch per layer: W_hh in registers, h exchanged in split form, slab barriers in
    device memory) against the float64 recurrence and against the per-step kernels; ragged slabs, both chunked (two streams) and
    whole-sequence forms, the training form's saved gates and cell states; no wait may have given up."""
    from musicfpaugment_amd import ops_demucs as D
    g = torch.Generator().manual_seed(B + Tn)
    x = torch.randn(B, Tn, H, generator=g) * 0.5
    skip = torch.randn(B, Tn, H, generator=g)
    wih = [torch.randn(4 * H, H, generator=g) / np.sqrt(H) for _ in range(2)]
    whh = [torch.randn(4 * H, H, generator=g) / np.sqrt(H) for _ in range(2)]
    bias = [torch.randn(4 * H, generator=g) * 0.1 for _ in range(2)]
    grouped = [w.reshape(4, H // 16, 16, H).permute(1, 0, 2, 3).reshape(4 * H, H).contiguous().cuda() for w in whh]
    want = _lstm_reference(x, skip, wih, bias, whh)
    dev = lambda ts: [t.cuda() for t in ts]
    runs = {}
    old = (D.PERSISTENT_LSTM, D.LSTM_CHUNK)
    try:
        for name, pers, chunk in [("seq", True, old[1]), ("seq-chunked", True, 4), ("steps", False, old[1])]:
            D.PERSISTENT_LSTM, D.LSTM_CHUNK = pers, chunk
            xsum, saved = D.lstm_two_layers(x.cuda(), skip.cuda(), dev(wih), dev(bias), grouped, 1, train)
            torch.cuda.synchronize()
            runs[name] = (xsum.cpu(), [tuple(None if s is None else s.cpu() for s in lay) for lay in saved])
    finally:
        D.PERSISTENT_LSTM, D.LSTM_CHUNK = old
    assert not D.lstm_seq_error()
    for name, (xsum, saved) in runs.items():
        err = (xsum.double() - want).abs().max().item()
        assert err < 2e-4, (name, err)
    for name in ("seq-chunked", "steps"):
        assert (runs[name][0] - runs["seq"][0]).abs().max() < 1e-4
        for la, lb in zip(runs[name][1], runs["seq"][1]):
            for ta, tb in zip(la, lb):
                assert (ta is None) == (tb is None)
                if ta is not None:
                    assert (ta - tb).abs().max() < 1e-4           # hseq, saved gates, cell states


@pytest.mark.parametrize("B,L", [(3, 1000), (2, 127), (1, 126), (2, 128), (1, 1), (2, 5 * 127 * 8 + 3)])
def test_last_decoder_level_in_one_launch(B, L):
    """mfpa_glu_convT1d_c1 = Conv1d(48, 96, 1) + GLU + ConvTranspose1d(48, 1, 8, 4) (model.py:80-88) against torch in float64;
    tile edges (127 output groups per tile, 8 tiles per workgroup) and one-row inputs."""
    import torch.nn.functional as F
    from musicfpaugment_amd import ops_demucs as D
    from musicfpaugment_amd._lib import check, lib, ptr, stream
    C = 48
    g = torch.Generator().manual_seed(L)
    x = torch.randn(B, L, C, generator=g)
    w1 = torch.randn(2 * C, C, generator=g) / np.sqrt(C)
    b1 = torch.randn(2 * C, generator=g) * 0.3
    wt = torch.randn(C, 1, 8, generator=g) / np.sqrt(C)
    bt = 0.123
    want = F.conv_transpose1d(F.glu(F.conv1d(x.double().permute(0, 2, 1), w1.double()[:, :, None], b1.double()), dim=1),
                              wt.double(), torch.tensor([bt], dtype=torch.float64), stride=4)[:, 0]
    gw, gb = D._pack_glu(w1, b1)
    wl = wt[:, 0, :].t().contiguous()
    y = torch.full((B, 4 * (L + 1)), float("nan"), device="cuda")
    xd, gwd, gbd, wld = x.cuda(), gw.cuda(), gb.cuda(), wl.cuda()
    check(lib().mfpa_glu_convT1d_c1(ptr(xd), B, L, C, ptr(gwd), ptr(gbd), ptr(wld), bt, ptr(y), stream()), "tail")
    assert want.shape == y.shape
    err = (y.cpu().double() - want).abs().max().item()
    assert err < 2e-5 * max(1.0, want.abs().max().item()), err
    assert lib().mfpa_glu_convT1d_c1(ptr(xd), B, L, 32, ptr(gwd), ptr(gbd), ptr(wld), bt, ptr(y), stream()) == -22


@pytest.mark.parametrize("B,Lout", [(3, 1000), (2, 128), (1, 129), (2, 1), (2, 8 * 128 * 3 + 5)])
def test_first_encoder_level_in_one_launch(B, Lout):
    """mfpa_conv1d_c1_glu = Conv1d(1, 48, 8, 4) + ReLU + Conv1d(48, 96, 1) + GLU (model.py:66-75) against torch in float64; tile
    edges (128 rows per tile, 8 tiles per workgroup), inputs longer than the last window needs."""
    import torch.nn.functional as F
    from musicfpaugment_amd import ops_demucs as D
    from musicfpaugment_amd._lib import check, lib, ptr, stream
    C = 48
    Lin = 4 * (Lout - 1) + 8 + 4 * (Lout % 3)                       # a multiple of 4: the rows' samples are read as aligned float4
    g = torch.Generator().manual_seed(Lout)
    x = torch.randn(B, Lin, generator=g)
    w0 = torch.randn(C, 1, 8, generator=g) / np.sqrt(8)
    b0 = torch.randn(C, generator=g) * 0.3
    w1 = torch.randn(2 * C, C, generator=g) / np.sqrt(C)
    b1 = torch.randn(2 * C, generator=g) * 0.3
    a = F.relu(F.conv1d(x.double()[:, None, :4 * (Lout - 1) + 8], w0.double(), b0.double(), stride=4))     # the first Lout windows
    want = F.glu(F.conv1d(a, w1.double()[:, :, None], b1.double()), dim=1).permute(0, 2, 1)
    assert want.shape == (B, Lout, C)
    gw, gb = D._pack_glu(w1, b1)
    xd, w0d, b0d, gwd, gbd = x.cuda(), w0[:, 0, :].t().contiguous().cuda(), b0.cuda(), gw.cuda(), gb.cuda()
    y = torch.full((B, Lout, C), float("nan"), device="cuda")
    check(lib().mfpa_conv1d_c1_glu(ptr(xd), B, Lin, Lout, C, ptr(w0d), ptr(b0d), ptr(gwd), ptr(gbd), ptr(y), stream()), "head")
    err = (y.cpu().double() - want).abs().max().item()
    assert err < 2e-5 * max(1.0, want.abs().max().item()), err
    assert lib().mfpa_conv1d_c1_glu(ptr(xd), B, 4 * (Lout - 1) + 4, Lout, C, ptr(w0d), ptr(b0d), ptr(gwd), ptr(gbd), ptr(y), stream()) == -22
    assert lib().mfpa_conv1d_c1_glu(ptr(xd), B, Lin - 1, Lout, C, ptr(w0d), ptr(b0d), ptr(gwd), ptr(gbd), ptr(y), stream()) == -22


def test_first_encoder_level_with_two_workgroups_per_cu_is_bit_exact():
    """More workgroups than CUs (two share a CU, i.e. a SIMD runs this kernel's first-convolution FMAs next to another wave's
    MFMAs): every run must give the bits of the 128 x 64-tile GEMM form, which evaluates the same arithmetic in the same order.
    This is the case in which the packed-fp32 form of the convolution (v_pk_fma_f32 op_sel:[0,1,0]) returned sporadically wrong
    rows -- small shapes and one-workgroup-per-CU launches never showed it (profiles/r02_pk_fma_op_sel.md)."""
    from musicfpaugment_amd import ops_demucs as D
    from musicfpaugment_amd._lib import check, lib, ptr, stream
    C, B, Lout = 48, 48, 64084
    Lin = 4 * (Lout - 1) + 8
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, Lin, generator=g).cuda()
    w0 = (torch.randn(8, C, generator=g) / np.sqrt(8)).cuda()
    b0 = (torch.randn(C, generator=g) * 0.3).cuda()
    gw, gb = D._pack_glu(torch.randn(2 * C, C, generator=g) / np.sqrt(C), torch.randn(2 * C, generator=g) * 0.3)
    gw, gb = gw.cuda(), gb.cuda()
    ref = torch.empty(B, Lout, C, device="cuda")
    D.gemm(0, C, Lout * C, B, Lout, gw, gb, C, D._p(ref), C, Lout * C, mode=1, c1=(x, w0, b0))
    for _ in range(4):
        y = torch.full((B, Lout, C), float("nan"), device="cuda")
        check(lib().mfpa_conv1d_c1_glu(ptr(x), B, Lin, Lout, C, ptr(w0), ptr(b0), ptr(gw), ptr(gb), ptr(y), stream()), "head")
        assert torch.equal(y, ref)


def test_one_channel_kernels_beside_an_mfma_gemm_on_a_second_stream_are_bit_exact():
    """mfpa_conv1d_c1 and mfpa_c1_wgrad (the two one-channel ends of the Demucs training step) at > 256 workgroups on a side
    stream while a bf16x3 MFMA GEMM keeps the main stream busy -- DemucsTrainEngine.train_step overlaps exactly these
    (ops_demucs_train.py: the clean signal's STFT GEMMs run on a side stream while forward() starts with mfpa_conv1d_c1).  Both
    kernels contained `v_pk_fma_f32 ... op_sel:[0,1,0]`, the form that returned wrong low lanes next to MFMA waves
    (profiles/r02_pk_fma_op_sel.md); they are now compiled without packed fp32 (MFPA_NO_PK_F32, tests/test_isa_scan.py).
    Four overlapped runs must reproduce the bits of a run with the GPU to itself."""
    from musicfpaugment_amd import ops_demucs as D
    from musicfpaugment_amd._lib import check, lib, ptr, stream
    C, B, Lout = 48, 16, 64084                                      # 16 x 64084 x 12 quads / 256 threads = 48 063 workgroups' worth
    Lin = 4 * (Lout - 1) + 8
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, Lin, generator=g).cuda()
    w0 = (torch.randn(8, C, generator=g) / np.sqrt(8)).cuda()
    b0 = (torch.randn(C, generator=g) * 0.3).cuda()
    gr = torch.randn(B, Lout, C, generator=g).cuda()
    # the MFMA work: a K = 768, N = 768 GEMM over 64 x 2048 rows (about 1 ms per launch), bf16x3
    M, K, N = 2048, 768, 768
    A = torch.randn(64, M, K, generator=g).cuda()
    Wg = (torch.randn(N, K, generator=g) / np.sqrt(K)).cuda()
    bg = torch.zeros(N, device="cuda")
    Cg = torch.empty(64, M, N, device="cuda")

    def head(y):
        check(lib().mfpa_conv1d_c1(ptr(x), B, Lin, Lout, C, ptr(w0), ptr(b0), 1, ptr(y), stream()), "conv1d_c1")

    def wgrad(dw):
        dw.zero_()
        check(lib().mfpa_c1_wgrad(ptr(x), Lin, ptr(gr), C, Lout * C, B, Lout, C, ptr(dw), stream()), "c1_wgrad")

    y_ref = torch.empty(B, Lout, C, device="cuda")
    head(y_ref)
    want = torch.relu(torch.nn.functional.conv1d(x[:, None, :], w0.t()[:, None, :], b0, stride=4)).permute(0, 2, 1)
    assert (y_ref - want).abs().max().item() < 1e-5
    dw_runs = []
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    for _ in range(4):
        y = torch.full((B, Lout, C), float("nan"), device="cuda")
        dw = torch.empty(8, C, device="cuda")
        side.wait_stream(torch.cuda.current_stream())
        for _ in range(3):                                           # main stream: MFMA waves on every CU for ~3 ms
            D.gemm(D._p(A), K, M * K, 64, M, Wg, bg, N, D._p(Cg), N, M * N, precision=1)
        with torch.cuda.stream(side):
            head(y)
            wgrad(dw)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        assert torch.equal(y, y_ref)
        dw_runs.append(dw.clone())
    # the weight gradient adds partial sums with float atomics (order varies): equal to rounding, and right against torch
    want_dw = torch.einsum("btj,btc->jc", x.double().unfold(1, 8, 4)[:, :Lout].cpu(), gr.double().cpu())
    for dw in dw_runs:
        assert (dw.double().cpu() - want_dw).abs().max().item() < 2e-3 * want_dw.abs().max().item()


def test_persistent_lstm_failure_is_caught_before_the_result_leaves_and_the_call_is_rerun(net):
    """A persistent LSTM launch whose waits gave up (grid not co-resident: another process on the GPU) returns garbage and raises the
    error word of its scratch.  Simulated by raising the word by hand -- the kernel then skips every wait, exactly the state after a
    give-up.  demucs_forward must notice BEFORE returning (ops_demucs.lstm_results_ok), clear the word, switch the persistent path
    off for the process, re-run on the per-step kernels and return the right audio -- also for the only / last call of a run."""
    from musicfpaugment_amd import ops_demucs as D
    from musicfpaugment_amd._lib import lib
    from oracle.unet import relative_l1
    x = torch.from_numpy(synth.batch(3, seed=77, n=16000)).cuda()
    old = (D.PERSISTENT_LSTM, D.PERSISTENT_LSTM_BWD)
    try:
        D.PERSISTENT_LSTM = True
        want = net(x)                                                 # healthy run (creates the scratch buffers of this shape)
        assert D.PERSISTENT_LSTM and not D.lstm_seq_error()
        off = lib().mfpa_lstm_seq_error_offset() // 4
        for ent in D._LSTM_WORK.values():
            ent[0][off] = 1
        with pytest.warns(RuntimeWarning, match="persistent LSTM"):
            got = net(x)
        assert not D.PERSISTENT_LSTM and not D.PERSISTENT_LSTM_BWD    # off for the rest of the process
        assert not D.lstm_seq_error()                                 # words cleared
        assert relative_l1(got.cpu(), want.cpu()) <= 1e-5             # the per-step kernels' result (same arithmetic, other schedule)
        assert relative_l1(net(x).cpu(), want.cpu()) <= 1e-5          # and later calls stay right
    finally:
        D.PERSISTENT_LSTM, D.PERSISTENT_LSTM_BWD = old


def test_resident_guard_orders_persistent_grids_that_do_not_fit_side_by_side():
    """ops_demucs._ResidentGuard: persistent grids in flight on OTHER streams of this process are accounted for; a new one that
    would push the resident workgroups past the CU count makes its stream wait (device-side) for the oldest of them, one that fits
    does not wait.  Checked with events around a long-running kernel standing in for the first grid."""
    from musicfpaugment_amd import ops_demucs as D
    dev = torch.device("cuda", torch.cuda.current_device())
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    guard = D._ResidentGuard()
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    big = torch.randn(8192, 8192, device="cuda")
    torch.cuda.synchronize()

    def run(first_wgs, second_wgs):
        """-> how many device-side waits did the guard put in front of the second launch (stream b) while the first (stream a)
        was still running?  (Observing the ordering through event timestamps does not work: two streams may share a hardware
        queue, which orders them anyway.)"""
        waits = []
        orig = torch.cuda.Stream.wait_event
        with torch.cuda.stream(a):
            done = guard.admit(dev, first_wgs)
            for _ in range(8):
                big @ big                                              # ~100 ms of work on stream a
            done()
        torch.cuda.Stream.wait_event = lambda self, ev: (waits.append(self.cuda_stream), orig(self, ev))[1]
        try:
            with torch.cuda.stream(b):
                guard.admit(dev, second_wgs)()
        finally:
            torch.cuda.Stream.wait_event = orig
        torch.cuda.synchronize()
        assert all(w == b.cuda_stream for w in waits)
        return len(waits)

    assert run(cus - 64, 128) == 1                                     # 192 + 128 > 256 CUs: b waits (device-side) for a's grid
    assert run(64, 64) == 0                                            # fits side by side: no wait
    assert run(cus, 1) == 1 and run(0, cus) == 0                       # a full chip admits nothing beside it; nothing in flight, no wait
    assert D.lstm_seq_workgroups(256, 768) == 4 * 48 and D.lstm_seq_workgroups(64, 768, cus // 2) == 2 * 48
    assert D.lstm_seq_workgroups(96, 768, cus // 2) == 2 * 48 and D.lstm_seq_workgroups(4096, 768) == 0      # too many slabs: per-step path
