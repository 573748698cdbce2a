"""GPU parity: Demucs training step (forward with kept activations, L1 + MultiResolutionSTFTLoss, hand-written backward,
Adam) against torch autograd through the CPU oracle (oracle/demucs.py, oracle/loss.py) -- training/train.py:275-312."""
import os

import numpy as np
import pytest
import torch

from musicfpaugment_amd import synth
from musicfpaugment_amd.training.demucs_weights import formula_state_dict

pytestmark = pytest.mark.gpu


def _rel(a: torch.Tensor, b: torch.Tensor) -> float:
    return float((a.double() - b.double()).abs().sum() / b.double().abs().sum().clamp_min(1e-30))


@pytest.mark.parametrize("precision,tol", [(0, 1e-5), (1, 1e-4), (2, 1e-2)])
def test_gemm_tn_window_weight_gradient(precision, tol):
    """dW[co][j*cin + c] = sum_{b,t} dy[b][t][co] * x[b][4t + j][c]: the Conv1d(k8, s4) weight gradient as one TN GEMM
    (fp32 MFMA / bf16x3 / plain bf16 through the transposing LDS reads)."""
    from musicfpaugment_amd import ops_demucs_train as T
    g = torch.Generator().manual_seed(1)
    for (B, L, cin, cout) in [(3, 37, 48, 96), (2, 130, 96, 192), (2, 9, 48, 48), (2, 1000, 192, 384)]:
        Lin = 4 * (L - 1) + 8
        x = torch.randn(B, Lin, cin, generator=g)
        dy = torch.randn(B, L, cout, generator=g)
        win = torch.stack([x[:, 4 * t:4 * t + 8, :].reshape(B, -1) for t in range(L)], dim=1)        # (B, L, 8 cin)
        want = torch.einsum("btm,btn->mn", dy.double(), win.double())
        out = torch.zeros(cout, 8 * cin, device="cuda")
        xd, dyd = x.cuda(), dy.cuda()
        fused = torch.zeros(cout, device="cuda")                  # the bias gradient read off the staged dy tiles
        T.gemm_tn(T._p(dyd), cout, L * cout, T._p(xd), 4 * cin, Lin * cin, out, 8 * cin, B, L, cout, 8 * cin, precision=precision,
                  colsum=fused)
        assert _rel(out.cpu(), want) < tol
        assert _rel(fused.cpu(), dy.double().sum((0, 1))) < 1e-5
        bias = torch.zeros(cout, device="cuda")
        T.colsum(T._p(dyd), B * L, cout, cout, bias)
        assert _rel(bias.cpu(), dy.double().sum((0, 1))) < 1e-5


def test_glu_bwd_and_mask_epilogue():
    import torch.nn.functional as F
    from musicfpaugment_amd import ops_demucs as D
    from musicfpaugment_amd import ops_demucs_train as T
    from musicfpaugment_amd._lib import check, lib, ptr, stream
    g = torch.Generator().manual_seed(2)
    B, L, C = 2, 50, 48
    a = torch.relu(torch.randn(B, L, C, generator=g)).requires_grad_(True)
    wg = (torch.randn(2 * C, C, generator=g) / np.sqrt(C)).requires_grad_(True)
    bg = torch.randn(2 * C, generator=g).requires_grad_(True)
    h = F.glu(a @ wg.t() + bg, dim=-1)
    dh = torch.randn(B, L, C, generator=g)
    h.backward(dh)
    wgp, bgp = D._pack_glu(wg.detach(), bg.detach())
    npad = wgp.shape[0]
    ad = a.detach().cuda()
    u = torch.empty(B, L, npad, device="cuda")
    hd = torch.empty(B, L, C, device="cuda")
    wgd, bgd = wgp.cuda(), bgp.cuda()
    D.gemm(D._p(ad), C, L * C, B, L, wgd, bgd, C, D._p(hd), C, L * C, mode=1, C2=D._p(u), ldc2=npad, strideC2=L * npad)
    assert _rel(hd.cpu(), h.detach()) < 1e-5
    dhd = dh.cuda()
    check(lib().mfpa_glu_bwd(ptr(u), B * L, npad, C, ptr(dhd), C, stream()), "glu_bwd")
    dwp = torch.zeros(npad, C, device="cuda")
    T.gemm_tn(T._p(u), npad, 0, T._p(ad), C, 0, dwp, C, 1, B * L, npad, C)
    dbp = torch.zeros(npad, device="cuda")
    T.colsum(T._p(u), B * L, npad, npad, dbp)
    dw, db = T.DemucsTrainEngine._unpack_glu(dwp, dbp, C)
    assert _rel(dw.cpu(), wg.grad) < 1e-5 and _rel(db.cpu(), bg.grad) < 1e-5
    # input gradient, masked by the ReLU that produced `a` (mode 3), unmasked copy through C2
    da, da_raw = torch.empty(B, L, C, device="cuda"), torch.empty(B, L, C, device="cuda")
    gwT = T._t_pad(wgd, C)
    D.gemm(D._p(u), npad, L * npad, B, L, gwT, None, C, D._p(da), C, L * C, mode=3, addend=D._p(ad), ldadd=C,
           strideAdd=L * C, C2=D._p(da_raw), ldc2=C, strideC2=L * C)
    assert _rel(da_raw.cpu(), a.grad) < 1e-5
    assert _rel(da.cpu(), a.grad * (a.detach() > 0)) < 1e-5


def test_downsample_adjoint_and_c1_kernels():
    from musicfpaugment_amd import ops_demucs as D
    from musicfpaugment_amd._lib import check, lib, ptr, stream
    from oracle import demucs as od
    g = torch.Generator().manual_seed(3)
    B, T_, keep = 2, 1001, 400
    x = torch.randn(B, T_, generator=g).requires_grad_(True)
    sc = torch.rand(B, generator=g) + 0.5
    y = od.downsample2(x)[:, :keep] * sc[:, None]
    dy = torch.randn(B, keep, generator=g)
    y.backward(dy)
    dx = torch.empty(B, T_, device="cuda")
    dyd, scd, ker = dy.cuda(), sc.cuda(), D.sinc_kernel("cuda")          # keep the device copies alive across the launch
    check(lib().mfpa_downsample2_adjoint(ptr(dyd), B, keep, keep, ptr(ker), ptr(scd), T_, ptr(dx), stream()), "adj")
    assert _rel(dx.cpu(), x.grad) < 1e-5
    # weight gradient of the 1 -> C convolution
    L, C = 333, 48
    Lin = 4 * (L - 1) + 8
    xs = torch.randn(B, Lin, generator=g)
    gr = torch.randn(B, L, C, generator=g)
    want = torch.stack([torch.einsum("bt,btc->c", xs[:, j:j + 4 * L:4].double(), gr.double()) for j in range(8)])
    dw = torch.zeros(8, C, device="cuda")
    xsd, grd = xs.cuda(), gr.cuda()
    check(lib().mfpa_c1_wgrad(ptr(xsd), Lin, ptr(grd), C, L * C, B, L, C, ptr(dw), stream()), "c1_wgrad")
    assert _rel(dw.cpu(), want) < 1e-5


def test_lstm_train_step_pair():
    """mfpa_lstm_step_train / mfpa_lstm_step_bwd over a short sequence vs autograd through the oracle's LSTM layer."""
    from musicfpaugment_amd._lib import check, lib, ptr, stream
    g = torch.Generator().manual_seed(4)
    B, Tn, H = 3, 5, 768
    wih = (torch.randn(4 * H, H, generator=g) * 0.03).requires_grad_(True)
    whh = (torch.randn(4 * H, H, generator=g) * 0.03).requires_grad_(True)
    b = (torch.randn(4 * H, generator=g) * 0.03).requires_grad_(True)
    x = torch.randn(B, Tn, H, generator=g).requires_grad_(True)
    h = torch.zeros(B, H); c = torch.zeros(B, H)
    outs = []
    for t in range(Tn):
        gt = x[:, t] @ wih.t() + b + h @ whh.t()
        i, f, gg, o = gt.chunk(4, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        outs.append(h)
    hs = torch.stack(outs, dim=1)
    dh = torch.randn(B, Tn, H, generator=g)
    hs.backward(dh)
    L = lib()
    dev = "cuda"
    gates = (x.detach() @ wih.detach().t() + b.detach()).contiguous().to(dev)            # input projection (B, Tn, 4H)
    whh_d = whh.detach().to(dev)
    grouped = whh_d.view(4, H // 16, 16, H).permute(1, 0, 2, 3).reshape(4 * H, H).contiguous()
    hseq, cseq = torch.empty(B, Tn, H, device=dev), torch.empty(B, Tn, H, device=dev)
    p = lambda t_, off=0: ptr(t_) + 4 * off
    for t in range(Tn):
        check(L.mfpa_lstm_step_train(p(hseq, (t - 1) * H) if t else 0, Tn * H, ptr(grouped), p(gates, t * 4 * H), Tn * 4 * H,
                                     p(cseq, (t - 1) * H) if t else 0, Tn * H, p(cseq, t * H), Tn * H, B, H, p(hseq, t * H), Tn * H,
                                     0, 0, 0, p(gates, t * 4 * H), Tn * 4 * H, stream()), "step_train")
    assert _rel(hseq.cpu(), hs.detach()) < 2e-5
    whhT = whh_d.t().contiguous()
    dc = torch.zeros(B, H, device=dev)
    dhd = dh.to(dev)
    for t in range(Tn - 1, -1, -1):
        check(L.mfpa_lstm_step_bwd(p(gates, (t + 1) * 4 * H) if t + 1 < Tn else 0, Tn * 4 * H, ptr(whhT), p(gates, t * 4 * H),
                                   Tn * 4 * H, p(cseq, t * H), Tn * H, p(cseq, (t - 1) * H) if t else 0, Tn * H, p(dhd, t * H),
                                   Tn * H, ptr(dc), B, H, stream()), "step_bwd")
    dg = gates.cpu().double()                                                               # now d loss / d gate pre-activations
    assert _rel(dg.sum((0, 1)), b.grad) < 1e-4
    assert _rel(torch.einsum("btg,bti->gi", dg, x.detach().double()), wih.grad) < 1e-4
    assert _rel(torch.einsum("btg,bti->gi", dg[:, 1:], hs.detach().double()[:, :-1]), whh.grad) < 1e-4
    assert _rel(dg @ wih.detach().double(), x.grad) < 1e-4


class _ReluWithMasks:
    """Stands in for torch.nn.functional inside oracle.demucs while one forward runs: the k-th ReLU multiplies by the k-th given
    mask instead of by (x > 0), and notes where the two disagree and how far from zero the pre-activation is there."""

    def __init__(self, F, masks):
        self._F, self.masks, self.k, self.disagree = F, masks, 0, []

    def __getattr__(self, name):
        return getattr(self._F, name)

    def relu(self, x):
        m = self.masks[self.k]
        self.k += 1
        dis = (x.detach() > 0) != m
        self.disagree.append((int(dis.sum()), float(x.detach().abs()[dis].max()) if bool(dis.any()) else 0.0))
        return x * m.to(x.dtype)


def _relu_masks(eng):
    """The ReLU decisions the device forward took (5 encoder convolutions, 4 transposed convolutions), as (B, C, L) masks."""
    return [(t > 0).permute(0, 2, 1).cpu() for t in list(eng.S["a"]) + list(eng.S["r"])]


def _oracle_step(sd, clean, aug, masks=None):
    """One float64 autograd step through the oracle.  The gradient of a ReLU network jumps where a pre-activation crosses zero,
    and among ~7e5 ReLU sites a few sit within float32 rounding of it: with `masks` (the device's decisions) those ties are taken
    the device's way -- after checking that they ARE ties (|pre-activation| < 2e-6, at most a handful) -- so the comparison is
    about the arithmetic, not about which side of zero a 2e-8 landed."""
    from oracle import demucs as od
    from oracle import loss as ol
    params = {k: v.clone().double().requires_grad_(True) for k, v in sd.items()}
    shim = None
    if masks is not None:
        shim = _ReluWithMasks(od.F, masks)
        od.F = shim
    try:
        pred = od.forward(aug.double(), params)[:, 0]
    finally:
        if shim is not None:
            od.F = shim._F
    if shim is not None:
        assert shim.k == len(masks)
        assert sum(n for n, _ in shim.disagree) <= 16 and max(v for _, v in shim.disagree) < 2e-6, shim.disagree
    l1 = torch.nn.functional.l1_loss(pred, clean.double())
    sc, mag, _ = ol.multi_resolution_stft_loss(pred, clean.double())
    (l1 + sc + mag).backward()
    return pred.detach(), (float(l1.detach()), float(sc.detach()), float(mag.detach())), {k: p.grad for k, p in params.items()}


def _inputs():
    n = 4000
    clean = torch.from_numpy(synth.batch(2, seed=31, n=n))
    aug = (clean + 0.05 * torch.from_numpy(synth.batch(2, seed=77, n=n))).float()
    return clean, aug


def test_train_step_gradients_vs_autograd():
    """Every parameter gradient of one step (B = 2, 0.5 s clips, exact-fp32 products) vs float64 autograd through the oracle;
    then the Adam update."""
    from musicfpaugment_amd.ops_demucs_train import DemucsTrainEngine
    sd = formula_state_dict(0)
    clean, aug = _inputs()
    eng = DemucsTrainEngine(sd, "cuda", precision=0)
    back = eng.state_dict()                                              # the reference layout round-trips
    assert all(torch.equal(back[k].cpu(), sd[k]) for k in sd)
    pred = eng.forward(aug.cuda())
    pred_w, (l1_w, sc_w, mag_w), grads = _oracle_step(sd, clean, aug, _relu_masks(eng))
    assert _rel(pred.cpu(), pred_w) < 1e-5
    l1, sc, mag, dpred = eng.loss_and_grad(pred, clean.cuda())
    np.testing.assert_allclose([float(l1), float(sc), float(mag)], [l1_w, sc_w, mag_w], rtol=1e-3)
    eng.backward(dpred)
    got = eng.grad_dict()
    worst = {k: _rel(got[k].cpu(), grads[k]) for k in sd}
    print(f"fp32: worst relative L1 {max(worst.values()):.2e} ({max(worst, key=worst.get)})")
    bad = {k: v for k, v in worst.items() if v > 1e-3}
    assert not bad, f"gradient mismatch: {bad}"
    # Adam: first step moves every parameter by -lr * sign(g) (bias-corrected m / sqrt(v) = g / |g|)
    before = eng.flat_p.clone()
    eng.adam_step()
    delta = (eng.flat_p - before).cpu()
    gflat = eng.flat_g.cpu()
    nz = gflat.abs() > 1e-6
    assert torch.allclose(delta[nz], -1e-3 * torch.sign(gflat[nz]), atol=2e-5)
    assert float(delta[gflat == 0].abs().max()) == 0.0          # zero-padded rows stay zero


@pytest.mark.parametrize("B,n", [(1, 4001), (3, 2731)])
def test_ragged_lengths_gradients_vs_autograd(B, n):
    """Odd clip lengths and batch sizes (valid_length padding, odd-length resampler adjoints, partial tiles everywhere)."""
    from musicfpaugment_amd.ops_demucs_train import DemucsTrainEngine
    sd = formula_state_dict(0)
    clean = torch.from_numpy(synth.batch(B, seed=41, n=n))
    aug = (clean + 0.05 * torch.from_numpy(synth.batch(B, seed=87, n=n))).float()
    eng = DemucsTrainEngine(sd, "cuda", precision=0)
    pred = eng.forward(aug.cuda())
    pred_w, _, grads = _oracle_step(sd, clean, aug, _relu_masks(eng))
    assert pred.shape == (B, n) and _rel(pred.cpu(), pred_w) < 1e-5
    _, _, _, dpred = eng.loss_and_grad(pred, clean.cuda())
    eng.backward(dpred)
    got = eng.grad_dict()
    worst = {k: _rel(got[k].cpu(), grads[k]) for k in sd}
    assert max(worst.values()) < 1e-3, {k: v for k, v in worst.items() if v > 1e-3}


@pytest.mark.parametrize("precision,wgrad,tol", [(1, 0, 1e-4), (1, 1, 1e-4), (1, 2, 1e-2)])
def test_backward_arithmetic_variants_on_one_forward_state(precision, wgrad, tol):
    """The backward pass is linear in dpred GIVEN the forward state (ReLU masks, gates).  On one fp32 forward state, the bf16x3
    input-gradient GEMMs (precision 1) and the bf16x3 / plain-bf16 weight-gradient GEMMs (wgrad 1 / 2: 2^-9 product rounding,
    averaged over the time steps) are compared with the exact-fp32 backward."""
    from musicfpaugment_amd.ops_demucs_train import DemucsTrainEngine
    sd = formula_state_dict(0)
    clean, aug = _inputs()

    def grads(prec, wg):
        eng = DemucsTrainEngine(sd, "cuda", precision=0, wgrad_precision=0)
        pred = eng.forward(aug.cuda())
        _, _, _, dpred = eng.loss_and_grad(pred, clean.cuda())
        eng.precision, eng.wgrad_precision = prec, wg               # arithmetic of the backward GEMMs only
        eng.backward(dpred)
        return eng.grad_dict()

    want, got = grads(0, 0), grads(precision, wgrad)
    worst = {k: _rel(got[k], want[k]) for k in want}
    print(f"backward precision {precision} wgrad {wgrad}: worst relative L1 {max(worst.values()):.2e} ({max(worst, key=worst.get)})")
    bad = {k: v for k, v in worst.items() if v > (tol if "weight" in k else 1e-4)}
    assert not bad, f"gradient mismatch: {bad}"


def test_bf16x3_forward_step_vs_autograd():
    """The whole step with bf16x3 forward GEMMs (the bench default).  The prediction stays within 1e-4; a gradient can differ
    from float64 autograd by more than the product rounding where a ReLU pre-activation lies within that rounding of zero and
    its mask flips (one flipped unit of the 768-channel level moves the deep layers' gradients by ~5e-3 on this 2-clip batch; a few
    flips by a few percent) --
    hence the looser bound here and the exact comparison of the backward arithmetic on a fixed forward state above."""
    from musicfpaugment_amd.ops_demucs_train import DemucsTrainEngine
    sd = formula_state_dict(0)
    clean, aug = _inputs()
    pred_w, (l1_w, sc_w, mag_w), grads = _oracle_step(sd, clean, aug)
    eng = DemucsTrainEngine(sd, "cuda", precision=1)
    pred = eng.forward(aug.cuda())
    assert _rel(pred.cpu(), pred_w) < 1e-4
    l1, sc, mag, dpred = eng.loss_and_grad(pred, clean.cuda())
    np.testing.assert_allclose([float(l1), float(sc), float(mag)], [l1_w, sc_w, mag_w], rtol=1e-3)
    eng.backward(dpred)
    got = eng.grad_dict()
    worst = {k: _rel(got[k].cpu(), grads[k]) for k in sd}
    cos = {k: float(torch.nn.functional.cosine_similarity(got[k].cpu().double().reshape(1, -1), grads[k].reshape(1, -1))) for k in sd}
    print(f"bf16x3 step: worst relative L1 {max(worst.values()):.2e} ({max(worst, key=worst.get)}), "
          f"lowest cosine {min(cos.values()):.6f} ({min(cos, key=cos.get)})")
    # which units flip depends on the last bits of the forward (it moved from 5.8e-3 to 4.5e-2 when the K = 48 / 96 layers went
    # from fp32 to bf16x3 products): bound the direction tightly and the size loosely
    assert min(cos.values()) > 0.995 and max(worst.values()) < 0.15


def test_train_step_runs_and_reduces_loss():
    from musicfpaugment_amd.ops_demucs_train import DemucsTrainEngine
    eng = DemucsTrainEngine(formula_state_dict(0), "cuda", lr=3e-4, precision=1)
    clean = torch.from_numpy(synth.batch(4, seed=5, n=8000)).cuda()
    aug = (clean + 0.05 * torch.from_numpy(synth.batch(4, seed=9, n=8000)).cuda()).contiguous()
    losses = [float(eng.train_step(clean, aug)) for _ in range(6)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


def test_train_step_golden_of_the_real_reference(golden):
    """g12: one training step of the REAL reference (Demucs.train(), L1 + MultiResolutionSTFTLoss, Adam) -- losses, gradient
    norms and the updated parameters."""
    from musicfpaugment_amd.ops_demucs_train import DemucsTrainEngine
    g = golden("g12_demucs_train_step")
    n = int(g["n"])
    clean = torch.from_numpy(synth.batch(2, seed=int(g["seed_clean"]), n=n))
    aug = (clean + float(g["noise_gain"]) * torch.from_numpy(synth.batch(2, seed=int(g["seed_noise"]), n=n))).float()
    sd = formula_state_dict(int(g["weight_seed"]))
    eng = DemucsTrainEngine(sd, "cuda", lr=float(g["lr"]), precision=0)
    before = eng.state_dict()
    loss = eng.train_step(clean.cuda(), aug.cuda())
    l1, sc, mag = (float(v) for v in eng.last_losses)
    np.testing.assert_allclose([l1, sc, mag], [float(g["l1"]), float(g["sc"]), float(g["mag"])], rtol=1e-4)
    assert abs(float(loss) - (float(g["l1"]) + float(g["sc"]) + float(g["mag"]))) < 1e-4
    grads, after = eng.grad_dict(), eng.state_dict()
    names = [str(k) for k in g["names"]]
    gn = np.array([float(grads[k].double().norm()) for k in names])
    np.testing.assert_allclose(gn, g["grad_norm"], rtol=5e-3)
    for r, k in enumerate(names):
        m = min(8, sd[k].numel())
        gh = g["grad_head"][r][:m]
        big = np.abs(gh) > 1e-5                                    # Adam's first step is -lr * sign(g): compare where the sign is safe
        got = after[k].reshape(-1)[:m].cpu().numpy()
        np.testing.assert_array_equal(before[k].reshape(-1)[:m].cpu().numpy(), g["param_head_before"][r][:m])
        np.testing.assert_allclose(got[big], g["param_head_after"][r][:m][big], rtol=0, atol=3e-6)


def test_trainer_audio_branch():
    """training.train.Trainer(input_type="audio") drives the engine; Demucs.train() forward goes through it too."""
    from musicfpaugment_amd.training.model import Demucs
    from musicfpaugment_amd.training.train import Trainer
    net = Demucs()
    net.load_state_dict(formula_state_dict(0))

    def loader(seed0):
        k = 0
        while True:
            clean = torch.from_numpy(synth.batch(2, seed=seed0 + k, n=4000))
            aug = clean + 0.05 * torch.from_numpy(synth.batch(2, seed=seed0 + 500 + k, n=4000))
            k += 1
            yield clean.unsqueeze(-1), aug.float().unsqueeze(-1)           # (B, T, 1) like the reference's loader

    tr = Trainer(net, loader(40), loader(90), train_steps=3, val_steps=2, device="cuda", input_type="audio", learning_rate=3e-4)
    start, start_m = Trainer(net, loader(40), loader(90), val_steps=2, device="cuda", input_type="audio").start_epoch()
    gen, want = loader(90), 0.0                               # train.py:470-578: un-denoised inputs, all val_steps batches
    for _ in range(2):
        c, a = next(gen)
        want += float(torch.mean(torch.abs(a.double() - c.double()))) / 2
    assert abs(start["l1_loss"] - want) < 1e-7 and np.isfinite(start_m["psnr"])
    assert abs(start["loss"] - (start["l1_loss"] + start["sc_loss"] + start["mag_loss"])) < 1e-9
    out = tr.train_epoch(1)
    assert np.isfinite(out["loss"]) and set(out) == {"loss", "l1_loss", "sc_loss", "mag_loss"}
    val, _ = tr.validation_epoch()
    assert np.isfinite(val["loss"])
    # the module's parameters follow the engine after a sync, and eval forward uses them
    tr.engine.sync_to_module()
    sd = net.state_dict()
    assert not torch.equal(sd["encoder.0.0.weight"].cpu(), formula_state_dict(0)["encoder.0.0.weight"])
    net.eval()
    y = net(torch.from_numpy(synth.batch(1, seed=3, n=4000)).cuda())
    assert y.shape == (1, 1, 4000) and bool(torch.isfinite(y).all())
    # checkpoints with the reference's keys (train.py:197-221) round-trip the parameters and the optimiser state
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        tr.ckpt_path = tmp
        tr.save_checkpoint(val["loss"])
        ck = torch.load(os.path.join(tmp, "best_epoch.pt"))
        assert set(ck["model_state_dict"].keys()) == set(formula_state_dict(0).keys())
        net2 = Demucs()
        tr2 = Trainer(net2, loader(40), loader(90), train_steps=3, val_steps=2, device="cuda", input_type="audio", ckpt_path=tmp)
        assert tr2.load_checkpoint()
        assert torch.equal(tr2.engine.flat_p, tr.engine.flat_p) and torch.equal(tr2.engine.flat_m, tr.engine.flat_m)
        assert tr2.engine.step_count == tr.engine.step_count


@pytest.mark.parametrize("B,Tn,H", [(70, 9, 256), (64, 12, 768), (33, 7, 512), (130, 6, 768)])
def test_persistent_lstm_backward_layer(B, Tn, H):
    """mfpa_lstm_layer_bwd_seq (one persistent launch: W_hh^T rows in registers on 16 x 16 x 32 MFMAs, dgates exchanged in split form,
    slab barriers in device memory) against the per-step kernels on the same saved state, whole range and two chained ranges
    (the chunked pipeline's use), ragged slabs; and against the float64 recurrence of the same backward."""
    from musicfpaugment_amd._lib import check, lib, ptr, stream
    from musicfpaugment_amd import ops_demucs as D
    L = lib()
    g = torch.Generator().manual_seed(B * Tn)
    gates0 = torch.cat([torch.rand(B, Tn, H, generator=g), torch.rand(B, Tn, H, generator=g), torch.rand(B, Tn, H, generator=g) * 2 - 1,
                        torch.rand(B, Tn, H, generator=g)], dim=2).contiguous()                 # [sig i | sig f | tanh g | sig o]
    cseq = (torch.randn(B, Tn, H, generator=g) * 0.7).contiguous()
    dhout = (torch.randn(B, Tn, H, generator=g) * 0.1).contiguous()
    whh = torch.randn(4 * H, H, generator=g) / np.sqrt(H)
    whhT = whh.t().contiguous()                                                                # (H, 4H)
    # float64 recurrence
    W = whh.double()
    dgn = torch.zeros(B, 4 * H, dtype=torch.float64)
    dc = torch.zeros(B, H, dtype=torch.float64)
    want = torch.zeros(B, Tn, 4 * H, dtype=torch.float64)
    for t in range(Tn - 1, -1, -1):
        vi, vf, vg, vo = [x.double() for x in gates0[:, t].split(H, dim=1)]
        ct = cseq[:, t].double()
        cp = cseq[:, t - 1].double() if t else torch.zeros_like(ct)
        dh = dhout[:, t].double() + dgn @ W
        tc = torch.tanh(ct)
        dO = dh * tc * vo * (1 - vo)
        dcv = dc + dh * vo * (1 - tc * tc)
        di = dcv * vg * vi * (1 - vi)
        df = dcv * cp * vf * (1 - vf)
        dg = dcv * vi * (1 - vg * vg)
        dc = dcv * vf
        dgn = torch.cat([di, df, dg, dO], dim=1)
        want[:, t] = dgn
    dev = lambda t: t.clone().cuda()
    whhT_d, cseq_d, dhout_d = dev(whhT), dev(cseq), dev(dhout)
    outs = {}
    nbytes = __import__("ctypes").c_longlong(0)
    check(L.mfpa_lstm_bwd_seq_work_bytes(B, H, __import__("ctypes").addressof(nbytes)), "bytes")
    for name, ranges, seq, budget in [("steps", [(0, Tn)], False, 0), ("seq", [(0, Tn)], True, 0), ("seq-two-ranges", [(Tn // 2, Tn), (0, Tn // 2)], True, 0),
                                      ("seq-half-the-chip", [(0, Tn)], True, 128), ("seq-64-clip-slabs", [(0, Tn)], True, 48 * ((B + 63) // 64))]:
        gd, dcs = dev(gates0), torch.full((B, H), float("nan"), device="cuda")
        work = torch.zeros(nbytes.value // 4, dtype=torch.int32, device="cuda")
        for (a, b) in ranges:
            if seq:
                check(L.mfpa_lstm_layer_bwd_seq(ptr(whhT_d), ptr(gd), ptr(cseq_d), ptr(dhout_d), ptr(dcs), B, Tn, H, a, b, budget, ptr(work), stream()), "seq")
            else:
                check(L.mfpa_lstm_layer_bwd_range(ptr(whhT_d), ptr(gd), ptr(cseq_d), ptr(dhout_d), ptr(dcs), B, Tn, H, a, b, stream()), "range")
        torch.cuda.synchronize()
        assert int(work[L.mfpa_lstm_seq_error_offset() // 4]) == 0
        outs[name] = (gd.cpu(), dcs.cpu())
    scale = float(want.abs().max())
    for name, (gd, dcs) in outs.items():
        err = float((gd.double() - want).abs().max())
        assert err < 2e-4 * scale, (name, err, scale)
        assert float((dcs.double() - dc).abs().max()) < 2e-4 * max(1.0, float(dc.abs().max())), name
    for name in ("seq", "seq-two-ranges", "seq-half-the-chip", "seq-64-clip-slabs"):
        assert float((outs[name][0] - outs["steps"][0]).abs().max()) < 1e-4 * scale
