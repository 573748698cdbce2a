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
