"""CPU: the algebra of the folded decoder level (oracle.unet.upconv_composite*, the restatement behind mfpa_upconv_fused / mfpa_upconv_pack,
csrc/unet_up.hip) against the reference formulation of Up.forward (training/unet.py:58-65) in float64 -- odd and even sizes, the padding row /
column, borders, a bias, a per-channel scale, C_up != C_skip."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import unet as ou


def _reference(skip, low, wt, bt, w3):
    dd = torch.float64
    up = F.conv_transpose2d(low.to(dd), wt.to(dd), bt.to(dd), stride=2)
    dY, dX = skip.shape[2] - up.shape[2], skip.shape[3] - up.shape[3]
    up = F.pad(up, [dX // 2, dX - dX // 2, dY // 2, dY - dY // 2])          # unet.py:60-63
    return F.conv2d(torch.cat([skip.to(dd), up], dim=1), w3.to(dd), padding=1)


@pytest.mark.parametrize("H,W,Hl,Wl", [(25, 23, 12, 11), (24, 22, 12, 11), (25, 22, 12, 11), (9, 33, 4, 16), (4, 5, 2, 2)])
def test_composite_equals_the_reference_formulation(H, W, Hl, Wl):
    g = torch.Generator().manual_seed(H * 100 + W)
    dd = torch.float64
    B, Cs, Cu, Cl, Cout = 2, 5, 3, 7, 4
    skip = torch.randn(B, Cs, H, W, generator=g, dtype=dd)
    low = torch.randn(B, Cl, Hl, Wl, generator=g, dtype=dd)
    wt = torch.randn(Cl, Cu, 2, 2, generator=g, dtype=dd)
    bt = torch.randn(Cu, generator=g, dtype=dd)
    w3 = torch.randn(Cout, Cs + Cu, 3, 3, generator=g, dtype=dd)
    want = _reference(skip, low, wt, bt, w3)
    wc, bias = ou.upconv_composite(w3, wt, bt)
    got = ou.upconv_composite_forward(skip, low, w3, wc, bias)
    assert float((got - want).abs().max()) < 1e-12 * float(want.abs().max())
    # a per-output-channel scale (the folded eval BatchNorm) commutes with the fold
    sc = torch.rand(Cout, generator=g, dtype=dd) + 0.5
    wcs, biass = ou.upconv_composite(w3, wt, bt, sc)
    gots = ou.upconv_composite_forward(skip, low, w3 * sc[:, None, None, None], wcs, biass)
    assert float((gots - want * sc[None, :, None, None]).abs().max()) < 1e-12 * float(want.abs().max())


def test_composite_tap_structure():
    """Each phase uses exactly four low-resolution taps; the centre tap (0, 0) of the low-resolution grid serves all four phases, the edge taps
    two, the corner taps one: 16 composite matrices in all (the kernel's weight image has no zero blocks to skip)."""
    g = torch.Generator().manual_seed(3)
    w3 = torch.randn(4, 6, 3, 3, generator=g)
    wt = torch.randn(5, 3, 2, 2, generator=g)
    wc, bias = ou.upconv_composite(w3, wt, torch.zeros(3))
    assert wc.shape == (16, 4, 5) and bias.shape == (4, 4, 4)
    assert all(float(wc[t].abs().max()) > 0 for t in range(16)) and float(bias.abs().max()) == 0.0
    used = {}
    for py in range(2):
        for px in range(2):
            for ty in range(2):
                for tx in range(2):
                    used.setdefault((ty - 1 + py, tx - 1 + px), []).append((py, px))
    assert sorted(len(v) for v in used.values()) == [1, 1, 1, 1, 2, 2, 2, 2, 4]
