#!/usr/bin/env python3
"""Algebra check (CPU, float64) for folding a decoder level's ConvTranspose2d(k = 2, s = 2) into the 3x3 convolution that consumes it
(training/unet.py:41-65: up -> pad to the skip's size -> cat([skip, up]) -> conv3x3): the up half of that convolution is, per output PHASE
(py, px) = (Y % 2, X % 2), a 2 x 2 convolution of the LOW-resolution tensor with composite weights

    Wc[py, px][ry, rx][co][ci] = sum over the 3x3 taps (a, b) that land on low-resolution offset (ry, rx) for this phase of
                                 sum_cu W3[a, b][co][cu] * Wt[(py + a) % 2, (px + b) % 2][cu][ci]

(4 taps x C_low instead of 9 taps x C_up products per output: 512 vs 576 for up4: -11 %, and no up-sampled tensor in memory), plus a bias
term that depends only on which of the nine taps fall inside the up-sampled extent (interior: one constant vector; border rows / columns and the
padding row / column of odd sizes: a 4 x 4 table).  Zero padding of the low-resolution tensor reproduces the zero padding of the up-sampled one.
Prints the maximum deviation from the reference formulation on an odd-sized example (257 x 251 skip from a 128 x 125 low-resolution tensor)."""
import torch
import torch.nn.functional as F
torch.manual_seed(0)
dd = torch.float64
B, Clow, Cup, Cskip, Cout = 1, 8, 4, 4, 6
Hl, Wl, H, W = 12, 11, 25, 23                      # 2 Hl = 24 < H = 25, 2 Wl = 22 < W = 23: one padding row / column (diff // 2 = 0 in front)
xl = torch.randn(B, Clow, Hl, Wl, dtype=dd)
skip = torch.randn(B, Cskip, H, W, dtype=dd)
Wt = torch.randn(Clow, Cup, 2, 2, dtype=dd)        # ConvTranspose2d weight (in, out, kH, kW)
bt = torch.randn(Cup, dtype=dd)
W3 = torch.randn(Cout, Cskip + Cup, 3, 3, dtype=dd)
# ---- reference (unet.py:58-65)
up = F.conv_transpose2d(xl, Wt, bt, stride=2)
dY, dX = H - up.shape[2], W - up.shape[3]
up_p = F.pad(up, [dX // 2, dX - dX // 2, dY // 2, dY - dY // 2])
ref = F.conv2d(torch.cat([skip, up_p], dim=1), W3, padding=1)
# ---- composite
oy, ox = dY // 2, dX // 2                           # offset of the up-sampled extent inside the output grid
W3u = W3[:, Cskip:]                                 # (Cout, Cup, 3, 3)
out = F.conv2d(skip, W3[:, :Cskip], padding=1)      # the skip half stays an ordinary 3x3 convolution
valid = torch.zeros(H, W, dtype=dd)
valid[oy:oy + 2 * Hl, ox:ox + 2 * Wl] = 1.0         # where the up-sampled tensor (and its bias) exists
xl_p = F.pad(xl, [2, 2, 2, 2])                      # zero-padded low-resolution tensor (two pixels: the padding row maps past the edge)
for Y in range(H):
    for X in range(W):
        acc = torch.zeros(B, Cout, dtype=dd)
        for a in (-1, 0, 1):
            for b in (-1, 0, 1):
                yy, xx = Y + a - oy, X + b - ox     # position in the up-sampled extent
                # low-resolution pixel and phase of that position (floor division also for the negative ones: they hit the zero padding)
                ly, lx, py, px = yy // 2, xx // 2, yy % 2, xx % 2
                comp = torch.einsum("ou,iu->oi", W3u[:, :, a + 1, b + 1], Wt[:, :, py, px])     # (Cout, Clow): one term of Wc
                inside = 0 <= yy < 2 * Hl and 0 <= xx < 2 * Wl
                v = xl_p[:, :, ly + 2, lx + 2] if (-2 <= ly < Hl + 2 and -2 <= lx < Wl + 2) else torch.zeros(B, Clow, dtype=dd)
                acc += v @ comp.T
                if inside:
                    acc += (W3u[:, :, a + 1, b + 1] @ bt)[None]
        out[:, :, Y, X] += acc
print("max |composite - reference|:", float((out - ref).abs().max()), " (reference scale", float(ref.abs().max()), ")")
# ---- how many distinct (phase, low-res offset) weight sets and bias vectors there are
sets = set()
for Y in range(oy, oy + 2 * Hl):
    for X in range(ox, ox + 2 * Wl):
        py, px = (Y - oy) % 2, (X - ox) % 2
        offs = tuple(sorted({(((Y + a - oy) // 2) - (Y - oy) // 2, ((X + b - ox) // 2) - (X - ox) // 2) for a in (-1, 0, 1) for b in (-1, 0, 1)}))
        sets.add((py, px, offs))
print("distinct (phase, low-resolution offsets) patterns inside the extent:", len(sets), "-> 4 phases x 4 taps each")
