#!/usr/bin/env python3
"""The other kernels whose ISA contains packed fp32 instructions with a low lane selecting a high half (profiles/r02_pk_fma_op_sel.md):
run each at full size several times, compare the runs bit for bit and against a float64 reference."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import torch.nn.functional as F
from musicfpaugment_amd import ops_demucs as D
from musicfpaugment_amd._lib import check, lib, ptr, stream
L = lib()
g = torch.Generator().manual_seed(0)
B, Lout, C = 64, 64084, 48
Lin = 4 * (Lout - 1) + 8
x = torch.randn(B, Lin, generator=g).cuda()
w = (torch.randn(8, C, generator=g) / np.sqrt(8)).cuda()
b = (torch.randn(C, generator=g) * 0.3).cuda()
def report(name, runs, want=None, tol=1e-5):
    same = all(torch.equal(runs[0], r) for r in runs[1:])
    msg = f"{name}: {len(runs)} runs bit-identical: {same}"
    if want is not None:
        err = float((runs[0].double().cpu() - want).abs().max()); sc = float(want.abs().max())
        msg += f"; max |error| vs float64 {err:.3e} (scale {sc:.3f})"
    print(msg, flush=True)
# conv1d_c1_kernel (v_pk_fma_f32 op_sel:[0,1,0])
runs = []
for _ in range(4):
    y = torch.full((B, Lout, C), float("nan"), device="cuda")
    check(L.mfpa_conv1d_c1_relu(ptr(x), B, Lin, Lout, C, ptr(w), ptr(b), ptr(y), stream()), "c1"); torch.cuda.synchronize(); runs.append(y)
want = F.relu(F.conv1d(x[:2].double().cpu()[:, None], w.t().double().cpu()[:, None, :], b.double().cpu(), stride=4)).permute(0, 2, 1)
report("conv1d_c1_kernel", [r[:2] for r in runs], want)
print("   all clips identical across runs:", all(torch.equal(runs[0], r) for r in runs[1:]))
# c1_wgrad_kernel (v_pk_fma_f32 op_sel:[0,1,0]); accumulates with atomics -> compare with tolerance only
gz = torch.randn(B, Lout, C, generator=g).cuda() * 0.01
outs = []
for _ in range(3):
    dw = torch.zeros(8, C, device="cuda")
    check(L.mfpa_c1_wgrad(ptr(x), Lin, ptr(gz), C, Lout * C, B, Lout, C, ptr(dw), stream()), "c1w"); torch.cuda.synchronize(); outs.append(dw)
xs = x[:, :4 * (Lout - 1) + 8].double().cpu().unfold(1, 8, 4)          # (B, Lout, 8)
wantw = torch.einsum("btj,btc->jc", xs, gz.double().cpu())
print(f"c1_wgrad_kernel: max relative error of three runs vs float64 {[float(((o.double().cpu() - wantw).abs().max() / wantw.abs().max())) for o in outs]}", flush=True)
# upsample2 / downsample2 (v_pk_mul_f32 op_sel:[0,1])
ker = D.sinc_kernel("cuda")
T = 64085
xs_ = torch.randn(256, T, generator=g).cuda()
ru, rd = [], []
for _ in range(4):
    y = torch.empty(256, 2 * T, device="cuda"); check(L.mfpa_upsample2(ptr(xs_), 256, T, ptr(ker), ptr(y), stream()), "up"); ru.append(y)
    z = torch.empty(256, (T + 1) // 2, device="cuda"); check(L.mfpa_downsample2(ptr(xs_), 256, T, ptr(ker), ptr(z), (T + 1) // 2, 0, 0, stream()), "down"); rd.append(z)
torch.cuda.synchronize()
report("upsample2_kernel", ru); report("downsample2_kernel", rd)
