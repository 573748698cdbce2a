#!/usr/bin/env python3
"""The transposed convolution's weight gradient as two strided TN GEMMs (mfpa_gemm_tn) against the current kernel (mfpa_wgrad_mfma mode 1):
dW[(ty, tx)][co][ci] = sum_{b, y, x} d_u[b, 2y + ty, 2x + tx, co] * a[b, y, x, ci]  ==  for ty in 0, 1:  C[tx * Cout + co][ci] += sum_{g = (b, y), x}
A[g][x][tx * Cout + co] * Bm[g][x][ci] with A rows 2 * Cout floats apart (d_u's row 2y + ty) and Bm = a."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicfpaugment_amd import ops_train as T
from musicfpaugment_amd.ops_demucs_train import gemm_tn
def ev(fn, n=10):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n
B = 64
for name, H, W, Cin, Cout in (("up4.up", 128, 125, 128, 64), ("up3.up", 64, 62, 256, 128), ("up2.up", 32, 31, 512, 256), ("up1.up", 16, 15, 1024, 512)):
    a = torch.relu(torch.randn(B, H, W, Cin, device="cuda"))
    du = torch.randn(B, 2 * H, 2 * W, Cout, device="cuda")
    a16, du16 = a.bfloat16(), du.bfloat16()
    dw_ref = torch.zeros(4, Cout, Cin, device="cuda")
    def cur():
        T.wgrad_mfma(du, a, dw_ref, Cout, mode=1, precision=2, dz_bf16=du16, x0_bf16=a16)
    dw_ref.zero_(); cur(); want = dw_ref.clone()
    dw = torch.zeros(4, Cout, Cin, device="cuda")
    def new(prec=2):
        for ty in (0, 1):
            gemm_tn(du.data_ptr() + 4 * ty * 2 * W * Cout, 2 * Cout, 2 * 2 * W * Cout, a.data_ptr(), Cin, W * Cin, dw[2 * ty:], Cin, B * H, W, 2 * Cout, Cin, precision=prec)
    dw.zero_(); new(); got = dw.clone()
    err = float((got - want).abs().sum() / want.abs().sum())
    t_cur, t_new = ev(cur), ev(new)
    gf = 2.0 * B * H * W * 4 * Cout * Cin / 1e9
    print(f"{name}: current {t_cur:7.1f} us ({gf / t_cur * 1e-3:.3f} PFLOP/s)   two gemm_tn (fp32 operands, plain bf16 products) {t_new:7.1f} us ({gf / t_new * 1e-3:.3f} PFLOP/s)   rel L1 {err:.2e}")
