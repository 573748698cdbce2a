#!/usr/bin/env python3
"""Reproduces what profiles/r02_pk_fma_op_sel.md describes: c1_glu_kernel (csrc/demucs.hip) with its first convolution written as
`v += x * w` -- hipcc then emits v_pk_fma_f32 ... op_sel:[0,1,0] -- returns sporadically wrong rows once two workgroups share a CU.
Builds musicfpaugment_amd/libmfpa_exp.so with -DMFPA_HEAD_PACKED_FMA unless --product is given (the shipped scalar-FMA form).
usage: head_kernel_two_wg_per_cu.py [--product]
       MFPA_HEAD_LDS=40000 head_kernel_two_wg_per_cu.py      (extra dynamic LDS -> one workgroup per CU -> no differences)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from musicfpaugment_amd import _lib
if "--build" in sys.argv:                                  # (run once where hipcc is: the GPU boxes use the prebuilt file)
    from musicfpaugment_amd.csrc import build
    build.build(verbose=False, experiments=True, extra_flags=[] if "--product" in sys.argv else ["-DMFPA_HEAD_PACKED_FMA"], force=True)
    sys.exit(0)
exp = os.path.join(os.path.dirname(_lib.__file__), "libmfpa_exp.so")
_lib.set_library_path(exp)
from musicfpaugment_amd import ops_demucs as D
from musicfpaugment_amd._lib import ptr, stream
h = ctypes.CDLL(exp)
V = ctypes.c_void_p
h.mfpa_conv1d_c1_glu.argtypes = [V, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, V, V, V, V, V, V]
C = 48
g = torch.Generator().manual_seed(0)
w0 = (torch.randn(8, C, generator=g) / np.sqrt(8)).cuda()
b0 = (torch.randn(C, generator=g) * 0.3).cuda()
w1 = torch.randn(2 * C, C, generator=g) / np.sqrt(C)
b1 = torch.randn(2 * C, generator=g) * 0.3
gw, gb = D._pack_glu(w1, b1); gw, gb = gw.cuda(), gb.cuda()
for B, Lout in [(1, 64084), (64, 1024), (8, 64084), (64, 64084)]:
    Lin = 4 * (Lout - 1) + 8
    x = torch.randn(B, Lin, generator=g).cuda()
    def fused():
        y = torch.full((B, Lout, C), float("nan"), device="cuda")
        rc = h.mfpa_conv1d_c1_glu(ptr(x), B, Lin, Lout, C, ptr(w0), ptr(b0), ptr(gw), ptr(gb), ptr(y), stream())
        assert rc == 0, rc
        torch.cuda.synchronize(); return y
    def unfused():
        y = torch.full((B, Lout, C), float("nan"), device="cuda")
        D.gemm(0, C, Lout * C, B, Lout, gw, gb, C, D._p(y), C, Lout * C, mode=1, c1=(x, w0, b0))
        torch.cuda.synchronize(); return y
    ref = unfused()
    ys = [fused() for _ in range(4)]
    print(f"B {B} Lout {Lout} ({B * ((Lout + 1023) // 1024)} workgroups): elements differing from the GEMM form per run "
          f"{[int((y != ref).sum()) for y in ys]}, max |diff| {max(float((y - ref).abs().max()) for y in ys):.3e}", flush=True)
