#!/usr/bin/env python3
"""Time c1_glu_kernel of libmfpa_exp.so (whatever -DMFPA_HEAD_PACKED_FMA it was built with) against the product library's and the GEMM form."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from musicfpaugment_amd import _lib
from musicfpaugment_amd import ops_demucs as D
from musicfpaugment_amd._lib import ptr, stream
root = os.path.dirname(_lib.__file__)
V = ctypes.c_void_p
libs = {n: ctypes.CDLL(os.path.join(root, f)) for n, f in (("product", "libmfpa.so"), ("experiment", "libmfpa_exp.so"))}
for h in libs.values():
    h.mfpa_conv1d_c1_glu.argtypes = [V, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, V, V, V, V, V, V]
C = 48
g = torch.Generator().manual_seed(0)
w0 = (torch.randn(8, C, generator=g) / np.sqrt(8)).cuda(); b0 = (torch.randn(C, generator=g) * 0.3).cuda()
gw, gb = D._pack_glu(torch.randn(2 * C, C, generator=g) / np.sqrt(C), torch.randn(2 * C, generator=g) * 0.3); gw, gb = gw.cuda(), gb.cuda()
B, Lout = 256, 64084
Lin = 4 * (Lout - 1) + 8
x = torch.randn(B, Lin, generator=g).cuda()
y = torch.empty(B, Lout, C, device="cuda")
fns = {n: (lambda h=h: h.mfpa_conv1d_c1_glu(ptr(x), B, Lin, Lout, C, ptr(w0), ptr(b0), ptr(gw), ptr(gb), ptr(y), stream())) for n, h in libs.items()}
fns["GEMM form"] = lambda: D.gemm(0, C, Lout * C, B, Lout, gw, gb, C, D._p(y), C, Lout * C, mode=1, c1=(x, w0, b0))
for rep in range(2):
    for name, f in fns.items():
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        print(name, round(e0.elapsed_time(e1) / 5, 3), "ms", flush=True)
