#!/usr/bin/env python3
"""Launch c1_glu_kernel out of hand-edited code objects (ISA-level experiments for profiles/r02_pk_fma_op_sel.md) and compare the
result bit for bit with the product's 128 x 64-tile GEMM form.   usage: run_hsaco_head.py a.hsaco [b.hsaco ...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from musicfpaugment_amd import ops_demucs as D
from musicfpaugment_amd._lib import ptr
hip = ctypes.CDLL("libamdhip64.so")
KNAME = b"_ZN12_GLOBAL__N_113c1_glu_kernelEPKfiiS1_S1_S1_S1_Pfi"
C = 48
g = torch.Generator().manual_seed(0)
w0 = (torch.randn(8, C, generator=g) / np.sqrt(8)).cuda(); b0 = (torch.randn(C, generator=g) * 0.3).cuda()
gw, gb = D._pack_glu(torch.randn(2 * C, C, generator=g) / np.sqrt(C), torch.randn(2 * C, generator=g) * 0.3); gw, gb = gw.cuda(), gb.cuda()
B, Lout = 64, 64084
Lin = 4 * (Lout - 1) + 8
x = torch.randn(B, Lin, generator=g).cuda()
ref = torch.empty(B, Lout, C, device="cuda")
D.gemm(0, C, Lout * C, B, Lout, gw, gb, C, D._p(ref), C, Lout * C, mode=1, c1=(x, w0, b0))
torch.cuda.synchronize()
tiles = (Lout + 127) // 128
for path in sys.argv[1:]:
    mod = ctypes.c_void_p(); fn = ctypes.c_void_p()
    assert hip.hipModuleLoad(ctypes.byref(mod), path.encode()) == 0, path
    assert hip.hipModuleGetFunction(ctypes.byref(fn), mod, KNAME) == 0
    res = []
    for _ in range(3):
        y = torch.full((B, Lout, C), float("nan"), device="cuda")
        args = [ctypes.c_void_p(ptr(x)), ctypes.c_int(Lin), ctypes.c_int(Lout), ctypes.c_void_p(ptr(w0)), ctypes.c_void_p(ptr(b0)),
                ctypes.c_void_p(ptr(gw)), ctypes.c_void_p(ptr(gb)), ctypes.c_void_p(ptr(y)), ctypes.c_int(tiles)]
        arr = (ctypes.c_void_p * len(args))(*[ctypes.cast(ctypes.byref(a), ctypes.c_void_p) for a in args])
        rc = hip.hipModuleLaunchKernel(fn, (tiles + 7) // 8, B, 1, 256, 1, 1, 0, None, arr, None)
        assert rc == 0, rc
        torch.cuda.synchronize()
        res.append(int((y != ref).sum()))
    print(f"{os.path.basename(path):28s} elements differing from the GEMM form in three runs: {res}", flush=True)
    hip.hipModuleUnload(mod)
