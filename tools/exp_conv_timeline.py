#!/usr/bin/env python3
"""In-kernel timeline of the 64-channel full-resolution launches (experiments build, libmfpa_exp.so): s_memrealtime stamps of every workgroup's
first wave at kernel start / patch staged / first tiles in LDS / main loop done / end -> mean cycles per phase and workgroup lifetime."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicfpaugment_amd import _lib
_lib.set_library_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "musicfpaugment_amd", "libmfpa_exp.so"))
from musicfpaugment_amd import ops_unet as K
h = ctypes.CDLL(_lib.LIB_PATH)
B, H, W = 64, 257, 251
sc = torch.ones(64, device="cuda"); sh = torch.zeros(64, device="cuda")
spec = torch.rand(B, H, W, device="cuda", dtype=torch.float64); den = torch.ones(B, device="cuda", dtype=torch.float64)
w1 = torch.randn(9, 64, device="cuda") * 0.1
w64 = K.split_bf16x3(torch.randn(9, 64, 64, device="cuda") * 0.05)
w128 = K.split_bf16x3(torch.randn(9, 64, 128, device="cuda") * 0.05)
x = torch.randn(B, H, W, 64, device="cuda"); u = torch.randn(B, H - 1, W - 1, 64, device="cuda")
wo = torch.randn(64, device="cuda")
runs = [("inc.3  c1src + pool", lambda: K.conv3x3_fused(None, w64, sc, sh, precision=1, pool=True, c1=dict(spec64=spec, denom=den, w=w1, scale=sc, shift=sh))),
        ("up4.0  concat", lambda: K.conv3x3_fused(x, w128, sc, sh, x1=u, precision=1)),
        ("up4.3  + OutConv", lambda: K.conv3x3_fused(x, w64, sc, sh, precision=1, out1x1=(wo, 0.1), store=False))]
def wd(H_, W_, ci, co):
    xx = torch.randn(B, H_, W_, ci, device="cuda"); w = torch.randn(9, co, ci, device="cuda") * 0.05
    w3, wf = K.split_bf16x3(w), (K.frag_layout(), K.split_bf16x3_frag(w, K.frag_layout()))
    s1 = torch.ones(co, device="cuda"); h1 = torch.zeros(co, device="cuda")
    return lambda: K.conv3x3_fused(xx, w3, s1, h1, precision=1, wf=wf)
if "--direct" in sys.argv:          # the weights-direct kernel on representative layers
    runs = [("d1.3  128->128 @128x125", wd(128, 125, 128, 128)), ("up3.0 256->128 @128x125", wd(128, 125, 256, 128)),
            ("d2.3  256->256 @64x62", wd(64, 62, 256, 256)), ("d3.3  512->512 @32x31", wd(32, 31, 512, 512)),
            ("up1.0 1024->512 @32x31", wd(32, 31, 1024, 512)), ("d4.3 1024->1024 @16x15", wd(16, 15, 1024, 1024))]
nwg = 8 * 33 * B
buf = torch.zeros(nwg * 8, dtype=torch.int64, device="cuda")
for name, fn in runs:
    fn(); torch.cuda.synchronize()
    buf.zero_()
    assert h.mfpa_exp_conv_stamps(ctypes.c_void_p(buf.data_ptr())) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    h.mfpa_exp_conv_stamps(ctypes.c_void_p(0))
    t = buf.view(nwg, 8).double()
    ok = t[:, 4] > 0
    t = t[ok]
    d = [(t[:, i + 1] - t[:, i]).mean().item() for i in range(4)]
    life = (t[:, 4] - t[:, 0]).mean().item()
    span = (t[:, 4].max() - t[:, 0].min()).item()
    us = e0.elapsed_time(e1) * 1e3
    print(f"{name:22s} {us:8.1f} us  | s_memrealtime ticks (10 ns): start->staged {d[0]:7.0f}  ->first tiles {d[1]:7.0f}  ->loop done {d[2]:7.0f}  ->end {d[3]:7.0f}  "
          f"| lifetime {life:7.0f} ticks = {life / 100:6.1f} us; kernel span {span / 100:7.1f} us; workgroups {int(ok.sum())}; mean resident = {life * int(ok.sum()) / span / 256:.2f} per CU")
