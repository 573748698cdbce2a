#!/usr/bin/env python3
"""Tap-level timeline of conv_wd16_kernel<8, 32, false, 4> (experiments build): wave 0 of one workgroup stamps s_memtime at every tap start,
after the chunk barrier, and around the epilogue, for its first tiles.  Prints shader-clock cycles per tap position averaged over tiles."""
import ctypes, os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicfpaugment_amd import _lib
_lib.set_library_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "musicfpaugment_amd", "libmfpa_exp.so"))
from musicfpaugment_amd import ops_unet as K
h = ctypes.CDLL(_lib.LIB_PATH)
B, H, W = 64, 257, 251
lay = K.frag_layout()
sc = torch.ones(64, device="cuda"); sh = torch.zeros(64, device="cuda")
def packs(co, ci):
    w = torch.randn(9, co, ci, device="cuda") * 0.05
    return K.split_bf16x3(w), (lay, K.split_bf16x3_frag(w, lay))
w64, wf64 = packs(64, 64); w128, wf128 = packs(64, 128)
x = torch.relu(torch.randn(B, H, W, 64, device="cuda")); u = torch.randn(B, H - 1, W - 1, 64, device="cuda")
wo = torch.randn(64, device="cuda")
runs = [("up4.0", lambda: K.conv3x3_fused(x, w128, sc, sh, x1=u, precision=1, wf=wf128)),
        ("up4.3", lambda: K.conv3x3_fused(x, w64, sc, sh, precision=1, out1x1=(wo, 0.1), store=False, wf=wf64)),
        ("plain 64->64 store", lambda: K.conv3x3_fused(x, w64, sc, sh, precision=1, wf=wf64))]
buf = torch.zeros(1024, dtype=torch.int64, device="cuda")
for name, fn in runs:
    fn(); torch.cuda.synchronize(); buf.zero_()
    assert h.mfpa_exp_conv_stamps(ctypes.c_void_p(buf.data_ptr())) == 0
    fn(); torch.cuda.synchronize()
    h.mfpa_exp_conv_stamps(ctypes.c_void_p(0))
    t = buf.cpu().numpy()
    n = int(t[0]); st = t[1:1 + n]
    tags = st & 0xff; tm = st & ~0xff
    # durations between consecutive stamps, keyed by (tag_from -> tag_to, index of chunk within tile)
    seq = []
    chunk = 0
    for i in range(n - 1):
        seq.append((int(tags[i]), int(tags[i + 1]), int(tm[i + 1] - tm[i])))
    agg = collections.OrderedDict()
    chunk = 0
    for a, b, d in seq:
        key = (chunk, a, b)
        agg.setdefault(key, []).append(d)
        if a == 8 and b == 0: chunk += 1
        if b == 10: pass
        if a == 11: chunk = 0
        if a == 8 and b == 10: chunk = 0
    print(f"== {name}: {n} stamps")
    tot = 0
    for (c, a, b), v in agg.items():
        v = v[1:] if len(v) > 2 else v           # drop the first tile (prologue)
        m = sum(v) / len(v); tot += m
        lab = {9: "bar", 10: "epi", 11: "end"}
        print(f"  chunk {c} {lab.get(a, 'tap%d' % a):>5s} -> {lab.get(b, 'tap%d' % b):>5s}: {m:8.0f} cycles  (n={len(v)}, min {min(v)}, max {max(v)})")
    print(f"  sum per tile: {tot:.0f} cycles")
