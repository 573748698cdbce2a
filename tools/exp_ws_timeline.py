#!/usr/bin/env python3
"""Timeline of conv_ws64_kernel (experiments build): wave 0 (compute) and wave 4 (loader) of workgroup 17 stamp s_memtime.
Compute tags: 0..8 tap start, 9 before the chunk barrier, 10 after it, 11 epilogue start, 12 tile end.
Loader tags: 20 iteration start, 21 first slot split (= this chunk's halo, requested one iteration ago, has arrived), 22 split done, 24 stores done
(barrier next), 23 after the barrier.
Prints mean shader-clock cycles per interval (first tile dropped)."""
import collections, ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicfpaugment_amd import _lib
# default: the product code + stamps (tools/build_ws_variants.py 0:MFPA_WS_STAMPS=1); or any other build given as argv[1]
_lib.set_library_path(sys.argv[1] if len(sys.argv) > 1 else
                      os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "musicfpaugment_amd", "libmfpa_ws_0_MFPA_WS_STAMPS1.so"))
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
if os.environ.get("WS_ZERO") == "1":                         # all-zero operands: same instruction stream, no data toggling (the clock's ceiling)
    x.zero_(); u.zero_(); w64.zero_(); w128.zero_(); wf64[1].zero_(); wf128[1].zero_()
runs = [("up4.0", 4, lambda: K.conv3x3_fused(x, w128, sc, sh, x1=u, precision=1, wf=wf128)),
        ("up4.3", 2, lambda: K.conv3x3_fused(x, w64, sc, sh, precision=1, out1x1=(wo, 0.1), store=False, wf=wf64)),
        ("plain 64->64 store", 2, lambda: K.conv3x3_fused(x, w64, sc, sh, precision=1, wf=wf64))]
buf = torch.zeros(4096, dtype=torch.int64, device="cuda")
lab = {9: "bar<", 10: "bar>", 11: "epi", 12: "end", 20: "turn", 21: "slot0", 22: "split", 23: "bar>", 24: "duty"}
for name, nch, fn in runs:
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    n_rep = 1500                                           # >= 2 s of back-to-back launches before the stamped one
    for _ in range(n_rep): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"== {name}: {e0.elapsed_time(e1) * 1e3 / n_rep:.1f} us per launch over {n_rep} back-to-back launches (stamps compiled in, none recorded)")
    buf.zero_()
    assert h.mfpa_exp_ws_stamps(ctypes.c_void_p(buf.data_ptr())) == 0
    fn(); torch.cuda.synchronize()
    h.mfpa_exp_ws_stamps(ctypes.c_void_p(0))
    t = buf.cpu().numpy()
    ck = t[3000:3000 + 512].reshape(256, 2).astype(float)
    ck = ck[ck[:, 1] > 0]
    if len(ck):
        ghz = np.sort(ck[:, 0] / ck[:, 1] * 0.1)
        print(f"== {name}: in-kernel clock (d s_memtime / d s_memrealtime x 100 MHz, {len(ck)} workgroups): median {np.median(ghz):.3f} GHz, "
              f"min {ghz[0]:.3f}, max {ghz[-1]:.3f}; workgroup life median {np.median(ck[:, 1]) / 100:.1f} us = {np.median(ck[:, 0]):.0f} shader cycles")
    for who, base in (("compute wave 0", 0), ("loader wave 4", 2048)):
        n = int(t[base]); st = t[base + 1:base + 1 + n]
        tags = (st & 0xff).astype(int); tm = st & ~0xff
        agg = collections.OrderedDict()
        pos = 0                                               # chunk index within the tile (compute) / turn index mod nch (loader)
        skip = 9 * nch + 8 if base == 0 else 5 * nch          # the first tile's stamps
        for i in range(n - 1):
            a_, b_, d = tags[i], tags[i + 1], int(tm[i + 1] - tm[i])
            if i >= skip: agg.setdefault((pos, a_, b_), []).append(d)
            if base == 0:
                if a_ == 8 and b_ == 0: pos += 1
                if a_ == 12 or (a_ == 8 and b_ == 11): pos = 0
            else:
                if a_ == 23: pos = (pos + 1) % nch
        print(f"== {name}: {who}, {n} stamps")
        tot = 0.0
        for (c, a_, b_), v in agg.items():
            m = sum(v) / len(v); tot += m
            print(f"  [{c}] {lab.get(a_, 'tap%d' % a_):>6s} -> {lab.get(b_, 'tap%d' % b_):>6s}: {m:8.0f} cycles  (n={len(v)}, min {min(v)}, max {max(v)})")
        print(f"  sum: {tot:.0f} cycles per tile")
