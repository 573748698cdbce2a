#!/usr/bin/env python3
"""Timing-only variants of conv_ws64_kernel (csrc/unet_ws.hip) as compile-time skips of the PRODUCT code: for every <bits> given, compile
unet_ws.hip with -DMFPA_SKIP_BITS=<bits> and link it with the product build's other objects -> musicfpaugment_amd/libmfpa_ws_<bits>.so
(never loaded by the package; wrong results by design).  Bits: 1 loaders skip the halo (loads + split), 2 compute waves skip the epilogue,
4 loaders skip the stores, 8 no fragment reads in the loop, 16 no weight loads in the loop, 32 no halo REQUESTS (split + LDS stores stay),
64 no LDS stores of the split halo (requests + split stay).  Extra -D flags after `--`."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from musicfpaugment_amd.csrc import build as B
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--"); extra = args[i + 1:]; args = args[:i]
B.build(verbose=False)                                           # the product objects
objs = [os.path.join(B.HERE, "build", f.replace(".hip", ".o")) for f in B._sources() if f != "unet_ws.hip"]
for spec in args:                                                # "<bits>" or "<bits>:NAME=VALUE,NAME=VALUE" (-D macros of unet_ws.hip)
    bits, _, defs = spec.partition(":")
    tag = spec.replace(":", "_").replace("=", "").replace(",", "_")
    flags = B.COMMON + [f"-DMFPA_SKIP_BITS={bits}"] + [f"-D{d}" for d in defs.split(",") if d] + extra
    obj = f"/tmp/unet_ws_{tag}.o"
    subprocess.run([B._hipcc()] + flags + ["-c", os.path.join(B.HERE, "unet_ws.hip"), "-o", obj], check=True)
    out = os.path.join(B.PKG, f"libmfpa_ws_{tag}.so")
    subprocess.run([B._hipcc(), f"--offload-arch={B.ARCH}", "-shared", "-fPIC", "-o", out] + objs + [obj], check=True)
    print(out)
