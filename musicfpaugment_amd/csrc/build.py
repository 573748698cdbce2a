"""Build libmfpa.so (all HIP kernels + the C ABI of include/mfpa.h) for gfx950 with hipcc.

In-tree, no torch dependency: `python -m musicfpaugment_amd.csrc.build`.  hipcc cross-compiles
without a GPU; the resulting musicfpaugment_amd/libmfpa.so travels to the GPU box as is.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT = os.path.join(PKG, "libmfpa.so")
OBJ = os.path.join(HERE, "build")
ARCH = "gfx950"

COMMON = ["-O3", f"--offload-arch={ARCH}", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
# The peak pickers compare float64 values that must round exactly like numpy's un-fused
# multiply / add sequence: no FMA contraction there.
PER_FILE = {
    "audfprint.hip": ["-ffp-contract=off"],
    "dejavu.hip": ["-ffp-contract=off"],
}


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libmfpa.so cannot be built")


def _sources():
    return sorted(f for f in os.listdir(HERE) if f.endswith(".hip"))


def _stamp(src: str, flags) -> str:
    h = hashlib.sha1()
    h.update(" ".join(flags).encode())
    for f in [src] + sorted(x for x in os.listdir(HERE) if x.endswith(".h")) + ["../../include/mfpa.h"]:
        with open(os.path.join(HERE, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True, experiments: bool = False, extra_flags=(), out: str = None) -> str:
    """experiments=True: the A/B build of tools/ (-DMFPA_EXPERIMENTS: the environment switches of mfpa_common.h are live) ->
    libmfpa_exp.so, which the package never loads on its own.  `extra_flags` / `out`: further tools-only variants."""
    OUT = out or (os.path.join(PKG, "libmfpa_exp.so") if experiments else globals()["OUT"])
    OBJ = os.path.join(HERE, "build" + ("_exp" if experiments else "") + ("_" + hashlib.sha1(" ".join(extra_flags).encode()).hexdigest()[:8] if extra_flags else ""))
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    objs = []
    for src in _sources():
        flags = COMMON + PER_FILE.get(src, []) + (["-DMFPA_EXPERIMENTS"] if experiments else []) + list(extra_flags)
        obj = os.path.join(OBJ, src.replace(".hip", ".o"))
        stamp_file = obj + ".stamp"
        stamp = _stamp(src, flags)
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
            continue
        jobs.append((src, [hipcc] + flags + ["-c", os.path.join(HERE, src), "-o", obj], stamp_file, stamp))

    def run(job):
        src, cmd, stamp_file, stamp = job
        if verbose:
            print("[mfpa build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip() and verbose:
            print(r.stderr, file=sys.stderr)
        with open(stamp_file, "w") as fh:
            fh.write(stamp)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(OUT) or force:
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", OUT] + objs
        if verbose:
            print("[mfpa build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return OUT


def record(path: str = None) -> dict:
    """Identity of a built library: sha256, size, hipcc version, flags (written to profiles/BUILD_rNN.txt by __graft_entry__.build())."""
    path = path or OUT
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 20), b""):
            h.update(blk)
    ver = subprocess.run([_hipcc(), "--version"], capture_output=True, text=True).stdout.strip().splitlines()
    src = hashlib.sha256()
    for f in _sources() + sorted(x for x in os.listdir(HERE) if x.endswith(".h")) + ["../../include/mfpa.h"]:
        with open(os.path.join(HERE, f), "rb") as fh:
            src.update(f.encode() + b"\0" + fh.read())
    return {"library": os.path.relpath(path, os.path.dirname(PKG)), "sha256": h.hexdigest(), "bytes": os.path.getsize(path),
            "sources_sha256": src.hexdigest(), "hipcc": " | ".join(v for v in ver[:2]), "flags": " ".join(COMMON),
            "per_file_flags": {k: " ".join(v) for k, v in PER_FILE.items()}}


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, experiments="--experiments" in sys.argv))
