"""Disassemble the gfx950 code objects inside a built libmfpa.so and look for instruction forms the library must not contain.

Used by tests/test_isa_scan.py (CPU) and by `__graft_entry__.build()`: the check runs on the SHIPPED binary, not on a re-compile.

The one form banned today: a packed-fp32 VALU instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) with an `op_sel:[...]`
operand selection, i.e. whose LOW lane reads the HIGH half of a source pair.  hipcc (ROCm 7.2.0) emits it freely and one
instance returned sporadically wrong low lanes next to MFMA waves (profiles/r02_pk_fma_op_sel.md).  `op_sel_hi:[...]` alone (the
broadcast of a low half into the high lane) is the form that has always been bit-exact and is allowed.
"""
from __future__ import annotations

import os
import re
import shutil
import struct
import subprocess
import tempfile
from typing import Dict, List, Tuple

_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
_PK_F32 = re.compile(r"\bv_pk_[a-z]+_f32\b")
_OP_SEL = re.compile(r"\bop_sel:\[[01,]+\]")
_LABEL = re.compile(r"^[0-9a-f]+ <([^>]+)>:")


def _objdump() -> str:
    for cand in ("/opt/rocm/lib/llvm/bin/llvm-objdump", shutil.which("llvm-objdump")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("llvm-objdump not found")


def code_objects(lib_path: str, arch: str = "gfx950") -> List[bytes]:
    """Every device ELF for `arch` in the clang offload bundles of a host library (one bundle per translation unit)."""
    data = open(lib_path, "rb").read()
    out, pos = [], 0
    while True:
        i = data.find(_MAGIC, pos)
        if i < 0:
            return out
        (n,) = struct.unpack_from("<Q", data, i + len(_MAGIC))
        p = i + len(_MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            p += 24
            triple = data[p:p + tl].decode()
            p += tl
            if arch in triple and size:
                out.append(data[i + off:i + off + size])
        pos = i + len(_MAGIC)


_CACHE: Dict[Tuple[str, float, str], List[Tuple[str, str]]] = {}


def disassemble(lib_path: str, arch: str = "gfx950") -> List[Tuple[str, str]]:
    """[(kernel symbol, instruction text)] for every instruction of every device code object in the library."""
    key = (os.path.abspath(lib_path), os.path.getmtime(lib_path), arch)
    if key in _CACHE:
        return _CACHE[key]
    rows: List[Tuple[str, str]] = []
    with tempfile.TemporaryDirectory() as tmp:
        for k, co in enumerate(code_objects(lib_path, arch)):
            path = os.path.join(tmp, f"co{k}.co")
            with open(path, "wb") as fh:
                fh.write(co)
            txt = subprocess.run([_objdump(), "-d", path], capture_output=True, text=True, check=True).stdout
            sym = "?"
            for line in txt.splitlines():
                m = _LABEL.match(line)
                if m:
                    sym = m.group(1)
                elif line.startswith("\t"):
                    rows.append((sym, line.split("//")[0].strip()))
    _CACHE[key] = rows
    return rows


def packed_fp32_op_sel(lib_path: str) -> Dict[str, List[str]]:
    """kernel symbol -> packed-fp32 instructions whose low lane selects a high half (must be empty for the product library)."""
    bad: Dict[str, List[str]] = {}
    for sym, ins in disassemble(lib_path):
        if _PK_F32.search(ins) and _OP_SEL.search(ins):
            bad.setdefault(sym, []).append(ins)
    return bad


def summary(lib_path: str) -> Dict[str, int]:
    rows = disassemble(lib_path)
    return {"code_objects": len(code_objects(lib_path)), "instructions": len(rows),
            "packed_fp32": sum(1 for _, i in rows if _PK_F32.search(i)),
            "packed_fp32_op_sel": sum(1 for _, i in rows if _PK_F32.search(i) and _OP_SEL.search(i)),
            "mfma": sum(1 for _, i in rows if i.startswith("v_mfma"))}


if __name__ == "__main__":
    import json
    import sys
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libmfpa.so")
    print(json.dumps(summary(lib)))
    for k, v in packed_fp32_op_sel(lib).items():
        print(k, len(v), v[0])
