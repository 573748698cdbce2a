"""Instruction-order sketch of the kernels in a disassembly (tools/disasm_obj.py output): M mfma, r / W LDS read / write, G global or buffer load,
S global store, A atomic, $ scratch, v VALU, s SALU, [w:..] s_waitcnt, |BAR| s_barrier, <br> branches.  isa_sketch.py file.s [kernel-substring]"""
import re, sys
txt = open(sys.argv[1]).read().splitlines()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
cur, name = [], None
def flush():
    if name is None or pat not in name or not cur: return
    out = []
    for l in cur:
        op = l.split()[0]
        if op.startswith("v_mfma"): out.append("M")
        elif op.startswith(("global_load", "buffer_load")): out.append("G")
        elif op.startswith(("global_store", "buffer_store")): out.append("S")
        elif op.startswith("global_atomic"): out.append("A")
        elif op.startswith("scratch_"): out.append("$")
        elif op.startswith("ds_read") or op.startswith("ds_bpermute") or op.startswith("ds_swizzle"): out.append("r")
        elif op.startswith("ds_write"): out.append("W")
        elif op.startswith("s_waitcnt"): out.append("[w:" + l.split(None, 1)[1].replace(" ", "") + "]")
        elif op.startswith("s_barrier"): out.append("|BAR|")
        elif op.startswith("s_cbranch") or op.startswith("s_branch"): out.append("<br>")
        elif op.startswith("s_endpgm"): out.append("<END>")
        elif op.startswith("v_"): out.append("v")
        elif op.startswith("s_"): out.append("s")
        else: out.append("?")
    print(name, len(cur)); print("".join(out)); print()
for line in txt:
    m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
    if m:
        flush(); name, cur = m.group(1), []
    elif line.startswith("\t"):
        cur.append(line.split("//")[0].strip())
flush()
