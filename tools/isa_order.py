"""Instruction-order sketch of a kernel of the built library (M mfma, r/W LDS read/write, G global load, A atomic, v VALU, s SALU): isa_order.py <mangled-name substring>"""
import os
sys_path = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import sys
sys.path.insert(0, sys_path)
from musicfpaugment_amd.csrc import isa_scan
rows = isa_scan.disassemble("musicfpaugment_amd/libmfpa.so")
pat = sys.argv[1]
cur=[r[1] for r in rows if pat in r[0]]
idx=[i for i,l in enumerate(cur) if "v_mfma" in l]
print(len(cur), idx[0], idx[-1], len(idx))
out=[]
for i in range(max(idx[0]-60,0), min(idx[-1]+10,len(cur))):
    l=cur[i].strip(); op=l.split()[0]
    if op.startswith("v_mfma"): out.append("M")
    elif op.startswith("global_load"): out.append("G")
    elif op.startswith("global_atomic"): out.append("A")
    elif op.startswith("scratch_"): out.append("$")
    elif op.startswith("ds_read"): out.append("r")
    elif op.startswith("ds_write"): out.append("W")
    elif op.startswith("s_waitcnt"): out.append("[w:"+l.split(None,1)[1].replace(" ","")+"]")
    elif op.startswith("s_barrier"): out.append("|BAR|")
    elif op.startswith("s_cbranch") or op.startswith("s_branch"): out.append("<br>")
    elif op.startswith("v_"): out.append("v")
    elif op.startswith("s_"): out.append("s")
    else: out.append("?")
print("".join(out))
