"""Disassemble the gfx950 code object inside a hipcc-built .o / .so: disasm_obj.py <file> > out.s"""
import os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicfpaugment_amd.csrc import isa_scan
for k, co in enumerate(isa_scan.code_objects(sys.argv[1])):
    with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as fh:
        fh.write(co)
    sys.stdout.write(subprocess.run([isa_scan._objdump(), "-d", fh.name], capture_output=True, text=True, check=True).stdout)
    os.unlink(fh.name)
