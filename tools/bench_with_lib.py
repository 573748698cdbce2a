#!/usr/bin/env python3
"""bench.py against another build of the library (tools/ A/B runs): bench_with_lib.py LIB [bench.py arguments]"""
import os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicfpaugment_amd import _lib
_lib.set_library_path(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
