#!/usr/bin/env python3
"""Developer tool: run a script of this repository against another build of the library (A/B of compile-time variants on ONE box).
    python tools/with_lib.py boxinstseg_amd/lib/libboxinst_hip_var.so bench.py --no-extras ..."""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from boxinstseg_amd import build as hb
hb.LIB_PATH = os.path.abspath(sys.argv[1])
hb.is_stale = lambda: False
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name='__main__')
