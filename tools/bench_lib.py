#!/usr/bin/env python3
"""Developer helper: run bench.py against an alternative build of the library (BXI_LIB=<file name in boxinstseg_amd/lib>)."""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from boxinstseg_amd import build as hb
if os.environ.get('BXI_LIB'):
    hb.LIB_PATH = os.path.join(hb.LIB_DIR, os.environ['BXI_LIB'])
sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name='__main__')
