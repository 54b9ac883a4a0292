#!/usr/bin/env python3
"""Developer tool: pytest against another build of the library.  python tools/pytest_with_lib.py lib.so tests/test_gpu_parity.py -m gpu -x -q"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from boxinstseg_amd import build as hb
hb.LIB_PATH = os.path.abspath(sys.argv[1])
hb.is_stale = lambda: False
hb.build = lambda force=False, verbose=False: hb.LIB_PATH
import pytest
sys.exit(pytest.main(sys.argv[2:]))
