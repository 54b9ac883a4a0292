#!/usr/bin/env python3
"""Developer tool (GPU box): the seeded fuzz tests of the suite over many more seeds than the suite carries
(`python tools/extended_fuzz.py 24 400`): main loss path, dynamic head, DiscoBox, tree_filter.  Prints failures, exits 1 on any."""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as entry
entry.build()
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (24, 200)
dev = torch.device('cuda:0')
from tests import test_gpu_parity as tp, test_gpu_dynamic_head as td, test_gpu_discobox as tdb, test_gpu_tree_filter as tt
suites = [('loss', lambda s: tp.test_loss_fuzz(dev, s)), ('dynamic_head', lambda s: td.test_dynamic_head_fuzz(dev, s)),
          ('discobox', lambda s: tdb.test_meanfield_fuzz(True, dev, s)),
          ('tree_filter', lambda s: tt.test_tree_filter_fuzz(True, dev, s))]
bad = 0
for name, fn in suites:
    n = 0
    for seed in range(lo, hi):
        try:
            fn(seed); n += 1
        except Exception as e:          # noqa: BLE001
            bad += 1
            print(f'FAIL {name} seed {seed}: {type(e).__name__}: {str(e)[:300]}')
    print(f'{name}: {n}/{hi - lo} seeds passed', flush=True)
sys.exit(1 if bad else 0)
