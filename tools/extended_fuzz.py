#!/usr/bin/env python3
"""Developer tool (GPU box): the seeded fuzz tests of the suite over many more seeds than the suite carries
(`python tools/extended_fuzz.py 24 400`): main loss path (default form; every other form + the targets-ahead split against it), dynamic head, DiscoBox, tree_filter.  Prints failures, exits 1 on any."""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as entry
entry.build()
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (24, 200)
dev = torch.device('cuda:0')
from tests import test_gpu_parity as tp, test_gpu_dynamic_head as td, test_gpu_discobox as tdb, test_gpu_tree_filter as tt, test_gpu_levelset as tl
import numpy as np
suites = [('loss', lambda s: tp.test_loss_fuzz(dev, s)), ('loss_forms', lambda s: tp.test_loss_fuzz_forms_and_targets_ahead(dev, s)), ('dynamic_head', lambda s: td.test_dynamic_head_fuzz(dev, s)),
          ('discobox', lambda s: tdb.test_meanfield_fuzz(True, dev, s)),
          ('tree_filter', lambda s: tt.test_tree_filter_fuzz(True, dev, s)), ('levelset', lambda s: levelset_case(s)), ('lcm', lambda s: lcm_case(s))]


def levelset_case(seed):        # projection (mil kernels) + level set at random shapes / channel counts
    r = np.random.default_rng(31000 + seed)
    tl.test_projection_and_levelset_vs_oracle(True, dev, int(r.integers(1, 12)), int(r.integers(2, 120)), int(r.integers(2, 340)), int(r.integers(1, 25)))   # target channels: up to three groups of 8


def lcm_case(seed):             # LCM forward + adjoint: cached, LDS and per-iteration paths
    r = np.random.default_rng(32000 + seed)
    big = r.random() < 0.2
    h, w = (int(r.integers(97, 160)), int(r.integers(97, 200))) if big else (int(r.integers(1, 97)), int(r.integers(1, 97)))
    tl.test_lcm_vs_oracle(True, dev, int(r.integers(1, 5)), h, w, int(r.integers(0, 6 if big else 11)), int(r.integers(1, 4)))
only = sys.argv[3].split(',') if len(sys.argv) > 3 else None
bad = 0
for name, fn in suites:
    if only and name not in only: continue
    n = 0
    for seed in range(lo, hi):
        try:
            fn(seed); n += 1
        except Exception as e:          # noqa: BLE001
            bad += 1
            print(f'FAIL {name} seed {seed}: {type(e).__name__}: {str(e)[:300]}')
    print(f'{name}: {n}/{hi - lo} seeds passed', flush=True)
sys.exit(1 if bad else 0)
