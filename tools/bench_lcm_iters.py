#!/usr/bin/env python3
"""Developer measurement (GPU box): LCM refine forward / adjoint time against the iteration count (prologue vs per-iteration cost)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as entry
entry.build()
from boxinstseg_amd import levelset as ls
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
N = 16
aff = torch.softmax(torch.rand(N, 8, 96, 96, generator=g), 1).to(dev).contiguous()
phi = torch.rand(N, 96, 96, generator=g).to(dev)
def ev(fn, n=200, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
out = {}
only = os.environ.get('LCM_ITERS')
for tr in (0, 1):
    for it in ([int(only)] if only else (0, 1, 2, 5, 10, 20, 40)):
        out['%s_iters_%d' % ('adjoint' if tr else 'forward', it)] = round(ev(lambda: ls._refine(aff, phi, 2, it, tr)), 2)
print(json.dumps(out))
