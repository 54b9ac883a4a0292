#!/usr/bin/env python3
"""Developer measurement of the op-level pairwise_nlog kernels at cfg-2 size."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as entry
entry.build()
from boxinstseg_amd import pairwise_nlog_forward, pairwise_nlog_backward
dev = torch.device('cuda:0')
def ev(fn, n=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
res = {}
for dt in (torch.float32, torch.float64):
    x = torch.randn(32, 1, 200, 256, device=dev, dtype=dt)
    pw = pairwise_nlog_forward(3, 2, x); g = torch.randn_like(pw)
    es = x.element_size()
    f = ev(lambda: pairwise_nlog_forward(3, 2, x)); b = ev(lambda: pairwise_nlog_backward(3, 2, x, pw, g))
    nb = x.numel() * es
    res[str(dt)] = dict(fwd_us=f, fwd_GBps=9 * nb / f / 1e3, bwd_us=b, bwd_GBps=(1 + 8 + 1) * nb / b / 1e3)
print(json.dumps(res, indent=1))
