#!/usr/bin/env python3
"""Developer measurement of the op-level pairwise_nlog kernels at cfg-2 size, f32 and f64.

Timed as batches of back-to-back calls (HIP events around the batch), rotating over 6 independent sets of tensors so that every call
reads COLD data (6 x 118 MB > the 256 MB Infinity Cache; `*_warm_us`: one set re-used, labelled as such).  `*_sol_us`: the same bytes
moved by a copy kernel (bxi_dev_sol_pairwise_f32, include/boxinst_hip_dev.h), timed the same way on the same sets."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as entry
entry.build()
from boxinstseg_amd import pairwise_nlog_forward, pairwise_nlog_backward, _lib
dev = torch.device('cuda:0')
SETS = 6


def ev(fn, n=60, warm=6):
    for i in range(warm): fn(i)
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n): fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


res = {}
lib = _lib.load()
for dt in (torch.float32, torch.float64):
    xs = [torch.randn(32, 1, 200, 256, device=dev, dtype=dt) for _ in range(SETS)]
    pws = [pairwise_nlog_forward(3, 2, x) for x in xs]
    gs = [torch.randn_like(p) for p in pws]
    es = xs[0].element_size()
    nb = xs[0].numel() * es
    f = ev(lambda i: pairwise_nlog_forward(3, 2, xs[i % SETS]))
    b = ev(lambda i: pairwise_nlog_backward(3, 2, xs[i % SETS], pws[i % SETS], gs[i % SETS]))
    fw = ev(lambda i: pairwise_nlog_forward(3, 2, xs[0]))
    bw = ev(lambda i: pairwise_nlog_backward(3, 2, xs[0], pws[0], gs[0]))
    res[str(dt)] = dict(fwd_us=f, fwd_GBps=9 * nb / f / 1e3, bwd_us=b, bwd_GBps=(1 + 8 + 1) * nb / b / 1e3, fwd_warm_us=fw, bwd_warm_us=bw,
                        note='fwd_us / bwd_us: cold inputs (6 rotating sets); *_warm_us: one set re-used (Infinity-Cache resident)')
    if dt == torch.float32:       # the same bytes moved by a copy: what this part does for them with nothing computed
        st = torch.cuda.current_stream().cuda_stream
        gls = [torch.empty_like(x) for x in xs]
        res[str(dt)]['fwd_sol_us'] = ev(lambda i: lib.bxi_dev_sol_pairwise_f32(xs[i % SETS].data_ptr(), pws[i % SETS].data_ptr(), gls[i % SETS].data_ptr(), 32, 200, 256, 0, st))
        res[str(dt)]['bwd_sol_us'] = ev(lambda i: lib.bxi_dev_sol_pairwise_f32(xs[i % SETS].data_ptr(), gs[i % SETS].data_ptr(), gls[i % SETS].data_ptr(), 32, 200, 256, 1, st))
    del xs, pws, gs
    torch.cuda.empty_cache()
print(json.dumps(res, indent=1))
