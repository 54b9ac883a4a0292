#!/usr/bin/env python3
"""Developer tool (GPU box): per-workgroup phase timestamps of the dynamic-head kernels from the -DBXI_TRACE
library (build it first: hipcc ... -DBXI_TRACE -o boxinstseg_amd/lib/libboxinst_hip_trace.so)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from boxinstseg_amd import _lib, build as hb
hb.LIB_PATH = os.path.join(hb.LIB_DIR, 'libboxinst_hip_trace.so')
from boxinstseg_amd import dynamic_mask_forward
lib = _lib.load()
lib.bxi_debug_set_trace_dyn.argtypes = [C.c_void_p]
dev = torch.device('cuda:0')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
B, Cc, H, W = 2, 16, 100, 128
g = torch.Generator().manual_seed(0)
feat = torch.randn(B, Cc, H, W, generator=g).to(dev).requires_grad_(True)
params = (torch.randn(N, 233, generator=g) * 0.3).to(dev).requires_grad_(True)
coors = (torch.rand(N, 2, generator=g) * 1000).to(dev); lvl = torch.randint(0, 5, (N,), generator=g).to(dev)
img = torch.randint(0, B, (N,), generator=g).to(dev)
gout = torch.randn(N, 1, 2 * H, 2 * W, generator=g).to(dev)
SOI = torch.tensor([64, 128, 256, 512, 1024], device=dev)
def step():
    feat.grad = None; params.grad = None
    dynamic_mask_forward(feat, params, coors, lvl, img, SOI).backward(gout)
for _ in range(10): step()
torch.cuda.synchronize()
trace = torch.zeros((5, 8192, 8), dtype=torch.int64, device=dev)
assert lib.bxi_debug_set_trace_dyn(trace.data_ptr()) == 0
step(); torch.cuda.synchronize()
t = trace.cpu().numpy()
for kid, name, nph in ((3, 'dyn_fwd', 5), (4, 'dyn_bwd', 5)):     # dyn_fwd: start, inputs loaded, MLP done, barrier, stores issued
    a = t[kid]; used = a[:, 0] > 0; a = a[used].astype(np.float64)
    if not len(a): continue
    t0 = a[:, 0].min()
    print(f'{name}: {len(a)} traced workgroups, span {(a[:, nph - 1].max() - t0) / 100:.2f} us '
          f'(first start 0, last start {(a[:, 0].max() - t0) / 100:.2f} us)')
    for ph in range(1, nph):
        d = (a[:, ph] - a[:, ph - 1]) / 100
        print(f'   phase {ph - 1}->{ph}: median {np.median(d):.2f} us  p90 {np.percentile(d, 90):.2f}  max {d.max():.2f}')
    d = (a[:, nph - 1] - a[:, 0]) / 100
    print(f'   workgroup life: median {np.median(d):.2f} us  max {d.max():.2f}')
