#!/usr/bin/env python3
"""Developer measurement of the dynamic mask head on the GPU box: HIP forward / backward
vs the reference's own op sequence (grouped F.conv2d + pad/interpolate) run by PyTorch-ROCm on the same GPU."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as entry
entry.build()
from boxinstseg_amd import dynamic_mask_forward
from oracle import torch_oracle as to

dev = torch.device('cuda:0')
SOI = torch.tensor([64, 128, 256, 512, 1024], device=dev)
def ev(fn, n=100, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
res = {}
for N in (32, 128):
    B, C, H, W = 2, 16, 100, 128
    g = torch.Generator(device='cpu').manual_seed(0)
    feat = torch.randn(B, C, H, W, generator=g).to(dev).requires_grad_(True)
    params = (torch.randn(N, 233, generator=g) * 0.3).to(dev).requires_grad_(True)
    coors = (torch.rand(N, 2, generator=g) * 1000).to(dev); lvl = torch.randint(0, 5, (N,), generator=g).to(dev)
    img = torch.randint(0, B, (N,), generator=g).to(dev)
    gout = torch.randn(N, 1, 2 * H, 2 * W, generator=g).to(dev)
    def hip_fwd(): return dynamic_mask_forward(feat, params, coors, lvl, img, SOI)
    def hip_fb():
        y = hip_fwd(); y.backward(gout); feat.grad = None; params.grad = None
    def ref_fwd(): return to.dynamic_mask_forward(feat, params, coors, lvl, img, SOI)
    def ref_fb():
        y = ref_fwd(); y.backward(gout); feat.grad = None; params.grad = None
    with torch.no_grad():
        t_hf = ev(hip_fwd); t_rf = ev(ref_fwd)
    res[f'N{N}'] = dict(hip_fwd_us=t_hf, hip_fwd_bwd_us=ev(hip_fb), torch_rocm_fwd_us=t_rf, torch_rocm_fwd_bwd_us=ev(ref_fb),
                        logits_MB=N * 4 * H * W * 4 / 1e6)
print(json.dumps(res, indent=1))
