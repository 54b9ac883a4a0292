#!/usr/bin/env python3
"""GPU box only: record what the PRODUCT (the HIP path through CondInstMaskHead.loss) produced for two small batches -- one per
rank of the 2-rank gloo test (tests/test_dist_gloo.py), which runs where there is no GPU and therefore cannot evaluate the product
itself.  Writes gpurun_out/hip_run_2ranks.npz; the committed copy is tests/golden/hip_run_2ranks.npz.

    gpurun -- 'python tests/golden/make_hip_run.py'
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import __graft_entry__ as entry

entry.build()
from boxinstseg_amd import CondInstMaskHead, synthetic

dev = torch.device('cuda:0')
out = {}
for rank in range(2):
    d = synthetic.make_batch(B=1, H=64, W=64, boxes_per_img=2, seed=100 + rank, min_box=16, max_box=40)
    head = CondInstMaskHead(in_channels=16, boxinst_enabled=True, topk_per_img=64, max_proposals=-1, pairwise_warmup=10000).to(dev)
    head.set_iter(2499)                                    # -> warm-up factor 0.25 on this call
    x = torch.from_numpy(d['mask_logits']).to(dev).requires_grad_(True)
    losses = head.loss(torch.from_numpy(d['imgs']).to(dev), d['img_metas'], x, torch.from_numpy(d['gt_inds']).to(dev),
                       [torch.from_numpy(b).to(dev) for b in d['gt_bboxes']], None, None)
    (losses['loss_prj'] + losses['loss_pairwise']).backward()
    torch.cuda.synchronize()
    out[f'rank{rank}_loss_prj'] = np.float32(losses['loss_prj'].item())
    out[f'rank{rank}_loss_pairwise'] = np.float32(losses['loss_pairwise'].item())
    out[f'rank{rank}_grad_abs_sum'] = np.float64(x.grad.double().abs().sum().item())
    out[f'rank{rank}_iter_after'] = np.float32(head._iter.item())
    out[f'rank{rank}_seed'] = np.int64(100 + rank)
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
np.savez(os.path.join(ROOT, 'gpurun_out', 'hip_run_2ranks.npz'), **out)
print({k: float(v) for k, v in out.items()})
