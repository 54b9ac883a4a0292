#!/usr/bin/env python3
"""Developer measurement (GPU box): CondInstMaskHead.forward_loss (head evaluated inside the loss evaluation's first launch)
against forward() + loss(), at cfg-2 size (2 x 800 x 1024, 32 instances, 16 mask-feature channels).  Run under
`rocprofv3 --kernel-trace --stats` for the per-kernel durations (prep_head vs dyn_fwd + prep)."""
import os, sys, json, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as entry
entry.build()
from boxinstseg_amd import CondInstMaskHead, synthetic
dev = torch.device('cuda:0')
d = synthetic.cfg2(0)
imgs = torch.from_numpy(d['imgs']).to(dev)
B, H, W = imgs.shape[0], imgs.shape[2], imgs.shape[3]
boxes = [torch.from_numpy(b).to(dev) for b in d['gt_bboxes']]
gt_inds = torch.from_numpy(d['gt_inds']).to(dev)
n = gt_inds.numel()
counts = np.cumsum([0] + [b.shape[0] for b in boxes])
img_inds = torch.tensor([int(np.searchsorted(counts, int(g), side='right') - 1) for g in gt_inds.cpu()], device=dev)
torch.manual_seed(0)
head = CondInstMaskHead(in_channels=16, boxinst_enabled=True, max_proposals=-1, topk_per_img=64).to(dev)
head.set_iter(5000)
feat = torch.randn(B, 16, H // 8, W // 8, device=dev, requires_grad=True)
params = (0.3 * torch.randn(n, head.num_gen_params, device=dev)).requires_grad_(True)
coors = torch.rand(n, 2, device=dev) * torch.tensor([W, H], device=dev)
lvl = torch.randint(0, 5, (n,), device=dev)


def step(fused):
    if fused:
        _, losses = head.forward_loss(feat, params, coors, lvl, img_inds, imgs, d['img_metas'], gt_inds, boxes, fuse_head=True)
    else:
        logits = head(feat, params, coors, lvl, img_inds)
        losses = head.loss(imgs, d['img_metas'], logits, gt_inds, boxes, None, None)
    (losses['loss_prj'] + losses['loss_pairwise']).backward()
    feat.grad = None; params.grad = None


def ev(fn, k=200, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / k * 1e3


print(json.dumps({'forward_loss_backward_us': ev(lambda: step(True)), 'forward_then_loss_backward_us': ev(lambda: step(False)),
                  'note': 'wall time per training-side call (host-bound under eager autograd); kernel durations: rocprofv3'}, indent=1))
