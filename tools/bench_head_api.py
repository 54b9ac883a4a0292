#!/usr/bin/env python3
"""Developer measurement: the drop-in Python API (CondInstMaskHead.loss + backward, get_targets) end to end at cfg-2,
i.e. what a maintainer who swaps the class sees per training iteration, against the raw C-ABI step of bench.py."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as entry
entry.build()
from boxinstseg_amd import CondInstMaskHead, synthetic
dev = torch.device('cuda:0')
d = synthetic.cfg2(0)
imgs = torch.from_numpy(d['imgs']).to(dev); logits0 = torch.from_numpy(d['mask_logits']).to(dev)
gi = torch.from_numpy(d['gt_inds']).to(dev); boxes = [torch.from_numpy(b).to(dev) for b in d['gt_bboxes']]
head = CondInstMaskHead(boxinst_enabled=True).to(dev)
def step():
    lg = logits0.clone().requires_grad_(True)
    out = head.loss(imgs, d['img_metas'], lg, gi, boxes, None, None)
    (out['loss_prj'] + out['loss_pairwise']).backward()
    return lg.grad
def wall(fn, n=200, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
res = dict(head_loss_plus_backward_us=wall(step), clone_only_us=wall(lambda: logits0.clone().requires_grad_(True)),
           get_targets_us=wall(lambda: head.get_targets(boxes, None, imgs, d['img_metas']), n=50, warm=5))
print(json.dumps(res, indent=1))
