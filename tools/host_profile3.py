"""Developer helper (GPU box): does the host time of loss() + backward() drift over a process's life?"""
import sys, time, gc
sys.path.insert(0, '.')
import torch
from boxinstseg_amd import CondInstMaskHead, synthetic

dev = torch.device('cuda:0')
sets = []
for seed in range(8):
    d = synthetic.cfg2(seed)
    sets.append((torch.from_numpy(d['imgs']).to(dev), d['img_metas'], torch.from_numpy(d['mask_logits']).to(dev).requires_grad_(True),
                 torch.from_numpy(d['gt_inds']).to(dev), [torch.from_numpy(b).to(dev) for b in d['gt_bboxes']]))
head = CondInstMaskHead(in_channels=16, boxinst_enabled=True, topk_per_img=64, max_proposals=-1).to(dev)
head.set_iter(20000)


def full(i):
    imgs, metas, x, gi, boxes = sets[i % 8]
    out = head.loss(imgs, metas, x, gi, boxes, None, None)
    (out['loss_prj'] + out['loss_pairwise']).backward()
    x.grad = None


w = torch.randn(32, 1, 200, 256, device=dev, requires_grad=True)


def torch_only(i):
    (w.sum() + w.mean()).backward()
    w.grad = None


def bench(fn, n):
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    el = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return el


for rnd in range(6):
    print('round %d: loss+backward %.1f us   torch-only %.1f us' % (rnd, bench(full, 1000), bench(torch_only, 1000)))
gc.collect(); gc.freeze()
for rnd in range(3):
    print('gc frozen %d: loss+backward %.1f us   torch-only %.1f us' % (rnd, bench(full, 1000), bench(torch_only, 1000)))
import os
print('threads', torch.get_num_threads(), 'cpus', os.cpu_count())
