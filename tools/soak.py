#!/usr/bin/env python3
"""Developer tool (GPU box): soak of the evaluation -- N repeats of a few batches through the module path, every repeat must
report status 0 and the same bits as the first (`python tools/soak.py 3000`); then the suite's loss fuzz over more seeds."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from boxinstseg_amd import functional as Fh, synthetic
from tests.helpers import to_dev
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
hog = len(sys.argv) > 2 and sys.argv[2] == 'hog'      # a second stream keeps every CU busy with matrix products meanwhile
dev = torch.device('cuda:0')
Fh.DEBUG_KEEP_LAST = True
cases = [synthetic.cfg2(0), synthetic.cfg1(0), synthetic.make_batch(B=2, H=800, W=1024, boxes_per_img=16, inst_per_box=2, seed=3),
         synthetic.make_batch(B=3, H=160, W=224, boxes_per_img=3, seed=5, min_box=12, max_box=120, img_shapes=[[150, 200], [160, 224], [121, 183]])]
bad = 0
t0 = time.time()
hs = torch.cuda.Stream(device=dev)
hm = torch.randn(4096, 4096, device=dev)
for ci, d in enumerate(cases):
    t = to_dev(d, dev)
    first = None
    for i in range(n):
        if hog and i % 4 == 0:
            with torch.cuda.stream(hs):
                hm = (hm @ hm).clamp_(-1.0, 1.0)
        x = t['logits'].clone().requires_grad_(True)
        out = Fh.boxinst_mask_loss(x, t['gt_inds'], t['gt_bboxes'], imgs=t['imgs'], img_metas=d['img_metas'], out_stride=d['stride'])
        (out['loss_prj'] + out['loss_pairwise']).backward()
        if i % 50 == 0 or i == n - 1:          # synchronise and compare now and then (the runs in between overlap on the GPU)
            st = Fh.last_eval_status()[0]
            got = (float(out['loss_prj'].detach()), float(out['loss_pairwise'].detach()), x.grad.clone())
            if first is None: first = got
            same = got[0] == first[0] and got[1] == first[1] and torch.equal(got[2], first[2])
            if st != 0 or not same:
                bad += 1; print('case', ci, 'repeat', i, 'status', st, 'same bits', same)
    print('case', ci, 'N', d['N'], 'done', n, 'repeats', flush=True)
print('soak: %d bad of %d cases x %d repeats in %.0f s' % (bad, len(cases), n, time.time() - t0))
sys.exit(1 if bad else 0)
