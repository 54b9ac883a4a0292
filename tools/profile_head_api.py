import os, sys, cProfile, pstats, io
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
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
for _ in range(20): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(32); print(s.getvalue()[:6000])
