"""Developer helper (GPU box): host time of the module-level training call, piece by piece (loss(), the caller's sum, backward())."""
import sys, time, gc
sys.path.insert(0, '.')
import torch
from boxinstseg_amd import CondInstMaskHead, synthetic, functional as Fh

dev = torch.device('cuda:0')
sets = []
for seed in range(8):
    d = synthetic.cfg2(seed)
    sets.append((torch.from_numpy(d['imgs']).to(dev), d['img_metas'], torch.from_numpy(d['mask_logits']).to(dev).requires_grad_(True),
                 torch.from_numpy(d['gt_inds']).to(dev), [torch.from_numpy(b).to(dev) for b in d['gt_bboxes']]))
head = CondInstMaskHead(in_channels=16, boxinst_enabled=True, topk_per_img=64, max_proposals=-1).to(dev)
head.set_iter(20000)
one = torch.ones((), device=dev)
N = 1500


def t(name, fn, n=N):
    for i in range(100):
        fn(i)
    torch.cuda.synchronize()
    gc.collect(); gc.freeze()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    el = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) / n * 1e6
    gc.unfreeze()
    print(f'{name:70s} host {el:7.2f} us   (with final sync {tot:7.2f})')
    return el


def loss_only(i):
    imgs, metas, x, gi, boxes = sets[i % 8]
    return head.loss(imgs, metas, x, gi, boxes, None, None)


def full(i):
    out = loss_only(i)
    (out['loss_prj'] + out['loss_pairwise']).backward()
    sets[i % 8][2].grad = None


def full_explicit(i):
    out = loss_only(i)
    torch.autograd.backward((out['loss_prj'], out['loss_pairwise']), (one, one))
    sets[i % 8][2].grad = None


def loss_and_sum(i):
    out = loss_only(i)
    return out['loss_prj'] + out['loss_pairwise']


with torch.no_grad():
    t('loss() under no_grad', loss_only)
t('loss() with grad (autograd node built)', loss_only)
t('loss() + the caller\'s sum of the two', loss_and_sum)
t('loss() + (a + b).backward()', full)
t('loss() + autograd.backward((a, b), (one, one))', full_explicit)
# a plain torch op chain of the same graph shape, for scale: what the engine costs whatever the node does
w = torch.randn(32, 1, 200, 256, device=dev, requires_grad=True)


def torch_only(i):
    a, b = w.sum(), w.mean()
    (a + b).backward()
    w.grad = None


t('torch only: (w.sum() + w.mean()).backward()', torch_only)

# inside our backward: body time vs what the engine / the Function wrapper add around it
import boxinstseg_amd.functional as F2
orig = F2.BoxInstMaskLoss.backward
acc = [0.0, 0]
raw = orig.__wrapped__ if hasattr(orig, '__wrapped__') else None
print('backward has __wrapped__:', raw is not None)


class Timed(F2.BoxInstMaskLoss):
    @staticmethod
    def backward(ctx, a, b):
        t0 = time.perf_counter()
        r = F2.BoxInstMaskLoss.backward(ctx, a, b)
        acc[0] += time.perf_counter() - t0; acc[1] += 1
        return r


def full_timed(i):
    imgs, metas, x, gi, boxes = sets[i % 8]
    cfg = dict(out_stride=4, bottom_pixels_removed=10, pairwise_size=3, pairwise_dilation=2, pairwise_color_thresh=0.3, warmup_factor=1.0)
    lp, lw = Timed.apply(x, imgs, metas, gi, boxes, cfg, None)
    (lp + lw).backward()
    x.grad = None


t('Timed.apply + (a + b).backward()', full_timed)
print('  of which inside backward (incl. once_differentiable wrapper of the parent): %.2f us per call' % (acc[0] / max(acc[1], 1) * 1e6))


def via_functional(i, **kw):
    imgs, metas, x, gi, boxes = sets[i % 8]
    out = Fh.boxinst_mask_loss(x, gi, boxes, imgs=imgs, img_metas=metas, **kw)
    (out['loss_prj'] + out['loss_pairwise']).backward()
    x.grad = None


t('functional.boxinst_mask_loss(warmup_factor=1.0) + backward', lambda i: via_functional(i, warmup_factor=1.0))
it = torch.full((1,), 20000.0, device=dev)
t('functional.boxinst_mask_loss(iter_counter) + backward', lambda i: via_functional(i, warmup_factor=1.0, iter_counter=it))
t('functional.boxinst_mask_loss(iter_counter, warmup_iters) + backward', lambda i: via_functional(i, iter_counter=it, warmup_iters=10000.0))
t('loss() + (a + b).backward() again', full)
