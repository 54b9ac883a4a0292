"""Developer helper (GPU box): where the host time of CondInstMaskHead.loss() goes -- each piece timed alone, 4000 calls."""
import sys, time, gc, ctypes as C
sys.path.insert(0, '.')
import torch
from boxinstseg_amd import CondInstMaskHead, synthetic, functional as Fh, _lib

dev = torch.device('cuda:0')
d = synthetic.cfg2(0)
imgs = torch.from_numpy(d['imgs']).to(dev)
boxes = [torch.from_numpy(b).to(dev) for b in d['gt_bboxes']]
gi = torch.from_numpy(d['gt_inds']).to(dev)
x = torch.from_numpy(d['mask_logits']).to(dev) if 'mask_logits' in d else torch.randn(gi.numel(), 1, d['h'], d['w'], device=dev)
metas = d['img_metas']
head = CondInstMaskHead(in_channels=16, boxinst_enabled=True, topk_per_img=64, max_proposals=-1).to(dev)
head.set_iter(20000)
N = 4000


def t(name, fn, n=N):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    gc.collect(); gc.freeze()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    el = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    gc.unfreeze()
    print(f'{name:55s} {el:7.2f} us')
    return el


with torch.no_grad():
    t('head.loss (no_grad)', lambda: head.loss(imgs, metas, x, gi, boxes, None, None))
    t('functional.boxinst_mask_loss (no_grad)', lambda: Fh.boxinst_mask_loss(x, gi, boxes, imgs=imgs, img_metas=metas, warmup_factor=1.0))
    cfg = dict(out_stride=4, bottom_pixels_removed=10, pairwise_size=3, pairwise_dilation=2, pairwise_color_thresh=0.3, warmup_factor=1.0)
    t('BoxInstMaskLoss.apply (no_grad)', lambda: Fh.BoxInstMaskLoss.apply(x, imgs, metas, gi, boxes, cfg, None))

    class Ctx:
        pass
    ctx = Ctx(); ctx.cfg, ctx.logits, ctx.imgs, ctx.gt_inds, ctx.boxes, ctx.metas, ctx.calls = cfg, x, imgs, gi, boxes, metas, 0
    t('BoxInstMaskLoss._evaluate (no grad buffer)', lambda: Fh.BoxInstMaskLoss._evaluate(ctx, False))
    t('BoxInstMaskLoss._evaluate (grad buffer)', lambda: Fh.BoxInstMaskLoss._evaluate(ctx, True))
    stream = torch.cuda.current_stream(dev).cuda_stream
    t('torch.cuda.current_stream(dev).cuda_stream', lambda: torch.cuda.current_stream(dev).cuda_stream)
    t('_f32c x2', lambda: (Fh._f32c(imgs), Fh._f32c(x)))
    t('boxes checks', lambda: [b if (b.dtype == torch.float32 and b.device == dev and b.is_contiguous()) else None for b in boxes])
    t('gt_inds checks', lambda: gi.dtype != torch.int64 or gi.device != dev or not gi.is_contiguous())
    t('_eval_plan', lambda: Fh._eval_plan(imgs, metas, x, boxes, 4, 10, stream))
    plan = Fh._eval_plan(imgs, metas, x, boxes, 4, 10, stream)
    nfl = 64 + plan.state_bytes // 4 + plan.grad_elems
    t('torch.empty(buf)', lambda: torch.empty(nfl, dtype=torch.float32, device=dev))
    buf = torch.empty(nfl, dtype=torch.float32, device=dev)
    t('buf.data_ptr()', lambda: buf.data_ptr())
    t('grad view: buf[a:].view(shape)', lambda: buf[64 + plan.state_bytes // 4:].view(x.shape))
    t('losses view + unbind', lambda: buf[:2].unbind(0))
    t('torch.cuda.current_device()', lambda: torch.cuda.current_device())
    base = buf.data_ptr()
    plan.batch.imgs, plan.inst.logits, plan.inst.gt_inds = imgs.data_ptr(), x.data_ptr(), gi.data_ptr()
    args = (plan.batch_ref, plan.inst_ref, 3, 2, 0.3, 1.0, 0, 0, base, base + 256 + plan.state_bytes, base + 256, plan.ws_ptr, plan.ws_bytes, 0, stream)
    t('plan.eval(*args) [ctypes + C side + launch]', lambda: plan.eval(*args))
    lib = _lib.load()
    t('ctypes call of a trivial entry (bxi_last_hip_error)', lambda: lib.bxi_last_hip_error())
    t('3 struct field stores', lambda: (setattr(plan.batch, 'imgs', base), setattr(plan.inst, 'logits', base), setattr(plan.inst, 'gt_inds', base)))
    pass
