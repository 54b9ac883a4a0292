#!/usr/bin/env python3
"""Developer tool (GPU box): microseconds per evaluation (bxi_boxinst_eval_f32 at the C ABI, back to back, cold-ish inputs) over a
sweep of batch / canvas / instance-count / dilation / stride shapes -- looks for pathologies off the headline size."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from boxinstseg_amd import _lib, functional as Fh, synthetic
lib = _lib.load()
dev = torch.device('cuda:0')
ones = torch.ones(2, device=dev)
st = torch.cuda.current_stream().cuda_stream


def build(B, H, W, bpi, ipb, stride, seed, dil):
    d = synthetic.make_batch(B=B, H=H, W=W, boxes_per_img=bpi, inst_per_box=ipb, stride=stride, seed=seed, min_box=min(64, H // 4), max_box=min(512, H // 2 + W // 4))
    imgs = torch.from_numpy(d['imgs']).to(dev); logits = torch.from_numpy(d['mask_logits']).to(dev)
    gi = torch.from_numpy(d['gt_inds']).to(dev); boxes = [torch.from_numpy(b).to(dev) for b in d['gt_bboxes']]
    batch = Fh._Batch(imgs, d['img_metas'], 10); inst = Fh._Inst(logits, gi, boxes, d['H'], d['W'], stride)
    losses = torch.zeros(2, device=dev); grad = torch.empty_like(inst.logits)
    state = torch.empty(max(lib.bxi_boxinst_loss_state_bytes(inst.N, inst.h, inst.w), 256), dtype=torch.uint8, device=dev)
    ws = torch.zeros(max(lib.bxi_boxinst_eval_workspace_bytes(B, H, W, stride, inst.N), 256), dtype=torch.uint8, device=dev)
    return (batch, inst, losses, grad, state, ws, imgs, logits, gi, boxes, d)


def run(s, dil):
    batch, inst, losses, grad, state, ws = s[:6]
    rc = lib.bxi_boxinst_eval_f32(C.byref(batch.struct), C.byref(inst.struct), 3, dil, 0.3, 1.0, ones.data_ptr(), ones.data_ptr() + 4,
                                  losses.data_ptr(), grad.data_ptr(), state.data_ptr(), ws.data_ptr(), ws.numel(), int(os.environ.get('BXI_FLAGS', '0')), st)
    assert rc == 0, rc


cases = [  # B, H, W, boxes/img, inst/box, stride, dilation
    (2, 800, 1024, 16, 1, 4, 2), (2, 800, 1344, 16, 1, 4, 2), (4, 800, 1024, 16, 1, 4, 2), (1, 1344, 1344, 8, 2, 4, 2),
    (8, 512, 512, 8, 2, 4, 2), (2, 800, 1024, 16, 1, 4, 4), (2, 800, 1024, 16, 1, 4, 1), (2, 800, 1024, 16, 1, 8, 2),
    (2, 480, 640, 4, 1, 4, 2), (16, 256, 256, 4, 1, 4, 2), (2, 800, 1024, 1, 1, 4, 2), (2, 800, 1024, 2, 64, 4, 2)]
for (B, H, W, bpi, ipb, stride, dil) in cases:
    sets = [build(B, H, W, bpi, ipb, stride, 100 + k, dil) for k in range(4)]
    for i in range(20): run(sets[i % 4], dil)
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for i in range(n): run(sets[i % 4], dil)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / n * 1e6
    N = sets[0][1].N
    mb = (12 * B * H * W + 8 * N * (H // stride) * (W // stride)) / 1e6
    print('B=%-2d %4dx%-4d N=%-4d stride %d dil %d : %7.2f us / evaluation   (%.0f MB compulsory, %.2f TB/s)  losses %.4f %.4f' %
          (B, H, W, N, stride, dil, us, mb, mb / us / 1e6 * 1e6 / 1e6, float(sets[0][2][0]), float(sets[0][2][1])))
