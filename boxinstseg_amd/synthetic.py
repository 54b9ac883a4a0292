"""Seeded synthetic inputs for the BoxInst loss path (SURVEY.md 8(d) recipe).

No dataset or checkpoint is reachable, so tests, ``bench.py`` and ``smoke()`` all draw from here:
  image   uint8 RGB, 16x16-pixel blocks of uniform random colour + N(0,6) pixel noise, fed to the
          "network" as ((u + 0.25) - mean) / std in fp32 -- the +0.25 keeps the de-normalised value
          away from an integer, so the uint8 truncation of condinst_head.py:180 is unambiguous.
          ``pixel_offset=0`` instead feeds exactly what the reference pipeline's Normalize step
          (mmcv imnormalize: f32 subtract of mean, double multiply by 1/std rounded to f32) makes of
          a uint8 image, so the de-normalised value lands within an ulp of an integer -- the regime
          real images are in, where every rounding step of the restatement shows;
  boxes   per image, x1,y1 uniform, width/height ~ U(64,512) px clipped to the image (xyxy fp32);
  logits  2*N(0,1) + (4*bitmask - 2): weakly correlated with the box, no sigmoid saturation.
All arrays are numpy (CPU); callers move what they need to the device.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np

MEAN = (123.675, 116.28, 103.53)      # configs/boxinst/boxinst_r50_fpn_1x_coco.py:93-94
STD = (58.395, 57.12, 57.375)


def make_batch(B: int = 2, H: int = 800, W: int = 1024, boxes_per_img: int = 16, inst_per_box: int = 1,
               stride: int = 4, seed: int = 0, img_shapes: Optional[Sequence[Sequence[int]]] = None,
               ori_shapes: Optional[Sequence[Sequence[int]]] = None, block: int = 16,
               logit_scale: float = 2.0, min_box: float = 64.0, max_box: float = 512.0,
               pixel_offset: float = 0.25) -> Dict:
    rng = np.random.default_rng(seed)
    h, w = H // stride, W // stride
    mean = np.asarray(MEAN, np.float32).reshape(1, 3, 1, 1)
    std = np.asarray(STD, np.float32).reshape(1, 3, 1, 1)
    gh, gw = (H + block - 1) // block, (W + block - 1) // block
    coarse = rng.integers(0, 256, size=(B, 3, gh, gw)).astype(np.float32)
    u = np.repeat(np.repeat(coarse, block, axis=2), block, axis=3)[:, :, :H, :W]
    u = np.clip(np.rint(u + rng.normal(0.0, 6.0, size=u.shape)), 0, 255).astype(np.float32)
    if pixel_offset:
        imgs = ((u + np.float32(pixel_offset)) - mean) / std
    else:
        inv = 1.0 / np.asarray(STD, np.float64).reshape(1, 3, 1, 1)
        imgs = ((u - mean).astype(np.float64) * inv).astype(np.float32)
    img_metas, gt_bboxes = [], []
    for b in range(B):
        ish = tuple(img_shapes[b]) if img_shapes is not None else (H, W)
        osh = tuple(ori_shapes[b]) if ori_shapes is not None else ish
        if ish != (H, W):                       # canvas padding is zero in network-input space
            imgs[b, :, ish[0]:, :] = 0.0
            imgs[b, :, :, ish[1]:] = 0.0
        img_metas.append(dict(img_shape=(ish[0], ish[1], 3), ori_shape=(osh[0], osh[1], 3),
                              pad_shape=(H, W, 3),
                              img_norm_cfg=dict(mean=np.asarray(MEAN, np.float32),
                                                std=np.asarray(STD, np.float32), to_rgb=True)))
        n = boxes_per_img
        bw = rng.uniform(min(min_box, ish[1] / 2), min(max_box, ish[1]), size=n)
        bh = rng.uniform(min(min_box, ish[0] / 2), min(max_box, ish[0]), size=n)
        x1 = rng.uniform(0, np.maximum(ish[1] - bw, 1.0))
        y1 = rng.uniform(0, np.maximum(ish[0] - bh, 1.0))
        x2 = np.minimum(x1 + bw, ish[1] - 1.0)
        y2 = np.minimum(y1 + bh, ish[0] - 1.0)
        gt_bboxes.append(np.stack([x1, y1, x2, y2], axis=1).astype(np.float32))
    G = B * boxes_per_img
    gt_inds = np.repeat(np.arange(G, dtype=np.int64), inst_per_box)
    N = gt_inds.shape[0]
    allb = np.concatenate(gt_bboxes, axis=0) if G else np.zeros((0, 4), np.float32)
    logits = (logit_scale * rng.standard_normal((N, 1, h, w))).astype(np.float32)
    start = stride // 2
    ys = (np.arange(h) * stride + start)[:, None]
    xs = (np.arange(w) * stride + start)[None, :]
    for n in range(N):
        x1, y1, x2, y2 = [int(v) for v in allb[gt_inds[n]]]
        inside = (ys >= y1) & (ys <= y2) & (xs >= x1) & (xs <= x2)
        logits[n, 0] += np.where(inside, 2.0, -2.0).astype(np.float32)
    return dict(imgs=imgs.astype(np.float32), img_metas=img_metas, gt_bboxes=gt_bboxes, gt_inds=gt_inds,
                mask_logits=logits, B=B, H=H, W=W, h=h, w=w, N=N, G=G, stride=stride,
                mean=MEAN, std=STD, to_rgb=True)


def cfg1(seed: int = 0) -> Dict:
    """BASELINE config 0: one 256x256 image, 4 boxes, 4 instances (the CPU-runnable plumbing case)."""
    return make_batch(B=1, H=256, W=256, boxes_per_img=4, seed=seed, min_box=32.0, max_box=160.0)


def cfg2(seed: int = 0, inst_per_box: int = 1) -> Dict:
    """BASELINE config 1 (the headline): 2 x 800 x 1024 images, 16 boxes each, 32 instances."""
    return make_batch(B=2, H=800, W=1024, boxes_per_img=16, inst_per_box=inst_per_box, seed=seed)
