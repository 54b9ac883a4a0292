"""Shared helpers for the parity tests: run the CPU oracle / the HIP path on a synthetic batch."""
from __future__ import annotations

import numpy as np
import torch

from boxinstseg_amd import functional as F_hip
from boxinstseg_amd.functional import rows_removed
from oracle import c_oracle


def oracle_path(d, warmup=1.0, g_prj=1.0, g_pw=1.0, bottom_pixels_removed=10, size=3, dil=2, thresh=0.3,
                want_targets=True):
    hw = np.array([[m['img_shape'][0], m['img_shape'][1]] for m in d['img_metas']], np.int32).reshape(-1, 2)
    rr = np.array([rows_removed(bottom_pixels_removed, m['img_shape'], m['ori_shape']) for m in d['img_metas']],
                  np.int32)
    boxes = np.concatenate(d['gt_bboxes'], axis=0) if len(d['gt_bboxes']) else np.zeros((0, 4), np.float32)
    cfg = d['img_metas'][0]['img_norm_cfg'] if d['img_metas'] else dict(mean=d['mean'], std=d['std'], to_rgb=True)
    return c_oracle.boxinst_path(d['imgs'], hw, rr, cfg['mean'], cfg['std'], cfg['to_rgb'], boxes,
                                 np.array([len(b) for b in d['gt_bboxes']], np.int32), d['gt_inds'],
                                 d['mask_logits'][:, 0], stride=d['stride'], size=size, dil=dil,
                                 color_thresh=thresh, warmup=warmup, g_prj=g_prj, g_pw=g_pw,
                                 want_targets=want_targets)


def oracle_path_f64(d, warmup=1.0, g_prj=1.0, g_pw=1.0, bottom_pixels_removed=10, size=3, dil=2, thresh=0.3):
    hw = np.array([[m['img_shape'][0], m['img_shape'][1]] for m in d['img_metas']], np.int32).reshape(-1, 2)
    rr = np.array([rows_removed(bottom_pixels_removed, m['img_shape'], m['ori_shape']) for m in d['img_metas']], np.int32)
    boxes = np.concatenate(d['gt_bboxes'], axis=0)
    cfg = d['img_metas'][0]['img_norm_cfg']
    return c_oracle.boxinst_path_f64(d['imgs'], hw, rr, cfg['mean'], cfg['std'], cfg['to_rgb'], boxes,
                                     np.array([len(b) for b in d['gt_bboxes']], np.int32), d['gt_inds'], d['mask_logits'][:, 0],
                                     stride=d['stride'], size=size, dil=dil, color_thresh=thresh, warmup=warmup, g_prj=g_prj,
                                     g_pw=g_pw)


def to_dev(d, dev):
    return dict(imgs=torch.from_numpy(d['imgs']).to(dev),
                logits=torch.from_numpy(d['mask_logits']).to(dev),
                gt_inds=torch.from_numpy(d['gt_inds']).to(dev),
                gt_bboxes=[torch.from_numpy(b).to(dev) for b in d['gt_bboxes']])


def hip_loss(d, dev, warmup=1.0, up=None, **kw):
    """-> (loss_prj, loss_pairwise, grad[N,h,w] numpy)."""
    F_hip.DEBUG_KEEP_LAST = True
    t = to_dev(d, dev)
    logits = t['logits'].clone().requires_grad_(True)
    out = F_hip.boxinst_mask_loss(logits, t['gt_inds'], t['gt_bboxes'], imgs=t['imgs'], img_metas=d['img_metas'],
                                  out_stride=d['stride'], warmup_factor=warmup, **kw)
    if up is None:
        (out['loss_prj'] + out['loss_pairwise']).backward()
    else:
        (up[0] * out['loss_prj'] + up[1] * out['loss_pairwise']).backward()
    torch.cuda.synchronize()
    if kw.get('pairwise_dilation', 2) <= 4 and logits.size(0) > 0:
        status, rows = F_hip.last_eval_status()
        assert status == 0 and rows in (4, 8), f'in-kernel wait timed out: status {status}'
    return float(out['loss_prj'].detach()), float(out['loss_pairwise'].detach()), logits.grad.cpu().numpy()[:, 0]


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-12)


def grad_report(got, want, logits=None, tie_eps=2.5e-7):
    """max-abs error relative to max|want|; positions where the arg-max of the projection term is
    ambiguous in fp32 (top-2 sigmoid values of a row/column within a few ulp) are excluded and counted."""
    scale = np.abs(want).max() + 1e-30
    err = np.abs(got - want)
    n_tie = 0
    if logits is not None:
        s = 1.0 / (1.0 + np.exp(-logits.astype(np.float64)))
        mask = np.zeros_like(err, dtype=bool)
        for axis in (1, 2):
            top2 = np.sort(s, axis=axis)
            gap = (np.take(top2, -1, axis=axis) - np.take(top2, -2, axis=axis))
            amb = gap <= tie_eps * np.take(top2, -1, axis=axis)          # [N, other]
            if amb.any():
                idx = np.argwhere(amb)
                for n, o in idx:
                    if axis == 1:
                        mask[n, :, o] = True
                    else:
                        mask[n, o, :] = True
                n_tie += len(idx)
        err = np.where(mask, 0.0, err)
    return float(err.max() / scale), n_tie
