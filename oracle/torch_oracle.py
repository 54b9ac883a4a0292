"""CPU restatement of the BoxInst mask-loss path in torch ops (the reference's own CPU-runnable form).

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module, and only as the checker / the timed CPU baseline; the
product (``boxinstseg_amd``) never imports it.

Each function cites the reference lines it follows (``condinst_head.py`` =
``mmdet/models/dense_heads/condinst_head.py`` in LiWentomng/BoxInstSeg).  The op sequence is kept
the same as the reference's (``F.unfold`` -> reshape -> drop centre, ``F.logsigmoid`` ...), so that
timing this module on the host cores is timing the reference's CPU loss path.

Parity status: the five torch-only functions are PINNED against the reference's functions
(AST-extracted and executed by ``tests/golden/make_golden.py``; fixtures in ``tests/golden``).
``denormalize_u8`` (mmcv/cv2 ``tensor2imgs``) and ``rgb2lab`` (scikit-image) restate third-party
code that is not in the reference tree and not installed here: parity UNPINNED for those two.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------
# neighbourhood gather -- condinst_head.py:190-217 (unfold_wo_center)
# --------------------------------------------------------------------------------------------
def neighbours(x: torch.Tensor, size: int, dilation: int) -> torch.Tensor:
    """[N,C,H,W] -> [N,C,size^2-1,H,W]; zero ("SAME") padding; window order row-major, centre dropped."""
    assert x.dim() == 4 and size % 2 == 1
    pad = (size + (dilation - 1) * (size - 1)) // 2
    cols = F.unfold(x, kernel_size=size, padding=pad, dilation=dilation)
    cols = cols.reshape(x.size(0), x.size(1), size * size, x.size(2), x.size(3))
    mid = (size * size) // 2
    return torch.cat((cols[:, :, :mid], cols[:, :, mid + 1:]), dim=2)


# --------------------------------------------------------------------------------------------
# pairwise term -- condinst_head.py:86-114 (compute_pairwise_term); equals the CUDA op
# (pairwise.cu:38-50) including the zero-padded log-prob convention at the border.
# --------------------------------------------------------------------------------------------
def pairwise_term(mask_logits: torch.Tensor, size: int, dilation: int) -> torch.Tensor:
    """[N,1,H,W] -> [N,size^2-1,H,W]: -log P(y_i == y_j)."""
    assert mask_logits.dim() == 4
    lf = F.logsigmoid(mask_logits)
    lb = F.logsigmoid(-mask_logits)
    same_fg = lf[:, :, None] + neighbours(lf, size, dilation)
    same_bg = lb[:, :, None] + neighbours(lb, size, dilation)
    top = torch.max(same_fg, same_bg)
    log_same = torch.log(torch.exp(same_fg - top) + torch.exp(same_bg - top)) + top
    return -log_same[:, 0]


# --------------------------------------------------------------------------------------------
# projection term -- condinst_head.py:117-143
# --------------------------------------------------------------------------------------------
def dice(x: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    n = x.size(0)
    x = x.reshape(n, -1)
    target = target.reshape(n, -1)
    inter = (x * target).sum(dim=1)
    union = (x ** 2.0).sum(dim=1) + (target ** 2.0).sum(dim=1) + 1e-5
    return 1.0 - (2 * inter / union)


def project_term(mask_scores: torch.Tensor, gt_bitmasks: torch.Tensor) -> torch.Tensor:
    ly = dice(mask_scores.max(dim=2, keepdim=True)[0], gt_bitmasks.max(dim=2, keepdim=True)[0])
    lx = dice(mask_scores.max(dim=3, keepdim=True)[0], gt_bitmasks.max(dim=3, keepdim=True)[0])
    return (lx + ly).mean()


# --------------------------------------------------------------------------------------------
# colour similarity -- condinst_head.py:220-246
# --------------------------------------------------------------------------------------------
def color_similarity(lab: torch.Tensor, mask: torch.Tensor, size: int, dilation: int) -> torch.Tensor:
    """lab [1,3,h,w], mask [h,w] -> [1,size^2-1,h,w]."""
    assert lab.dim() == 4 and lab.size(0) == 1
    diff = lab.unsqueeze(2) - neighbours(lab, size, dilation)
    sim = torch.exp(-torch.norm(diff, dim=1) * 0.5)
    wgt = neighbours(mask[None, None], size, dilation)[:, 0]
    return sim * wgt


# --------------------------------------------------------------------------------------------
# third-party pieces, restated (UNPINNED)
# --------------------------------------------------------------------------------------------
def denormalize_u8(img: torch.Tensor, img_shape: Sequence[int], mean, std, to_rgb: bool) -> torch.Tensor:
    """condinst_head.py:170-186.  img [3,Hc,Wc] normalised -> [3,img_h,img_w] float RGB in 0..255.

    mmcv ``imdenormalize``: ``cv2.multiply(img, std_f64)`` (OpenCV works in double for mul/div by a
    scalar row and rounds once to the f32 image) then ``cv2.add(img, mean_f64)`` (OpenCV demotes a
    float64 scalar to float32 against a float32 array: a plain f32 add) -- then optional RGB->BGR;
    ``astype(uint8)`` truncates; the caller reverses channels again (:182).  Net channel map:
    c -> (c if to_rgb else 2-c).
    """
    ih, iw = int(img_shape[0]), int(img_shape[1])
    x = img[:, :ih, :iw].detach().cpu().to(torch.float32).numpy()
    mean = np.asarray(mean, dtype=np.float64).reshape(3, 1, 1)
    std = np.asarray(std, dtype=np.float64).reshape(3, 1, 1)
    t = (x.astype(np.float64) * std).astype(np.float32)
    v = t + mean.astype(np.float32)
    u8 = v.astype(np.int32).astype(np.uint8)
    if not to_rgb:
        u8 = u8[::-1]
    return torch.from_numpy(np.ascontiguousarray(u8)).float()


_M = np.array([[0.412453, 0.357580, 0.180423],
               [0.212671, 0.715160, 0.072169],
               [0.019334, 0.119193, 0.950227]], dtype=np.float64)
_WHITE = np.array([0.95047, 1.0, 1.08883], dtype=np.float64)


def rgb2lab(rgb_u8: np.ndarray) -> np.ndarray:
    """skimage.color.rgb2lab (D65, 2 deg) on [...,3] uint8 -> float64 (call site condinst_head.py:1413)."""
    assert rgb_u8.dtype == np.uint8 and rgb_u8.shape[-1] == 3
    c = rgb_u8.astype(np.float64) / 255.0
    hi = c > 0.04045
    lin = np.where(hi, np.power((c + 0.055) / 1.055, 2.4), c / 12.92)
    xyz = lin @ _M.T / _WHITE
    big = xyz > 0.008856
    f = np.where(big, np.cbrt(xyz), 7.787 * xyz + 16.0 / 116.0)
    L = 116.0 * f[..., 1] - 16.0
    a = 500.0 * (f[..., 0] - f[..., 1])
    b = 200.0 * (f[..., 1] - f[..., 2])
    return np.stack([L, a, b], axis=-1)


# --------------------------------------------------------------------------------------------
# targets -- condinst_head.py:1345-1448
# --------------------------------------------------------------------------------------------
def rows_removed(bottom_pixels_removed: int, img_h: int, ori_h: int) -> int:
    """condinst_head.py:1358-1361."""
    return int(bottom_pixels_removed * float(img_h) / float(ori_h))


def get_targets(imgs: torch.Tensor, img_metas: List[dict], gt_bboxes: List[torch.Tensor], *,
                out_stride: int = 4, bottom_pixels_removed: int = 10, pairwise_size: int = 3,
                pairwise_dilation: int = 2) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
    """-> (similarities: list[B] of [G_i,K,h,w], bitmasks: list[B] of [G_i,h,w]).  bitmasks_full is
    omitted: the reference's loss() never reads it."""
    B, _, Hc, Wc = imgs.shape
    stride, start = out_stride, out_stride // 2
    assert Hc % stride == 0 and Wc % stride == 0
    masks, images = [], []
    for i in range(B):                                               # :1353-1375
        ih, iw = img_metas[i]['img_shape'][:2]
        m = torch.ones((ih, iw), dtype=torch.float32)
        pr = rows_removed(bottom_pixels_removed, ih, img_metas[i]['ori_shape'][0])
        if pr > 0:
            m[-pr:, :] = 0
        pad = (0, Wc - iw, 0, Hc - ih)
        masks.append(F.pad(m, pad))
        cfg = img_metas[i]['img_norm_cfg']
        rgb = denormalize_u8(imgs[i], (ih, iw), cfg['mean'], cfg['std'], cfg['to_rgb'])
        images.append(F.pad(rgb, pad))
    masks = torch.stack(masks)
    images = torch.stack(images)
    small = F.avg_pool2d(images.float(), kernel_size=stride, stride=stride, padding=0)   # :1403
    small_masks = masks[:, start::stride, start::stride]                                # :1405
    sims, bitmasks = [], []
    for i, boxes in enumerate(gt_bboxes):                                               # :1412
        lab = rgb2lab(small[i].byte().permute(1, 2, 0).numpy())
        lab = torch.as_tensor(lab, dtype=torch.float32).permute(2, 0, 1)[None]
        sim = color_similarity(lab, small_masks[i], pairwise_size, pairwise_dilation)
        per_box = []
        for box in boxes.detach().cpu():                                                # :1426-1432
            full = torch.zeros((Hc, Wc), dtype=torch.float32)
            full[int(box[1]):int(box[3]) + 1, int(box[0]):int(box[2]) + 1] = 1.0
            per_box.append(full[start::stride, start::stride])
        n = len(per_box)
        bitmasks.append(torch.stack(per_box) if n else torch.zeros((0, Hc // stride, Wc // stride)))
        sims.append(sim.expand(n, -1, -1, -1))      # reference cats n copies (:1443); same values
    return sims, bitmasks


# --------------------------------------------------------------------------------------------
# loss glue -- condinst_head.py:1297-1337
# --------------------------------------------------------------------------------------------
def loss_given_targets(mask_logits: torch.Tensor, sim_per_inst: torch.Tensor, bitmask_per_inst: torch.Tensor,
                       *, pairwise_size: int = 3, pairwise_dilation: int = 2,
                       pairwise_color_thresh: float = 0.3, warmup_factor: float = 1.0) -> Dict[str, torch.Tensor]:
    """mask_logits [N,1,h,w]; sim_per_inst [N,K,h,w]; bitmask_per_inst [N,1,h,w]."""
    if mask_logits.size(0) == 0:            # documented deviation: SURVEY 8a quirk 1
        z = 0 * mask_logits.sum()
        return {'loss_prj': z, 'loss_pairwise': z}
    scores = mask_logits.sigmoid()
    bm = bitmask_per_inst.to(scores.dtype)
    loss_prj = project_term(scores, bm)
    pw = pairwise_term(mask_logits, pairwise_size, pairwise_dilation)
    weights = (sim_per_inst >= pairwise_color_thresh).to(scores.dtype) * bm
    loss_pw = (pw * weights).sum() / weights.sum().clamp(min=1.0)
    return {'loss_prj': loss_prj, 'loss_pairwise': loss_pw * warmup_factor}


def mask_loss(imgs: torch.Tensor, img_metas: List[dict], mask_logits: torch.Tensor, gt_inds: torch.Tensor,
              gt_bboxes: List[torch.Tensor], *, out_stride: int = 4, bottom_pixels_removed: int = 10,
              pairwise_size: int = 3, pairwise_dilation: int = 2, pairwise_color_thresh: float = 0.3,
              warmup_factor: float = 1.0, targets: Optional[tuple] = None) -> Dict[str, torch.Tensor]:
    """Whole path: condinst_head.py:1288-1343 with boxinst_enabled=True (all CPU)."""
    if targets is None:
        targets = get_targets(imgs.detach().cpu(), img_metas, gt_bboxes, out_stride=out_stride,
                              bottom_pixels_removed=bottom_pixels_removed, pairwise_size=pairwise_size,
                              pairwise_dilation=pairwise_dilation)
    sims, bitmasks = targets
    bm = torch.cat(bitmasks, dim=0)[gt_inds].unsqueeze(1)
    sim = torch.cat([s for s in sims], dim=0)[gt_inds]
    return loss_given_targets(mask_logits, sim.to(mask_logits.dtype), bm,
                              pairwise_size=pairwise_size, pairwise_dilation=pairwise_dilation,
                              pairwise_color_thresh=pairwise_color_thresh, warmup_factor=warmup_factor)


# --------------------------------------------------------------------------------------------
# producer of the logits -- condinst_head.py:146-167 (aligned_bilinear), :1139-1164 (forward)
# --------------------------------------------------------------------------------------------
def aligned_upsample(x: torch.Tensor, factor: int) -> torch.Tensor:
    """[N,C,h,w] -> [N,C,factor*h,factor*w] with the sampling grid of the reference's aligned_bilinear."""
    if factor == 1:
        return x
    h, w = x.shape[2:]
    x = F.pad(x, (0, 1, 0, 1), mode='replicate')
    x = F.interpolate(x, size=(factor * h + 1, factor * w + 1), mode='bilinear', align_corners=True)
    x = F.pad(x, (factor // 2, 0, factor // 2, 0), mode='replicate')
    return x[:, :, :factor * h, :factor * w]


def dynamic_mask_forward(feat, params, coors, level_inds, img_inds, sizes_of_interest, *, in_stride=8, out_stride=4,
                         dynamic_channels=8, disable_rel_coors=False):
    """feat [B,C,H,W], params [N,P] -> logits [N,1,H*f,W*f] (three per-instance 1x1 convs as grouped convs)."""
    x = feat[img_inds]
    n, c, h, w = x.shape
    if not disable_rel_coors:
        xs = torch.arange(0, w * in_stride, in_stride, dtype=x.dtype, device=x.device) + in_stride // 2
        ys = torch.arange(0, h * in_stride, in_stride, dtype=x.dtype, device=x.device) + in_stride // 2
        loc = torch.stack([xs[None, :].expand(h, w), ys[:, None].expand(h, w)], dim=0)          # [2,h,w] (x,y)
        rel = (coors[:, :, None, None] - loc[None]) / sizes_of_interest.float()[level_inds][:, None, None, None]
        x = torch.cat([rel, x], dim=1)
        c += 2
    sizes = [c * dynamic_channels, dynamic_channels * dynamic_channels, dynamic_channels,
             dynamic_channels, dynamic_channels, 1]
    w0, w1, w2, b0, b1, b2 = torch.split_with_sizes(params, sizes, dim=1)
    x = x.reshape(1, n * c, h, w)
    x = F.relu(F.conv2d(x, w0.reshape(n * dynamic_channels, c, 1, 1), b0.reshape(-1), groups=n))
    x = F.relu(F.conv2d(x, w1.reshape(n * dynamic_channels, dynamic_channels, 1, 1), b1.reshape(-1), groups=n))
    x = F.conv2d(x, w2.reshape(n, dynamic_channels, 1, 1), b2.reshape(-1), groups=n)
    return aligned_upsample(x.permute(1, 0, 2, 3), in_stride // out_stride)


def warmup_factor(iteration: float, warmup_iters: int) -> float:
    """condinst_head.py:1330-1331."""
    return min(iteration / float(warmup_iters), 1.0)

