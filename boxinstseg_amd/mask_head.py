"""``CondInstMaskHead`` with the BoxInst loss path running on MI355X HIP kernels.

Drop-in for the reference's ``CondInstMaskHead`` (``mmdet/models/dense_heads/condinst_head.py:1041-1448``)
as far as the box-supervised loss path is concerned: same registry name, same constructor keywords
(``configs/boxinst/*.py`` build it unchanged), same parameters/buffers in the state dict
(``param_conv.{weight,bias}``, ``_iter``, ``sizes_of_interest``), same ``loss`` / ``get_targets`` /
``get_bitmasks_from_boxes`` signatures and return values (keys ``loss_prj`` / ``loss_pairwise``), and the methods
``CondInst.forward_train`` / ``simple_test`` call around them: ``training_sample``, ``forward``, ``simple_test``.

What is different on purpose:
  * nothing on the path leaves the device: no ``tensor2imgs`` / ``rgb2lab`` round trip, no
    per-image or per-box Python loop, no ``.item()`` for the warm-up factor (a host-side mirror of
    ``_iter`` is kept, re-read from the buffer only after ``load_state_dict``);
  * ``loss()`` produces the gradient w.r.t. ``mask_logits`` in the same pass (fused fwd+bwd);
  * zero instances / an image without boxes give zero losses instead of NaN / an exception
    (SURVEY 8a quirks 1 and 8);
  * CPU tensors raise: there is no CPU implementation in this package (the CPU restatement lives
    in ``oracle/`` and is test infrastructure only).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn

from . import _lib
from . import functional as F_hip
from .dynamic import dynamic_mask_forward
from .pairwise import pairwise_nlog
from .registry import HEADS


@HEADS.register_module()
class CondInstMaskHead(nn.Module):

    def __init__(self,
                 in_channels: int = 8,
                 in_stride: int = 8,
                 out_stride: int = 4,
                 dynamic_convs: int = 3,
                 dynamic_channels: int = 8,
                 disable_rel_coors: bool = False,
                 bbox_head_channels: int = 256,
                 sizes_of_interest: Sequence[int] = (64, 128, 256, 512, 1024),
                 max_proposals: int = 500,
                 topk_per_img: int = -1,
                 boxinst_enabled: bool = False,
                 bottom_pixels_removed: int = 10,
                 pairwise_size: int = 3,
                 pairwise_dilation: int = 2,
                 pairwise_color_thresh: float = 0.3,
                 pairwise_warmup: int = 10000,
                 norm_cfg: Optional[dict] = None,
                 init_cfg: Optional[dict] = None):
        super().__init__()
        if in_stride < out_stride or in_stride % out_stride:
            raise AssertionError('in_stride must be a multiple of out_stride')
        if dynamic_channels <= 1:
            raise AssertionError('dynamic_channels must be > 1')
        if not (max_proposals == -1 or topk_per_img == -1):
            raise AssertionError('max_proposals and topk_per_img cannot be used at the same time')
        self.in_channels = in_channels
        self.in_stride = in_stride
        self.out_stride = out_stride
        self.dynamic_convs = dynamic_convs
        self.dynamic_channels = dynamic_channels
        self.disable_rel_coors = disable_rel_coors
        self.bbox_head_channels = bbox_head_channels
        self.max_proposals = max_proposals
        self.topk_per_img = topk_per_img
        self.boxinst_enabled = boxinst_enabled
        self.bottom_pixels_removed = bottom_pixels_removed
        self.pairwise_size = pairwise_size
        self.pairwise_dilation = pairwise_dilation
        self.pairwise_color_thresh = pairwise_color_thresh
        self._warmup_iters = pairwise_warmup
        self.norm_cfg = norm_cfg if norm_cfg is not None else dict(type='BN', requires_grad=True)
        self.init_cfg = init_cfg if init_cfg is not None else dict(type='Normal', layer='Conv2d', std=0.01, bias=0)
        self.fp16_enable = False   # sic: the reference sets this (mis-spelt) attribute, condinst_head.py:1108

        # layout of the generated dynamic-conv parameters (condinst_head.py:1079-1089)
        first_in = in_channels if disable_rel_coors else in_channels + 2
        self.dy_weights: List[int] = []
        self.dy_biases: List[int] = []
        for i in range(dynamic_convs):
            cin = first_in if i == 0 else dynamic_channels
            cout = 1 if i == dynamic_convs - 1 else dynamic_channels
            self.dy_weights.append(cin * cout)
            self.dy_biases.append(cout)
        self.num_gen_params = sum(self.dy_weights) + sum(self.dy_biases)

        self.register_buffer('sizes_of_interest', torch.tensor(list(sizes_of_interest)))
        self.register_buffer('_iter', torch.zeros([1]))
        # `_iter` lives on the device only: the loss evaluation adds the 1 (condinst_head.py:1297) in its last launch and evaluates
        # the warm-up factor (:1330-1331) from it in the kernels -- no `.item()` per iteration (the reference's sync), no host copy
        # to keep in step with load_state_dict / DDP buffer broadcasts / fill_().
        self.param_conv = nn.Conv2d(bbox_head_channels, self.num_gen_params, 3, stride=1, padding=1)
        self.init_weights()

    # ---- initialisation / checkpoint compatibility ------------------------------------------------
    def init_weights(self) -> None:
        """init_cfg=dict(type='Normal', layer='Conv2d', std=0.01, bias=0) (condinst_head.py:1064-1068)."""
        cfg = self.init_cfg or {}
        if cfg.get('type') == 'Normal':
            nn.init.normal_(self.param_conv.weight, mean=cfg.get('mean', 0.0), std=cfg.get('std', 0.01))
            if self.param_conv.bias is not None:
                nn.init.constant_(self.param_conv.bias, cfg.get('bias', 0))

    def set_iter(self, value: float) -> None:
        """Set the iteration counter, e.g. when resuming by hand (``load_state_dict`` restores it like any buffer)."""
        self._iter.fill_(float(value))

    def _tick(self) -> float:
        """``self._iter += 1`` and ``min(self._iter.item() / warmup_iters, 1)`` exactly as the reference (condinst_head.py:1297,
        1330-1331), host sync included.  Only the paths that do not go through the fused evaluation use it (other window sizes, the
        fully supervised branch, a counter that is not a float32 scalar on the logits' device)."""
        self._iter += 1
        return min(self._iter.item() / float(self._warmup_iters), 1.0)

    def _counts_in_evaluation(self, like: torch.Tensor) -> bool:
        it = self._iter
        return it.dtype == torch.float32 and it.device == like.device and it.numel() == 1

    # ---- the producer of mask_logits -------------------------------------------------------
    def parse_dynamic_params(self, params):
        """condinst_head.py:1120-1137: split [N,P] into the grouped-conv weights / biases of the three
        dynamic layers (kept for callers that want the reference's view of the parameters)."""
        n = params.size(0)
        parts = list(torch.split_with_sizes(params, self.dy_weights + self.dy_biases, dim=1))
        weights, biases = parts[:self.dynamic_convs], parts[self.dynamic_convs:]
        for i in range(self.dynamic_convs):
            cout = 1 if i == self.dynamic_convs - 1 else self.dynamic_channels
            weights[i] = weights[i].reshape(n * cout, -1, 1, 1)
            biases[i] = biases[i].reshape(n * cout)
        return weights, biases

    def forward(self, feat, params, coors, level_inds, img_inds):
        """condinst_head.py:1139-1164 -> mask logits [N,1,H*f,W*f], f = in_stride // out_stride."""
        if not feat.is_cuda:
            raise RuntimeError('CondInstMaskHead.forward: feat must be a CUDA (HIP) tensor; no CPU path')
        if self.dynamic_convs != 3 or self.dynamic_channels != 8 or feat.size(1) not in (8, 16):
            # every shipped config is 3 layers x 8 channels on 8 / 16 feature channels: that is what the tuned HIP kernels are
            # instantiated for.  The reference leaves the three numbers free (:1079-1089): up to 4 layers x 16 channels on 32
            # feature channels run the general HIP kernels (csrc/dynamic_head_generic.hip); beyond that the same arithmetic
            # composed of PyTorch-ROCm ops (still on the GPU; autograd supplies the backward).
            from .dynamic import dynamic_mask_forward_generic, generic_supported
            if generic_supported(self.dynamic_convs, self.dynamic_channels, feat.size(1), self.disable_rel_coors):
                return dynamic_mask_forward_generic(feat, params, coors, level_inds, img_inds, self.sizes_of_interest,
                                                    self.dynamic_convs, self.dynamic_channels, in_stride=self.in_stride,
                                                    out_stride=self.out_stride, disable_rel_coors=self.disable_rel_coors)
            return self._composed_forward(feat, params, coors, level_inds, img_inds)
        return dynamic_mask_forward(feat, params, coors, level_inds, img_inds, self.sizes_of_interest,
                                    in_stride=self.in_stride, out_stride=self.out_stride,
                                    disable_rel_coors=self.disable_rel_coors)

    def forward_loss(self, feat, params, coors, level_inds, img_inds, imgs, img_metas, gt_inds, gt_bboxes, gt_masks=None,
                     gt_labels=None, fuse_head: bool = True):
        """``mask_logits = self(feat, params, coors, level_inds, img_inds)`` followed by ``self.loss(imgs, img_metas, mask_logits,
        gt_inds, gt_bboxes, gt_masks, gt_labels)`` -- the two calls ``CondInst.forward_train`` makes back to back
        (``mmdet/models/detectors/condinst.py:71-74``) -- as ONE call.  With ``fuse_head=True`` (and where the shapes allow) the
        dynamic head is evaluated inside the loss evaluation's first launch (``bxi_boxinst_head_eval_f32``): its tiles run side by
        side with the image pooling and leave, besides the logits, the projection maxima and the zero-filled gradient -- one read of
        the logits fewer.  Measured: 28.5 us against 30.1 us of GPU time for the two calls at the C ABI (5 %), and through this module
        146 us against 148 us of host time per forward_loss + backward (bench.py `module_api.forward_loss`, variants alternated: the
        variant a process times FIRST is up to twice as slow, which is what round 3's 655-vs-375 us figure had measured).
        Returns ``(mask_logits, losses)``; every configuration the fused launch is not built for takes the two calls."""
        factor = self.in_stride // self.out_stride
        # (after a fault in the two-launch form this thread takes the path without any in-kernel wait -- functional.note_fault -- which the
        # head-fused launch is not: the two calls then)
        # (targets prepared for this batch -- prepare_targets -- are consumed by loss(): the head-fused launch computes the image side itself,
        # so with targets waiting the two calls are the cheaper path and the prepared ones do not stay alive unused)
        fused = (fuse_head and getattr(self, '_prepared', None) is None and not F_hip._TLS.wait_free and self.boxinst_enabled and feat.is_cuda and params.size(0) > 0 and factor == 2 and self.dynamic_convs == 3 and
                 self.dynamic_channels == 8 and feat.size(1) in (8, 16) and feat.size(3) % 2 == 0 and
                 F_hip.fused_supported(self.pairwise_size, self.pairwise_dilation) and self.out_stride == 4 and
                 imgs.size(2) % 4 == 0 and imgs.size(3) % 4 == 0 and
                 feat.size(2) * self.in_stride == imgs.size(2) and feat.size(3) * self.in_stride == imgs.size(3))
        if not fused:
            logits = self(feat, params, coors, level_inds, img_inds)
            return logits, self.loss(imgs, img_metas, logits, gt_inds, gt_bboxes, gt_masks, gt_labels)
        # the checks the two calls make (DynamicMaskHead.forward, BoxInstMaskLoss): the fused entry point gets raw pointers
        n = params.size(0)
        expect = sum(self.dy_weights) + sum(self.dy_biases)
        if params.dim() != 2 or params.size(1) != expect:
            raise RuntimeError(f'params must be [N,{expect}], got {tuple(params.shape)}')
        if feat.size(1) != self.in_channels:
            raise RuntimeError(f'mask features have {feat.size(1)} channels, the head was built for {self.in_channels}')
        for name, t in (('coors', coors), ('level_inds', level_inds), ('img_inds', img_inds), ('gt_inds', gt_inds)):
            if t.size(0) != n:
                raise RuntimeError(f'{name} has {t.size(0)} entries for {n} instances')
        if len(gt_bboxes) != imgs.size(0) or len(img_metas) != imgs.size(0):
            raise RuntimeError(f'{imgs.size(0)} images but {len(gt_bboxes)} box lists / {len(img_metas)} img_metas')
        in_eval = self._counts_in_evaluation(feat)
        cfg = dict(out_stride=self.out_stride, bottom_pixels_removed=self.bottom_pixels_removed, pairwise_size=self.pairwise_size,
                   pairwise_dilation=self.pairwise_dilation, pairwise_color_thresh=self.pairwise_color_thresh,
                   warmup_factor=1.0 if in_eval else self._tick())
        if in_eval:          # counted and ramped on the device
            cfg['iter_counter'] = self._iter
            cfg['warmup_iters'] = float(self._warmup_iters)
        try:
            logits, loss_prj, loss_pw = F_hip.HeadBoxInstLoss.apply(
                feat, params, coors, level_inds, img_inds, self.sizes_of_interest, (self.in_stride, factor, self.disable_rel_coors),
                imgs, img_metas, gt_inds, gt_bboxes, cfg)
        except _lib.BoxInstHipError as e:
            if e.status != _lib.BXI_ERR_UNSUPPORTED:
                raise
            # a configuration the fused launch is not built for after all (e.g. an image tensor that is not 16-byte aligned):
            # the two calls, as documented.  The library refuses BEFORE its first launch, so nothing has been enqueued or counted:
            # the evaluation below is the one that adds the 1 (a host-side _tick() above has counted already and is not repeated).
            logits = self(feat, params, coors, level_inds, img_inds)
            losses = F_hip.boxinst_mask_loss(logits, gt_inds, gt_bboxes, imgs=imgs, img_metas=img_metas, **cfg)
            return logits, losses
        return logits, {'loss_prj': loss_prj, 'loss_pairwise': loss_pw}

    def _composed_forward(self, feat, params, coors, level_inds, img_inds):
        """The dynamic head for layer counts / widths the HIP kernels are not built for: a per-instance 1x1 convolution is
        a [cout, cin] x [cin, H*W] product, so the layers are batched matrix products over the instances."""
        n = params.size(0)
        expect = sum(self.dy_weights) + sum(self.dy_biases)
        if params.dim() != 2 or (n > 0 and params.size(1) != expect):
            raise RuntimeError(f'params must be [N,{expect}], got {tuple(params.shape)}')
        x = feat[img_inds]                                                  # [N,C,H,W]
        H, W = x.shape[2:]
        if not self.disable_rel_coors:
            half = self.in_stride // 2
            xs = torch.arange(W, device=x.device, dtype=x.dtype) * self.in_stride + half
            ys = torch.arange(H, device=x.device, dtype=x.dtype) * self.in_stride + half
            scale = self.sizes_of_interest.to(device=x.device, dtype=x.dtype)[level_inds].view(n, 1, 1)
            rel_x = ((coors[:, 0].view(n, 1, 1) - xs.view(1, 1, W)) / scale).expand(n, H, W)
            rel_y = ((coors[:, 1].view(n, 1, 1) - ys.view(1, H, 1)) / scale).expand(n, H, W)
            x = torch.cat([rel_x.unsqueeze(1), rel_y.unsqueeze(1), x], dim=1)
        x = x.flatten(2)                                                    # [N,cin,H*W]
        w_off, b_off = 0, sum(self.dy_weights)
        for i, (nw, nb) in enumerate(zip(self.dy_weights, self.dy_biases)):
            w = params[:, w_off:w_off + nw].reshape(n, nb, nw // nb)       # [N,cout,cin], rows = output channels
            b = params[:, b_off:b_off + nb].reshape(n, nb, 1)
            x = torch.baddbmm(b, w, x)
            if i < self.dynamic_convs - 1:
                x = torch.relu(x)
            w_off += nw
            b_off += nb
        from .dynamic import aligned_bilinear
        return aligned_bilinear(x.reshape(n, 1, H, W), self.in_stride // self.out_stride)

    # ---- the callers either side of the path: instance sampling (training) and mask post-processing (test) -----
    def training_sample(self, cls_scores, centernesses, param_preds, coors, level_inds, img_inds, gt_inds):
        """condinst_head.py:1166-1232: pick the positive locations whose dynamic parameters become instances.

        ``max_proposals``: the first ``min(max_proposals, P)`` positives in a random order (the reference draws
        ``randperm`` of that count, not a random subset).  ``topk_per_img``: per image, every ground-truth box keeps its
        ``max(int(topk_per_img / boxes_in_image), 1)`` best locations by ``sigmoid(cls).max * sigmoid(centerness)``;
        output order = image, then box index, then (boxes over the quota) descending score / (others) location order --
        the reference's nested Python loops, here as fixed-shape sorts / scatter-adds: two host synchronisations in all (the
        positive mask and the final selection have data-dependent lengths, as in the reference), none per image or per box.
        Returns ``(param_preds [N,P], coors, level_inds, img_inds, gt_inds)``."""
        def flat(ts):
            return torch.cat([t.permute(0, 2, 3, 1).flatten(end_dim=2) for t in ts], dim=0)

        params = flat(param_preds)
        pos = (gt_inds != -1).nonzero(as_tuple=True)[0]
        params, coors, level_inds, img_inds, gt_inds = (t[pos] for t in (params, coors, level_inds, img_inds, gt_inds))
        n = params.size(0)
        if self.max_proposals != -1:
            keep = torch.randperm(min(self.max_proposals, n), device=params.device)
        elif self.topk_per_img != -1:
            score = flat(cls_scores)[pos].sigmoid().max(dim=1)[0] * flat(centernesses).reshape(-1)[pos].sigmoid()
            keep = _topk_per_box(img_inds, gt_inds, score, self.topk_per_img, int(cls_scores[0].size(0)))
        else:       # the reference leaves `sampled_inds` unbound here (an exception); keep every positive instead
            keep = torch.arange(n, device=params.device)
        return params[keep], coors[keep], level_inds[keep], img_inds[keep], gt_inds[keep]

    def simple_test(self, mask_feat, det_labels, det_params, det_coors, det_level_inds, img_metas, num_classes,
                    rescale=False):
        """condinst_head.py:1234-1286: masks of the detections -> per image, per class lists of uint8 [h,w] arrays."""
        import numpy as np
        from .dynamic import aligned_bilinear
        counts = [int(p.size(0)) for p in det_params]
        if sum(counts) == 0:
            return [[[] for _ in range(num_classes)] for _ in img_metas]
        img_inds = torch.cat([torch.full((c,), i, dtype=torch.long, device=mask_feat.device) for i, c in enumerate(counts)])
        logits = self.forward(mask_feat, torch.cat(det_params), torch.cat(det_coors), torch.cat(det_level_inds), img_inds)
        probs = aligned_bilinear(logits.sigmoid(), self.out_stride)
        results = []
        for cur, labels, meta in zip(probs.split(counts, dim=0), det_labels, img_metas):
            ih, iw = meta['img_shape'][:2]
            cur = cur[:, :, :ih, :iw]
            if rescale and cur.size(0):
                oh, ow = meta['ori_shape'][:2]
                cur = torch.nn.functional.interpolate(cur, (oh, ow), mode='bilinear', align_corners=False)
            masks = (cur.squeeze(1) > 0.5).cpu().numpy().astype(np.uint8)
            lab = labels.detach().cpu().numpy()
            results.append([masks[lab == c] for c in range(num_classes)])
        return results

    # ---- targets ----------------------------------------------------------------------------------
    def get_targets(self, gt_bboxes, gt_masks, img, img_metas):
        """condinst_head.py:1345-1393 -> (similarities, bitmasks, bitmasks_full).

        BoxInst: ``similarities[i]`` is ``[G_i,K,h,w]`` (a zero-copy expand of the image's map; the
        reference concatenates G_i copies), ``bitmasks[i]`` ``[G_i,h,w]``, ``bitmasks_full[i]``
        ``[G_i,H,W]``.  Fully supervised: the strided ground-truth masks (:1384-1391)."""
        if not self.boxinst_enabled:
            start = int(self.out_stride // 2)
            bitmasks = [m[:, start::self.out_stride, start::self.out_stride] for m in gt_masks]
            return None, bitmasks, gt_masks
        sim, _, _ = F_hip.color_affinity(
            img, img_metas, out_stride=self.out_stride, bottom_pixels_removed=self.bottom_pixels_removed,
            pairwise_size=self.pairwise_size, pairwise_dilation=self.pairwise_dilation,
            pairwise_color_thresh=self.pairwise_color_thresh, want_similarity=True, want_bits=False)
        return self._per_image_targets(gt_bboxes, sim, img.shape[2], img.shape[3])

    def get_bitmasks_from_boxes(self, gt_bboxes, padded_images, padded_image_masks):
        """condinst_head.py:1395-1448.  ``padded_images`` [B,3,H,W] RGB in 0..255 (integer valued),
        ``padded_image_masks`` [B,H,W]."""
        stride = self.out_stride
        assert padded_images.size(2) % stride == 0
        assert padded_images.size(3) % stride == 0
        B, _, H, W = padded_images.shape
        metas = [dict(img_shape=(H, W, 3), ori_shape=(H, W, 3)) for _ in range(B)]
        sim, _, _ = F_hip.color_affinity(
            padded_images.float(), metas, out_stride=stride, bottom_pixels_removed=0,
            pairwise_size=self.pairwise_size, pairwise_dilation=self.pairwise_dilation,
            pairwise_color_thresh=self.pairwise_color_thresh, want_similarity=True, want_bits=False,
            image_masks=padded_image_masks, denormalize=False)
        return self._per_image_targets(gt_bboxes, sim, H, W)

    def _per_image_targets(self, gt_bboxes, sim: torch.Tensor, H: int, W: int):
        stride, start = self.out_stride, int(self.out_stride // 2)
        counts = [int(b.size(0)) for b in gt_bboxes]
        small = F_hip.box_bitmasks(gt_bboxes, H, W, stride, start).split(counts, dim=0)
        full = F_hip.box_bitmasks(gt_bboxes, H, W, 1, 0).split(counts, dim=0)
        similarities = [sim[i:i + 1].expand(n, -1, -1, -1) for i, n in enumerate(counts)]
        return similarities, list(small), list(full)

    def prepare_targets(self, imgs, img_metas, gt_bboxes, stream=None) -> bool:
        """OPTIONAL, ahead of :meth:`loss`: the reference computes its targets at the top of ``loss`` from ``imgs`` and ``gt_bboxes``
        alone (``self.get_targets(gt_bboxes, gt_masks, imgs, img_metas)``, condinst_head.py:1298-1299), and both exist before the backbone
        runs (``mmdet/models/detectors/condinst.py:53`` vs ``:73``).  A detector that calls this at the top of ``forward_train`` -- with
        ``stream`` a side stream -- takes the image -> Lab -> colour predicates -> pair-count chain off the loss's critical path;
        ``loss()`` finds the result if it is called with the same ``imgs`` / ``gt_bboxes`` and uses it (``BXI_EVAL_TARGETS_READY``), and
        behaves exactly as before otherwise.  Returns whether targets were prepared (not for windows / thresholds the fused path is
        not built for)."""
        self._prepared = None
        if not (self.boxinst_enabled and imgs.is_cuda and F_hip.fused_supported(self.pairwise_size, self.pairwise_dilation)):
            return False
        self._prepared = F_hip.prepare_targets(
            imgs, img_metas, gt_bboxes, out_stride=self.out_stride, bottom_pixels_removed=self.bottom_pixels_removed,
            pairwise_size=self.pairwise_size, pairwise_dilation=self.pairwise_dilation,
            pairwise_color_thresh=self.pairwise_color_thresh, stream=stream)
        return self._prepared is not None

    # ---- loss ---------------------------------------------------------------------------------------
    def loss(self, imgs, img_metas, mask_logits, gt_inds, gt_bboxes, gt_masks, gt_labels) -> Dict[str, torch.Tensor]:
        """condinst_head.py:1288-1343."""
        if not mask_logits.is_cuda:
            raise RuntimeError('CondInstMaskHead.loss: mask_logits must be a CUDA (HIP) tensor; '
                               'boxinstseg_amd has no CPU loss path')
        if self.boxinst_enabled and F_hip.fused_supported(self.pairwise_size, self.pairwise_dilation):
            in_eval = self._counts_in_evaluation(mask_logits)
            prepared, self._prepared = getattr(self, '_prepared', None), None       # one batch's targets serve one loss() call
            return F_hip.boxinst_mask_loss(
                mask_logits, gt_inds, gt_bboxes, targets=prepared, imgs=imgs, img_metas=img_metas, out_stride=self.out_stride,
                bottom_pixels_removed=self.bottom_pixels_removed, pairwise_size=self.pairwise_size,
                pairwise_dilation=self.pairwise_dilation, pairwise_color_thresh=self.pairwise_color_thresh,
                warmup_factor=1.0 if in_eval else self._tick(), iter_counter=self._iter if in_eval else None,
                warmup_iters=float(self._warmup_iters) if in_eval else None)
        warmup = self._tick()
        if not self.boxinst_enabled:
            return {'loss_mask': self._supervised_loss(mask_logits, gt_inds, gt_masks)}
        return self._composed_loss(imgs, img_metas, mask_logits, gt_inds, gt_bboxes, warmup)

    def _composed_loss(self, imgs, img_metas, mask_logits, gt_inds, gt_bboxes, warmup: float):
        """Other window sizes: the reference's own composition (:1314-1332) over the HIP op and the HIP
        target kernels (the fused kernel is specialised for the 3x3 window every config uses)."""
        if mask_logits.size(0) == 0:
            zero = 0 * mask_logits.sum()
            return {'loss_prj': zero, 'loss_pairwise': zero}
        similarities, bitmasks, _ = self.get_targets(gt_bboxes, None, imgs, img_metas)
        scores = mask_logits.sigmoid()
        bm = torch.cat(bitmasks, dim=0)[gt_inds].unsqueeze(1).to(scores.dtype)
        sim = torch.cat(similarities, dim=0)[gt_inds].to(scores.dtype)
        loss_prj = (_dice(scores.max(dim=2, keepdim=True)[0], bm.max(dim=2, keepdim=True)[0]) +
                    _dice(scores.max(dim=3, keepdim=True)[0], bm.max(dim=3, keepdim=True)[0])).mean()
        pw = pairwise_nlog(mask_logits, self.pairwise_size, self.pairwise_dilation)
        weights = (sim >= self.pairwise_color_thresh).to(scores.dtype) * bm
        loss_pw = (pw * weights).sum() / weights.sum().clamp(min=1.0) * warmup
        return {'loss_prj': loss_prj, 'loss_pairwise': loss_pw}

    def _supervised_loss(self, mask_logits, gt_inds, gt_masks):
        """Fully supervised CondInst branch (:1338-1341); not part of the BoxInst hot path, plain torch."""
        if mask_logits.size(0) == 0:
            return 0 * mask_logits.sum()
        start = int(self.out_stride // 2)
        bm = torch.cat([m[:, start::self.out_stride, start::self.out_stride] for m in gt_masks], dim=0)
        bm = bm[gt_inds].unsqueeze(1).to(mask_logits.dtype)
        return _dice(mask_logits.sigmoid(), bm).mean()


def _topk_per_box(img_inds: torch.Tensor, gt_inds: torch.Tensor, score: torch.Tensor, topk_per_img: int,
                  num_imgs: int) -> torch.Tensor:
    """Indices kept by the ``topk_per_img`` rule of ``training_sample`` (condinst_head.py:1201-1225), in its order.

    Fixed-shape tensor ops only (sorts, cumulative sums, scatter-adds over ``n`` or ``num_imgs`` slots): the one host
    synchronisation is the final boolean selection, whose length is data dependent (as the reference's output is)."""
    n = score.numel()
    dev = score.device
    if n == 0:
        return torch.zeros(0, dtype=torch.long, device=dev)
    ar = torch.arange(n, device=dev)
    key = img_inds.long() * (1 << 32) + gt_inds.long()                        # (image, box), ascending like the loops
    by_key = torch.argsort(key, stable=True)                                   # location order inside a group
    skey = key[by_key]
    first = torch.ones(n, dtype=torch.bool, device=dev)
    first[1:] = skey[1:] != skey[:-1]
    seg = torch.cumsum(first.long(), 0) - 1                                    # group id of every sorted element, 0..n-1
    size = torch.zeros(n, dtype=torch.long, device=dev).scatter_add_(0, seg, torch.ones_like(seg))
    # distinct boxes with a positive, per image -> quota of each group
    boxes_in_img = torch.zeros(num_imgs, dtype=torch.long, device=dev).scatter_add_(
        0, img_inds.long()[by_key], first.long())
    quota = torch.clamp(torch.div(topk_per_img, boxes_in_img[img_inds.long()[by_key]], rounding_mode='floor'), min=1)
    over = size[seg] > quota
    # inside a group: descending score if it is over its quota, location order otherwise
    by_score = torch.empty(n, dtype=torch.long, device=dev)
    by_score[torch.argsort(score, descending=True, stable=True)] = ar
    second = torch.where(over, by_score[by_key], by_key)
    inner = torch.argsort(second, stable=True)
    order = inner[torch.argsort(seg[inner], stable=True)]                      # positions in the key-sorted list
    start = torch.cumsum(size, 0) - size                                       # first position of group g (groups are contiguous)
    rank = ar - start[seg[order]]
    return by_key[order][rank < quota[order]]


def _dice(x: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    n = x.size(0)
    x, target = x.reshape(n, -1), target.reshape(n, -1)
    inter = (x * target).sum(dim=1)
    union = (x ** 2.0).sum(dim=1) + (target ** 2.0).sum(dim=1) + 1e-5
    return 1.0 - 2 * inter / union
