"""Host side of the BoxInst mask-loss path: thin torch wrappers over the C ABI of libboxinst_hip.so.

Everything here only *marshals*: it reads tensor pointers / shapes / the current HIP stream, lets
torch own the memory, and calls ``include/boxinst_hip.h`` entry points.  No arithmetic of the path
is done in Python or in torch ops, there is no CPU fallback, and nothing synchronises the host.

Reference interfaces mirrored (``condinst_head.py`` = ``mmdet/models/dense_heads/condinst_head.py``):
  color_affinity      get_original_image :170-186 + get_targets :1345-1393 +
                      get_bitmasks_from_boxes :1395-1424 + get_image_color_similarity :220-246
  box_bitmasks        get_bitmasks_from_boxes :1426-1438
  boxinst_mask_loss   CondInstMaskHead.loss :1297-1337 (boxinst branch), forward + backward
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch.autograd.function import once_differentiable

from . import _lib


def _require_cuda(**tensors: Optional[torch.Tensor]) -> None:
    for name, t in tensors.items():
        if t is not None and not t.is_cuda:
            raise RuntimeError(f'{name} must be a CUDA (HIP) tensor: boxinstseg_amd has no CPU path')


def _stream(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def rows_removed(bottom_pixels_removed: int, img_shape: Sequence[int], ori_shape: Sequence[int]) -> int:
    """``int(bottom_pixels_removed * float(img_h) / float(ori_h))`` -- condinst_head.py:1358-1361."""
    return int(bottom_pixels_removed * float(img_shape[0]) / float(ori_shape[0]))


class _Batch:
    """Keeps the ctypes arrays behind a ``bxi_image_batch`` alive for the duration of a call."""

    def __init__(self, imgs: torch.Tensor, img_metas: Sequence[dict], bottom_pixels_removed: int,
                 image_masks: Optional[torch.Tensor] = None, denormalize: bool = True):
        if imgs.dim() != 4 or imgs.size(1) != 3:
            raise RuntimeError(f'imgs must be [B,3,H,W], got {tuple(imgs.shape)}')
        if imgs.dtype != torch.float32:
            raise RuntimeError(f'imgs must be float32, got {imgs.dtype}')
        B, _, Hc, Wc = imgs.shape
        if B > _lib.BXI_MAX_IMAGES:
            raise RuntimeError(f'at most {_lib.BXI_MAX_IMAGES} images per call, got {B}')
        if len(img_metas) != B:
            raise RuntimeError(f'{B} images but {len(img_metas)} img_metas')
        self.imgs = imgs.contiguous()
        self.masks = None if image_masks is None else image_masks.to(torch.float32).contiguous()
        self._h = _lib.int_array(m['img_shape'][0] for m in img_metas)
        self._w = _lib.int_array(m['img_shape'][1] for m in img_metas)
        self._rm = _lib.int_array(rows_removed(bottom_pixels_removed, m['img_shape'], m['ori_shape'])
                                  for m in img_metas)
        s = _lib.ImageBatch()
        s.imgs = self.imgs.data_ptr()
        s.B, s.Hc, s.Wc = B, Hc, Wc
        s.img_h_host = C.cast(self._h, C.POINTER(C.c_int))
        s.img_w_host = C.cast(self._w, C.POINTER(C.c_int))
        s.rows_removed_host = C.cast(self._rm, C.POINTER(C.c_int))
        if denormalize and B > 0:
            cfg = img_metas[0]['img_norm_cfg']
            for m in img_metas[1:]:
                c2 = m['img_norm_cfg']
                if list(c2['mean']) != list(cfg['mean']) or list(c2['std']) != list(cfg['std']) or \
                        bool(c2['to_rgb']) != bool(cfg['to_rgb']):
                    raise RuntimeError('all images of a batch must share img_norm_cfg')
            mean, std, to_rgb = cfg['mean'], cfg['std'], bool(cfg['to_rgb'])
        else:  # already an RGB image in 0..255 (the padded_images argument of get_bitmasks_from_boxes)
            mean, std, to_rgb = (0.0, 0.0, 0.0), (1.0, 1.0, 1.0), True
        for i in range(3):
            s.mean[i] = float(mean[i])
            s.std[i] = float(std[i])
        s.to_rgb = int(to_rgb)
        s.image_masks = 0 if self.masks is None else self.masks.data_ptr()
        self.struct = s
        self.B, self.Hc, self.Wc = B, Hc, Wc


class _Inst:
    """Keeps the ctypes arrays behind a ``bxi_instances`` alive for the duration of a call."""

    def __init__(self, mask_logits: torch.Tensor, gt_inds: torch.Tensor, gt_bboxes: Sequence[torch.Tensor],
                 Hc: int, Wc: int, stride: int):
        if mask_logits.dim() != 4 or mask_logits.size(1) != 1:
            raise RuntimeError(f'mask_logits must be [N,1,h,w], got {tuple(mask_logits.shape)}')
        N, _, h, w = mask_logits.shape
        if h * stride != Hc or w * stride != Wc:
            raise RuntimeError(f'mask_logits {h}x{w} x stride {stride} != image canvas {Hc}x{Wc}')
        if len(gt_bboxes) > _lib.BXI_MAX_IMAGES:
            raise RuntimeError(f'at most {_lib.BXI_MAX_IMAGES} images per call')
        self.logits = mask_logits.detach().to(torch.float32).contiguous()
        self.gt_inds = gt_inds.to(device=mask_logits.device, dtype=torch.int64).contiguous()
        if self.gt_inds.numel() != N:
            raise RuntimeError(f'{N} instances but {self.gt_inds.numel()} gt_inds')
        self.boxes = [b.detach().to(device=mask_logits.device, dtype=torch.float32).contiguous().view(-1, 4)
                      for b in gt_bboxes]
        self._ptrs = _lib.ptr_array(b.data_ptr() if b.numel() else 0 for b in self.boxes)
        self._cnt = _lib.int_array(b.size(0) for b in self.boxes)
        s = _lib.Instances()
        s.logits = self.logits.data_ptr()
        s.N, s.h, s.w = N, h, w
        s.gt_inds = self.gt_inds.data_ptr()
        s.boxes_per_img_host = C.cast(self._ptrs, C.POINTER(C.c_void_p))
        s.gt_count_host = C.cast(self._cnt, C.POINTER(C.c_int))
        s.B = len(self.boxes)
        s.Hc, s.Wc, s.stride = Hc, Wc, stride
        self.struct = s
        self.N, self.h, self.w = N, h, w


# ------------------------------------------------------------------------------------------------
# targets
# ------------------------------------------------------------------------------------------------
def color_affinity(imgs: torch.Tensor, img_metas: Sequence[dict], *, out_stride: int = 4,
                   bottom_pixels_removed: int = 10, pairwise_size: int = 3, pairwise_dilation: int = 2,
                   pairwise_color_thresh: float = 0.3, want_similarity: bool = True, want_bits: bool = True,
                   image_masks: Optional[torch.Tensor] = None, denormalize: bool = True
                   ) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor], torch.Tensor]:
    """-> (sim [B,K,h,w] f32 | None, bits [B,h,w] u8/int32 | None, lab [B,3,h,w] f32)."""
    _require_cuda(imgs=imgs, image_masks=image_masks)
    batch = _Batch(imgs, img_metas, bottom_pixels_removed, image_masks, denormalize)
    B, Hc, Wc = batch.B, batch.Hc, batch.Wc
    if Hc % out_stride or Wc % out_stride:
        raise RuntimeError(f'canvas {Hc}x{Wc} is not a multiple of out_stride {out_stride}')
    h, w = Hc // out_stride, Wc // out_stride
    K = pairwise_size * pairwise_size - 1
    dev = imgs.device
    lab = torch.empty((B, 3, h, w), dtype=torch.float32, device=dev)
    sim = torch.empty((B, K, h, w), dtype=torch.float32, device=dev) if want_similarity else None
    bits = None
    if want_bits:
        bits = torch.empty((B, h, w), dtype=torch.uint8 if K <= 8 else torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check('bxi_color_affinity_f32', _lib.load().bxi_color_affinity_f32(
            C.byref(batch.struct), int(out_stride), int(pairwise_size), int(pairwise_dilation),
            float(pairwise_color_thresh), lab.data_ptr(), 0, 0 if sim is None else sim.data_ptr(),
            0 if bits is None else bits.data_ptr(), _stream(dev)))
    return sim, bits, lab


def box_bitmasks(gt_bboxes: Sequence[torch.Tensor], Hc: int, Wc: int, stride: int, start: int) -> torch.Tensor:
    """Per-box {0,1} masks sampled at ``[start::stride, start::stride]`` -> [G, ., .] f32 (:1426-1432)."""
    if not gt_bboxes:
        raise RuntimeError('gt_bboxes is empty')
    _require_cuda(**{f'gt_bboxes[{i}]': b for i, b in enumerate(gt_bboxes)})
    dev = gt_bboxes[0].device
    boxes = [b.detach().to(torch.float32).contiguous().view(-1, 4) for b in gt_bboxes]
    ptrs = _lib.ptr_array(b.data_ptr() if b.numel() else 0 for b in boxes)
    cnt = _lib.int_array(b.size(0) for b in boxes)
    G = sum(b.size(0) for b in boxes)
    h, w = (Hc - start + stride - 1) // stride, (Wc - start + stride - 1) // stride
    out = torch.empty((G, h, w), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check('bxi_box_bitmasks_f32', _lib.load().bxi_box_bitmasks_f32(
            C.cast(ptrs, C.POINTER(C.c_void_p)), C.cast(cnt, C.POINTER(C.c_int)), len(boxes), int(Hc), int(Wc),
            int(stride), int(start), out.data_ptr(), _stream(dev)))
    return out


# ------------------------------------------------------------------------------------------------
# loss
# ------------------------------------------------------------------------------------------------
class BoxInstMaskLoss(torch.autograd.Function):
    """(loss_prj, loss_pairwise) = f(mask_logits); forward and backward in ONE pass over the logits.

    forward  : bxi_boxinst_eval_f32 (image side + fused loss) -- or bxi_boxinst_loss_fwd_bwd_f32 when
               precomputed affinity bits are given -- writes both scalars and the un-finished
               gradient (zeros + the un-normalised pairwise gradient on the box tiles).
    backward : bxi_boxinst_loss_backward_f32 normalises, adds the projection gradient and folds the
               two upstream scalars in, reading them from device memory (no host sync).
    """

    @staticmethod
    def forward(ctx, mask_logits: torch.Tensor, imgs: Optional[torch.Tensor], img_metas, gt_inds: torch.Tensor,
                gt_bboxes, cfg: Dict, affinity_bits: Optional[torch.Tensor]):
        _require_cuda(mask_logits=mask_logits, imgs=imgs, gt_inds=gt_inds, affinity_bits=affinity_bits)
        dev = mask_logits.device
        stride, size, dil = int(cfg['out_stride']), int(cfg['pairwise_size']), int(cfg['pairwise_dilation'])
        if imgs is not None:
            Hc, Wc = imgs.shape[2], imgs.shape[3]
        else:
            Hc, Wc = mask_logits.shape[2] * stride, mask_logits.shape[3] * stride
        inst = _Inst(mask_logits, gt_inds, gt_bboxes, Hc, Wc, stride)
        need_grad = bool(ctx.needs_input_grad[0])
        lib = _lib.load()
        losses = torch.empty(2, dtype=torch.float32, device=dev)
        grad = torch.empty_like(inst.logits) if need_grad else None
        state = None
        if need_grad:
            state = torch.empty(max(lib.bxi_boxinst_loss_state_bytes(inst.N, inst.h, inst.w), 256),
                                dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            if affinity_bits is None:
                batch = _Batch(imgs, img_metas, int(cfg['bottom_pixels_removed']))
                nbytes = lib.bxi_boxinst_eval_workspace_bytes(batch.B, Hc, Wc, stride, inst.N)
                ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev)
                _lib.check('bxi_boxinst_eval_f32', lib.bxi_boxinst_eval_f32(
                    C.byref(batch.struct), C.byref(inst.struct), size, dil, float(cfg['pairwise_color_thresh']),
                    float(cfg['warmup_factor']), losses.data_ptr(), 0 if grad is None else grad.data_ptr(),
                    0 if state is None else state.data_ptr(), ws.data_ptr(), ws.numel(), _stream(dev)))
            else:
                bits = affinity_bits.contiguous()
                if bits.dtype != torch.uint8 or tuple(bits.shape) != (len(gt_bboxes), inst.h, inst.w):
                    raise RuntimeError('affinity_bits must be uint8 [B,h,w]')
                nbytes = lib.bxi_boxinst_loss_workspace_bytes(inst.N, inst.h, inst.w)
                ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev)
                _lib.check('bxi_boxinst_loss_fwd_bwd_f32', lib.bxi_boxinst_loss_fwd_bwd_f32(
                    C.byref(inst.struct), bits.data_ptr(), size, dil, float(cfg['warmup_factor']),
                    losses.data_ptr(), 0 if grad is None else grad.data_ptr(),
                    0 if state is None else state.data_ptr(), ws.data_ptr(), ws.numel(), _stream(dev)))
        ctx.inst = inst
        ctx.dil = dil
        ctx.grad = grad
        ctx.state = state
        ctx.in_dtype = mask_logits.dtype
        return losses[0], losses[1]

    @staticmethod
    @once_differentiable
    def backward(ctx, g_prj: torch.Tensor, g_pw: torch.Tensor):
        grad, inst = ctx.grad, ctx.inst
        if grad is None:
            raise RuntimeError('BoxInstMaskLoss.backward called twice (the fused gradient buffer is finished in '
                               'place by the first call) or without a gradient request')
        ctx.grad = None    # hand the buffer to autograd; a second backward must re-run the forward
        dev = grad.device
        if inst.N > 0:
            g_prj = g_prj.to(device=dev, dtype=torch.float32).contiguous()
            g_pw = g_pw.to(device=dev, dtype=torch.float32).contiguous()
            with torch.cuda.device(dev):
                _lib.check('bxi_boxinst_loss_backward_f32', _lib.load().bxi_boxinst_loss_backward_f32(
                    C.byref(inst.struct), g_prj.data_ptr(), g_pw.data_ptr(), ctx.dil, ctx.state.data_ptr(),
                    grad.data_ptr(), _stream(dev)))
        if grad.dtype != ctx.in_dtype:
            grad = grad.to(ctx.in_dtype)
        return grad, None, None, None, None, None, None


def boxinst_mask_loss(mask_logits: torch.Tensor, gt_inds: torch.Tensor, gt_bboxes: Sequence[torch.Tensor], *,
                      imgs: Optional[torch.Tensor] = None, img_metas: Optional[Sequence[dict]] = None,
                      affinity_bits: Optional[torch.Tensor] = None, out_stride: int = 4,
                      bottom_pixels_removed: int = 10, pairwise_size: int = 3, pairwise_dilation: int = 2,
                      pairwise_color_thresh: float = 0.3, warmup_factor: float = 1.0) -> Dict[str, torch.Tensor]:
    """The BoxInst branch of ``CondInstMaskHead.loss`` (condinst_head.py:1297-1337) as one call.

    Either ``imgs`` + ``img_metas`` (targets are computed on the device from the network input) or
    precomputed ``affinity_bits`` (from :func:`color_affinity`) must be given.
    Returns ``{'loss_prj', 'loss_pairwise'}`` attached to the autograd graph of ``mask_logits``.
    """
    if affinity_bits is None and (imgs is None or img_metas is None):
        raise RuntimeError('need imgs + img_metas or affinity_bits')
    if pairwise_size != 3:
        raise RuntimeError('the fused path is built for pairwise_size == 3; use boxinst_mask_loss_composed')
    cfg = dict(out_stride=out_stride, bottom_pixels_removed=bottom_pixels_removed, pairwise_size=pairwise_size,
               pairwise_dilation=pairwise_dilation, pairwise_color_thresh=pairwise_color_thresh,
               warmup_factor=warmup_factor)
    loss_prj, loss_pw = BoxInstMaskLoss.apply(mask_logits, imgs, img_metas, gt_inds, list(gt_bboxes), cfg,
                                              affinity_bits)
    return {'loss_prj': loss_prj, 'loss_pairwise': loss_pw}
