"""Box2Mask / BoxLevelSet loss pieces on the HIP kernels of ``csrc/levelset.hip`` / ``csrc/meanfield.hip``:
``BoxProjectionLoss`` (``mmdet/models/losses/box_projection_loss.py:5-43``), ``LevelsetLoss`` / ``region_levelset``
(``mmdet/models/losses/levelset_loss.py:7-45``), ``LocalConsistencyModule`` and ``LCM`` (``:53-126``).
Same class names, constructor keywords, call signatures and return values; thin marshalling only, no CPU path.
(The ``tree_filter`` extension of the same row is not built.)
"""
from __future__ import annotations

import torch
from torch.autograd.function import once_differentiable

from . import _lib
from .registry import LOSSES


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def _need_cuda(**tensors):
    for name, t in tensors.items():
        if t is not None and not t.is_cuda:
            raise RuntimeError(f'{name} must be a CUDA (HIP) tensor: boxinstseg_amd has no CPU path')


def _f32(t):
    return t.detach().to(torch.float32).contiguous()


class _Projection(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scores, bitmask, weight):
        _need_cuda(mask_scores=scores, box_bitmask=bitmask)
        dev = scores.device
        s, b = _f32(scores), _f32(bitmask)
        if s.dim() != 4 or s.size(1) != 1 or b.shape != s.shape:
            raise RuntimeError(f'BoxProjectionLoss expects [N,1,H,W] scores and bitmask, got {tuple(scores.shape)} / {tuple(bitmask.shape)}')
        N, _, H, W = s.shape
        lib = _lib.load()
        loss = torch.empty((N,), dtype=torch.float32, device=dev)
        state = torch.empty(max(lib.bxi_mil_loss_state_bytes(N, H, W), 16), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check('bxi_projection_loss_forward_f32', lib.bxi_projection_loss_forward_f32(
                s.data_ptr(), b.data_ptr(), N, H, W, float(weight), loss.data_ptr(), state.data_ptr(), _stream(dev)))
        ctx.save_for_backward(state)
        ctx.meta = (N, H, W, scores.dtype)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (state,) = ctx.saved_tensors
        N, H, W, dtype = ctx.meta
        g = g.to(torch.float32).contiguous()
        gi = torch.empty((N, 1, H, W), dtype=torch.float32, device=state.device)
        with torch.cuda.device(state.device):
            _lib.check('bxi_mil_loss_backward_f32', _lib.load().bxi_mil_loss_backward_f32(
                N, H, W, state.data_ptr(), g.data_ptr(), gi.data_ptr(), _stream(state.device)))
        return gi.to(dtype), None, None


@LOSSES.register_module()
class BoxProjectionLoss(torch.nn.Module):
    def __init__(self, loss_weight=1.0):
        super().__init__()
        self.loss_weight = loss_weight

    def forward(self, mask_scores, box_bitmask):
        """[N,1,H,W], [N,1,H,W] -> [N]"""
        return _Projection.apply(mask_scores, box_bitmask, self.loss_weight)


class _Levelset(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mask_score, target, pixel_num, weight):
        _need_cuda(mask_logits=mask_score, targets=target, pixel_num=pixel_num)
        dev = mask_score.device
        m, t, pn = _f32(mask_score), _f32(target), _f32(pixel_num).view(-1)
        if m.dim() != 4 or m.size(1) != 2 or t.dim() != 4 or t.shape[0] != m.shape[0] or t.shape[2:] != m.shape[2:]:
            raise RuntimeError(f'LevelsetLoss expects scores [N,2,H,W] and targets [N,C,H,W], got {tuple(mask_score.shape)} / {tuple(target.shape)}')
        N, C, H, W = t.shape
        if pn.numel() != N:
            raise RuntimeError('pixel_num must hold one value per instance')
        lib = _lib.load()
        loss = torch.empty((N,), dtype=torch.float32, device=dev)
        state = torch.empty(max(lib.bxi_levelset_state_bytes(N, C), 16), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check('bxi_levelset_loss_forward_f32', lib.bxi_levelset_loss_forward_f32(
                m.data_ptr(), t.data_ptr(), pn.data_ptr(), N, C, H, W, float(weight), loss.data_ptr(), state.data_ptr(),
                _stream(dev)))
        ctx.save_for_backward(m, t, pn, state)
        ctx.meta = (float(weight), mask_score.dtype, target.dtype)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        m, t, pn, state = ctx.saved_tensors
        weight, dm, dt = ctx.meta
        N, C, H, W = t.shape
        g = g.to(torch.float32).contiguous()
        gm = torch.empty_like(m)
        gt = torch.empty_like(t) if ctx.needs_input_grad[1] else None
        with torch.cuda.device(m.device):
            _lib.check('bxi_levelset_loss_backward_f32', _lib.load().bxi_levelset_loss_backward_f32(
                m.data_ptr(), t.data_ptr(), pn.data_ptr(), N, C, H, W, weight, state.data_ptr(), g.data_ptr(), gm.data_ptr(),
                0 if gt is None else gt.data_ptr(), _stream(m.device)))
        return gm.to(dm), (None if gt is None else gt.to(dt)), None, None


class region_levelset(torch.nn.Module):
    """``region_levelset()(mask_score, lst_target)`` (:21-45) -> [N]"""

    def forward(self, mask_score, lst_target):
        ones = torch.ones((mask_score.size(0),), dtype=torch.float32, device=mask_score.device)
        return _Levelset.apply(mask_score, lst_target, ones, 1.0)


@LOSSES.register_module()
class LevelsetLoss(torch.nn.Module):
    def __init__(self, loss_weight=1.0):
        super().__init__()
        self.loss_weight = loss_weight

    def forward(self, mask_logits, targets, pixel_num):
        return _Levelset.apply(mask_logits, targets, pixel_num, self.loss_weight)


def _refine(aff, phi, dilation, iters, transpose):
    lib = _lib.load()
    N, _, h, w = aff.shape
    out = torch.empty_like(phi)
    ws = torch.empty(max(lib.bxi_lcm_workspace_bytes(N, h, w), 16), dtype=torch.uint8, device=phi.device)
    with torch.cuda.device(phi.device):
        _lib.check('bxi_lcm_refine_f32', lib.bxi_lcm_refine_f32(
            aff.data_ptr(), phi.data_ptr(), N, h, w, int(dilation), int(iters), int(transpose), out.data_ptr(), ws.data_ptr(),
            ws.numel(), _stream(phi.device)))
    return out


class _LcmRefine(torch.autograd.Function):
    @staticmethod
    def forward(ctx, aff, phi, dilation, iters):
        p = _f32(phi)
        ctx.save_for_backward(aff)
        ctx.meta = (dilation, iters, phi.dtype, phi.shape)
        return _refine(aff, p.view(aff.size(0), aff.size(2), aff.size(3)), dilation, iters, 0).view(phi.shape)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (aff,) = ctx.saved_tensors
        dilation, iters, dtype, shape = ctx.meta
        gg = _f32(g).view(aff.size(0), aff.size(2), aff.size(3))
        return None, _refine(aff, gg, dilation, iters, 1).view(shape).to(dtype), None, None


class LocalConsistencyModule(torch.nn.Module):
    """``LocalConsistencyModule(dilations, num_iter)(imgs, pred_phis)`` (:63-126).  One dilation (the reference passes
    ``[2]``, :55); ``imgs`` [N,C,h,w] carries no gradient, ``pred_phis`` [N,1,h,w] does."""

    def __init__(self, dilations, num_iter):
        super().__init__()
        if len(dilations) != 1:
            raise RuntimeError('LocalConsistencyModule is built for one dilation (levelset_loss.py:55 passes [2])')
        self.dilations = list(dilations)
        self.num_iter = num_iter
        self.alpha = 0.3

    def affinity(self, imgs):
        _need_cuda(imgs=imgs)
        x = _f32(imgs)
        N, C, h, w = x.shape
        aff = torch.empty((N, 8, h, w), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check('bxi_lcm_affinity_f32', _lib.load().bxi_lcm_affinity_f32(
                x.data_ptr(), N, C, h, w, int(self.dilations[0]), float(self.alpha), aff.data_ptr(), _stream(x.device)))
        return aff

    def forward(self, imgs, pred_phis):
        _need_cuda(pred_phis=pred_phis)
        if pred_phis.dim() != 4 or pred_phis.size(1) != 1:
            raise RuntimeError('pred_phis must be [N,1,h,w]')
        return _LcmRefine.apply(self.affinity(imgs), pred_phis, self.dilations[0], self.num_iter)


def LCM(imgs, pred_phis, box_targets):
    """``LCM(imgs, pred_phis, box_targets)`` (:53-60)"""
    lcm = LocalConsistencyModule(num_iter=10, dilations=[2])
    refine_phis = lcm(imgs, pred_phis)
    local_consist = (torch.abs(refine_phis - pred_phis) * box_targets).sum()
    local_regions = box_targets.sum().clamp(min=1)
    return local_consist / local_regions
