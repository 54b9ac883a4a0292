"""The producer of ``mask_logits``: ``CondInstMaskHead.forward`` of the reference
(``mmdet/models/dense_heads/condinst_head.py:1139-1164``) as one HIP kernel forward and two backward
(SURVEY 8(f-2)): relative coordinates, the three per-instance dynamic 1x1 convolutions with ReLU,
``aligned_bilinear`` -- see ``csrc/dynamic_head.hip``.  Thin marshalling only; no CPU path.
"""
from __future__ import annotations

import torch
from torch.autograd.function import once_differentiable

from . import _lib


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


class DynamicMaskHead(torch.autograd.Function):
    """logits[N,1,fH,fW] = f(feat[B,C,H,W], params[N,P]); differentiable w.r.t. feat and params."""

    @staticmethod
    def forward(ctx, feat, params, coors, level_inds, img_inds, sizes_of_interest, in_stride, factor,
                disable_rel_coors):
        for name, t in (('feat', feat), ('params', params), ('coors', coors), ('level_inds', level_inds),
                        ('img_inds', img_inds), ('sizes_of_interest', sizes_of_interest)):
            if not t.is_cuda:
                raise RuntimeError(f'{name} must be a CUDA (HIP) tensor: boxinstseg_amd has no CPU path')
        dev = feat.device
        B, C, H, W = feat.shape
        N = params.size(0)
        f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
        i64 = lambda t: t.detach().to(device=dev, dtype=torch.int64).contiguous()
        feat_c, params_c, coors_c = f32(feat), f32(params), f32(coors).view(-1, 2)
        lvl, img, soi = i64(level_inds), i64(img_inds), f32(sizes_of_interest)
        expect = (C + (0 if disable_rel_coors else 2)) * 8 + 64 + 8 + 8 + 8 + 1
        if params_c.dim() != 2 or (N > 0 and params_c.size(1) != expect):
            raise RuntimeError(f'params must be [N,{expect}] for {C} feature channels, got {tuple(params.shape)}')
        out = torch.empty((N, 1, H * factor, W * factor), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check('bxi_dynamic_mask_forward_f32', _lib.load().bxi_dynamic_mask_forward_f32(
                feat_c.data_ptr(), B, C, H, W, params_c.data_ptr(), N, coors_c.data_ptr(), lvl.data_ptr(),
                img.data_ptr(), soi.data_ptr(), soi.numel(), int(in_stride), int(factor), int(bool(disable_rel_coors)),
                out.data_ptr(), _stream(dev)))
        ctx.save_for_backward(feat_c, params_c, coors_c, lvl, img, soi)
        ctx.cfg = (int(in_stride), int(factor), int(bool(disable_rel_coors)))
        ctx.dtypes = (feat.dtype, params.dtype)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        feat, params, coors, lvl, img, soi = ctx.saved_tensors
        in_stride, factor, no_rel = ctx.cfg
        dev = feat.device
        B, C, H, W = feat.shape
        N = params.size(0)
        g = g.to(torch.float32).contiguous()
        g_feat = torch.empty_like(feat)
        g_params = torch.empty_like(params)
        lib = _lib.load()
        ws = torch.empty(max(lib.bxi_dynamic_mask_backward_workspace_bytes(B, C, H, W, N, no_rel), 256),
                         dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check('bxi_dynamic_mask_backward_f32', lib.bxi_dynamic_mask_backward_f32(
                feat.data_ptr(), B, C, H, W, params.data_ptr(), N, coors.data_ptr(), lvl.data_ptr(), img.data_ptr(),
                soi.data_ptr(), soi.numel(), in_stride, factor, no_rel, g.data_ptr(), g_feat.data_ptr(),
                g_params.data_ptr(), ws.data_ptr(), ws.numel(), _stream(dev)))
        return (g_feat.to(ctx.dtypes[0]), g_params.to(ctx.dtypes[1]), None, None, None, None, None, None, None)


def generic_supported(dynamic_convs: int, dynamic_channels: int, in_channels: int, disable_rel_coors: bool) -> bool:
    """The shapes ``csrc/dynamic_head_generic.hip`` is built for."""
    return 1 <= dynamic_convs <= 4 and 1 <= dynamic_channels <= 16 and 1 <= in_channels and in_channels + (0 if disable_rel_coors else 2) <= 34


class GenericDynamicMaskHead(torch.autograd.Function):
    """``DynamicMaskHead`` for every head shape the reference's constructor admits (condinst_head.py:1079-1089): ``layers``
    dynamic convolutions of ``channels`` channels -- HIP forward and backward (``csrc/dynamic_head_generic.hip``)."""

    @staticmethod
    def forward(ctx, feat, params, coors, level_inds, img_inds, sizes_of_interest, in_stride, factor, disable_rel_coors, layers,
                channels):
        for name, t in (('feat', feat), ('params', params), ('coors', coors), ('level_inds', level_inds),
                        ('img_inds', img_inds), ('sizes_of_interest', sizes_of_interest)):
            if not t.is_cuda:
                raise RuntimeError(f'{name} must be a CUDA (HIP) tensor: boxinstseg_amd has no CPU path')
        dev = feat.device
        B, C, H, W = feat.shape
        N = params.size(0)
        f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
        i64 = lambda t: t.detach().to(device=dev, dtype=torch.int64).contiguous()
        feat_c, params_c, coors_c = f32(feat), f32(params), f32(coors).view(-1, 2)
        lvl, img, soi = i64(level_inds), i64(img_inds), f32(sizes_of_interest)
        cin = C + (0 if disable_rel_coors else 2)
        expect = (cin + 1) if layers == 1 else (cin * channels + (layers - 2) * channels * channels + channels +
                                                  (layers - 1) * channels + 1)
        if params_c.dim() != 2 or (N > 0 and params_c.size(1) != expect):
            raise RuntimeError(f'params must be [N,{expect}] for {layers} layers x {channels} channels on {C} feature channels, '
                               f'got {tuple(params.shape)}')
        for name, t in (('coors', coors_c), ('level_inds', lvl), ('img_inds', img)):
            if t.size(0) != N:
                raise RuntimeError(f'{name} has {t.size(0)} entries for {N} instances')
        out = torch.empty((N, 1, H * factor, W * factor), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check('bxi_dynamic_mask_generic_forward_f32', _lib.load().bxi_dynamic_mask_generic_forward_f32(
                feat_c.data_ptr(), B, C, H, W, params_c.data_ptr(), N, int(layers), int(channels), coors_c.data_ptr(), lvl.data_ptr(),
                img.data_ptr(), soi.data_ptr(), soi.numel(), int(in_stride), int(factor), int(bool(disable_rel_coors)),
                out.data_ptr(), _stream(dev)))
        ctx.save_for_backward(feat_c, params_c, coors_c, lvl, img, soi)
        ctx.cfg = (int(in_stride), int(factor), int(bool(disable_rel_coors)), int(layers), int(channels))
        ctx.dtypes = (feat.dtype, params.dtype)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        feat, params, coors, lvl, img, soi = ctx.saved_tensors
        in_stride, factor, no_rel, layers, channels = ctx.cfg
        dev = feat.device
        B, C, H, W = feat.shape
        N = params.size(0)
        g = g.to(torch.float32).contiguous()
        g_feat = torch.empty_like(feat)
        g_params = torch.empty_like(params)
        lib = _lib.load()
        ws = torch.empty(max(lib.bxi_dynamic_mask_generic_backward_workspace_bytes(B, C, H, W, N, layers, channels, no_rel), 256),
                         dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check('bxi_dynamic_mask_generic_backward_f32', lib.bxi_dynamic_mask_generic_backward_f32(
                feat.data_ptr(), B, C, H, W, params.data_ptr(), N, layers, channels, coors.data_ptr(), lvl.data_ptr(), img.data_ptr(),
                soi.data_ptr(), soi.numel(), in_stride, factor, no_rel, g.data_ptr(), g_feat.data_ptr(), g_params.data_ptr(),
                ws.data_ptr(), ws.numel(), _stream(dev)))
        return (g_feat.to(ctx.dtypes[0]), g_params.to(ctx.dtypes[1])) + (None,) * 9


def dynamic_mask_forward_generic(feat, params, coors, level_inds, img_inds, sizes_of_interest, dynamic_convs, dynamic_channels,
                                 in_stride=8, out_stride=4, disable_rel_coors=False):
    """``CondInstMaskHead.forward`` for a head of ``dynamic_convs`` layers x ``dynamic_channels`` channels -> ``[N,1,H*f,W*f]``."""
    if in_stride % out_stride:
        raise RuntimeError('in_stride must be a multiple of out_stride')
    return GenericDynamicMaskHead.apply(feat, params, coors, level_inds, img_inds, sizes_of_interest, in_stride,
                                        in_stride // out_stride, disable_rel_coors, dynamic_convs, dynamic_channels)


def dynamic_mask_forward(feat, params, coors, level_inds, img_inds, sizes_of_interest, in_stride=8, out_stride=4,
                         disable_rel_coors=False):
    """``CondInstMaskHead.forward(feat, params, coors, level_inds, img_inds)`` -> ``[N,1,H*f,W*f]``."""
    if in_stride % out_stride:
        raise RuntimeError('in_stride must be a multiple of out_stride')
    return DynamicMaskHead.apply(feat, params, coors, level_inds, img_inds, sizes_of_interest, in_stride,
                                 in_stride // out_stride, disable_rel_coors)


def aligned_bilinear(tensor, factor):
    """The module-level ``aligned_bilinear(tensor, factor)`` of condinst_head.py:146-167 for the test-time path
    (``simple_test`` up-samples the mask probabilities to the input resolution): output sample ``(y, x)`` sits at source
    coordinate ``((y - factor // 2) / factor, (x - factor // 2) / factor)``, clamped to the map.  Composed of torch ops on
    whatever device the tensor lives on; the training-time ``x factor`` step is fused into ``dyn_fwd_kernel`` instead."""
    import torch.nn.functional as F
    assert tensor.dim() == 4 and factor >= 1 and int(factor) == factor
    if factor == 1:
        return tensor
    h, w = tensor.shape[2:]
    x = F.pad(tensor, (0, 1, 0, 1), mode='replicate')
    x = F.interpolate(x, size=(factor * h + 1, factor * w + 1), mode='bilinear', align_corners=True)
    x = F.pad(x, (factor // 2, 0, factor // 2, 0), mode='replicate')
    return x[:, :, :factor * h, :factor * w]
