"""Pairwise -log P(y_i == y_j) op on MI355X: drop-in for the reference's ``mmdet.ops.pairwise``.

Two levels, same names and argument order as the reference:

* extension level -- ``pairwise_nlog_forward(size, dilation, logits)`` and
  ``pairwise_nlog_backward(size, dilation, logits, pairwise, g_pairwise)``: what the pybind11
  module ``pairwise_ext`` exports (``mmdet/ops/pairwise/csrc/pairwise/bind.cpp:15-36``).  Inputs that
  are not device-resident or not contiguous raise ``RuntimeError`` as the reference's
  ``CHECK_INPUT`` does (``pairwise.cu:7-13``); f32 and f64 are dispatched (``pairwise.cu:162``).
* autograd level -- ``pairwise_nlog(logits[N,1,H,W], size, dilation) -> [N,size^2-1,H,W]``
  (``mmdet/ops/pairwise/pairwise.py:6-26``), differentiable w.r.t. ``logits`` only.

Differences, all deliberate: kernels run on torch's *current* stream (the reference launches on
the legacy default stream); the backward is a deterministic gather (no atomics) and recomputes the
pair value, so the forward output is not kept alive for it; N == 0 is a no-op.
"""
from __future__ import annotations

import torch
from torch.autograd.function import once_differentiable

from . import _lib

_SUFFIX = {torch.float32: 'f32', torch.float64: 'f64'}


def _require_device_contiguous(**tensors: torch.Tensor) -> None:
    for name, t in tensors.items():
        if not t.is_cuda:
            raise RuntimeError(f'{name} must be a CUDA tensor')
        if not t.is_contiguous():
            raise RuntimeError(f'{name} must be contiguous')


def _geometry(logits: torch.Tensor, size: int):
    if logits.dim() != 4 or logits.size(1) != 1:
        raise RuntimeError(f'logits must be [N,1,H,W], got {tuple(logits.shape)}')
    if logits.dtype not in _SUFFIX:
        raise RuntimeError(f'"pairwise_nlog" not implemented for {logits.dtype}')
    n, _, h, w = logits.shape
    return n, h, w, size * size - 1


def _call(name: str, logits: torch.Tensor, *args) -> None:
    fn = f'{name}_{_SUFFIX[logits.dtype]}'
    with torch.cuda.device(logits.device):
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(fn, getattr(_lib.load(), fn)(*args, stream))


def pairwise_nlog_forward(pairwise_size: int, pairwise_dilation: int, logits: torch.Tensor) -> torch.Tensor:
    size, dil = int(pairwise_size), int(pairwise_dilation)
    _require_device_contiguous(logits=logits)
    n, h, w, k = _geometry(logits, size)
    out = logits.new_empty((n, k, h, w))
    _call('bxi_pairwise_nlog_forward', logits, logits.data_ptr(), n, h, w, size, dil, out.data_ptr())
    return out


def pairwise_nlog_backward(pairwise_size: int, pairwise_dilation: int, logits: torch.Tensor,
                           pairwise: torch.Tensor, g_pairwise: torch.Tensor) -> torch.Tensor:
    size, dil = int(pairwise_size), int(pairwise_dilation)
    _require_device_contiguous(logits=logits, g_pairwise=g_pairwise)
    if pairwise is not None:
        _require_device_contiguous(pairwise=pairwise)
    n, h, w, k = _geometry(logits, size)
    if tuple(g_pairwise.shape) != (n, k, h, w) or g_pairwise.dtype != logits.dtype:
        raise RuntimeError(f'g_pairwise must be {(n, k, h, w)} {logits.dtype}, got '
                           f'{tuple(g_pairwise.shape)} {g_pairwise.dtype}')
    grad = torch.empty_like(logits)
    _call('bxi_pairwise_nlog_backward', logits, logits.data_ptr(), 0 if pairwise is None else pairwise.data_ptr(),
          g_pairwise.data_ptr(), n, h, w, size, dil, grad.data_ptr())
    return grad


class PairwiseNLog(torch.autograd.Function):
    """autograd node behind :func:`pairwise_nlog`; keeps only the logits for the backward."""

    @staticmethod
    def forward(ctx, logits: torch.Tensor, pairwise_size: int, pairwise_dilation: int) -> torch.Tensor:
        x = logits.contiguous()
        ctx.window = (int(pairwise_size), int(pairwise_dilation))
        ctx.save_for_backward(x)
        return pairwise_nlog_forward(*ctx.window, x)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_pairwise: torch.Tensor):
        (x,) = ctx.saved_tensors
        return pairwise_nlog_backward(*ctx.window, x, None, grad_pairwise.contiguous()), None, None


def pairwise_nlog(logits: torch.Tensor, pairwise_size: int, pairwise_dilation: int) -> torch.Tensor:
    return PairwiseNLog.apply(logits, pairwise_size, pairwise_dilation)
