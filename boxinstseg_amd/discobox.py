"""DiscoBox pseudo-label path: ``MeanField``, ``dice_loss`` and ``mil_loss`` of the reference's
``mmdet/models/dense_heads/discobox_head.py`` (:585-655, :542-550, :552-562) on the HIP kernels of
``csrc/meanfield.hip``.  Same names, constructor keywords, call signatures and return values; thin marshalling
only, no CPU path.
"""
from __future__ import annotations

import torch
from torch.autograd.function import once_differentiable

from . import _lib


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def _need_cuda(**tensors):
    for name, t in tensors.items():
        if t is not None and not t.is_cuda:
            raise RuntimeError(f'{name} must be a CUDA (HIP) tensor: boxinstseg_amd has no CPU path')


def _target_arg(t: torch.Tensor):
    """0/1 targets as the kernels read them: uint8/bool without a copy, anything else as fp32."""
    if t.dtype in (torch.uint8, torch.bool):
        return t.contiguous().view(torch.uint8), 1
    return t.to(torch.float32).contiguous(), 0


def meanfield_kernel(feature_map: torch.Tensor, kernel_size: int = 3, alpha0: float = 3.0, theta0: float = 0.5,
                     theta1: float = 30.0) -> torch.Tensor:
    """``feature_map`` [B,C,H,W] -> neighbourhood kernel [B,k*k,H,W] (``MeanField.__init__`` :597-611)."""
    _need_cuda(feature_map=feature_map)
    f = feature_map.detach().to(torch.float32).contiguous()
    B, Cc, H, W = f.shape
    out = torch.empty((B, kernel_size * kernel_size, H, W), dtype=torch.float32, device=f.device)
    with torch.cuda.device(f.device):
        _lib.check('bxi_meanfield_kernel_f32', _lib.load().bxi_meanfield_kernel_f32(
            f.data_ptr(), B, Cc, H, W, int(kernel_size), float(alpha0), float(theta0), float(theta1), out.data_ptr(),
            _stream(f.device)))
    return out


def meanfield_forward(kernel: torch.Tensor, x: torch.Tensor, targets: torch.Tensor, iters: int, base: float,
                      img_inds: torch.Tensor = None, inter_img_mask: torch.Tensor = None, gamma: float = 0.01):
    """Batched ``MeanField.forward``: kernel [B,k*k,H,W], x / targets [N,(1,)H,W], img_inds [N] -> (ret like x, valid [N])."""
    _need_cuda(kernel=kernel, x=x, targets=targets, img_inds=img_inds, inter_img_mask=inter_img_mask)
    dev = x.device
    B, K2, H, W = kernel.shape
    ks = int(round(K2 ** 0.5))
    shape = x.shape
    xc = x.detach().to(torch.float32).contiguous().view(-1, H, W)
    N = xc.size(0)
    tc, t_u8 = _target_arg(targets.detach())
    if tc.numel() != N * H * W:
        raise RuntimeError(f'targets {tuple(targets.shape)} do not match x {tuple(shape)}')
    img = None if img_inds is None else img_inds.detach().to(device=dev, dtype=torch.int64).contiguous()
    inter = None
    if inter_img_mask is not None:
        inter = inter_img_mask.detach().to(torch.float32).contiguous()
        if inter.numel() != N * 2 * H * W:
            raise RuntimeError('inter_img_mask must be [N,2,H,W]')
    ret = torch.empty((N, H, W), dtype=torch.float32, device=dev)
    valid = torch.empty((N,), dtype=torch.float32, device=dev)
    kc = kernel.detach().to(torch.float32).contiguous()
    lib = _lib.load()
    ws = torch.empty(max(lib.bxi_meanfield_workspace_bytes(N, H, W), 8), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check('bxi_meanfield_forward_f32', lib.bxi_meanfield_forward_f32(
            kc.data_ptr(), B, H, W, ks, xc.data_ptr(), tc.data_ptr(), t_u8, 0 if img is None else img.data_ptr(), N,
            int(iters), float(base), 0 if inter is None else inter.data_ptr(), float(gamma), ret.data_ptr(),
            valid.data_ptr(), ws.data_ptr(), ws.numel(), _stream(dev)))
    return ret.view(shape), valid


class MeanField(torch.nn.Module):
    """Drop-in for the reference's ``MeanField`` (one object per image; ``feature_map`` [1,3,H,W])."""

    def __init__(self, feature_map, kernel_size=3, require_grad=False, theta0=0.5, theta1=30, theta2=10, alpha0=3,
                 iter=20, base=0.45, gamma=0.01):
        super().__init__()
        self.require_grad = require_grad
        self.kernel_size = kernel_size
        self.theta0, self.theta1, self.theta2, self.alpha0 = theta0, theta1, theta2, alpha0
        self.gamma, self.base, self.iter = gamma, base, iter
        with torch.no_grad():
            self.feature_map = feature_map + 10
            k = meanfield_kernel(feature_map, kernel_size, alpha0, theta0, theta1)      # [B,k*k,H,W]
            self._kernel = k
            self.kernel = k.view(k.size(0), 1, k.size(1), -1)                           # the reference's layout (:611-612)

    def forward(self, x, targets, inter_img_mask=None):
        """The reference broadcasts ``kernel [B,1,k*k,HW]`` against ``x [N,...]`` (discobox_head.py:617-655): B == 1 serves any
        number of instances (every call site of the reference), B == N pairs instance i with image i; anything else fails
        to broadcast there and raises here."""
        B = self._kernel.size(0)
        img_inds = None
        if B != 1:
            if x.size(0) != B:
                raise RuntimeError(f'MeanField built from {B} feature maps cannot serve {x.size(0)} instances '
                                   '(the reference broadcasts kernel [B,...] against x [N,...]: B must be 1 or N)')
            img_inds = torch.arange(B, device=x.device)
        with torch.no_grad():
            return meanfield_forward(self._kernel, x, targets, self.iter, self.base, img_inds, inter_img_mask, self.gamma)


class _DiceLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, target):
        _need_cuda(input=inp, target=target)
        dev = inp.device
        N = inp.size(0)
        L = 1
        for d in inp.shape[1:]:
            L *= int(d)
        ic = inp.detach().to(torch.float32).contiguous().view(N, L)
        tc, t_u8 = _target_arg(target.detach())
        if tc.numel() != ic.numel():
            raise RuntimeError(f'target {tuple(target.shape)} does not match input {tuple(inp.shape)}')
        loss = torch.empty((N,), dtype=torch.float32, device=dev)
        sums = torch.empty((max(N, 1), 2), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check('bxi_dice_loss_forward_f32', _lib.load().bxi_dice_loss_forward_f32(
                ic.data_ptr(), tc.data_ptr(), t_u8, N, max(L, 1), loss.data_ptr(), sums.data_ptr(), _stream(dev)))
        ctx.save_for_backward(ic, tc, sums)
        ctx.meta = (t_u8, inp.shape, inp.dtype)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        ic, tc, sums = ctx.saved_tensors
        t_u8, shape, dtype = ctx.meta
        N, L = ic.shape
        g = g.to(torch.float32).contiguous()
        gi = torch.empty_like(ic)
        with torch.cuda.device(ic.device):
            _lib.check('bxi_dice_loss_backward_f32', _lib.load().bxi_dice_loss_backward_f32(
                ic.data_ptr(), tc.data_ptr(), t_u8, N, max(L, 1), sums.data_ptr(), g.data_ptr(), gi.data_ptr(),
                _stream(ic.device)))
        return gi.view(shape).to(dtype), None


def dice_loss(input, target):
    """``dice_loss(input, target)`` (:542-550) -> [N]; differentiable w.r.t. ``input``."""
    return _DiceLoss.apply(input, target)


class _MilLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, target):
        _need_cuda(input=inp, target=target)
        if inp.dim() != 3:
            raise RuntimeError('mil_loss expects input [N,H,W]')
        dev = inp.device
        N, H, W = inp.shape
        ic = inp.detach().to(torch.float32).contiguous()
        tc, t_u8 = _target_arg(target.detach())
        if tc.numel() != ic.numel():
            raise RuntimeError(f'target {tuple(target.shape)} does not match input {tuple(inp.shape)}')
        lib = _lib.load()
        loss = torch.empty((N,), dtype=torch.float32, device=dev)
        state = torch.empty(max(lib.bxi_mil_loss_state_bytes(N, H, W), 16), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check('bxi_mil_loss_forward_f32', lib.bxi_mil_loss_forward_f32(
                ic.data_ptr(), tc.data_ptr(), t_u8, N, H, W, loss.data_ptr(), state.data_ptr(), _stream(dev)))
        ctx.save_for_backward(state)
        ctx.meta = (N, H, W, inp.dtype)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (state,) = ctx.saved_tensors
        N, H, W, dtype = ctx.meta
        g = g.to(torch.float32).contiguous()
        gi = torch.empty((N, H, W), dtype=torch.float32, device=state.device)
        with torch.cuda.device(state.device):
            _lib.check('bxi_mil_loss_backward_f32', _lib.load().bxi_mil_loss_backward_f32(
                N, H, W, state.data_ptr(), g.data_ptr(), gi.data_ptr(), _stream(state.device)))
        return gi.to(dtype), None


def mil_loss(loss_func, input, _, target):
    """``mil_loss(loss_func, input, _, target)`` (:552-562).  ``loss_func`` must be this module's ``dice_loss`` (the only
    one the reference passes, :1289); the row/column maxima and both dice terms are one launch."""
    if loss_func is not dice_loss:
        raise RuntimeError('mil_loss is built for loss_func = dice_loss (discobox_head.py:1289)')
    return _MilLoss.apply(input, target)
