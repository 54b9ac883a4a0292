"""ctypes binding of ``oracle/_ref/libpairwise_ref.so``: the reference's OWN ``pairwise.cu`` kernels
(mmdet/ops/pairwise/csrc/pairwise/pairwise.cu:15-147) executed on the CPU.

TEST INFRASTRUCTURE ONLY.  The library is built by ``make -C oracle ref`` in the build container, from the reference tree
(see oracle/ref_wrap/pairwise_kernels_wrap.cpp and cuda_on_cpu.h); it travels to the GPU box with the snapshot, the
reference tree does not.  Used to pin the C oracle (``bxo_pairwise_nlog_*``) and, through fixtures and live runs, the HIP op.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_ref', 'libpairwise_ref.so')


def available() -> bool:
    return os.path.exists(_SO)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _suf(dtype):
    return {np.dtype(np.float32): 'f32', np.dtype(np.float64): 'f64'}[np.dtype(dtype)]


def forward(logits: np.ndarray, size: int, dil: int) -> np.ndarray:
    """pairwise_nlog_forward_kernel: logits [B,1,H,W] (f32 or f64) -> [B, size^2-1, H, W]"""
    lg = np.ascontiguousarray(logits)
    B, _, H, W = lg.shape
    out = np.empty((B, size * size - 1, H, W), lg.dtype)
    getattr(C.CDLL(_SO), f'ref_pairwise_nlog_forward_{_suf(lg.dtype)}')(int(size), int(dil), _p(lg), B, H, W, _p(out))
    return out


def backward(logits: np.ndarray, pairwise: np.ndarray, g_pairwise: np.ndarray, size: int, dil: int) -> np.ndarray:
    """pairwise_nlog_backward_kernel on zeros_like(logits): -> g_logits [B,1,H,W]"""
    lg = np.ascontiguousarray(logits)
    pw = np.ascontiguousarray(pairwise, lg.dtype); gp = np.ascontiguousarray(g_pairwise, lg.dtype)
    B, _, H, W = lg.shape
    out = np.empty_like(lg)
    getattr(C.CDLL(_SO), f'ref_pairwise_nlog_backward_{_suf(lg.dtype)}')(int(size), int(dil), _p(lg), _p(pw), _p(gp), B, H, W, _p(out))
    return out
