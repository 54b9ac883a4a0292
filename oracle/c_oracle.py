"""ctypes binding of ``oracle/_build/libboxinst_oracle.so`` (plain-C CPU oracle).

TEST INFRASTRUCTURE ONLY -- see ``boxinst_oracle.h``.  Inputs/outputs are numpy arrays.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libboxinst_oracle.so')
_lib = None


def build(force: bool = False) -> str:
    """Compile the C oracle with the committed Makefile (gcc).  Returns the .so path."""
    # always go through make (dependency-driven, a no-op when up to date) so an edited source is never
    # checked against a stale library; under a file lock, so that concurrent test processes do not rebuild it at once
    import fcntl
    os.makedirs(os.path.join(_HERE, '_build'), exist_ok=True)
    with open(os.path.join(_HERE, '_build', '.lock'), 'w') as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        subprocess.run(['make', '-C', _HERE] + (['-B'] if force else []), check=True, stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.bxo_project_term_f32.restype = C.c_float
        _lib.bxo_project_term_f64.restype = C.c_double
        _lib.bxo_max_threads.restype = C.c_int
    return _lib


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dtype) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=dtype)


def _real(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return 'f32', C.c_float
    if dtype == np.float64:
        return 'f64', C.c_double
    raise TypeError(dtype)


def set_threads(n: int) -> None:
    lib().bxo_set_threads(C.c_int(n))


def max_threads() -> int:
    return lib().bxo_max_threads()


# ---- image side ---------------------------------------------------------------------------
def denormalize_u8(img: np.ndarray, img_h: int, img_w: int, mean, std, to_rgb: bool) -> np.ndarray:
    img = _c(img, np.float32)
    _, Hc, Wc = img.shape
    out = np.empty((3, Hc, Wc), np.uint8)
    m = (C.c_double * 3)(*[float(v) for v in mean])
    s = (C.c_double * 3)(*[float(v) for v in std])
    lib().bxo_denormalize_u8(_p(img), Hc, Wc, int(img_h), int(img_w), m, s, int(bool(to_rgb)), _p(out))
    return out


def pool_u8(rgb: np.ndarray, stride: int) -> np.ndarray:
    rgb = _c(rgb, np.uint8)
    _, Hc, Wc = rgb.shape
    out = np.empty((3, Hc // stride, Wc // stride), np.uint8)
    lib().bxo_pool_u8(_p(rgb), Hc, Wc, int(stride), _p(out))
    return out


def rgb2lab_u8(rgb_planar: np.ndarray) -> np.ndarray:
    """[3,h,w] u8 -> [3,h,w] f32."""
    rgb = _c(rgb_planar, np.uint8)
    n = rgb.shape[1] * rgb.shape[2]
    out = np.empty(rgb.shape, np.float32)
    lib().bxo_rgb2lab_u8(_p(rgb), C.c_int64(n), _p(out))
    return out


def rgb2lab_one(r: int, g: int, b: int) -> Tuple[float, float, float]:
    out = (C.c_double * 3)()
    lib().bxo_rgb2lab_one(C.c_uint8(r), C.c_uint8(g), C.c_uint8(b), out)
    return tuple(out)


def image_mask(Hc: int, Wc: int, img_h: int, img_w: int, rows_removed: int, stride: int) -> np.ndarray:
    out = np.empty((Hc // stride, Wc // stride), np.float32)
    lib().bxo_image_mask(Hc, Wc, int(img_h), int(img_w), int(rows_removed), int(stride), _p(out))
    return out


def color_similarity(lab: np.ndarray, mask: np.ndarray, size: int, dilation: int) -> np.ndarray:
    lab = _c(lab, np.float32)
    mask = _c(mask, np.float32)
    _, h, w = lab.shape
    out = np.empty((size * size - 1, h, w), np.float32)
    lib().bxo_color_similarity(_p(lab), _p(mask), h, w, int(size), int(dilation), _p(out))
    return out


def box_bitmask(box: Sequence[float], Hc: int, Wc: int, stride: int) -> np.ndarray:
    b = (C.c_float * 4)(*[float(v) for v in box])
    out = np.empty((Hc // stride, Wc // stride), np.float32)
    lib().bxo_box_bitmask(b, Hc, Wc, int(stride), _p(out))
    return out


# ---- loss side ----------------------------------------------------------------------------
def pairwise_nlog_fwd(logits: np.ndarray, size: int, dil: int) -> np.ndarray:
    """logits [N,H,W] (f32|f64) -> [N,K,H,W]."""
    suf, _ = _real(logits.dtype)
    logits = np.ascontiguousarray(logits)
    N, H, W = logits.shape
    out = np.empty((N, size * size - 1, H, W), logits.dtype)
    getattr(lib(), f'bxo_pairwise_nlog_fwd_{suf}')(_p(logits), N, H, W, int(size), int(dil), _p(out))
    return out


def pairwise_nlog_bwd(logits: np.ndarray, pairwise: np.ndarray, g_pairwise: np.ndarray, size: int,
                      dil: int) -> np.ndarray:
    suf, _ = _real(logits.dtype)
    logits = np.ascontiguousarray(logits)
    pairwise = _c(pairwise, logits.dtype)
    g_pairwise = _c(g_pairwise, logits.dtype)
    N, H, W = logits.shape
    out = np.empty_like(logits)
    getattr(lib(), f'bxo_pairwise_nlog_bwd_{suf}')(_p(logits), _p(pairwise), _p(g_pairwise), N, H, W,
                                                  int(size), int(dil), _p(out))
    return out


def project_term(logits: np.ndarray, bitmask: np.ndarray, g_out: float = 1.0, want_grad: bool = True):
    """-> (loss_prj, d loss/d logits or None).  logits, bitmask [N,H,W]."""
    suf, ct = _real(logits.dtype)
    logits = np.ascontiguousarray(logits)
    bitmask = _c(bitmask, logits.dtype)
    N, H, W = logits.shape
    g = np.zeros_like(logits) if want_grad else None
    v = getattr(lib(), f'bxo_project_term_{suf}')(_p(logits), _p(bitmask), N, H, W, ct(g_out), _p(g))
    return float(v), g


def boxinst_loss(logits: np.ndarray, sim: np.ndarray, bitmask: np.ndarray, size: int = 3, dil: int = 2,
                 color_thresh: float = 0.3, warmup: float = 1.0, g_prj: float = 1.0, g_pw: float = 1.0,
                 want_grad: bool = True):
    """logits [N,H,W], sim [N,K,H,W], bitmask [N,H,W] -> ((loss_prj, loss_pairwise), grad|None)."""
    suf, ct = _real(logits.dtype)
    logits = np.ascontiguousarray(logits)
    sim = _c(sim, logits.dtype)
    bitmask = _c(bitmask, logits.dtype)
    N, H, W = logits.shape
    losses = np.zeros(2, logits.dtype)
    g = np.zeros_like(logits) if want_grad else None
    getattr(lib(), f'bxo_boxinst_loss_{suf}')(_p(logits), _p(sim), _p(bitmask), N, H, W, int(size), int(dil),
                                              ct(color_thresh), ct(warmup), ct(g_prj), ct(g_pw),
                                              _p(losses), _p(g))
    return (float(losses[0]), float(losses[1])), g


def boxinst_path(imgs: np.ndarray, img_hw: np.ndarray, rows_removed: np.ndarray, mean, std, to_rgb: bool,
                 boxes: np.ndarray, gt_count: np.ndarray, gt_inds: np.ndarray, logits: np.ndarray,
                 stride: int = 4, size: int = 3, dil: int = 2, color_thresh: float = 0.3,
                 warmup: float = 1.0, g_prj: float = 1.0, g_pw: float = 1.0, want_grad: bool = True,
                 want_targets: bool = False):
    """Whole path (f32).  imgs [B,3,Hc,Wc]; logits [N,h,w].
    -> dict(loss_prj, loss_pairwise, grad, sim [B,K,h,w], bitmask [G,h,w])."""
    imgs = _c(imgs, np.float32)
    B, _, Hc, Wc = imgs.shape
    h, w = Hc // stride, Wc // stride
    K = size * size - 1
    img_hw = _c(img_hw, np.int32).reshape(B, 2)
    rows_removed = _c(rows_removed, np.int32).reshape(B)
    gt_count = _c(gt_count, np.int32).reshape(B)
    G = int(gt_count.sum())
    boxes = _c(boxes, np.float32).reshape(G, 4)
    gt_inds = _c(gt_inds, np.int64).reshape(-1)
    logits = _c(logits, np.float32)
    N = logits.shape[0]
    assert logits.shape == (N, h, w) and gt_inds.shape[0] == N
    m = (C.c_double * 3)(*[float(v) for v in mean])
    s = (C.c_double * 3)(*[float(v) for v in std])
    losses = np.zeros(2, np.float32)
    g = np.zeros_like(logits) if want_grad else None
    sim = np.empty((B, K, h, w), np.float32) if want_targets else None
    bm = np.empty((max(G, 1), h, w), np.float32) if want_targets else None
    lib().bxo_boxinst_path_f32(_p(imgs), B, Hc, Wc, _p(img_hw), _p(rows_removed), m, s, int(bool(to_rgb)),
                               _p(boxes), _p(gt_count), _p(gt_inds), _p(logits), N, int(stride), int(size),
                               int(dil), C.c_float(color_thresh), C.c_float(warmup), C.c_float(g_prj),
                               C.c_float(g_pw), _p(losses), _p(g), _p(sim), _p(bm))
    return dict(loss_prj=float(losses[0]), loss_pairwise=float(losses[1]), grad=g, sim=sim,
                bitmask=None if bm is None else bm[:G])


def boxinst_path_f64(imgs, img_hw, rows_removed, mean, std, to_rgb, boxes, gt_count, gt_inds, logits, stride=4, size=3, dil=2,
                     color_thresh=0.3, warmup=1.0, g_prj=1.0, g_pw=1.0):
    """The fp64 oracle of the whole path (SURVEY 8(d) parity gate): the targets are what the reference computes them in
    (uint8 / f32: de-normalise, pool, Lab, similarity -- condinst_head.py:1345-1448 -- those types are part of its
    semantics), the loss and its gradient in float64 on the f32 logits (condinst_head.py:1297-1337 with
    compute_pairwise_term substituted for the CUDA op, which the f64 kernels of pairwise.cu equal bit for bit).
    -> dict(loss_prj, loss_pairwise, grad [N,h,w] f64, sim, bitmask)."""
    t = boxinst_path(imgs, img_hw, rows_removed, mean, std, to_rgb, boxes, gt_count, gt_inds, logits, stride=stride, size=size,
                     dil=dil, color_thresh=color_thresh, want_grad=False, want_targets=True)
    gt_inds = np.asarray(gt_inds, np.int64).reshape(-1)
    gt_count = np.asarray(gt_count, np.int64).reshape(-1)
    img_of_gt = np.repeat(np.arange(len(gt_count)), gt_count)
    sim = t['sim'][img_of_gt[gt_inds]].astype(np.float64)              # [N,K,h,w], :1316-1317
    bm = t['bitmask'][gt_inds].astype(np.float64)
    (lp, lw), g = boxinst_loss(np.asarray(logits, np.float64), sim, bm, size=size, dil=dil, color_thresh=color_thresh,
                               warmup=warmup, g_prj=g_prj, g_pw=g_pw)
    return dict(loss_prj=lp, loss_pairwise=lw, grad=g, sim=t['sim'], bitmask=t['bitmask'])
