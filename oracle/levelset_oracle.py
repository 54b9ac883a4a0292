"""CPU restatement of the Box2Mask / BoxLevelSet loss pieces of SURVEY 8(f-4): ``BoxProjectionLoss``,
``region_levelset`` / ``LevelsetLoss`` and the local consistency module ``LCM``.

TEST INFRASTRUCTURE ONLY (imported by tests/ and tools/ only).  numpy fp64, explicit loops over neighbours /
channels with hand-derived gradients, i.e. an independent statement of the algorithm.  Pinned:
tests/test_oracle_golden.py checks values AND gradients against fixtures produced by executing the reference's own
classes under autograd (tests/golden/make_golden.py, oracle/reference_extract.py::load_levelset).

Reference: mmdet/models/losses/box_projection_loss.py:5-43, mmdet/models/losses/levelset_loss.py:7-126
"""
from __future__ import annotations

import numpy as np


# ---- BoxProjectionLoss (box_projection_loss.py:18-43) ------------------------------------------------------------------
def _dice(x, t, eps=1e-5):
    inter = (x * t).sum(1)
    union = (x * x).sum(1) + (t * t).sum(1) + eps
    return 1.0 - 2.0 * inter / union, inter, union


def box_projection_loss(scores: np.ndarray, bitmask: np.ndarray, loss_weight: float = 1.0):
    """scores, bitmask [N,H,W] -> (loss [N], d loss / d scores [N,H,W]); max(dim=2) of the reference's [N,1,H,W] is over
    rows (one value per column), max(dim=3) over columns; the gradient of a max goes to its first arg-max."""
    s = scores.astype(np.float64); t = bitmask.astype(np.float64)
    n, H, W = s.shape
    grad = np.zeros_like(s)
    loss = np.zeros(n)
    for axis in (1, 2):
        m, tm = s.max(axis), t.max(axis)
        l, inter, union = _dice(m, tm)
        loss += l
        g = -2.0 * tm / union[:, None] + 4.0 * inter[:, None] * m / (union[:, None] ** 2)
        arg = s.argmax(axis)
        for k in range(n):
            if axis == 1:
                grad[k, arg[k], np.arange(W)] += g[k]
            else:
                grad[k, np.arange(H), arg[k]] += g[k]
    return loss_weight * loss, loss_weight * grad


# ---- region_levelset / LevelsetLoss (levelset_loss.py:13-45) --------------------------------------------------------------
def levelset_loss(mask_score: np.ndarray, target: np.ndarray, pixel_num: np.ndarray, loss_weight: float = 1.0):
    """mask_score [N,2,H,W] (foreground, background), target [N,C,H,W], pixel_num [N]
    -> (loss [N], d/d mask_score [N,2,H,W], d/d target [N,C,H,W])."""
    m = mask_score.astype(np.float64); T = target.astype(np.float64)
    N, C = T.shape[:2]
    loss = np.zeros(N); gm = np.zeros_like(m); gT = np.zeros_like(T)
    for n in range(N):
        w = loss_weight / (C * float(pixel_num[n]))
        for side in (0, 1):
            f = m[n, side]
            S = f.sum(); Sc = max(S, 1e-5)
            for c in range(C):
                A = (f * T[n, c]).sum()
                a = A / Sc                                         # interior_ / exterior_ (:34-35)
                d = T[n, c] - a
                loss[n] += w * (d * d * f).sum()
                R = (d * f).sum()                                  # = A - a S : 0 unless the clamp is active
                da_df = (T[n, c] - (a if S >= 1e-5 else 0.0)) / Sc  # d a / d f_p
                gm[n, side] += w * (d * d - 2.0 * R * da_df)
                gT[n, c] += w * (2.0 * d * f - 2.0 * R * f / Sc)
    return loss, gm, gT


# ---- LocalConsistencyModule / LCM (levelset_loss.py:63-126) -------------------------------------------------------------
_OFFS = [(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1)]     # get_kernel (:82-92)


def _neighbours(x: np.ndarray, d: int) -> np.ndarray:
    """x [...,h,w] -> [8,...,h,w]: the 8 dilated neighbours with replicate padding (:96-104)."""
    h, w = x.shape[-2:]
    rr = np.arange(h)[:, None]; cc = np.arange(w)[None, :]
    return np.stack([x[..., np.clip(rr + dy * d, 0, h - 1), np.clip(cc + dx * d, 0, w - 1)] for dy, dx in _OFFS])


def lcm_affinity(imgs: np.ndarray, d: int = 2, alpha: float = 0.3) -> np.ndarray:
    """imgs [N,C,h,w] -> aff [N,8,h,w] (:108-120): softmax over the neighbours of the channel mean of
    -(|I_q - I_p| / (std_k I_q + 1e-8) / alpha)^2, std unbiased over the 8 neighbours."""
    I = imgs.astype(np.float64)
    nb = _neighbours(I, d)                                  # [8,N,C,h,w]
    std = nb.std(axis=0, ddof=1)
    a = -((np.abs(nb - I[None]) / (std[None] + 1e-8) / alpha) ** 2)
    a = a.mean(axis=2)                                      # [8,N,h,w]
    a = a - a.max(axis=0, keepdims=True)
    e = np.exp(a)
    return np.moveaxis(e / e.sum(axis=0, keepdims=True), 0, 1)


def lcm_refine(aff: np.ndarray, phi: np.ndarray, iters: int = 10, d: int = 2) -> np.ndarray:
    """aff [N,8,h,w], phi [N,h,w] -> refined phi (:122-126)"""
    p = phi.astype(np.float64)
    for _ in range(iters):
        p = (np.moveaxis(_neighbours(p, d), 0, 1) * aff).sum(1)
    return p


def lcm_refine_backward(aff: np.ndarray, g: np.ndarray, iters: int = 10, d: int = 2) -> np.ndarray:
    """transpose of lcm_refine: d (sum g * refined) / d phi, by explicit scatter."""
    N, _, h, w = aff.shape
    rr = np.arange(h)[:, None]; cc = np.arange(w)[None, :]
    g = g.astype(np.float64)
    for _ in range(iters):
        out = np.zeros_like(g)
        for k, (dy, dx) in enumerate(_OFFS):
            r2 = np.broadcast_to(np.clip(rr + dy * d, 0, h - 1), (h, w)); c2 = np.broadcast_to(np.clip(cc + dx * d, 0, w - 1), (h, w))
            for n in range(N):
                np.add.at(out[n], (r2, c2), aff[n, k] * g[n])
        g = out
    return g


def lcm_loss(imgs: np.ndarray, phi: np.ndarray, box: np.ndarray, iters: int = 10, d: int = 2):
    """LCM (:53-60): imgs [N,C,h,w], phi, box [N,h,w] -> (scalar loss, d loss / d phi)"""
    aff = lcm_affinity(imgs, d)
    ref = lcm_refine(aff, phi, iters, d)
    b = box.astype(np.float64)
    regions = max(b.sum(), 1.0)
    diff = ref - phi
    loss = (np.abs(diff) * b).sum() / regions
    s = np.sign(diff) * b / regions
    grad = lcm_refine_backward(aff, s, iters, d) - s
    return loss, grad
