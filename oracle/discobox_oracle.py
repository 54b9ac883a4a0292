"""CPU restatement of the DiscoBox pieces of SURVEY 8(f-3): ``MeanField`` (Gaussian-bilateral 3x3 kernel +
mean-field iterations), ``dice_loss`` and ``mil_loss``.

TEST INFRASTRUCTURE ONLY (imported by tests/ and tools/ only).  numpy, fp32 step by step in the order the reference's
torch ops evaluate, written as explicit stencil loops rather than unfold/sum so that it is an independent statement
of the algorithm.  Pinned: tests/test_oracle_golden.py checks it against fixtures produced by executing the
reference's own classes/functions (tests/golden/make_golden.py, oracle/reference_extract.py::load_discobox).

Reference: mmdet/models/dense_heads/discobox_head.py
  dice_loss :542-550     mil_loss :552-562     MeanField.__init__ :591-613  .forward :617-638  .simple_forward :640-655
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


def meanfield_kernel(feat: np.ndarray, ksize: int = 3, alpha0: float = 3.0, theta0: float = 0.5,
                     theta1: float = 30.0) -> np.ndarray:
    """feat [3,H,W] (any channel count) -> K [ksize^2, H, W]   (:597-611)

    K[k,p] = alpha0 * exp( sum_c -(F[c,q]-F[c,p])^2 / (2 theta0^2) + -(|delta_k|^2 / (2 theta1^2)) ),
    F = feat + 10 inside the map and 0 outside (nn.Unfold pads the shifted map with zeros)."""
    C, H, W = feat.shape
    F_ = (feat.astype(f32) + f32(10.0)).astype(f32)
    half = ksize // 2
    pad = np.zeros((C, H + 2 * half, W + 2 * half), f32)
    pad[:, half:half + H, half:half + W] = F_
    K = np.zeros((ksize * ksize, H, W), f32)
    d0 = f32(2 * theta0 ** 2)
    d1 = f32(2 * theta1 ** 2)
    for k in range(ksize * ksize):
        dy, dx = k // ksize - half, k % ksize - half
        q = pad[:, half + dy:half + dy + H, half + dx:half + dx + W]
        acc = np.zeros((H, W), f32)
        for c in range(C):                         # .sum(1): channel order
            d = (q[c] - F_[c]).astype(f32)
            acc = (acc + (-(d * d).astype(f32))).astype(f32)
        spatial = f32(dy * dy + dx * dx)
        e = (acc / d0).astype(f32) + (-(spatial / d1)).astype(f32)
        K[k] = (f32(alpha0) * np.exp(e.astype(f32)).astype(f32)).astype(f32)
    return K


def state_levels(base: float):
    """the two values a thresholded probability takes (:622, :653) and -log of (1-x, x) for both."""
    lo = f32(f32(0.0) * f32(1 - base * 2) + f32(base))
    hi = f32(f32(1.0) * f32(1 - base * 2) + f32(base))
    nl = {}
    for name, x in (('lo', lo), ('hi', hi)):
        nl[name] = (f32(-np.log(f32(f32(1.0) - x))), f32(-np.log(x)))      # (channel 0, channel 1)
    return lo, hi, nl


def meanfield_step(K: np.ndarray, state: np.ndarray, targets: np.ndarray, base: float, inter=None, gamma: float = 0.01):
    """One simple_forward (:640-655) on a binary state.  K [9,H,W]; state [n,H,W] bool (True: x = hi);
    targets [n,H,W] (0/1) -> (new state [n,H,W] bool, margin [n,H,W] = |r1 - 0.5|)."""
    n, H, W = state.shape
    ks = int(round(np.sqrt(K.shape[0])))
    half = ks // 2
    _, _, nl = state_levels(base)
    out = np.zeros_like(state)
    margin = np.zeros(state.shape, f32)
    for i in range(n):
        u = [np.where(state[i], nl['hi'][c], nl['lo'][c]).astype(f32) for c in range(2)]
        f = []
        for c in range(2):
            pad = np.zeros((H + 2 * half, W + 2 * half), f32)
            pad[half:half + H, half:half + W] = u[c]
            acc = np.zeros((H, W), f32)
            for k in range(ks * ks):           # .sum(2): neighbour order
                dy, dx = k // ks - half, k % ks - half
                acc = (acc + (pad[half + dy:half + dy + H, half + dx:half + dx + W] * K[k]).astype(f32)).astype(f32)
            fc = np.exp((-acc).astype(f32)).astype(f32)
            if inter is not None:
                fc = (fc + (inter[i, c].astype(f32) * f32(gamma)).astype(f32)).astype(f32)
            f.append(fc)
        f[1] = (f[1] * targets[i].astype(f32)).astype(f32)
        f = [(x + f32(1e-6)).astype(f32) for x in f]
        s = (f[0] + f[1]).astype(f32)
        r1 = (f[1] / s).astype(f32)
        out[i] = r1 > f32(0.5)
        margin[i] = np.abs(r1 - f32(0.5))
    return out, margin


def meanfield_forward(K: np.ndarray, x: np.ndarray, targets: np.ndarray, iters: int, base: float, inter=None,
                      gamma: float = 0.01, return_states: bool = False):
    """MeanField.forward (:617-638): x, targets [n,H,W] -> (ret [n,H,W] f32 in {0,1}, valid [n] f32)."""
    t = targets.astype(f32)
    state = (x.astype(f32) * t).astype(f32) > f32(0.5)
    states = [state]
    margins = []
    for _ in range(iters):
        state, m = meanfield_step(K, state, t, base, inter, gamma)
        states.append(state); margins.append(m)
    ret = state.astype(f32)
    count = ret.reshape(ret.shape[0], -1).sum(1).astype(f32)
    hw = ret.shape[1] * ret.shape[2]
    valid = ((count >= f32(hw * 0.05)) & (count <= f32(hw * 0.95))).astype(f32)
    if return_states:
        return ret, valid, states, margins
    return ret, valid


def dice_loss(inp: np.ndarray, target: np.ndarray) -> np.ndarray:
    """:542-550   [n, ...] -> [n]    1 - 2 a / (b + c), b and c carry +0.001"""
    i = inp.reshape(inp.shape[0], -1).astype(np.float64)
    t = target.reshape(target.shape[0], -1).astype(np.float64)
    a = (i * t).sum(1)
    b = (i * i).sum(1) + 0.001
    c = (t * t).sum(1) + 0.001
    return 1.0 - (2 * a) / (b + c)


def dice_loss_grad(inp: np.ndarray, target: np.ndarray) -> np.ndarray:
    """d dice_loss[n] / d inp[n, ...]"""
    shp = inp.shape
    i = inp.reshape(shp[0], -1).astype(np.float64)
    t = target.reshape(shp[0], -1).astype(np.float64)
    a = (i * t).sum(1, keepdims=True)
    bc = (i * i).sum(1, keepdims=True) + (t * t).sum(1, keepdims=True) + 0.002
    return (-(2 * t) / bc + 4 * a * i / (bc * bc)).reshape(shp)


def mil_loss(inp: np.ndarray, target: np.ndarray):
    """:552-562  inp, target [n,H,W] -> (loss [n], grad [n,H,W]); max(1) is over rows (one value per column),
    max(2) over columns; the gradient of a max goes to its first arg-max (torch CPU)."""
    n, H, W = inp.shape
    i = inp.astype(np.float64); t = target.astype(np.float64)
    col_in, col_t = i.max(1), t.max(1)          # [n,W]   (the reference calls these "row_*")
    row_in, row_t = i.max(2), t.max(2)          # [n,H]
    loss = dice_loss(row_in, row_t) + dice_loss(col_in, col_t)
    g_col = dice_loss_grad(col_in, col_t)       # [n,W]
    g_row = dice_loss_grad(row_in, row_t)       # [n,H]
    grad = np.zeros_like(i)
    arg_c = i.argmax(1)                         # first maximum
    arg_r = i.argmax(2)
    for k in range(n):
        grad[k, arg_c[k], np.arange(W)] += g_col[k]
        grad[k, np.arange(H), arg_r[k]] += g_row[k]
    return loss, grad
