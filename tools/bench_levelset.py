#!/usr/bin/env python3
"""Developer measurement of the Box2Mask loss pieces on the GPU box: HIP BoxProjectionLoss /
LevelsetLoss / LCM forward+backward vs the reference's op sequence run by PyTorch-ROCm on the same GPU
(restated here, in tools/, for timing only)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import torch.nn.functional as F
import __graft_entry__ as entry
entry.build()
from boxinstseg_amd import LCM, BoxProjectionLoss, LevelsetLoss

dev = torch.device('cuda:0')


def ev(fn, n=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def t_dice(x, t):
    n = x.size(0); x = x.reshape(n, -1); t = t.reshape(n, -1)
    return 1. - (2 * (x * t).sum(1) / ((x ** 2.0).sum(1) + (t ** 2.0).sum(1) + 1e-5))


def t_proj(s, b):
    return t_dice(s.max(dim=3, keepdim=True)[0], b.max(dim=3, keepdim=True)[0]) + t_dice(s.max(dim=2, keepdim=True)[0], b.max(dim=2, keepdim=True)[0])


def t_levelset(ms, T, pn, w):
    f = ms[:, 0:1]; b = ms[:, 1:2]
    i_ = torch.sum(f * T, (2, 3)) / torch.sum(f, (2, 3)).clamp(min=0.00001)
    e_ = torch.sum(b * T, (2, 3)) / torch.sum(b, (2, 3)).clamp(min=0.00001)
    r = torch.pow(T - i_.unsqueeze(-1).unsqueeze(-1), 2) * f + torch.pow(T - e_.unsqueeze(-1).unsqueeze(-1), 2) * b
    return w * torch.sum(r, (1, 2, 3)) / T.shape[1] / pn


KER = torch.zeros(8, 1, 3, 3, device=dev)
for k, (i, j) in enumerate([(0, 0), (0, 1), (0, 2), (1, 0), (1, 2), (2, 0), (2, 1), (2, 2)]): KER[k, 0, i, j] = 1


def t_nb(x, d=2):
    b, c, h, w = x.shape
    xp = F.pad(x, [d] * 4, mode='replicate').reshape(b * c, -1, h + 2 * d, w + 2 * d)
    return F.conv2d(xp, KER, dilation=d).view(b, c, -1, h, w)


def t_lcm(imgs, phi, box):
    nb = t_nb(imgs); rep = imgs.unsqueeze(2).repeat(1, 1, 8, 1, 1)
    aff = -((nb - rep).abs() / (torch.std(nb, dim=2, keepdim=True) + 1e-8) / 0.3) ** 2
    aff = F.softmax(aff.mean(dim=1, keepdim=True), dim=2)
    p = phi
    for _ in range(10): p = (t_nb(p) * aff).sum(2)
    return ((p - phi).abs() * box).sum() / box.sum().clamp(min=1)


res = {}
g = torch.Generator().manual_seed(0)
for N in (16, 100):
    H, W = 200, 304
    s = torch.rand(N, 1, H, W, generator=g).to(dev).requires_grad_(True)
    box = torch.zeros(N, 1, H, W)
    rng = np.random.default_rng(N)
    for i in range(N):
        hh, ww = int(rng.integers(20, 120)), int(rng.integers(20, 160)); r0, c0 = int(rng.integers(0, H - hh)), int(rng.integers(0, W - ww))
        box[i, 0, r0:r0 + hh, c0:c0 + ww] = 1
    box = box.to(dev)
    ms = (torch.rand(N, 2, H, W, generator=g).to(dev) * box).requires_grad_(True)
    T = torch.rand(N, 3, H, W, generator=g).to(dev)
    pn = box.sum((1, 2, 3)).clamp(min=1)
    img = torch.rand(N, 3, 96, 96, generator=g).to(dev); phi = torch.rand(N, 1, 96, 96, generator=g).to(dev).requires_grad_(True)
    b96 = (torch.rand(N, 1, 96, 96, generator=g) > 0.5).float().to(dev)
    P, L = BoxProjectionLoss(), LevelsetLoss()
    def fb(fn, *leaves):
        def run():
            fn().sum().backward()
            for t in leaves: t.grad = None
        return run
    res[f'N{N}'] = dict(
        hip_projection_fwd_bwd_us=ev(fb(lambda: P(s, box), s)), torch_rocm_projection_fwd_bwd_us=ev(fb(lambda: t_proj(s, box), s)),
        hip_levelset_fwd_bwd_us=ev(fb(lambda: L(ms, T, pn), ms)), torch_rocm_levelset_fwd_bwd_us=ev(fb(lambda: t_levelset(ms, T, pn, 1.0), ms)),
        hip_lcm_fwd_bwd_us=ev(fb(lambda: LCM(img, phi, b96), phi)), torch_rocm_lcm_fwd_bwd_us=ev(fb(lambda: t_lcm(img, phi, b96), phi), n=10, warm=2),
        lcm_value_rel_diff=float((LCM(img, phi, b96) - t_lcm(img, phi, b96)).abs() / t_lcm(img, phi, b96)))
print(json.dumps(res, indent=1))
