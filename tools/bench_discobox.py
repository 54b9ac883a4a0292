#!/usr/bin/env python3
"""Developer measurement of the DiscoBox pseudo-label path on the GPU box: HIP MeanField / mil_loss /
dice_loss vs the reference's op sequence run by PyTorch-ROCm on the same GPU (restated in tools/ only for timing)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import torch.nn.functional as F
import __graft_entry__ as entry
entry.build()
from boxinstseg_amd import MeanField, dice_loss, mil_loss

dev = torch.device('cuda:0')


def ev(fn, n=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


class TorchMeanField:
    """the reference's op sequence (discobox_head.py:591-655) on torch-ROCm tensors"""
    def __init__(self, fm, ks, alpha0, theta0, theta1, iters, base):
        self.unfold = torch.nn.Unfold(ks, stride=1, padding=ks // 2)
        fm = fm + 10
        u = self.unfold(fm).view(fm.size(0), fm.size(1), ks * ks, -1)
        sp = torch.tensor((np.arange(ks * ks) // ks - ks // 2) ** 2 + (np.arange(ks * ks) % ks - ks // 2) ** 2).to(fm.device).float()
        self.kernel = (alpha0 * torch.exp((-(u - fm.view(fm.size(0), fm.size(1), 1, -1)) ** 2).sum(1) / (2 * theta0 ** 2)
                                          + (-(sp.view(1, -1, 1) / (2 * theta1 ** 2))))).unsqueeze(1)
        self.ks, self.iters, self.base = ks, iters, base

    def step(self, x, targets):
        h, w = x.size(2), x.size(3)
        ux = self.unfold(-torch.log(x)).view(x.size(0) // 2, 2, self.ks ** 2, -1)
        f = torch.exp(-(ux * self.kernel).sum(2).view(-1, 1, h, w)).view(-1, 2, h, w)
        f[:, 1:] *= targets
        f = f + 1e-6
        f = f / f.sum(1, keepdim=True)
        return ((f > 0.5).float() * (1 - self.base * 2) + self.base).view(-1, 1, h, w)

    def __call__(self, x, targets):
        with torch.no_grad():
            x = x * targets
            x = (x > 0.5).float() * (1 - self.base * 2) + self.base
            U = torch.cat([1 - x, x], 1).view(-1, 1, x.size(2), x.size(3))
            for _ in range(self.iters):
                U = self.step(U, targets)
            ret = (U.view(-1, 2, U.size(2), U.size(3))[:, 1:] > 0.5).float()
            cnt = ret.reshape(ret.shape[0], -1).sum(1)
            hw = ret.shape[2] * ret.shape[3]
            return ret, ((cnt >= hw * 0.05) * (cnt <= hw * 0.95)).float()


def t_dice(i, t):
    i = i.contiguous().view(i.size(0), -1).float(); t = t.contiguous().view(t.size(0), -1).float()
    return 1 - (2 * torch.sum(i * t, 1)) / (torch.sum(i * i, 1) + 0.001 + torch.sum(t * t, 1) + 0.001)


def t_mil(i, t):
    return t_dice(i.max(2)[0], t.max(2)[0]) + t_dice(i.max(1)[0], t.max(1)[0])


res = {}
H, W = 200, 304
g = torch.Generator().manual_seed(0)
yy, xx = np.mgrid[0:H, 0:W]
feat = torch.from_numpy(np.stack([np.sin(xx / 7.0), np.cos(xx / 9.0 + yy / 11.0), 0.5 * np.sin(yy / 4.0)]).astype(np.float32))[None].to(dev)
for n in (16, 64):
    x = torch.rand(n, 1, H, W, generator=g).to(dev)
    t = torch.zeros(n, 1, H, W)
    rng = np.random.default_rng(n)
    for i in range(n):
        hh, ww = int(rng.integers(20, 120)), int(rng.integers(20, 160))
        r0, c0 = int(rng.integers(0, H - hh)), int(rng.integers(0, W - ww))
        t[i, 0, r0:r0 + hh, c0:c0 + ww] = 1
    t = t.to(dev)
    mf = MeanField(feat, alpha0=2.0, theta0=0.5, theta1=30.0, iter=10, kernel_size=3, base=0.1)
    tm = TorchMeanField(feat, 3, 2.0, 0.5, 30.0, 10, 0.1)
    a, _ = mf(x, t); b, _ = tm(x, t)
    mism = int((a != b).sum())
    xi = torch.rand(n, H, W, generator=g).to(dev).requires_grad_(True)
    tb = t[:, 0].byte()
    def hip_mil():
        l = mil_loss(dice_loss, xi, xi, tb); l.sum().backward(); xi.grad = None
    def ref_mil():
        l = t_mil(xi, tb); l.sum().backward(); xi.grad = None
    res[f'n{n}'] = dict(box_px_mean=float(t.sum() / n), hip_meanfield_us=ev(lambda: mf(x, t)), torch_rocm_meanfield_us=ev(lambda: tm(x, t), n=10, warm=2),
                        label_mismatch_vs_torch_rocm=mism, hip_kernel_build_us=ev(lambda: MeanField(feat, alpha0=2.0, theta0=0.5, theta1=30.0, iter=10, base=0.1)),
                        hip_mil_fwd_bwd_us=ev(hip_mil), torch_rocm_mil_fwd_bwd_us=ev(ref_mil))
print(json.dumps(res, indent=1))
