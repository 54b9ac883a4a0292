"""GPU parity of the DiscoBox pseudo-label path: HIP MeanField / dice_loss / mil_loss against the
fixtures produced by the reference's own code (tests/golden/discobox.npz) and against the numpy oracle on larger,
seeded inputs.  Everything goes through the C ABI (boxinstseg_amd.discobox is marshalling only)."""
import os

import numpy as np
import pytest
import torch

from oracle import discobox_oracle as do

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def _cfg(g, case):
    ks, iters, base, alpha0, theta0, theta1, gamma = [float(v) for v in g[f'{case}_cfg']]
    return int(ks), int(iters), base, alpha0, theta0, theta1, gamma


@pytest.mark.parametrize('case', ['a', 'b', 'c'])
def test_meanfield_reference_fixture(built, dev, case):
    from boxinstseg_amd import MeanField
    g = np.load(os.path.join(GOLD, 'discobox.npz'))
    ks, iters, base, alpha0, theta0, theta1, gamma = _cfg(g, case)
    feat = torch.from_numpy(g[f'{case}_feat'])[None].to(dev)
    mf = MeanField(feat, alpha0=alpha0, theta0=theta0, theta1=theta1, theta2=20.0, iter=iters, kernel_size=ks, base=base)
    K = mf.kernel[0, 0].reshape(g[f'{case}_kernel'].shape).cpu().numpy()
    Kref = g[f'{case}_kernel']
    assert (np.abs(K - Kref) <= 4e-7 * np.abs(Kref) + 1e-37).all()                  # exp: a couple of ulp
    x = torch.from_numpy(g[f'{case}_x'])[:, None].to(dev)
    t = torch.from_numpy(g[f'{case}_t'])[:, None].to(dev)
    inter = torch.from_numpy(g[f'{case}_inter']).to(dev) if f'{case}_inter' in g else None
    # with the reference's own kernel values the decisions must be the reference's
    mf._kernel = torch.from_numpy(Kref)[None].to(dev)
    ret, valid = mf(x, t, inter)
    assert ret.shape == x.shape and ret.dtype == torch.float32
    bad = int((ret[:, 0].cpu().numpy() != g[f'{case}_ret']).sum())
    assert bad == 0, f'{bad} of {ret.numel()} labels differ'
    assert np.array_equal(valid.cpu().numpy(), g[f'{case}_valid'])
    # and with the kernel built on the GPU
    mf2 = MeanField(feat, alpha0=alpha0, theta0=theta0, theta1=theta1, iter=iters, kernel_size=ks, base=base)
    ret2, _ = mf2(x.float(), t.float(), inter)
    assert int((ret2[:, 0].cpu().numpy() != g[f'{case}_ret']).sum()) <= 2e-4 * ret2.numel()


@pytest.mark.parametrize('H,W,n,ks,iters,base', [(100, 136, 12, 3, 10, 0.10), (200, 304, 6, 3, 10, 0.10), (64, 64, 3, 5, 3, 0.45),
                                                 (37, 65, 4, 3, 0, 0.10), (50, 129, 5, 3, 1, 0.30)])
def test_meanfield_vs_oracle(built, dev, H, W, n, ks, iters, base):
    """Seeded image-like features, several instances over two images (img_inds), step-by-step agreement."""
    from boxinstseg_amd import meanfield_forward, meanfield_kernel
    rng = np.random.default_rng(H * 1000 + W)
    yy, xx = np.mgrid[0:H, 0:W]
    feats = []
    for b in range(2):
        f = np.stack([np.sin(xx / (7.0 + b)) + 0.3 * np.cos(yy / 5.0), np.cos(xx / 9.0 + yy / 11.0), 0.5 * np.sin(yy / (4.0 + b))])
        feats.append((f + 0.05 * rng.standard_normal(f.shape)).astype(np.float32))
    feats = np.stack(feats)
    K = meanfield_kernel(torch.from_numpy(feats).to(dev), ks, 2.0, 0.5, 30.0)
    Ko = np.stack([do.meanfield_kernel(feats[b], ks, 2.0, 0.5, 30.0) for b in range(2)])
    assert (np.abs(K.cpu().numpy() - Ko) <= 4e-7 * np.abs(Ko) + 1e-37).all()
    x = rng.uniform(0, 1, size=(n, H, W)).astype(np.float32)
    t = np.zeros((n, H, W), np.uint8)
    for i in range(n):
        r0, c0 = int(rng.integers(0, H // 2)), int(rng.integers(0, W // 2))
        t[i, r0:r0 + int(rng.integers(4, H // 2 + 1)), c0:c0 + int(rng.integers(4, W // 2 + 1))] = 1
    t[0, :, :] = 1                                   # a target covering the whole map (touches every border)
    img = rng.integers(0, 2, size=n)
    Kd = torch.from_numpy(Ko).to(dev)                # same kernel values on both sides: decisions must then agree
    ret, valid = meanfield_forward(Kd, torch.from_numpy(x).to(dev), torch.from_numpy(t).to(dev), iters, base,
                                   img_inds=torch.from_numpy(img).to(dev))
    want = np.zeros_like(x); wv = np.zeros(n, np.float32)
    for b in range(2):
        m = img == b
        if m.any():
            want[m], wv[m] = do.meanfield_forward(Ko[b], x[m], t[m], iters, base)
    bad = int((ret.cpu().numpy() != want).sum())
    assert bad <= 1e-4 * want.size, f'{bad} of {want.size} labels differ'
    if bad == 0:
        assert np.array_equal(valid.cpu().numpy(), wv)
    # float targets give the same answer as uint8 targets
    ret_f, _ = meanfield_forward(Kd, torch.from_numpy(x).to(dev), torch.from_numpy(t).float().to(dev), iters, base,
                                 img_inds=torch.from_numpy(img).to(dev))
    assert torch.equal(ret_f, ret)


@pytest.mark.parametrize('case', ['a', 'b', 'c'])
def test_mil_and_dice_reference_fixture(built, dev, case):
    from boxinstseg_amd import dice_loss, mil_loss
    g = np.load(os.path.join(GOLD, 'discobox.npz'))
    t = torch.from_numpy(g[f'{case}_t']).to(dev)
    gl = torch.from_numpy(g[f'{case}_gl']).to(dev)
    inp = torch.from_numpy(g[f'{case}_mil_in']).to(dev).requires_grad_(True)
    l = mil_loss(dice_loss, inp, inp, t)
    (l * gl).sum().backward()
    assert np.abs(l.detach().cpu().numpy() - g[f'{case}_mil_loss']).max() < 2e-6
    assert np.abs(inp.grad.cpu().numpy() - g[f'{case}_mil_grad']).max() < 2e-6
    inp2 = torch.from_numpy(g[f'{case}_dice_in']).to(dev).requires_grad_(True)
    d = dice_loss(inp2 * t, torch.from_numpy(g[f'{case}_ret']).to(dev))
    (d * gl).sum().backward()
    assert np.abs(d.detach().cpu().numpy() - g[f'{case}_dice_loss']).max() < 2e-6
    assert np.abs(inp2.grad.cpu().numpy() - g[f'{case}_dice_grad']).max() < 2e-6


def test_mil_and_dice_full_size_vs_oracle(built, dev):
    from boxinstseg_amd import dice_loss, mil_loss
    rng = np.random.default_rng(5)
    n, H, W = 40, 200, 304
    x = rng.uniform(0, 1, size=(n, H, W)).astype(np.float32)
    t = np.zeros((n, H, W), np.uint8)
    for i in range(n):
        r0, c0 = int(rng.integers(0, H - 8)), int(rng.integers(0, W - 8))
        t[i, r0:r0 + int(rng.integers(4, H - r0)), c0:c0 + int(rng.integers(4, W - c0))] = 1
    xd = torch.from_numpy(x).to(dev).requires_grad_(True)
    l = mil_loss(dice_loss, xd, xd, torch.from_numpy(t).to(dev))
    l.sum().backward()
    lo, go = do.mil_loss(x, t)
    assert np.abs(l.detach().cpu().numpy() - lo).max() < 2e-6
    assert np.abs(xd.grad.cpu().numpy() - go).max() < 2e-6
    xd2 = torch.from_numpy(x).to(dev).requires_grad_(True)
    d = dice_loss(xd2, torch.from_numpy(t).float().to(dev))
    d.sum().backward()
    assert np.abs(d.detach().cpu().numpy() - do.dice_loss(x, t)).max() < 2e-6
    assert np.abs(xd2.grad.cpu().numpy() - do.dice_loss_grad(x, t)).max() < 2e-6


def test_discobox_errors_and_empty(built, dev):
    from boxinstseg_amd import MeanField, dice_loss, mil_loss
    with pytest.raises(RuntimeError):
        MeanField(torch.zeros(1, 3, 8, 8))                       # CPU tensor: no fallback
    with pytest.raises(RuntimeError):
        mil_loss(lambda a, b: a, torch.zeros(1, 4, 4, device=dev), None, torch.zeros(1, 4, 4, device=dev))
    mf = MeanField(torch.zeros(1, 3, 8, 70, device=dev), iter=3, base=0.1)
    ret, valid = mf(torch.zeros(0, 1, 8, 70, device=dev), torch.zeros(0, 1, 8, 70, device=dev))
    assert ret.shape == (0, 1, 8, 70) and valid.shape == (0,)
    assert dice_loss(torch.zeros(0, 5, device=dev), torch.zeros(0, 5, device=dev)).shape == (0,)


@pytest.mark.parametrize('seed', list(range(10)))
def test_meanfield_fuzz(built, dev, seed):
    """Seeded sweep over map sizes (widths around the 64-pixel word boundary), kernel 3 / 5, iteration counts, bases, one or two
    images, empty / full / thin targets, with and without inter_img_mask."""
    from boxinstseg_amd import meanfield_forward
    rng = np.random.default_rng(4000 + seed)
    H = int(rng.integers(2, 60)); W = int(rng.choice([1, 5, 63, 64, 65, 127, 128, 129, int(rng.integers(2, 200))]))
    ks = int(rng.choice([3, 3, 5])); iters = int(rng.integers(0, 12)); base = float(rng.choice([0.05, 0.1, 0.3, 0.45]))
    n = int(rng.integers(1, 9)); B = int(rng.integers(1, 3))
    yy, xx = np.mgrid[0:H, 0:W]
    feats = np.stack([np.stack([np.sin(xx / (5.0 + b)), np.cos(yy / 6.0 + xx / 9.0), 0.3 * np.sin(yy / 3.0)]) for b in range(B)])
    feats = (feats + 0.1 * rng.standard_normal(feats.shape)).astype(np.float32)
    Ko = np.stack([do.meanfield_kernel(feats[b], ks, 2.0, 0.5, 30.0) for b in range(B)])
    x = rng.uniform(0, 1, size=(n, H, W)).astype(np.float32)
    t = np.zeros((n, H, W), np.uint8)
    for i in range(n):
        kind = int(rng.integers(0, 5))
        if kind == 0:
            continue                                   # no target
        if kind == 1:
            t[i] = 1                                   # everything
        elif kind == 2:
            t[i, int(rng.integers(0, H)), :] = 1       # one row
        else:
            r0, c0 = int(rng.integers(0, H)), int(rng.integers(0, W))
            t[i, r0:r0 + int(rng.integers(1, H + 1)), c0:c0 + int(rng.integers(1, W + 1))] = 1
    img = rng.integers(0, B, size=n)
    inter = rng.uniform(0, 20, size=(n, 2, H, W)).astype(np.float32) if rng.integers(0, 2) else None
    ret, valid = meanfield_forward(torch.from_numpy(Ko).to(dev), torch.from_numpy(x).to(dev), torch.from_numpy(t).to(dev), iters, base,
                                   img_inds=torch.from_numpy(img).to(dev),
                                   inter_img_mask=None if inter is None else torch.from_numpy(inter).to(dev), gamma=0.01)
    want = np.zeros_like(x); wv = np.zeros(n, np.float32)
    for b in range(B):
        m = img == b
        if m.any():
            want[m], wv[m] = do.meanfield_forward(Ko[b], x[m], t[m], iters, base, None if inter is None else inter[m], 0.01)
    bad = int((ret.cpu().numpy() != want).sum())
    assert bad <= max(1, int(2e-4 * want.size)), f'{bad} of {want.size} labels differ ({H}x{W} ks{ks} it{iters} base{base})'
    if bad == 0:
        assert np.array_equal(valid.cpu().numpy(), wv)
