"""GPU: the HIP path against the golden fixtures (outputs of the reference's own functions)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return np.load(os.path.join(G, name))


@pytest.mark.parametrize('case', ['a', 'b', 'c'])
def test_pairwise_op_f64_vs_reference(dev, case):
    from boxinstseg_amd import pairwise_nlog
    g = load('pairwise_f64.npz')
    x = torch.from_numpy(g[f'{case}_logits'][:, None]).to(dev).requires_grad_(True)
    y = pairwise_nlog(x, int(g[f'{case}_size']), int(g[f'{case}_dil']))
    y.backward(torch.from_numpy(g[f'{case}_gp']).to(dev))
    assert np.abs(y.detach().cpu().numpy() - g[f'{case}_pairwise']).max() < 1e-12
    assert np.abs(x.grad.cpu().numpy()[:, 0] - g[f'{case}_grad']).max() < 1e-11
    x32 = torch.from_numpy(g[f'{case}_logits'][:, None].astype(np.float32)).to(dev)
    y32 = pairwise_nlog(x32, int(g[f'{case}_size']), int(g[f'{case}_dil'])).cpu().numpy()
    assert np.abs(y32 - g[f'{case}_pairwise']).max() < 2e-6 * max(1.0, np.abs(g[f'{case}_pairwise']).max())


def test_pairwise_op_extreme_vs_reference(dev):
    from boxinstseg_amd import pairwise_nlog
    g = load('pairwise_f64.npz')
    y = pairwise_nlog(torch.from_numpy(g['ext_logits'][:, None]).to(dev), 3, 1).cpu().numpy()
    assert np.isfinite(y).all() and np.abs(y - g['ext_pairwise']).max() < 1e-11 * np.abs(g['ext_pairwise']).max()


@pytest.mark.parametrize('name', ['loss_cfg1.npz', 'loss_ragged.npz'])
def test_loss_vs_reference(dev, name):
    """CondInstMaskHead.loss (reference source, CPU) vs boxinstseg_amd.CondInstMaskHead.loss (HIP)."""
    from boxinstseg_amd import CondInstMaskHead
    g = load(name)
    head = CondInstMaskHead(in_channels=16, boxinst_enabled=True, max_proposals=-1, topk_per_img=64).to(dev)
    it = round(float(g['warmup']) * 10000) - 1
    head._iter.fill_(float(it)); head._iter_host = float(it)
    counts = [int(c) for c in g['gt_count']]
    boxes = [torch.from_numpy(b).to(dev) for b in np.split(g['boxes'], np.cumsum(counts)[:-1])]
    cfg = dict(mean=np.array([123.675, 116.28, 103.53], np.float32), std=np.array([58.395, 57.12, 57.375], np.float32),
               to_rgb=True)
    metas = [dict(img_shape=(int(s[0]), int(s[1]), 3), ori_shape=(int(o[0]), int(o[1]), 3), img_norm_cfg=cfg)
             for s, o in zip(g['img_shapes'], g['ori_shapes'])]
    x = torch.from_numpy(g['mask_logits']).to(dev).requires_grad_(True)
    imgs = torch.from_numpy(g['imgs']).to(dev)
    out = head.loss(imgs, metas, x, torch.from_numpy(g['gt_inds']).to(dev), boxes, None, None)
    (out['loss_prj'] + out['loss_pairwise']).backward()
    assert abs(out['loss_prj'].item() - float(g['loss_prj'])) <= 1e-4 * abs(float(g['loss_prj']))
    assert abs(out['loss_pairwise'].item() - float(g['loss_pairwise'])) <= 1e-4 * abs(float(g['loss_pairwise']))
    assert np.abs(x.grad.cpu().numpy()[:, 0] - g['grad']).max() <= 1e-4 * np.abs(g['grad']).max()
    sims, bms, _ = head.get_targets(boxes, None, imgs, metas)
    assert np.abs(np.stack([s[0].cpu().numpy() for s in sims]) - g['sim']).max() <= 2e-6
    assert np.array_equal(torch.cat(bms).cpu().numpy(), g['bitmask'])
