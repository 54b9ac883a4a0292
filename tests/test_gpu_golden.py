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
    head._iter.fill_(float(it))
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


def test_lab_kernels_match_real_scikit_image(dev):
    """Both Lab producers on the GPU -- pool_rgb (get_targets API) and the pool blocks of prep_kernel (the evaluation) -- against
    skimage.color.rgb2lab itself (scikit-image 0.18.3; tests/golden/lab_skimage.npz, made by make_lab_golden.py in the build
    container): 65 536 colours as a 1024x1024 image of constant 4x4 blocks, so that every pooled pixel is one fixture colour."""
    import os
    from boxinstseg_amd import boxinst_mask_loss, color_affinity, functional as Fh
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'lab_skimage.npz'))
    rgb, want = g['rgb'][:65536], g['lab'][:65536]
    small = rgb.reshape(256, 256, 3).transpose(2, 0, 1).astype(np.float32)                  # [3,256,256]
    img = torch.from_numpy(np.repeat(np.repeat(small, 4, axis=1), 4, axis=2)[None]).to(dev)  # [1,3,1024,1024], integer valued
    cfg = dict(mean=np.zeros(3, np.float32), std=np.ones(3, np.float32), to_rgb=True)
    metas = [dict(img_shape=(1024, 1024, 3), ori_shape=(1024, 1024, 3), img_norm_cfg=cfg)]
    want_planes = want.reshape(256, 256, 3).transpose(2, 0, 1)

    def check(lab, what):
        bad = lab != want_planes
        assert bad.sum() <= 2, f'{what}: {int(bad.sum())} of {lab.size} Lab values differ from scikit-image'
        if bad.any():
            assert np.abs(lab.view(np.int32).astype(np.int64) - want_planes.view(np.int32).astype(np.int64)).max() <= 1

    _, _, lab = color_affinity(img, metas, want_similarity=False, want_bits=False, bottom_pixels_removed=0)
    check(lab.cpu().numpy()[0], 'pool_rgb kernel')
    Fh.DEBUG_KEEP_LAST = True
    x = torch.zeros((1, 1, 256, 256), device=dev, requires_grad=True)
    boxinst_mask_loss(x, torch.zeros(1, dtype=torch.long, device=dev), [torch.tensor([[100., 100., 400., 300.]], device=dev)],
                      imgs=img, img_metas=metas, bottom_pixels_removed=0)
    torch.cuda.synchronize()
    ws = Fh._LAST['plan'].ws
    from boxinstseg_amd import _lib
    off = _lib.load().bxi_boxinst_eval_workspace_lab_offset()
    lab4 = ws[off:off + 256 * 256 * 16].view(torch.float32).view(256, 256, 4).cpu().numpy()        # (L, a, b, tag) per pooled pixel
    check(np.ascontiguousarray(lab4[:, :, :3].transpose(2, 0, 1)), 'prep_kernel pool blocks')


def test_recorded_hip_run_for_the_gloo_test_is_what_the_product_gives(dev):
    """tests/golden/hip_run_2ranks.npz (what the 2-rank gloo test on the CPU feeds dist.parse_losses) is a recording of THIS product:
    re-evaluate the two batches and compare."""
    import os
    from boxinstseg_amd import CondInstMaskHead, synthetic
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'hip_run_2ranks.npz'))
    for rank in range(2):
        d = synthetic.make_batch(B=1, H=64, W=64, boxes_per_img=2, seed=100 + rank, min_box=16, max_box=40)
        head = CondInstMaskHead(in_channels=16, boxinst_enabled=True, topk_per_img=64, max_proposals=-1, pairwise_warmup=10000).to(dev)
        head.set_iter(2499)
        x = torch.from_numpy(d['mask_logits']).to(dev).requires_grad_(True)
        losses = head.loss(torch.from_numpy(d['imgs']).to(dev), d['img_metas'], x, torch.from_numpy(d['gt_inds']).to(dev),
                           [torch.from_numpy(b).to(dev) for b in d['gt_bboxes']], None, None)
        (losses['loss_prj'] + losses['loss_pairwise']).backward()
        assert abs(float(losses['loss_prj'].detach()) - float(g[f'rank{rank}_loss_prj'])) <= 1e-6
        assert abs(float(losses['loss_pairwise'].detach()) - float(g[f'rank{rank}_loss_pairwise'])) <= 1e-6
        assert abs(float(x.grad.double().abs().sum()) - float(g[f'rank{rank}_grad_abs_sum'])) <= 1e-6 * float(g[f'rank{rank}_grad_abs_sum'])
        assert float(head._iter) == float(g[f'rank{rank}_iter_after'])
