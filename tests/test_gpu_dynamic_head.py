"""GPU: the dynamic mask head -- HIP forward/backward vs the golden fixture produced by the
reference's own CondInstMaskHead.forward and vs the torch restatement at larger shapes."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
SOI = [64, 128, 256, 512, 1024]


def _run_hip(dev, feat, params, coors, lvl, img, fac, no_rel, g):
    from boxinstseg_amd import dynamic_mask_forward
    f = torch.from_numpy(feat.astype(np.float32)).to(dev).requires_grad_(True)
    p = torch.from_numpy(params.astype(np.float32)).to(dev).requires_grad_(True)
    y = dynamic_mask_forward(f, p, torch.from_numpy(coors.astype(np.float32)).to(dev), torch.from_numpy(lvl).to(dev),
                             torch.from_numpy(img).to(dev), torch.tensor(SOI, device=dev), in_stride=8,
                             out_stride=8 // fac, disable_rel_coors=bool(no_rel))
    y.backward(torch.from_numpy(g.astype(np.float32)).to(dev))
    return y.detach().cpu().numpy(), f.grad.cpu().numpy(), p.grad.cpu().numpy()


def _close(got, want, tol):
    return np.abs(got - want).max() <= tol * max(np.abs(want).max(), 1e-30)


def _close_except_kinks(got, want, tol, groups, max_groups=2):
    """Like _close, but a ReLU pre-activation that is zero to within fp32 rounding (seen once in ~10 000 random
    instance-pixels: 6.8e-8 in fp64, <= 0 in fp32) gates one hidden unit differently and changes the gradient of that
    one (instance, pixel): mismatches confined to at most `max_groups` groups (`groups` = the group id of every element:
    the pixel for d feat, the instance for d params) are tolerated."""
    bad = np.abs(got - want) > tol * max(np.abs(want).max(), 1e-30)
    return len(np.unique(groups[bad])) <= max_groups


@pytest.mark.parametrize('case', ['a', 'b', 'c'])
def test_dynamic_head_vs_reference_fixture(dev, case):
    g = np.load(os.path.join(G, 'dynamic_head_f64.npz'))
    C, no_rel, fac = [int(v) for v in g[f'{case}_cfg']]
    y, gf, gp = _run_hip(dev, g[f'{case}_feat'], g[f'{case}_params'], g[f'{case}_coors'], g[f'{case}_level'],
                         g[f'{case}_img'], fac, no_rel, g[f'{case}_g'])
    assert y.shape == g[f'{case}_logits'].shape
    assert _close(y, g[f'{case}_logits'], 2e-5)
    assert _close(gf, g[f'{case}_gfeat'], 2e-5)
    assert _close(gp, g[f'{case}_gparams'], 2e-5)


@pytest.mark.parametrize('shape', [(2, 16, 100, 128, 32), (3, 8, 37, 50, 70), (1, 16, 8, 32, 1), (2, 16, 100, 128, 128), (4, 8, 100, 128, 9)])
def test_dynamic_head_vs_oracle_large(dev, shape):
    """BoxInst R-50 shape (2 x 16 x 100 x 128 features, 32 instances -> 200 x 256 logits; 128 instances = topk 64 x 2 images: eight
    slots per (image, tile), eight instances per backward workgroup) and ragged ones (4 images x 52 tiles: four slots fit the launch)."""
    from oracle import torch_oracle as to
    B, C, H, W, N = shape
    rng = np.random.default_rng(sum(shape))
    feat = rng.standard_normal((B, C, H, W)).astype(np.float32)
    params = (rng.standard_normal((N, (C + 2) * 8 + 64 + 8 + 17)) * 0.3).astype(np.float32)
    coors = rng.uniform(0, 8 * W, size=(N, 2)).astype(np.float32)
    lvl = rng.integers(0, 5, size=N)
    img = rng.integers(0, B, size=N)
    g = rng.standard_normal((N, 1, 2 * H, 2 * W)).astype(np.float32)
    f64 = lambda a: torch.from_numpy(a.astype(np.float64))
    ft, pt = f64(feat).requires_grad_(True), f64(params).requires_grad_(True)
    yo = to.dynamic_mask_forward(ft, pt, f64(coors), torch.from_numpy(lvl), torch.from_numpy(img), torch.tensor(SOI))
    yo.backward(f64(g))
    y, gf, gp = _run_hip(dev, feat, params, coors, lvl, img, 2, 0, g)
    assert _close(y, yo.detach().numpy(), 2e-5)
    assert _close(gf, ft.grad.numpy(), 5e-5)
    assert _close(gp, pt.grad.numpy(), 5e-5)


@pytest.mark.parametrize('case', ['a', 'b'])
def test_simple_test_masks_vs_reference(dev, case):
    """CondInstMaskHead.simple_test (HIP dynamic head + torch up-sampling) against the masks the reference's own
    simple_test / forward / aligned_bilinear produced on the CPU (tests/golden/simple_test.npz); a probability within
    float noise of 0.5 may land on the other side, so a handful of pixels per instance is tolerated."""
    import boxinstseg_amd as bx
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'simple_test.npz'))
    feat = torch.from_numpy(g[f'{case}_feat']).to(dev)
    ncls, rescale = int(g[f'{case}_ncls']), bool(int(g[f'{case}_rescale']))
    head = bx.CondInstMaskHead(in_channels=feat.size(1), in_stride=8, out_stride=4).to(dev)
    metas = [dict(img_shape=tuple(int(v) for v in sh[0]) + (3,), ori_shape=tuple(int(v) for v in sh[1]) + (3,)) for sh in g[f'{case}_shapes']]
    t = lambda k: torch.from_numpy(g[f'{case}_{k}']).to(dev)
    res = head.simple_test(feat, [t(f'labels{i}') for i in range(2)], [t(f'params{i}') for i in range(2)], [t(f'coors{i}') for i in range(2)],
                           [t(f'lvl{i}') for i in range(2)], metas, ncls, rescale=rescale)
    assert len(res) == 2 and all(len(r) == ncls for r in res)
    for i in range(2):
        for c in range(ncls):
            want = g[f'{case}_masks{i}_{c}']
            got = np.asarray(res[i][c], np.uint8)
            assert got.shape == want.shape and got.dtype == np.uint8
            if want.size:
                assert int((got != want).sum()) <= 3 * want.shape[0], (i, c)
    empty = head.simple_test(feat, [t('labels0')[:0], t('labels1')[:0]], [t('params0')[:0], t('params1')[:0]], [t('coors0')[:0], t('coors1')[:0]],
                             [t('lvl0')[:0], t('lvl1')[:0]], metas, ncls)
    assert empty == [[[] for _ in range(ncls)] for _ in range(2)]


def test_dynamic_head_module_and_errors(dev):
    from boxinstseg_amd import CondInstMaskHead
    head = CondInstMaskHead(in_channels=16, boxinst_enabled=True, max_proposals=-1, topk_per_img=64).to(dev)
    feat = torch.randn(2, 16, 12, 20, device=dev)
    params = torch.randn(3, head.num_gen_params, device=dev)
    out = head(feat, params, torch.rand(3, 2, device=dev) * 100, torch.tensor([0, 2, 4], device=dev),
               torch.tensor([1, 0, 1], device=dev))
    assert out.shape == (3, 1, 24, 40)
    empty = head(feat, params[:0], torch.zeros(0, 2, device=dev), torch.zeros(0, dtype=torch.long, device=dev),
                 torch.zeros(0, dtype=torch.long, device=dev))
    assert empty.shape == (0, 1, 24, 40)
    with pytest.raises(RuntimeError, match='CUDA'):
        head(feat.cpu(), params.cpu(), torch.zeros(3, 2), torch.zeros(3, dtype=torch.long), torch.zeros(3, dtype=torch.long))
    with pytest.raises(RuntimeError):
        head(feat, params[:, :100], torch.zeros(3, 2, device=dev), torch.zeros(3, dtype=torch.long, device=dev),
             torch.zeros(3, dtype=torch.long, device=dev))
    w, b = head.parse_dynamic_params(params)
    assert [tuple(t.shape) for t in w] == [(24, 18, 1, 1), (24, 8, 1, 1), (3, 8, 1, 1)] and [t.numel() for t in b] == [24, 24, 3]


@pytest.mark.parametrize('seed', list(range(12)))
def test_dynamic_head_fuzz(dev, seed):
    """Seeded sweep: feature maps that are / are not multiples of the 8x32 tile, C in {8,16}, relative coordinates on/off,
    up-sampling factors 1/2/4 and the run-time path (3), 0..40 instances spread unevenly over 1..4 images (including
    images with no instance and more than 8 instances per image: several per slot)."""
    from oracle import torch_oracle as to
    rng = np.random.default_rng(7000 + seed)
    B = int(rng.integers(1, 5)); C = int(rng.choice([8, 16])); no_rel = bool(rng.integers(0, 2))
    H = int(rng.integers(3, 40)); W = int(rng.integers(3, 70)); N = int(rng.integers(0, 41))
    fac = int(rng.choice([1, 2, 2, 4, 3]))
    stride = {1: 8, 2: 8, 4: 8, 3: 9}[fac]
    feat = rng.standard_normal((B, C, H, W)).astype(np.float32)
    P = (C + (0 if no_rel else 2)) * 8 + 64 + 8 + 17
    params = (rng.standard_normal((N, P)) * 0.3).astype(np.float32)
    coors = rng.uniform(0, stride * W, size=(N, 2)).astype(np.float32)
    lvl = rng.integers(0, 5, size=N)
    img = rng.integers(0, max(1, B - int(rng.integers(0, 2))), size=N)       # sometimes the last image gets no instance
    g = rng.standard_normal((N, 1, fac * H, fac * W)).astype(np.float32)
    from boxinstseg_amd import dynamic_mask_forward
    f = torch.from_numpy(feat).to(dev).requires_grad_(True)
    p = torch.from_numpy(params).to(dev).requires_grad_(True)
    y = dynamic_mask_forward(f, p, torch.from_numpy(coors).to(dev), torch.from_numpy(lvl).to(dev), torch.from_numpy(img).to(dev),
                             torch.tensor(SOI, device=dev), in_stride=stride, out_stride=stride // fac, disable_rel_coors=no_rel)
    assert y.shape == (N, 1, fac * H, fac * W)
    if N == 0:
        return
    y.backward(torch.from_numpy(g).to(dev))
    f64 = lambda a: torch.from_numpy(a.astype(np.float64))
    ft, pt = f64(feat).requires_grad_(True), f64(params).requires_grad_(True)
    yo = to.dynamic_mask_forward(ft, pt, f64(coors), torch.from_numpy(lvl), torch.from_numpy(img), torch.tensor(SOI),
                                 in_stride=stride, out_stride=stride // fac, disable_rel_coors=no_rel)
    yo.backward(f64(g))
    cfg = f'B{B} C{C} {H}x{W} N{N} f{fac} rel{not no_rel}'
    assert _close(y.detach().cpu().numpy(), yo.detach().numpy(), 3e-5), cfg
    pix = np.broadcast_to(np.arange(B * H * W).reshape(B, 1, H, W), feat.shape)
    inst = np.broadcast_to(np.arange(N)[:, None], params.shape)
    assert _close_except_kinks(f.grad.cpu().numpy(), ft.grad.numpy(), 1e-4, pix), cfg
    assert _close_except_kinks(p.grad.cpu().numpy(), pt.grad.numpy(), 1e-4, inst), cfg


@pytest.mark.parametrize('convs,ch,cin,no_rel,fac,shape', [
    (1, 8, 8, False, 2, (2, 12, 20, 5)),        # a single layer: [1 x (C + 2)]
    (2, 4, 6, False, 2, (2, 12, 20, 4)),
    (3, 8, 16, False, 2, (2, 23, 37, 9)),       # the shipped shape through the general kernels
    (3, 5, 7, True, 1, (1, 9, 33, 3)),          # channels padded 5 -> 8, no relative coordinates, no up-sampling
    (4, 16, 32, False, 4, (2, 10, 35, 6)),      # the largest built: 4 layers x 16 channels on 32 + 2 inputs
    (4, 12, 3, False, 3, (3, 17, 40, 7)),       # odd factor, channels padded 12 -> 16
    (2, 2, 1, True, 2, (1, 8, 32, 2)),
])
def test_general_dynamic_head_shapes_in_hip(dev, convs, ch, cin, no_rel, fac, shape):
    """Every head shape the reference's constructor admits up to 4 layers x 16 channels (condinst_head.py:1079-1089) runs HIP
    kernels (csrc/dynamic_head_generic.hip), forward and backward: checked against the fp64 CPU evaluation of the composition
    that tests/test_host_cpu.py pins to the reference's grouped-convolution formulation, values and both gradients."""
    from boxinstseg_amd import CondInstMaskHead
    from boxinstseg_amd import dynamic as dyn
    B, H, W, N = shape
    rng = np.random.default_rng(convs * 1000 + ch * 10 + cin)
    head = CondInstMaskHead(in_channels=cin, dynamic_convs=convs, dynamic_channels=ch, boxinst_enabled=True, disable_rel_coors=no_rel,
                            in_stride=8, out_stride=8 // fac if 8 % fac == 0 else 8)
    if 8 % fac:                                             # factor 3: strides 12 / 4
        head.in_stride, head.out_stride = 4 * fac, 4
    assert dyn.generic_supported(convs, ch, cin, no_rel)
    feat = rng.standard_normal((B, cin, H, W))
    params = rng.standard_normal((N, head.num_gen_params)) * 0.4
    coors = rng.uniform(0, head.in_stride * W, size=(N, 2))
    lvl = rng.integers(0, 5, size=N); img = rng.integers(0, B, size=N)
    g = rng.standard_normal((N, 1, fac * H, fac * W))
    f64 = lambda a: torch.from_numpy(np.asarray(a, np.float64))
    ft, pt = f64(feat).requires_grad_(True), f64(params).requires_grad_(True)
    cpu = head.double()
    want = cpu._composed_forward(ft, pt, f64(coors), torch.from_numpy(lvl), torch.from_numpy(img))
    want.backward(f64(g))
    f32 = lambda a: torch.from_numpy(np.asarray(a, np.float32)).to(dev)
    fd, pd = f32(feat).requires_grad_(True), f32(params).requires_grad_(True)
    gpu_head = CondInstMaskHead(in_channels=cin, dynamic_convs=convs, dynamic_channels=ch, boxinst_enabled=True, disable_rel_coors=no_rel).to(dev)
    gpu_head.in_stride, gpu_head.out_stride = head.in_stride, head.out_stride
    if (convs, ch) == (3, 8) and cin in (8, 16):            # the module would take the tuned kernels: call the general entry
        got = dyn.dynamic_mask_forward_generic(fd, pd, f32(coors), torch.from_numpy(lvl).to(dev), torch.from_numpy(img).to(dev),
                                               gpu_head.sizes_of_interest, convs, ch, in_stride=head.in_stride, out_stride=head.out_stride,
                                               disable_rel_coors=no_rel)
    else:
        got = gpu_head(fd, pd, f32(coors), torch.from_numpy(lvl).to(dev), torch.from_numpy(img).to(dev))
    assert got.shape == want.shape
    got.backward(f32(g))
    cfg = f'{convs} x {ch} on {cin} rel{not no_rel} f{fac}'
    assert _close(got.detach().cpu().numpy(), want.detach().numpy(), 3e-5), cfg
    pix = np.broadcast_to(np.arange(B * H * W).reshape(B, 1, H, W), feat.shape)
    inst = np.broadcast_to(np.arange(N)[:, None], params.shape)
    assert _close_except_kinks(fd.grad.cpu().numpy(), ft.grad.numpy(), 1e-4, pix), cfg
    assert _close_except_kinks(pd.grad.cpu().numpy(), pt.grad.numpy(), 1e-4, inst), cfg
    # run-to-run identical (no atomics anywhere)
    fd2, pd2 = f32(feat).requires_grad_(True), f32(params).requires_grad_(True)
    again = dyn.dynamic_mask_forward_generic(fd2, pd2, f32(coors), torch.from_numpy(lvl).to(dev), torch.from_numpy(img).to(dev),
                                             gpu_head.sizes_of_interest, convs, ch, in_stride=head.in_stride, out_stride=head.out_stride,
                                             disable_rel_coors=no_rel)
    again.backward(f32(g))
    assert torch.equal(again, got) and torch.equal(fd2.grad, fd.grad) and torch.equal(pd2.grad, pd.grad)


@pytest.mark.parametrize('seed', range(10))
def test_general_dynamic_head_fuzz(dev, seed):
    """Random head shapes inside the general kernels' limits (1-4 layers, 2-16 channels, 1-32 feature channels, with / without relative
    coordinates, factors 1-4, ragged instance -> image maps, maps that are not multiples of the 8 x 32 tile)."""
    r = np.random.default_rng(7700 + seed)
    convs, ch, cin = int(r.integers(1, 5)), int(r.integers(2, 17)), int(r.integers(1, 33))
    no_rel = bool(r.integers(0, 2))
    fac = int(r.integers(1, 5))
    shape = (int(r.integers(1, 4)), int(r.integers(1, 30)), int(r.integers(1, 70)), int(r.integers(1, 12)))
    test_general_dynamic_head_shapes_in_hip(dev, convs, ch, cin, no_rel, fac, shape)


def test_general_dynamic_head_empty_and_limits(dev):
    from boxinstseg_amd import CondInstMaskHead
    from boxinstseg_amd import dynamic as dyn
    assert not dyn.generic_supported(5, 8, 8, False) and not dyn.generic_supported(3, 17, 8, False) and not dyn.generic_supported(3, 8, 33, False)
    head = CondInstMaskHead(in_channels=6, dynamic_convs=2, dynamic_channels=4, boxinst_enabled=True).to(dev)
    feat = torch.randn(2, 6, 12, 20, device=dev, requires_grad=True)
    params = torch.zeros(0, head.num_gen_params, device=dev, requires_grad=True)
    out = head(feat, params, torch.zeros(0, 2, device=dev), torch.zeros(0, dtype=torch.long, device=dev), torch.zeros(0, dtype=torch.long, device=dev))
    assert out.shape == (0, 1, 24, 40)
    out.sum().backward()
    assert float(feat.grad.abs().max()) == 0.0
    with pytest.raises(RuntimeError, match='params must be'):
        head(feat, torch.zeros(3, head.num_gen_params + 1, device=dev), torch.zeros(3, 2, device=dev), torch.zeros(3, dtype=torch.long, device=dev),
             torch.zeros(3, dtype=torch.long, device=dev))
    big = CondInstMaskHead(in_channels=6, dynamic_convs=2, dynamic_channels=20, boxinst_enabled=True).to(dev)     # beyond the build: composed
    p = torch.randn(2, big.num_gen_params, device=dev)
    y = big(feat.detach(), p, torch.rand(2, 2, device=dev) * 50, torch.tensor([0, 1], device=dev), torch.tensor([0, 1], device=dev))
    assert y.shape == (2, 1, 24, 40) and torch.isfinite(y).all()


def test_dynamic_head_shapes_outside_the_hip_build_run_composed(dev):
    """dynamic_convs / dynamic_channels other than 3 / 8 (free in the reference, condinst_head.py:1079-1089) through the module:
    checked against the fp64 CPU evaluation of the composition (pinned on the CPU, tests/test_host_cpu.py), differentiable, and the
    built shape gives the same logits through the tuned kernels and through the composition of PyTorch-ROCm ops."""
    from boxinstseg_amd import CondInstMaskHead
    torch.manual_seed(3)
    head = CondInstMaskHead(in_channels=6, dynamic_convs=2, dynamic_channels=4, boxinst_enabled=True).to(dev)
    feat = torch.randn(2, 6, 12, 20, device=dev, requires_grad=True)
    params = torch.randn(4, head.num_gen_params, device=dev, requires_grad=True)
    coors = torch.rand(4, 2, device=dev) * 100
    lvl = torch.tensor([0, 2, 4, 1], device=dev); img = torch.tensor([1, 0, 1, 1], device=dev)
    out = head(feat, params, coors, lvl, img)
    assert out.shape == (4, 1, 24, 40)
    out.square().sum().backward()
    assert torch.isfinite(feat.grad).all() and torch.isfinite(params.grad).all()
    cpu = head.cpu().double()
    want = cpu._composed_forward(feat.detach().cpu().double(), params.detach().cpu().double(), coors.cpu().double(), lvl.cpu(), img.cpu())
    assert (out.detach().cpu().double() - want).abs().max() <= 1e-5 * max(1.0, float(want.abs().max()))
    built = CondInstMaskHead(in_channels=8, boxinst_enabled=True).to(dev)
    f8 = torch.randn(2, 8, 12, 20, device=dev); p8 = torch.randn(4, built.num_gen_params, device=dev)
    a = built(f8, p8, coors, lvl, img); b = built._composed_forward(f8, p8, coors, lvl, img)
    assert (a - b).abs().max() <= 2e-5 * max(1.0, float(b.abs().max()))


@pytest.mark.parametrize('C,no_rel,N', [(16, False, 32), (8, False, 5), (8, True, 3)])
def test_head_fused_into_the_loss_evaluation(dev, C, no_rel, N):
    """CondInstMaskHead.forward_loss = forward() + loss() with the dynamic head evaluated inside the evaluation's first launch
    (bxi_boxinst_head_eval_f32): same logits, same losses, same gradients w.r.t. the mask features and the dynamic parameters as
    the two calls (different kernels, same arithmetic: fp32 rounding differences only)."""
    import copy
    from boxinstseg_amd import CondInstMaskHead, synthetic
    d = synthetic.cfg2(3) if N == 32 else synthetic.cfg1(1)
    imgs = torch.from_numpy(d['imgs']).to(dev)
    B, H, W = imgs.shape[0], imgs.shape[2], imgs.shape[3]
    boxes = [torch.from_numpy(b).to(dev) for b in d['gt_bboxes']]
    gt_inds = torch.from_numpy(d['gt_inds']).to(dev)[:N]
    n = gt_inds.numel()
    counts = np.cumsum([0] + [b.shape[0] for b in boxes])
    img_inds = torch.tensor([int(np.searchsorted(counts, int(g), side='right') - 1) for g in gt_inds.cpu()], device=dev)
    torch.manual_seed(C + n)
    head = CondInstMaskHead(in_channels=C, boxinst_enabled=True, disable_rel_coors=no_rel, max_proposals=-1, topk_per_img=64).to(dev)
    head.set_iter(5000)
    feat = torch.randn(B, C, H // 8, W // 8, device=dev)
    params = 0.3 * torch.randn(n, head.num_gen_params, device=dev)
    coors = torch.rand(n, 2, device=dev) * torch.tensor([W, H], device=dev)
    lvl = torch.randint(0, 5, (n,), device=dev)

    def run(fused):
        h2 = copy.deepcopy(head)
        f = feat.clone().requires_grad_(True); p = params.clone().requires_grad_(True)
        if fused:
            logits, losses = h2.forward_loss(f, p, coors, lvl, img_inds, imgs, d['img_metas'], gt_inds, boxes, fuse_head=True)
        else:
            logits = h2(f, p, coors, lvl, img_inds)
            losses = h2.loss(imgs, d['img_metas'], logits, gt_inds, boxes, None, None)
        (losses['loss_prj'] + 2.0 * losses['loss_pairwise']).backward()
        return logits.detach(), losses['loss_prj'].detach(), losses['loss_pairwise'].detach(), f.grad, p.grad

    a, b = run(True), run(False)
    assert a[0].shape == b[0].shape
    assert (a[0] - b[0]).abs().max() <= 2e-6 * max(1.0, float(b[0].abs().max()))
    for i in (1, 2):
        assert abs(float(a[i]) - float(b[i])) <= 1e-5 * max(abs(float(b[i])), 1e-6), (i, float(a[i]), float(b[i]))
    for i in (3, 4):
        assert (a[i] - b[i]).abs().max() <= 2e-4 * max(float(b[i].abs().max()), 1e-8), (i, float((a[i] - b[i]).abs().max()), float(b[i].abs().max()))



def test_head_fused_backward_twice_with_retain_graph(dev):
    """The reference's composed graph (forward() then loss()) can be differentiated twice with retain_graph=True; the head-fused node
    finishes its gradient in place for the first call's upstream factors, so a second call evaluates the loss again from the kept logits:
    each call's gradients equal those of a fresh evaluation with the same weights."""
    from boxinstseg_amd import CondInstMaskHead, synthetic
    d = synthetic.cfg1(2)
    imgs = torch.from_numpy(d['imgs']).to(dev)
    B, H, W = imgs.shape[0], imgs.shape[2], imgs.shape[3]
    boxes = [torch.from_numpy(b).to(dev) for b in d['gt_bboxes']]
    gt_inds = torch.from_numpy(d['gt_inds']).to(dev)
    n = gt_inds.numel()
    counts = np.cumsum([0] + [b.shape[0] for b in boxes])
    img_inds = torch.tensor([int(np.searchsorted(counts, int(g), side='right') - 1) for g in gt_inds.cpu()], device=dev)
    torch.manual_seed(3)
    head = CondInstMaskHead(in_channels=8, boxinst_enabled=True, max_proposals=-1, topk_per_img=64).to(dev)
    head.set_iter(5000)
    feat = torch.randn(B, 8, H // 8, W // 8, device=dev)
    params = 0.3 * torch.randn(n, head.num_gen_params, device=dev)
    coors = torch.rand(n, 2, device=dev) * torch.tensor([W, H], device=dev)
    lvl = torch.randint(0, 5, (n,), device=dev)

    def evaluate():
        f = feat.clone().requires_grad_(True); p = params.clone().requires_grad_(True)
        it = int(head._iter.item())
        _, losses = head.forward_loss(f, p, coors, lvl, img_inds, imgs, d['img_metas'], gt_inds, boxes, fuse_head=True)
        assert float(head._iter) == it + 1              # counted inside the head-fused evaluation's last launch
        head.set_iter(it)                               # the same warm-up factor for every evaluation of this test
        return f, p, losses

    weights = [(1.0, 2.0), (0.5, 3.0), (1.0, 1.0)]
    f, p, losses = evaluate()
    got = []
    for k, (a, b) in enumerate(weights):
        f.grad = p.grad = None
        (a * losses['loss_prj'] + b * losses['loss_pairwise']).backward(retain_graph=k + 1 < len(weights))
        got.append((f.grad.clone(), p.grad.clone()))
    for (a, b), (gf, gp) in zip(weights, got):
        f2, p2, l2 = evaluate()
        (a * l2['loss_prj'] + b * l2['loss_pairwise']).backward()
        assert torch.isfinite(gf).all() and float(gf.abs().max()) > 0
        assert (gf - f2.grad).abs().max() <= 1e-5 * float(f2.grad.abs().max()), (a, b)
        assert (gp - p2.grad).abs().max() <= 1e-5 * float(p2.grad.abs().max()), (a, b)
