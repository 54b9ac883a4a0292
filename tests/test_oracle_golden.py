"""CPU: both oracles against the golden fixtures generated from the reference's own functions
(tests/golden/make_golden.py).  This is what pins the oracle (parity gate 1)."""
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle, torch_oracle as to

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return np.load(os.path.join(G, name))


@pytest.fixture(scope='module', autouse=True)
def _built(built):
    return built


@pytest.mark.parametrize('case', ['a', 'b', 'c'])
def test_pairwise_fwd_bwd_f64(case):
    g = load('pairwise_f64.npz')
    x, size, dil = g[f'{case}_logits'], int(g[f'{case}_size']), int(g[f'{case}_dil'])
    pw = c_oracle.pairwise_nlog_fwd(x, size, dil)
    assert np.abs(pw - g[f'{case}_pairwise']).max() < 1e-13
    grad = c_oracle.pairwise_nlog_bwd(x, pw, g[f'{case}_gp'], size, dil)
    assert np.abs(grad - g[f'{case}_grad']).max() < 1e-12
    xt = torch.from_numpy(x[:, None]).requires_grad_(True)
    y = to.pairwise_term(xt, size, dil)
    y.backward(torch.from_numpy(g[f'{case}_gp']))
    assert np.abs(y.detach().numpy() - g[f'{case}_pairwise']).max() < 1e-13
    assert np.abs(xt.grad.numpy()[:, 0] - g[f'{case}_grad']).max() < 1e-12
    # f32 flavour of the C oracle against the f64 truth
    pw32 = c_oracle.pairwise_nlog_fwd(x.astype(np.float32), size, dil)
    assert np.abs(pw32 - g[f'{case}_pairwise']).max() < 2e-6 * max(1.0, np.abs(g[f'{case}_pairwise']).max())


@pytest.mark.parametrize('case', ['f32_3_2', 'f64_3_2', 'f32_5_1', 'f64_3_1'])
def test_pairwise_oracle_is_bit_equal_to_reference_kernels(case):
    """The C oracle against the outputs of the reference's OWN pairwise.cu kernels (compiled from the reference tree against
    oracle/ref_wrap/cuda_on_cpu.h and run on the CPU; fixture pairwise_refk.npz): forward AND backward bit for bit, in f32
    and f64, saturated logits included (the backward's atomicAdds in CUDA-thread order = the oracle's scatter order).
    Where oracle/_ref is built, a fresh run of the reference kernels reproduces the fixture."""
    g = load('pairwise_refk.npz')
    x, size, dil = g[f'{case}_logits'], int(g[f'{case}_size']), int(g[f'{case}_dil'])
    pw = c_oracle.pairwise_nlog_fwd(x[:, 0], size, dil)
    assert pw.dtype == x.dtype and np.array_equal(pw, g[f'{case}_pairwise'])
    grad = c_oracle.pairwise_nlog_bwd(x[:, 0], g[f'{case}_pairwise'], g[f'{case}_gp'], size, dil)
    assert np.array_equal(grad, g[f'{case}_grad'][:, 0])
    from oracle import pairwise_ref as pr
    if pr.available():
        assert np.array_equal(pr.forward(x, size, dil), g[f'{case}_pairwise'])
        assert np.array_equal(pr.backward(x, g[f'{case}_pairwise'], g[f'{case}_gp'], size, dil), g[f'{case}_grad'])


def test_pairwise_extreme_logits_f64():
    g = load('pairwise_f64.npz')
    pw = c_oracle.pairwise_nlog_fwd(g['ext_logits'], 3, 1)
    assert np.isfinite(pw).all()
    assert np.abs(pw - g['ext_pairwise']).max() < 1e-12 * max(1.0, np.abs(g['ext_pairwise']).max())


def test_pairwise_known_answers():
    """pair = ln 2 at x=y=0; -> 0 for equal saturated logits; |x| - ln 2 for opposite ones; 0 at borders."""
    x = np.zeros((1, 5, 5), np.float64)
    pw = c_oracle.pairwise_nlog_fwd(x, 3, 1)
    assert abs(pw[0, 4, 2, 2] - np.log(2.0)) < 1e-15
    assert pw[0, 0, 0, 0] == 0.0 and pw[0, 7, 4, 4] == 0.0           # out-of-bounds neighbours
    x[:] = 30.0
    assert c_oracle.pairwise_nlog_fwd(x, 3, 1)[0, 4, 2, 2] < 1e-12
    x[0, 2, 2] = -30.0
    assert abs(c_oracle.pairwise_nlog_fwd(x, 3, 1)[0, 4, 2, 1] - (30.0 - np.log(2.0))) < 1e-9


def test_project_term_f64():
    g = load('project_f64.npz')
    loss, grad = c_oracle.project_term(g['logits'], g['bitmask'])
    assert abs(loss - float(g['loss'])) < 1e-13
    assert np.abs(grad - g['grad']).max() < 1e-13
    x = torch.from_numpy(g['logits'][:, None]).requires_grad_(True)
    out = to.project_term(x.sigmoid(), torch.from_numpy(g['bitmask'][:, None]))
    out.backward()
    assert abs(out.item() - float(g['loss'])) < 1e-14
    # dice known answers: zero prediction vs non-empty box -> 1 per axis
    z = np.full((1, 6, 6), -60.0)
    t = np.zeros((1, 6, 6)); t[0, 1:4, 2:5] = 1
    assert abs(c_oracle.project_term(z, t, want_grad=False)[0] - 2.0) < 1e-9


def test_color_similarity():
    g = load('similarity.npz')
    assert np.abs(c_oracle.color_similarity(g['lab'], g['mask'], 3, 2) - g['sim_3_2']).max() <= 6e-8
    assert np.abs(c_oracle.color_similarity(g['lab'], g['mask'], 5, 1) - g['sim_5_1']).max() <= 6e-8
    s = to.color_similarity(torch.from_numpy(g['lab'])[None], torch.from_numpy(g['mask']), 3, 2)[0].numpy()
    assert np.array_equal(s, g['sim_3_2'])


def test_lab_known_answers():
    g = load('lab_kat.npz')
    for rgb, want in zip(g['rgb'], g['lab']):
        got = np.array(c_oracle.rgb2lab_one(*[int(v) for v in rgb]))
        assert np.abs(got - want).max() < 6e-4, (rgb, got, want)
    # C and numpy restatements agree to the last bit after the f32 cast, on every grey and a colour sweep
    rng = np.random.default_rng(0)
    rgb = np.concatenate([np.repeat(np.arange(256, dtype=np.uint8)[:, None], 3, 1),
                          rng.integers(0, 256, size=(4096, 3)).astype(np.uint8)])
    a = c_oracle.rgb2lab_u8(np.ascontiguousarray(rgb.T).reshape(3, -1, 1))[:, :, 0].T
    b = to.rgb2lab(rgb).astype(np.float32)
    assert np.abs(a - b).max() <= 4e-6


@pytest.mark.parametrize('name', ['loss_cfg1.npz', 'loss_ragged.npz'])
def test_full_path(name):
    """CondInstMaskHead.loss of the reference (run on its own source) vs the C oracle."""
    g = load(name)
    hw = g['img_shapes']
    rr = np.array([to.rows_removed(10, int(s[0]), int(o[0])) for s, o in zip(g['img_shapes'], g['ori_shapes'])])
    out = c_oracle.boxinst_path(g['imgs'], hw, rr, (123.675, 116.28, 103.53), (58.395, 57.12, 57.375), True,
                                g['boxes'], g['gt_count'], g['gt_inds'], g['mask_logits'][:, 0],
                                warmup=float(g['warmup']), want_targets=True)
    assert abs(out['loss_prj'] - float(g['loss_prj'])) <= 2e-6 * abs(float(g['loss_prj']))
    assert abs(out['loss_pairwise'] - float(g['loss_pairwise'])) <= 2e-6 * abs(float(g['loss_pairwise']))
    assert np.abs(out['grad'] - g['grad']).max() <= 2e-6 * np.abs(g['grad']).max()
    assert np.abs(out['sim'] - g['sim']).max() <= 6e-8
    assert np.array_equal(out['bitmask'], g['bitmask'])


def test_box_bitmask_python_slices():
    """condinst_head.py:1429-1430 with negative / inverted / out-of-canvas coordinates."""
    H, W = 40, 56
    for box in ([10.7, 3.2, 30.9, 20.1], [-3.0, -2.0, 20.0, 10.0], [30.0, 2.0, 10.0, 30.0], [50.0, 35.0, 90.0, 80.0],
                [5.0, 5.0, 5.9, 5.9], [-60.0, -50.0, -1.0, -1.0]):
        full = np.zeros((H, W), np.float32)
        full[int(box[1]):int(box[3]) + 1, int(box[0]):int(box[2]) + 1] = 1.0
        assert np.array_equal(c_oracle.box_bitmask(box, H, W, 4), full[2::4, 2::4]), box


def test_image_mask_and_pool():
    m = c_oracle.image_mask(32, 48, 30, 41, 7, 4)
    full = np.ones((30, 41), np.float32); full[-7:, :] = 0
    pad = np.zeros((32, 48), np.float32); pad[:30, :41] = full
    assert np.array_equal(m, pad[2::4, 2::4])
    rng = np.random.default_rng(1)
    u = rng.integers(0, 256, size=(3, 8, 12)).astype(np.uint8)
    want = torch.nn.functional.avg_pool2d(torch.from_numpy(u).float()[None], 4, 4)[0].byte().numpy()
    assert np.array_equal(c_oracle.pool_u8(u, 4), want)


def test_denormalize_on_exact_pixels():
    """Real images de-normalise to within an ulp of an integer, so the uint8 truncation sees every
    rounding step (double multiply -> f32, then f32 add of (float)mean).  The C and the numpy
    restatements must agree bit for bit there, and a few values are worked by hand."""
    from boxinstseg_amd import synthetic
    d = synthetic.make_batch(B=1, H=64, W=96, boxes_per_img=1, seed=11, min_box=16, max_box=32, pixel_offset=0.0)
    mean, std = np.asarray(synthetic.MEAN, np.float64), np.asarray(synthetic.STD, np.float64)
    for to_rgb in (True, False):
        a = c_oracle.denormalize_u8(d['imgs'][0], 64, 96, mean, std, to_rgb)
        b = to.denormalize_u8(torch.from_numpy(d['imgs'][0]), (64, 96), mean, std, to_rgb).numpy()
        assert np.array_equal(a.astype(np.float32), b)
    # hand-worked: x = f32((f32(u) - f32(mean)) / std); t = f32(double(x) * std); v = t + f32(mean)
    for u, c in [(0, 0), (255, 0), (17, 1), (200, 2), (124, 0), (116, 1)]:
        mf = np.float32(mean[c])
        x = np.float32(np.float64(np.float32(u) - mf) * (1.0 / std[c]))
        t = np.float32(np.float64(x) * std[c])
        v = np.float32(t + mf)
        img = np.zeros((3, 1, 1), np.float32); img[c, 0, 0] = x
        got = c_oracle.denormalize_u8(img, 1, 1, mean, std, True)[c, 0, 0]
        assert int(got) == int(v) and abs(int(got) - u) <= 1


@pytest.mark.parametrize('case', ['a', 'b', 'c'])
def test_dynamic_head_restatement_vs_reference(case):
    """oracle.torch_oracle.dynamic_mask_forward vs CondInstMaskHead.forward run from the reference source."""
    g = load('dynamic_head_f64.npz')
    C, no_rel, fac = [int(v) for v in g[f'{case}_cfg']]
    feat = torch.from_numpy(g[f'{case}_feat']).requires_grad_(True)
    params = torch.from_numpy(g[f'{case}_params']).requires_grad_(True)
    y = to.dynamic_mask_forward(feat, params, torch.from_numpy(g[f'{case}_coors']), torch.from_numpy(g[f'{case}_level']),
                                torch.from_numpy(g[f'{case}_img']), torch.tensor([64, 128, 256, 512, 1024]),
                                in_stride=8, out_stride=8 // fac, disable_rel_coors=bool(no_rel))
    y.backward(torch.from_numpy(g[f'{case}_g']))
    assert np.abs(y.detach().numpy() - g[f'{case}_logits']).max() < 1e-12
    assert np.abs(feat.grad.numpy() - g[f'{case}_gfeat']).max() < 1e-12
    assert np.abs(params.grad.numpy() - g[f'{case}_gparams']).max() < 1e-11


# ---- DiscoBox MeanField / dice_loss / mil_loss ---------------------------------------------------------
@pytest.mark.parametrize('case', ['a', 'b', 'c'])
def test_discobox_restatement_vs_reference(case):
    """oracle.discobox_oracle vs the reference's MeanField / mil_loss / dice_loss (fixtures made by running them)."""
    from oracle import discobox_oracle as do
    g = load('discobox.npz')
    ks, iters, base, alpha0, theta0, theta1, gamma = [float(v) for v in g[f'{case}_cfg']]
    ks, iters = int(ks), int(iters)
    K = do.meanfield_kernel(g[f'{case}_feat'], ks, alpha0, theta0, theta1)
    Kref = g[f'{case}_kernel']
    assert np.abs(K - Kref).max() <= 2.4e-7 * alpha0 and (np.abs(K - Kref) <= 2e-7 * np.abs(Kref) + 1e-37).all()   # 1 ulp (exp)
    inter = g[f'{case}_inter'] if f'{case}_inter' in g else None
    ret, valid, states, margins = do.meanfield_forward(Kref, g[f'{case}_x'], g[f'{case}_t'], iters, base, inter, gamma,
                                                       return_states=True)
    # decisions are thresholds of fp32 expressions evaluated in the reference's op order: they must agree exactly
    assert np.array_equal(ret, g[f'{case}_ret'])
    assert np.array_equal(valid, g[f'{case}_valid'])
    t = g[f'{case}_t']
    l, gr = do.mil_loss(g[f'{case}_mil_in'], t)
    assert np.abs(l - g[f'{case}_mil_loss']).max() < 2e-6
    assert np.abs(gr * g[f'{case}_gl'][:, None, None] - g[f'{case}_mil_grad']).max() < 2e-6
    x = g[f'{case}_dice_in'] * t
    assert np.abs(do.dice_loss(x, g[f'{case}_ret']) - g[f'{case}_dice_loss']).max() < 2e-6
    gd = do.dice_loss_grad(x, g[f'{case}_ret']) * t * g[f'{case}_gl'][:, None, None]
    assert np.abs(gd - g[f'{case}_dice_grad']).max() < 2e-6


# ---- BoxProjectionLoss / LevelsetLoss / LCM ---------------------------------------------------------------
@pytest.mark.parametrize('case', ['a', 'b', 'c'])
def test_levelset_restatement_vs_reference(case):
    """oracle.levelset_oracle (values and hand-derived gradients) vs the reference's classes run under autograd."""
    from oracle import levelset_oracle as lo
    g = load('levelset.npz')
    gl = g[f'{case}_gl']
    l, gr = lo.box_projection_loss(g[f'{case}_scores'][:, 0], g[f'{case}_bitmask'][:, 0], 1.3)
    assert np.abs(l - g[f'{case}_prj_loss']).max() < 1e-12
    assert np.abs(gr * gl[:, None, None] - g[f'{case}_prj_grad'][:, 0]).max() < 1e-12
    l, gm, gT = lo.levelset_loss(g[f'{case}_ms'], g[f'{case}_T'], g[f'{case}_pn'], 0.7)
    assert np.abs(l - g[f'{case}_lst_loss']).max() < 1e-12
    assert np.abs(gm * gl[:, None, None, None] - g[f'{case}_lst_gms']).max() < 1e-12
    assert np.abs(gT * gl[:, None, None, None] - g[f'{case}_lst_gT']).max() < 1e-12
    aff = lo.lcm_affinity(g[f'{case}_img'])
    ref = lo.lcm_refine(aff, g[f'{case}_phi'][:, 0])
    assert np.abs(ref - g[f'{case}_refined'][:, 0]).max() < 5e-6          # the reference runs this one in fp32
    l, gr = lo.lcm_loss(g[f'{case}_img'], g[f'{case}_phi'][:, 0], g[f'{case}_box'][:, 0])
    assert abs(l - float(g[f'{case}_lcm_loss'])) < 2e-6
    assert np.abs(gr - g[f'{case}_lcm_grad'][:, 0]).max() < 2e-6 * max(1.0, np.abs(gr).max() * 1e3)


# ---- tree_filter ----------------------------------------------------------------------------------------------
def _edge_set(e):
    return set((int(min(a, b)), int(max(a, b))) for a, b in np.asarray(e).reshape(-1, 2).tolist())


@pytest.mark.parametrize('case', ['a', 'b', 'c', 'd'])
def test_tree_filter_mst_vs_reference_boruvka(case):
    """oracle mst (Kruskal under (weight, index)) == the edge set the reference's own boruvka.cpp produced (fixture), and,
    where oracle/_ref is built, == a fresh run of it."""
    from oracle import tree_filter_oracle as tfo
    g = load('tree_filter.npz')
    fm = g[f'{case}_fm']
    H, W = fm.shape[1:]
    idx, wt = tfo.grid_edges(H, W), tfo.grid_weights(fm)
    mine = tfo.mst_edges(idx, wt, H * W)
    assert len(mine) == H * W - 1
    assert _edge_set(idx[mine]) == _edge_set(g[f'{case}_tree'])
    if tfo.ref_available():
        assert _edge_set(tfo.ref_boruvka_mst(idx, wt, H * W)) == _edge_set(g[f'{case}_tree'])


def _per_vertex(gw_sorted, si):
    out = np.zeros_like(gw_sorted)
    out[si] = gw_sorted
    return out


@pytest.mark.parametrize('case', ['a', 'b', 'd'])
def test_tree_filter_refine_restatement_vs_reference_kernels(case):
    """oracle restatement of bfs.cu / refine.cu == what the reference's own kernels produced (run on the CPU through
    oracle/ref_wrap/cuda_on_cpu.h; fixture keys refk_*), on the reference's BFS order AND on the oracle's own order
    (results are order-independent in vertex order); where oracle/_ref is built, a fresh run reproduces the fixture."""
    from oracle import tree_filter_oracle as tfo
    g = load('tree_filter.npz')
    tree = g[f'{case}_tree']
    V = tree.shape[0] + 1
    si, sp, sc = g[f'refk_{case}_si'], g[f'refk_{case}_sp'], g[f'refk_{case}_sc']
    # the reference's order is a valid parent-before-child order of the same tree
    assert sorted(si.tolist()) == list(range(V)) and si[0] == 0 and (sp[1:] < np.arange(1, V)).all()
    assert _edge_set(np.stack([si[1:], si[sp[1:]]], 1)) == _edge_set(tree)
    for i in range(V):
        for c in sc[i]:
            assert c == 0 or sp[c] == i
    x, w, gr = (g[f'refk_{case}_{k}'].astype(np.float64) for k in ('x', 'w', 'g'))
    for order in ('reference', 'oracle'):
        o_si, o_sp, o_sc = (si, sp, sc) if order == 'reference' else tfo.bfs_order(tree, V)
        w_o = w if order == 'reference' else _per_vertex(w, si)[o_si]      # the same weight on the same tree edge
        out, saved = tfo.refine_forward(x, w_o, o_si, o_sp, o_sc)
        assert np.abs(out - g[f'refk_{case}_out']).max() < 2e-6 * max(1.0, np.abs(out).max())
        aggr = np.empty_like(saved['D']); aggr[:, o_si] = saved['D']
        assert np.abs(aggr - g[f'refk_{case}_aggr']).max() < 2e-6 * np.abs(aggr).max()
        wsum = np.empty(V); wsum[o_si] = saved['WD']
        assert np.abs(wsum - g[f'refk_{case}_wsum']).max() < 2e-6 * wsum.max()
        if order == 'reference':
            assert np.abs(saved['U'] - g[f'refk_{case}_aggr_up']).max() < 2e-6 * np.abs(saved['U']).max()
            assert np.abs(saved['WU'] - g[f'refk_{case}_wsum_up']).max() < 2e-6 * saved['WU'].max()
        gf = tfo.refine_backward_feature(gr, w_o, o_si, o_sp, o_sc, saved)
        assert np.abs(gf - g[f'refk_{case}_gf']).max() < 2e-6 * max(1.0, np.abs(gf).max())
        gw = _per_vertex(tfo.refine_backward_weight(x, gr, w_o, o_si, o_sp, o_sc, saved), o_si)
        want = _per_vertex(g[f'refk_{case}_gw'].astype(np.float64), si)
        assert np.abs(gw - want).max() < 5e-6 * max(1.0, np.abs(want).max())
    if tfo.ref_kernels_available():
        fwd = tfo.ref_refine_forward(g[f'refk_{case}_x'], g[f'refk_{case}_w'], si, sp, sc)
        assert np.array_equal(fwd['out'], g[f'refk_{case}_out'])
        gf2, gw2 = tfo.ref_refine_backward(g[f'refk_{case}_g'], g[f'refk_{case}_w'], si, sp, sc, fwd)
        assert np.array_equal(gf2, g[f'refk_{case}_gf']) and np.array_equal(gw2, g[f'refk_{case}_gw'])
        r_si, r_sp, r_sc = tfo.ref_bfs(tree, V)                            # arrival order may differ; validity must not
        assert sorted(r_si.tolist()) == list(range(V)) and (r_sp[1:] < np.arange(1, V)).all()


def test_tree_filter_refine_restatement_closed_form_and_autograd():
    """the recurrences of refine.cu restated (oracle) == the closed form they implement, and their gradients == autograd"""
    from oracle import tree_filter_oracle as tfo
    rng = np.random.default_rng(3)
    H, W = 6, 8
    V = H * W
    fm = rng.standard_normal((3, H, W)).astype(np.float32)
    idx = tfo.grid_edges(H, W)
    si, sp, sc = tfo.bfs_order(idx[tfo.mst_edges(idx, tfo.grid_weights(fm), V)], V)
    assert (sp[1:] < np.arange(1, V)).all() and sorted(si.tolist()) == list(range(V))
    w = tfo.edge_weights(rng.standard_normal((3, V)) * 0.3, si, sp, False)
    x = rng.standard_normal((2, V)); g = rng.standard_normal((2, V))
    out, saved = tfo.refine_forward(x, w, si, sp, sc)
    assert np.abs(out - tfo.refine_closed_form(x, w, si, sp)).max() < 1e-12
    xt = torch.tensor(x, requires_grad=True); wt_ = torch.tensor(w, requires_grad=True)
    S = [[None] * V for _ in range(V)]
    one = torch.ones((), dtype=torch.float64)
    for i in range(V):
        S[i][i] = one
        if i:
            for j in range(i):
                S[i][j] = S[int(sp[i])][j] * wt_[i]
                S[j][i] = S[i][j]
    Sf = torch.stack([torch.stack(r) for r in S])
    o = (Sf[None] * xt[:, torch.from_numpy(si).long()][:, None, :]).sum(2) / Sf.sum(1)[None]
    (o * torch.tensor(g[:, si])).sum().backward()
    assert np.abs(tfo.refine_backward_feature(g, w, si, sp, sc, saved) - xt.grad.numpy()).max() < 1e-12
    assert np.abs(tfo.refine_backward_weight(x, g, w, si, sp, sc, saved)[1:] - wt_.grad.numpy()[1:]).max() < 1e-12
