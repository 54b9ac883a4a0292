"""GPU parity of the tree_filter extension: HIP mst / bfs / refine and the MinimumSpanningTree /
TreeFilter2D modules against (a) the edge sets the reference's own boruvka.cpp produced (tests/golden/tree_filter.npz)
(b) the outputs of the reference's own bfs.cu / refine.cu kernels (fixture refk_* keys, and live at 96x96 where
oracle/_ref is present) and (c) the oracle restatement of bfs / refine.  Through the C ABI."""
import os

import numpy as np
import pytest
import torch

from oracle import tree_filter_oracle as tfo

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def _edge_set(e):
    return set((int(min(a, b)), int(max(a, b))) for a, b in np.asarray(e).reshape(-1, 2).tolist())


@pytest.mark.parametrize('case', ['a', 'b', 'c', 'd'])
def test_mst_matches_reference_boruvka(built, dev, case):
    from boxinstseg_amd import MinimumSpanningTree, TreeFilter2D
    g = np.load(os.path.join(GOLD, 'tree_filter.npz'))
    fm = g[f'{case}_fm']
    H, W = fm.shape[1:]
    tree = MinimumSpanningTree(TreeFilter2D.norm2_distance)(torch.from_numpy(fm)[None].to(dev).repeat(2, 1, 1, 1))
    assert tree.shape == (2, H * W - 1, 2) and tree.dtype == torch.int32
    t = tree.cpu().numpy()
    assert _edge_set(t[0]) == _edge_set(g[f'{case}_tree']), 'not the tree the reference Boruvka selects'
    assert np.array_equal(t[0], t[1])
    # listed in ascending edge order of the grid edge list
    idx = tfo.grid_edges(H, W)
    order = {(int(a), int(b)): i for i, (a, b) in enumerate(idx.tolist())}
    ids = [order[(int(a), int(b))] for a, b in t[0].tolist()]
    assert ids == sorted(ids)


def test_mst_with_labels(built, dev):
    """the label branch of MinimumSpanningTree.forward (tree_filter.py:58-61) against the same weights on the CPU"""
    from boxinstseg_amd import MinimumSpanningTree, TreeFilter2D
    rng = np.random.default_rng(2)
    H, W = 20, 26
    fm = rng.standard_normal((1, 4, H, W)).astype(np.float32)
    lab = (rng.uniform(size=(1, 2, H, W)) > 0.6).astype(np.float32)
    tree = MinimumSpanningTree(TreeFilter2D.norm2_distance)(torch.from_numpy(fm).to(dev), torch.from_numpy(lab).to(dev))
    wt = tfo.grid_weights(fm[0])
    both = np.concatenate([(lab[0, :, :-1, :] + lab[0, :, 1:, :]).sum(0).reshape(-1), (lab[0, :, :, :-1] + lab[0, :, :, 1:]).sum(0).reshape(-1)])
    diff = tfo.grid_weights(lab[0]) - 1
    m = (diff * both) > 0
    wt[m] = (1.0 / (1.0 + np.exp(-wt[m].astype(np.float64)))).astype(np.float32)
    idx = tfo.grid_edges(H, W)
    want = tfo.mst_edges(idx, wt, H * W)
    got = _edge_set(tree.cpu().numpy()[0])
    # sigmoid is evaluated in fp32 on the GPU: compare on the tree weight instead of edge by edge if they differ
    if got != _edge_set(idx[want]):
        wsum = lambda es: sum(float(wt[i]) for i, (a, b) in enumerate(idx.tolist()) if (min(a, b), max(a, b)) in es)
        assert abs(wsum(got) - wsum(_edge_set(idx[want]))) < 1e-3


@pytest.mark.parametrize('H,W', [(96, 96), (12, 17), (3, 40), (2, 2)])
def test_bfs_is_a_valid_deterministic_order(built, dev, H, W):
    from boxinstseg_amd import bfs, mst
    rng = np.random.default_rng(H * 7 + W)
    V = H * W
    idx = tfo.grid_edges(H, W)
    wt = (rng.uniform(size=(3, len(idx))) + 1).astype(np.float32)
    tree = mst(torch.from_numpy(idx)[None].repeat(3, 1, 1).to(dev), torch.from_numpy(wt).to(dev), V)
    si, sp, sc = bfs(tree, 4)
    si2, sp2, sc2 = bfs(tree, 4)
    assert torch.equal(si, si2) and torch.equal(sp, sp2) and torch.equal(sc, sc2)
    lv = si._bxi_levels.cpu().numpy()
    si, sp, sc, t = si.cpu().numpy(), sp.cpu().numpy(), sc.cpu().numpy(), tree.cpu().numpy()
    for b in range(3):
        assert sorted(si[b].tolist()) == list(range(V)) and si[b, 0] == 0
        assert (sp[b, 1:] < np.arange(1, V)).all() and sp[b, 0] == 0
        es = _edge_set(t[b])
        assert all((min(int(si[b, i]), int(si[b, sp[b, i]])), max(int(si[b, i]), int(si[b, sp[b, i]]))) in es for i in range(1, V))
        # children: contiguous, consistent with sorted_parent, breadth first (levels = depth)
        depth = np.zeros(V, np.int64)
        for i in range(1, V):
            depth[i] = depth[sp[b, i]] + 1
        assert (np.diff(depth) >= 0).all()
        D = lv[b, 0]
        assert D == depth.max() + 1 and lv[b, 1] == 0 and lv[b, 1 + D] == V
        assert np.array_equal(np.searchsorted(depth, np.arange(D)), lv[b, 1:1 + D])
        for i in range(V):
            ch = [c for c in sc[b, i] if c > 0]
            assert all(sp[b, c] == i for c in ch) and ch == list(range(ch[0], ch[0] + len(ch))) if ch else True
        assert sum((sc[b] > 0).sum(1)) == V - 1


@pytest.mark.parametrize('H,W,C,B,low', [(96, 96, 1, 3, True), (96, 96, 2, 2, False), (10, 13, 3, 2, False), (4, 5, 1, 1, True)])
def test_refine_forward_backward_vs_oracle(built, dev, H, W, C, B, low):
    from boxinstseg_amd import bfs, mst, refine
    rng = np.random.default_rng(H * 5 + W + C)
    V = H * W
    idx = tfo.grid_edges(H, W)
    fm = rng.standard_normal((B, 3, H, W)).astype(np.float32)
    wt = np.stack([tfo.grid_weights(fm[b]) for b in range(B)])
    tree = mst(torch.from_numpy(idx)[None].repeat(B, 1, 1).to(dev), torch.from_numpy(wt).to(dev), V)
    si, sp, sc = bfs(tree, 4)
    x = rng.standard_normal((B, C, V)).astype(np.float32)
    g = rng.standard_normal((B, C, V)).astype(np.float32)
    sin, spn, scn = si.cpu().numpy(), sp.cpu().numpy(), sc.cpu().numpy()
    emb = rng.standard_normal((B, 3, V)) * (0.05 if low else 0.4)
    w = np.stack([tfo.edge_weights(emb[b], sin[b], spn[b], low) for b in range(B)]).astype(np.float32)
    xd = torch.from_numpy(x).to(dev).requires_grad_(True)
    wd = torch.from_numpy(w).to(dev).requires_grad_(True)
    out = refine(xd, wd, si, sp, sc, low)
    out.backward(torch.from_numpy(g).to(dev))
    for b in range(B):
        want, saved = tfo.refine_forward(x[b].astype(np.float64), w[b].astype(np.float64), sin[b], spn[b], scn[b])
        assert np.abs(out[b].detach().cpu().numpy() - want).max() <= 2e-5 * max(np.abs(want).max(), 1.0)
        gf = tfo.refine_backward_feature(g[b].astype(np.float64), w[b].astype(np.float64), sin[b], spn[b], scn[b], saved)
        assert np.abs(xd.grad[b].cpu().numpy() - gf).max() <= 2e-5 * max(np.abs(gf).max(), 1.0)
        if low:
            assert wd.grad is None                  # functions/refine.py:33-35: no gradient to the weights of the low-level tree
        else:
            gw = tfo.refine_backward_weight(x[b].astype(np.float64), g[b].astype(np.float64), w[b].astype(np.float64), sin[b], spn[b],
                                            scn[b], saved)
            assert np.abs(wd.grad[b].cpu().numpy() - gw).max() <= 1e-4 * max(np.abs(gw).max(), 1.0)


def _hip_refine_per_vertex(dev, tree_np, x, w_vertex, g):
    """HIP bfs + refine (forward, both gradients) on one tree; the edge weight of vertex v's edge to its parent is
    w_vertex[v]; everything returned in vertex order"""
    from boxinstseg_amd import bfs, refine
    tree = torch.from_numpy(np.ascontiguousarray(tree_np, np.int32))[None].to(dev)
    si, sp, sc = bfs(tree, 4)
    sin = si[0].cpu().numpy()
    xd = torch.from_numpy(x)[None].to(dev).requires_grad_(True)
    wd = torch.from_numpy(np.ascontiguousarray(w_vertex[sin]))[None].to(dev).requires_grad_(True)
    out = refine(xd, wd, si, sp, sc, False)
    out.backward(torch.from_numpy(g)[None].to(dev))
    gw = np.zeros_like(w_vertex)
    gw[sin] = wd.grad[0].cpu().numpy()
    return out[0].detach().cpu().numpy(), xd.grad[0].cpu().numpy(), gw


@pytest.mark.parametrize('case', ['a', 'b', 'd'])
def test_refine_vs_reference_kernels_fixture(built, dev, case):
    """HIP bfs + refine against what the reference's OWN bfs.cu / refine.cu kernels produced (run on the CPU through
    oracle/ref_wrap/cuda_on_cpu.h, committed as the refk_* keys): the BFS orders differ, the results per vertex must not."""
    g = np.load(os.path.join(GOLD, 'tree_filter.npz'))
    si = g[f'refk_{case}_si']
    w_vertex = np.zeros_like(g[f'refk_{case}_w']); w_vertex[si] = g[f'refk_{case}_w']
    out, gf, gw = _hip_refine_per_vertex(dev, g[f'{case}_tree'], g[f'refk_{case}_x'], w_vertex, g[f'refk_{case}_g'])
    want_gw = np.zeros_like(w_vertex); want_gw[si] = g[f'refk_{case}_gw']
    assert np.abs(out - g[f'refk_{case}_out']).max() <= 5e-6 * max(1.0, np.abs(g[f'refk_{case}_out']).max())
    assert np.abs(gf - g[f'refk_{case}_gf']).max() <= 5e-6 * max(1.0, np.abs(g[f'refk_{case}_gf']).max())
    assert np.abs(gw - want_gw).max() <= 2e-5 * max(1.0, np.abs(want_gw).max())


@pytest.mark.skipif(not tfo.ref_kernels_available(), reason='oracle/_ref/libtreekernels_ref.so not built (make -C oracle ref)')
def test_refine_vs_reference_kernels_live_96x96(built, dev):
    """the same comparison at Box2Mask's size, the reference kernels run on this host's CPU (the .so travels with the
    snapshot; built from the reference tree in the build container only)"""
    from boxinstseg_amd import mst
    rng = np.random.default_rng(77)
    H = W = 96
    V = H * W
    idx = tfo.grid_edges(H, W)
    fm = rng.standard_normal((3, H, W)).astype(np.float32)
    tree = mst(torch.from_numpy(idx)[None].to(dev), torch.from_numpy(tfo.grid_weights(fm))[None].to(dev), V)[0].cpu().numpy()
    r_si, r_sp, r_sc = tfo.ref_bfs(tree, V)
    C = 2
    x = rng.standard_normal((C, V)).astype(np.float32)
    g = rng.standard_normal((C, V)).astype(np.float32)
    w_vertex = np.exp(-rng.random(V) * 0.8).astype(np.float32)
    fwd = tfo.ref_refine_forward(x, w_vertex[r_si], r_si, r_sp, r_sc)
    r_gf, r_gw = tfo.ref_refine_backward(g, w_vertex[r_si], r_si, r_sp, r_sc, fwd)
    want_gw = np.zeros(V, np.float32); want_gw[r_si] = r_gw
    out, gf, gw = _hip_refine_per_vertex(dev, tree, x, w_vertex, g)
    assert np.abs(out - fwd['out']).max() <= 1e-5 * max(1.0, np.abs(fwd['out']).max())
    assert np.abs(gf - r_gf).max() <= 1e-5 * max(1.0, np.abs(r_gf).max())
    assert np.abs(gw - want_gw).max() <= 1e-4 * max(1.0, np.abs(want_gw).max())


def test_tree_filter_module_end_to_end(built, dev):
    """TreeFilter2D.forward: bfs + build_edge_weight (torch, differentiable w.r.t. the embedding) + refine, two groups"""
    from boxinstseg_amd import MinimumSpanningTree, TreeFilter2D
    rng = np.random.default_rng(11)
    B, C, H, W = 2, 4, 16, 20
    V = H * W
    guide = rng.standard_normal((B, 3, H, W)).astype(np.float32)
    feat = rng.standard_normal((B, C, H, W)).astype(np.float32)
    emb = (rng.standard_normal((B, 6, H, W)) * 0.4).astype(np.float32)
    tree = MinimumSpanningTree(TreeFilter2D.norm2_distance)(torch.from_numpy(guide).to(dev))
    tf2 = TreeFilter2D(groups=2)
    fd = torch.from_numpy(feat).to(dev).requires_grad_(True)
    ed = torch.from_numpy(emb).to(dev).requires_grad_(True)
    out = tf2(fd, ed, tree, low_tree=False)
    assert out.shape == fd.shape
    gout = rng.standard_normal(out.shape).astype(np.float32)
    out.backward(torch.from_numpy(gout).to(dev))
    idx = tfo.grid_edges(H, W)
    for b in range(B):
        si, sp, sc = tfo.bfs_order(tree.cpu().numpy()[b], V)
        for gidx in range(2):
            e = emb[b, 3 * gidx:3 * gidx + 3].reshape(3, V)
            w = tfo.edge_weights(e, si, sp, False)
            x = feat[b, 2 * gidx:2 * gidx + 2].reshape(2, V).astype(np.float64)
            want, saved = tfo.refine_forward(x, w, si, sp, sc)
            got = out[b, 2 * gidx:2 * gidx + 2].detach().cpu().numpy().reshape(2, V)
            assert np.abs(got - want).max() <= 3e-5 * max(np.abs(want).max(), 1.0)       # independent of the BFS order used
            gf = tfo.refine_backward_feature(gout[b, 2 * gidx:2 * gidx + 2].reshape(2, V).astype(np.float64), w, si, sp, sc, saved)
            assert np.abs(fd.grad[b, 2 * gidx:2 * gidx + 2].cpu().numpy().reshape(2, V) - gf).max() <= 3e-5 * max(np.abs(gf).max(), 1.0)
    assert ed.grad is not None and torch.isfinite(ed.grad).all() and float(ed.grad.abs().sum()) > 0


def test_tree_filter_errors(built, dev):
    from boxinstseg_amd import MinimumSpanningTree, TreeFilter2D, mst
    with pytest.raises(RuntimeError):
        MinimumSpanningTree(TreeFilter2D.norm2_distance)(torch.zeros(1, 3, 4, 4))            # CPU tensor
    with pytest.raises(RuntimeError):                                                      # beyond the 2^24-vertex limit of the workspace kernels
        mst(torch.zeros(1, 10, 2, dtype=torch.int32, device=dev), torch.ones(1, 10, device=dev), (1 << 24) + 1)


@pytest.mark.parametrize('seed', list(range(8)))
def test_tree_filter_fuzz(built, dev, seed):
    """Seeded sweep: grids from 1xK strips to 100x100, weights with heavy ties, several graphs per call: the GPU tree is the
    unique minimum spanning tree under (weight, index), its BFS order is valid, and refine matches the oracle on it."""
    from boxinstseg_amd import bfs, mst, refine
    rng = np.random.default_rng(6000 + seed)
    H = int(rng.choice([1, 2, 3, 7, 20, 45, 100])); W = int(rng.choice([2, 3, 9, 33, 64, 100]))
    B = int(rng.integers(1, 4)); C = int(rng.integers(1, 4)); V = H * W
    idx = tfo.grid_edges(H, W)
    wt = rng.uniform(1, 2, size=(B, len(idx))).astype(np.float32)
    if seed % 2:
        wt = (np.round(wt * 4) / 4).astype(np.float32)            # ties everywhere
    tree = mst(torch.from_numpy(idx)[None].repeat(B, 1, 1).to(dev), torch.from_numpy(wt).to(dev), V)
    si, sp, sc = bfs(tree, 4)
    t, sin, spn, scn = tree.cpu().numpy(), si.cpu().numpy(), sp.cpu().numpy(), sc.cpu().numpy()
    x = rng.standard_normal((B, C, V)).astype(np.float32); g = rng.standard_normal((B, C, V)).astype(np.float32)
    w = rng.uniform(0.2, 1.0, size=(B, V)).astype(np.float32)
    xd = torch.from_numpy(x).to(dev).requires_grad_(True); wd = torch.from_numpy(w).to(dev).requires_grad_(True)
    out = refine(xd, wd, si, sp, sc, False)
    out.backward(torch.from_numpy(g).to(dev))
    for b in range(B):
        assert _edge_set(t[b]) == _edge_set(idx[tfo.mst_edges(idx, wt[b], V)]), f'{H}x{W}'
        assert sorted(sin[b].tolist()) == list(range(V)) and (spn[b, 1:] < np.arange(1, V)).all()
        want, saved = tfo.refine_forward(x[b].astype(np.float64), w[b].astype(np.float64), sin[b], spn[b], scn[b])
        assert np.abs(out[b].detach().cpu().numpy() - want).max() <= 3e-5 * max(np.abs(want).max(), 1.0)
        gf = tfo.refine_backward_feature(g[b].astype(np.float64), w[b].astype(np.float64), sin[b], spn[b], scn[b], saved)
        assert np.abs(xd.grad[b].cpu().numpy() - gf).max() <= 3e-5 * max(np.abs(gf).max(), 1.0)
        gw = tfo.refine_backward_weight(x[b].astype(np.float64), g[b].astype(np.float64), w[b].astype(np.float64), sin[b], spn[b], scn[b], saved)
        assert np.abs(wd.grad[b].cpu().numpy() - gw).max() <= 2e-4 * max(np.abs(gw).max(), 1.0)


# ---------------------------------------------------------------------------------------------------------------------
# graphs beyond the LDS-resident limit (tree_filter_large.hip): BoxLevelSet filters 200x304 mask features
# (box_solov2_head.py:354-358; the reference's kernels have no vertex limit)
# ---------------------------------------------------------------------------------------------------------------------
LARGE = [(200, 304), (120, 136)]


@pytest.mark.parametrize('H,W', LARGE)
def test_large_mst_selects_the_reference_tree(built, dev, H, W):
    from boxinstseg_amd import mst
    rng = np.random.default_rng(H + W)
    V = H * W
    assert V > 10200
    idx = tfo.grid_edges(H, W)
    fm = rng.standard_normal((3, H, W)).astype(np.float32)
    fm2 = np.round(rng.standard_normal((1, H, W)) * 2).astype(np.float32)          # heavy ties
    wts = np.stack([tfo.grid_weights(fm), tfo.grid_weights(fm2)])
    tree = mst(torch.from_numpy(idx)[None].repeat(2, 1, 1).to(dev), torch.from_numpy(wts).to(dev), V).cpu().numpy()
    order = {(int(a), int(b)): i for i, (a, b) in enumerate(idx.tolist())}
    for b in range(2):
        want = tfo.ref_boruvka_mst(idx, wts[b], V) if tfo.ref_available() else idx[tfo.mst_edges(idx, wts[b], V)]
        assert _edge_set(tree[b]) == _edge_set(want), 'not the tree the reference Boruvka selects'
        ids = [order[(int(a), int(c))] for a, c in tree[b].tolist()]
        assert ids == sorted(ids) and len(ids) == V - 1


def _check_bfs(tree, V):
    """bfs() of tree [B, V - 1, 2] twice: deterministic, a permutation rooted at 0, parents before children, the tree's own edges,
    levels = the depth histogram, children contiguous and pointing back."""
    from boxinstseg_amd import bfs, _lib
    si, sp, sc = bfs(tree, 4)
    si2, sp2, sc2 = bfs(tree, 4)
    assert torch.equal(si, si2) and torch.equal(sp, sp2) and torch.equal(sc, sc2)
    # the level walk (bxi_dev_set_tree_level_walk) and the Euler-tour ranking (the default) are two algorithms for the same order
    lib = _lib.load()
    lib.bxi_dev_set_tree_level_walk(1)
    try:
        si3, sp3, sc3 = bfs(tree, 4)
    finally:
        lib.bxi_dev_set_tree_level_walk(0)
    assert torch.equal(si, si3) and torch.equal(sp, sp3) and torch.equal(sc, sc3)
    for la, lb in zip(si._bxi_levels.cpu(), si3._bxi_levels.cpu()):
        assert int(la[0]) == int(lb[0]) > 0 and torch.equal(la[:int(la[0]) + 2], lb[:int(la[0]) + 2])
    lv = si._bxi_levels.cpu().numpy()
    si, sp, sc, t = si.cpu().numpy(), sp.cpu().numpy(), sc.cpu().numpy(), tree.cpu().numpy()
    widest = 0
    for b in range(t.shape[0]):
        assert np.array_equal(np.sort(si[b]), np.arange(V)) and si[b, 0] == 0
        assert (sp[b, 1:] < np.arange(1, V)).all() and sp[b, 0] == 0
        a, c = si[b, 1:], si[b, sp[b, 1:]]
        got = set(zip(np.minimum(a, c).tolist(), np.maximum(a, c).tolist()))
        assert got == _edge_set(t[b])
        depth = np.zeros(V, np.int64)
        for i in range(1, V):
            depth[i] = depth[sp[b, i]] + 1
        assert (np.diff(depth) >= 0).all()
        D = lv[b, 0]
        assert D == depth.max() + 1 and lv[b, 1] == 0 and lv[b, 1 + D] == V
        assert np.array_equal(np.searchsorted(depth, np.arange(D)), lv[b, 1:1 + D])
        widest = max(widest, int(np.diff(lv[b, 1:2 + D]).max()))
        nch = (sc[b] > 0).sum(1)
        assert nch.sum() == V - 1
        first = sc[b, :, 0]
        has = nch > 0
        for k in range(1, 4):                       # children contiguous and pointing back to their parent
            m = nch > k
            assert (sc[b, m, k] == first[m] + k).all()
        assert (sp[b, first[has]] == np.nonzero(has)[0]).all()
        # siblings in ascending vertex order (bfs.cu walks the adjacency list, which the reference fills in edge order = ascending)
        sib = sp[b, 2:] == sp[b, 1:-1]
        assert (si[b, 2:][sib] > si[b, 1:-1][sib]).all()
    return widest


def test_large_bfs_is_a_valid_deterministic_order(built, dev):
    from boxinstseg_amd import mst
    H, W = 200, 304
    rng = np.random.default_rng(5)
    V = H * W
    idx = tfo.grid_edges(H, W)
    wt = (rng.uniform(size=(2, len(idx))) + 1).astype(np.float32)
    tree = mst(torch.from_numpy(idx)[None].repeat(2, 1, 1).to(dev), torch.from_numpy(wt).to(dev), V)
    _check_bfs(tree, V)


def _comb(H, W):
    """the top row as a spine, every column a tooth: from vertex 0 the frontier grows to min(H, W) nodes"""
    e = [(c, c + 1) for c in range(W - 1)] + [(r * W + c, (r + 1) * W + c) for r in range(H - 1) for c in range(W)]
    return np.asarray(e, np.int32)


@pytest.mark.parametrize('seed', range(6))
def test_large_bfs_random_trees(built, dev, seed):
    """Random trees of degree <= 4 that are not grid trees at all (random attachment, random labels, edges in random order and direction):
    validity (a BFS order of this tree from vertex 0, siblings in ascending vertex order -- bfs.cu's own sibling order is the arrival order
    of its atomics) and the Euler-tour form against the level walk bit for bit (_check_bfs)."""
    rng = np.random.default_rng(100 + seed)
    V = int(rng.integers(10201, 30000))
    deg = np.zeros(V, np.int64)
    edges = np.zeros((V - 1, 2), np.int64)
    open_ = [0]                                               # vertices that can still take a neighbour
    for v in range(1, V):
        k = int(rng.integers(0, len(open_))) if seed % 2 else max(len(open_) - 1 - int(rng.integers(0, 3)), 0)   # bushy / deep trees
        u = open_[k]
        edges[v - 1] = (u, v)
        deg[u] += 1; deg[v] += 1
        if deg[u] == 4:
            open_[k] = open_[-1]; open_.pop()
        open_.append(v)
    perm = rng.permutation(V)
    edges = perm[edges]
    flip = rng.random(V - 1) < 0.5
    edges[flip] = edges[flip][:, ::-1]
    edges = edges[rng.permutation(V - 1)].astype(np.int32)
    tree = torch.from_numpy(np.ascontiguousarray(edges))[None].to(dev)
    _check_bfs(tree, V)


@pytest.mark.parametrize('form', [0, 16])
def test_large_bfs_reports_input_it_cannot_represent(built, dev, form):
    """More than 4 neighbours, or an edge list that is not one connected tree (V - 1 edges with a cycle somewhere): levels[0] = -1 from both
    forms of the large BFS (the Euler-tour ranking and the level walk), every output index in range, and a good graph in the same batch
    is not affected."""
    from boxinstseg_amd import bfs, _lib
    V = 12000
    path = np.stack([np.arange(V - 1), np.arange(1, V)], 1).astype(np.int32)
    star = path.copy()
    star[5:10, 0] = 5                                       # 5-6, 5-7, 5-8, 5-9, 5-10 next to 4-5: six neighbours
    star[5:10, 1] = np.arange(6, 11)
    loop = path.copy()
    loop[V - 4] = (V - 3, V - 1)                            # 0 .. V-4 a path; V-3, V-2, V-1 a triangle of their own
    trees = torch.from_numpy(np.stack([path, star, loop, path])).to(dev)
    lib = _lib.load()
    lib.bxi_dev_set_tree_level_walk(1 if form else 0)
    try:
        si, sp, sc = bfs(trees, 4)
        torch.cuda.synchronize()
    finally:
        lib.bxi_dev_set_tree_level_walk(0)
    lv = si._bxi_levels.cpu().numpy()
    assert lv[0, 0] == V and lv[3, 0] == V                  # a path from vertex 0: one vertex per level
    assert lv[1, 0] == -1 and lv[2, 0] == -1
    for t in (si, sp):
        assert int(t.min()) >= 0 and int(t.max()) < V
    assert int(sc.min()) >= 0 and int(sc.max()) < V
    assert np.array_equal(si[0].cpu().numpy(), np.arange(V)) and np.array_equal(si[3].cpu().numpy(), np.arange(V))


@pytest.mark.parametrize('kind', ['grid_wide_frontier', 'relabelled', 'relabelled_wide_frontier', 'strip', 'three_sort_passes_wide_frontier'])
def test_large_bfs_forms(built, dev, kind):
    """bfs() of a large tree ranks its Euler tour; the level walk (debug form 16, compared bit for bit inside _check_bfs) has a grid form
    (adjacency as four bits per vertex in LDS) and a general one (16-byte records), and each of them a one-wave form (frontier <= 256
    nodes) and a workgroup form: every combination, the 1 x V strip (grid width 1: general form; V levels), and a tree whose depths need
    three passes of the radix sort."""
    from boxinstseg_amd import mst
    rng = np.random.default_rng(11)
    if kind == 'strip':
        V = 20000
        t = np.stack([np.arange(V - 1), np.arange(1, V)], 1).astype(np.int32)
        widest = _check_bfs(torch.from_numpy(t)[None].to(dev), V)
        assert widest == 1
        return
    if kind == 'relabelled':
        H, W = 120, 136
        idx = tfo.grid_edges(H, W)
        wt = (rng.uniform(size=(1, len(idx))) + 1).astype(np.float32)
        t = mst(torch.from_numpy(idx)[None].to(dev), torch.from_numpy(wt).to(dev), H * W).cpu().numpy()[0]
    else:
        H, W = (520, 512) if kind.startswith('three') else (300, 300)      # 266 240 vertices: depths need 19 bits, three 9-bit passes
        t = _comb(H, W)
    V = H * W
    if kind.startswith('relabelled'):
        perm = rng.permutation(V).astype(np.int32)                  # the same tree, no longer v +- 1 / v +- W
        t = perm[t]
    widest = _check_bfs(torch.from_numpy(np.ascontiguousarray(t))[None].to(dev), V)
    assert (widest > 256) == kind.endswith('wide_frontier'), widest


@pytest.fixture(params=[0, 16], ids=['depth_free', 'level_walks'])
def large_form(request):
    """Both forms of the large-tree kernels: the default (Euler-tour BFS, leaf->root pass by doubling over the levels) and the level walks
    (bxi_dev_set_tree_level_walk, include/boxinst_hip_dev.h)."""
    from boxinstseg_amd import _lib
    lib = _lib.load()
    lib.bxi_dev_set_tree_level_walk(1 if request.param else 0)
    yield request.param
    lib.bxi_dev_set_tree_level_walk(0)


@pytest.mark.parametrize('low', [True, False])
def test_large_refine_forward_backward_vs_oracle(built, dev, low, large_form):
    from boxinstseg_amd import bfs, mst, refine
    H, W, C, B = 200, 304, 2, 2
    rng = np.random.default_rng(11 + low)
    V = H * W
    idx = tfo.grid_edges(H, W)
    fm = rng.standard_normal((B, 3, H, W)).astype(np.float32)
    wt = np.stack([tfo.grid_weights(fm[b]) for b in range(B)])
    tree = mst(torch.from_numpy(idx)[None].repeat(B, 1, 1).to(dev), torch.from_numpy(wt).to(dev), V)
    si, sp, sc = bfs(tree, 4)
    x = rng.standard_normal((B, C, V)).astype(np.float32)
    g = rng.standard_normal((B, C, V)).astype(np.float32)
    sin, spn, scn = si.cpu().numpy(), sp.cpu().numpy(), sc.cpu().numpy()
    emb = rng.standard_normal((B, 3, V)) * (0.05 if low else 0.4)
    w = np.stack([tfo.edge_weights(emb[b], sin[b], spn[b], low) for b in range(B)]).astype(np.float32)
    xd = torch.from_numpy(x).to(dev).requires_grad_(True)
    wd = torch.from_numpy(w).to(dev).requires_grad_(True)
    out = refine(xd, wd, si, sp, sc, low)
    out.backward(torch.from_numpy(g).to(dev))
    for b in range(B):
        want, saved = tfo.refine_forward(x[b].astype(np.float64), w[b].astype(np.float64), sin[b], spn[b], scn[b])
        assert np.abs(out[b].detach().cpu().numpy() - want).max() <= 2e-5 * max(np.abs(want).max(), 1.0)
        gf = tfo.refine_backward_feature(g[b].astype(np.float64), w[b].astype(np.float64), sin[b], spn[b], scn[b], saved)
        assert np.abs(xd.grad[b].cpu().numpy() - gf).max() <= 2e-5 * max(np.abs(gf).max(), 1.0)
        if low:
            assert wd.grad is None
        else:
            gw = tfo.refine_backward_weight(x[b].astype(np.float64), g[b].astype(np.float64), w[b].astype(np.float64), sin[b], spn[b],
                                            scn[b], saved)
            assert np.abs(wd.grad[b].cpu().numpy() - gw).max() <= 1e-4 * max(np.abs(gw).max(), 1.0)


@pytest.mark.parametrize('shape', ['strip', 'comb', 'bushy'])
def test_large_refine_tree_shapes(built, dev, shape, large_form):
    """The extremes of the depth-free leaf->root pass: a 1 x V strip (V levels of one node: every doubling round is effective), a comb
    (long teeth, levels up to 120 nodes wide, long descendant ranges near the root), a bushy random tree (few levels, wide ranges:
    the whole-wave sums) -- forward and the feature gradient against the fp64 oracle, in both forms."""
    from boxinstseg_amd import bfs, refine
    rng = np.random.default_rng({'strip': 1, 'comb': 2, 'bushy': 3}[shape])
    if shape == 'strip':
        V = 12000
        t = np.stack([np.arange(V - 1), np.arange(1, V)], 1)
    elif shape == 'comb':
        H, W = 110, 120
        V = H * W
        t = _comb(H, W)
    else:
        V = 20000
        deg = np.zeros(V, np.int64)
        t = np.zeros((V - 1, 2), np.int64)
        open_ = [0]
        for v in range(1, V):
            k = int(rng.integers(0, min(len(open_), 3)))          # attach near the oldest open vertex: shallow and wide
            u = open_[k]
            t[v - 1] = (u, v)
            deg[u] += 1; deg[v] += 1
            if deg[u] == 4:
                open_.pop(k)
            open_.append(v)
    tree = torch.from_numpy(np.ascontiguousarray(t.astype(np.int32)))[None].to(dev)
    si, sp, sc = bfs(tree, 4)
    sin, spn, scn = si.cpu().numpy(), sp.cpu().numpy(), sc.cpu().numpy()
    C = 2
    x = rng.standard_normal((1, C, V)).astype(np.float32)
    g = rng.standard_normal((1, C, V)).astype(np.float32)
    w = rng.uniform(0.55, 1.0, (1, V)).astype(np.float32)          # weights in sorted order (refine.cu: weight[0] is unused)
    xd = torch.from_numpy(x).to(dev).requires_grad_(True)
    out = refine(xd, torch.from_numpy(w).to(dev), si, sp, sc, True)
    out.backward(torch.from_numpy(g).to(dev))
    want, saved = tfo.refine_forward(x[0].astype(np.float64), w[0].astype(np.float64), sin[0], spn[0], scn[0])
    assert np.isfinite(want).all()
    assert np.abs(out[0].detach().cpu().numpy() - want).max() <= 2e-5 * max(np.abs(want).max(), 1.0)
    gf = tfo.refine_backward_feature(g[0].astype(np.float64), w[0].astype(np.float64), sin[0], spn[0], scn[0], saved)
    assert np.abs(xd.grad[0].cpu().numpy() - gf).max() <= 2e-5 * max(np.abs(gf).max(), 1.0)


@pytest.mark.skipif(not tfo.ref_kernels_available(), reason='oracle/_ref/libtreekernels_ref.so not built (make -C oracle ref)')
def test_large_refine_vs_reference_kernels_live_200x304(built, dev):
    """BoxLevelSet's size against the reference's OWN bfs.cu / refine.cu kernels run on this host's CPU
    (oracle/_ref/libtreekernels_ref.so travels with the snapshot): the BFS orders differ, the results per vertex must not."""
    from boxinstseg_amd import mst
    rng = np.random.default_rng(78)
    H, W = 200, 304
    V = H * W
    idx = tfo.grid_edges(H, W)
    fm = rng.standard_normal((3, H, W)).astype(np.float32)
    tree = mst(torch.from_numpy(idx)[None].to(dev), torch.from_numpy(tfo.grid_weights(fm))[None].to(dev), V)[0].cpu().numpy()
    r_si, r_sp, r_sc = tfo.ref_bfs(tree, V)
    C = 1
    x = rng.standard_normal((C, V)).astype(np.float32)
    g = rng.standard_normal((C, V)).astype(np.float32)
    w_vertex = np.exp(-rng.random(V) * 0.8).astype(np.float32)
    fwd = tfo.ref_refine_forward(x, w_vertex[r_si], r_si, r_sp, r_sc)
    r_gf, r_gw = tfo.ref_refine_backward(g, w_vertex[r_si], r_si, r_sp, r_sc, fwd)
    want_gw = np.zeros(V, np.float32); want_gw[r_si] = r_gw
    out, gf, gw = _hip_refine_per_vertex(dev, tree, x, w_vertex, g)
    assert np.abs(out - fwd['out']).max() <= 1e-5 * max(1.0, np.abs(fwd['out']).max())
    assert np.abs(gf - r_gf).max() <= 1e-5 * max(1.0, np.abs(r_gf).max())
    assert np.abs(gw - want_gw).max() <= 1e-4 * max(1.0, np.abs(want_gw).max())


def test_large_tree_filter_module_end_to_end(built, dev):
    """MinimumSpanningTree + TreeFilter2D on a 200x304 map with 2 groups, forward and backward, against the oracle."""
    from boxinstseg_amd import MinimumSpanningTree, TreeFilter2D
    rng = np.random.default_rng(3)
    B, C, H, W = 1, 4, 200, 304
    V = H * W
    guide = torch.from_numpy(rng.standard_normal((B, 3, H, W)).astype(np.float32)).to(dev)
    feat = torch.from_numpy(rng.standard_normal((B, C, H, W)).astype(np.float32)).to(dev).requires_grad_(True)
    emb = torch.from_numpy((rng.standard_normal((B, 4, H, W)) * 0.3).astype(np.float32)).to(dev).requires_grad_(True)
    tree = MinimumSpanningTree(TreeFilter2D.norm2_distance)(guide)
    out = TreeFilter2D(groups=2)(feat, emb, tree, low_tree=False)
    assert out.shape == feat.shape and torch.isfinite(out).all()
    gout = rng.standard_normal(out.shape).astype(np.float32)
    out.backward(torch.from_numpy(gout).to(dev))
    si, sp, sc = tfo.bfs_order(tree[0].cpu().numpy(), V)
    e = emb.detach().cpu().numpy()[0].reshape(4, V)
    f = feat.detach().cpu().numpy()[0].reshape(C, V)
    for gidx in range(2):
        w = tfo.edge_weights(e[2 * gidx:2 * gidx + 2].astype(np.float64), si, sp, False)
        want, saved = tfo.refine_forward(f[2 * gidx:2 * gidx + 2].astype(np.float64), w, si, sp, sc)
        got = out[0, 2 * gidx:2 * gidx + 2].detach().cpu().numpy().reshape(2, V)
        assert np.abs(got - want).max() <= 3e-5 * max(np.abs(want).max(), 1.0)
    assert emb.grad is not None and torch.isfinite(emb.grad).all() and float(emb.grad.abs().sum()) > 0
    assert torch.isfinite(feat.grad).all()
