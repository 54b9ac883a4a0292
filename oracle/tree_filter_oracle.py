"""CPU restatement of the reference's ``tree_filter`` extension: minimum spanning tree of the
4-connected pixel graph, breadth-first ordering, and the tree-filter refinement with its two gradients.

TEST INFRASTRUCTURE ONLY (imported by tests/ and tools/ only).  Reference: mmdet/ops/tree_filter/
  src/mst/boruvka.cpp (+ mst.cu:49-118), src/bfs/bfs.cu:19-90, src/refine/refine.cu:17-370,
  modules/tree_filter.py:10-150 (MinimumSpanningTree, TreeFilter2D).

Pinning status:
  * MST: PINNED.  ``ref_boruvka_mst`` runs the reference's own boruvka.cpp, compiled from the reference tree into
    oracle/_ref/libboruvka_ref.so (oracle/Makefile, target ``ref``); ``mst_edges`` (Kruskal under the total order
    (weight, edge index), i.e. the unique tree Boruvka's strict ``>`` comparisons select) is checked against it, and
    the edge sets it produced are committed in tests/golden/tree_filter.npz.
  * BFS order and refine (forward, backward w.r.t. feature and edge weight): PINNED to the reference's own KERNELS.
    bfs.cu / refine.cu are CUDA + ATen/THC and cannot run here as they are, but their `__global__` functions are plain
    C++ apart from the CUDA execution model, so oracle/Makefile (target ``ref``) compiles exactly those functions, from
    the reference tree, against oracle/ref_wrap/cuda_on_cpu.h (one OS thread per CUDA thread of a block, barrier =
    __syncthreads, fetch-add = atomicAdd) into oracle/_ref/libtreekernels_ref.so; the host functions around them (launch
    shapes, the divisions between launches) are restated in oracle/ref_wrap/tree_kernels_wrap.cpp with line citations.
    ``ref_bfs`` / ``ref_refine_*`` below run them; their outputs on seeded trees are committed in
    tests/golden/tree_filter.npz (``refk_*`` keys) and the restatement is checked against them.  What this does not pin
    is nvcc's FMA contraction (the kernels are compiled with contraction off): agreement is at fp32 rounding level.
    The reference's BFS order depends on the arrival order of atomics (bfs.cu:72) -- any breadth-first order is equally
    valid and everything downstream is compared in vertex order.  The restatement is additionally validated against
    (a) the closed form refine implements, out_i = sum_j S(i,j) x_j / sum_j S(i,j) with S the product of the edge
    weights on the tree path i..j (brute force), and (b) torch autograd through that closed form.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_REF = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_ref', 'libboruvka_ref.so')


def ref_available() -> bool:
    return os.path.exists(_REF)


def ref_boruvka_mst(index: np.ndarray, weight: np.ndarray, V: int) -> np.ndarray:
    """the reference's boruvka.cpp itself: index [E,2] int32, weight [E] f32 -> edges [V-1,2] in its emission order"""
    lib = ctypes.CDLL(_REF)
    idx = np.ascontiguousarray(index, np.int32); w = np.ascontiguousarray(weight, np.float32)
    out = np.zeros((V - 1, 2), np.int32)
    lib.ref_boruvka_mst(ctypes.c_int(V), ctypes.c_int(len(w)), idx.ctypes.data_as(ctypes.c_void_p),
                        w.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
    return out


_REFK = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_ref', 'libtreekernels_ref.so')


def ref_kernels_available() -> bool:
    return os.path.exists(_REFK)


def _vp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


_POOL = None


def _isolated(fn_name: str, *args):
    """Run one of the ``_ref_*`` functions below in a separate (spawned, re-used) process: the emulated CUDA blocks are
    64 real threads running foreign kernels, and a fault in there must fail one test, not take the test session down."""
    global _POOL
    import concurrent.futures as cf
    import multiprocessing as mp
    if _POOL is None:
        _POOL = cf.ProcessPoolExecutor(max_workers=1, mp_context=mp.get_context('spawn'))
    try:
        return _POOL.submit(_dispatch, fn_name, args).result(timeout=600)
    except cf.process.BrokenProcessPool as e:
        _POOL = None
        raise RuntimeError(f'reference kernel emulation ({fn_name}) crashed in its worker process') from e


def _dispatch(fn_name, args):
    return globals()[fn_name](*args)


def ref_bfs(edges: np.ndarray, V: int, max_adj: int = 4):
    """the reference's adj_vec_kernel + breadth_first_sort_kernel (bfs.cu:19-90) run on the CPU: edges [V-1,2] ->
    sorted_index [V], sorted_parent [V], sorted_child [V,max_adj] (one valid arrival order of its atomics)"""
    return _isolated('_ref_bfs', np.asarray(edges), int(V), int(max_adj))


def ref_refine_forward(x: np.ndarray, w: np.ndarray, si, sp, sc):
    """the reference's refine_forward kernels (refine.cu:19-134 under :201-249): x [C,V] f32 vertex order, w [V] f32 ->
    dict(out, aggr [C,V] vertex order, aggr_up [C,V] sorted, wsum [V] vertex, wsum_up [V] sorted), all f32"""
    return _isolated('_ref_refine_forward', *(np.asarray(a) for a in (x, w, si, sp, sc)))


def ref_refine_backward(g: np.ndarray, w: np.ndarray, si, sp, sc, fwd: dict):
    """the reference's refine_backward_feature / refine_backward_weight (refine.cu:251-370) on the tensors its forward
    saved: g [C,V] -> (grad_feature [C,V] vertex order, grad_weight [V] sorted order)"""
    return _isolated('_ref_refine_backward', *(np.asarray(a) for a in (g, w, si, sp, sc)), fwd)


def _ref_bfs(edges: np.ndarray, V: int, max_adj: int = 4):
    lib = ctypes.CDLL(_REFK)
    e = np.ascontiguousarray(edges, np.int32).reshape(1, V - 1, 2)
    si = np.zeros((1, V), np.int32); sp = np.zeros((1, V), np.int32); sc = np.zeros((1, V, max_adj), np.int32)
    lib.ref_bfs_forward(_vp(e), 1, V, max_adj, _vp(si), _vp(sp), _vp(sc))
    return si[0], sp[0], sc[0]


def _ref_refine_forward(x: np.ndarray, w: np.ndarray, si, sp, sc):
    lib = ctypes.CDLL(_REFK)
    C, V = x.shape
    A = sc.shape[1]
    xs = np.ascontiguousarray(x, np.float32); ws = np.ascontiguousarray(w, np.float32)
    si_, sp_, sc_ = (np.ascontiguousarray(t, np.int32) for t in (si, sp, sc))
    r = dict(out=np.zeros((C, V), np.float32), aggr=np.zeros((C, V), np.float32), aggr_up=np.zeros((C, V), np.float32),
             wsum=np.zeros(V, np.float32), wsum_up=np.zeros(V, np.float32))
    lib.ref_refine_forward(_vp(xs), _vp(ws), _vp(si_), _vp(sp_), _vp(sc_), 1, C, V, A, _vp(r['out']), _vp(r['aggr']),
                           _vp(r['aggr_up']), _vp(r['wsum']), _vp(r['wsum_up']))
    return r


def _ref_refine_backward(g: np.ndarray, w: np.ndarray, si, sp, sc, fwd: dict):
    lib = ctypes.CDLL(_REFK)
    C, V = g.shape
    A = sc.shape[1]
    gs = np.ascontiguousarray(g, np.float32); ws = np.ascontiguousarray(w, np.float32)
    si_, sp_, sc_ = (np.ascontiguousarray(t, np.int32) for t in (si, sp, sc))
    gf = np.zeros((C, V), np.float32); gw = np.zeros(V, np.float32)
    lib.ref_refine_backward_feature(_vp(ws), _vp(si_), _vp(sp_), _vp(sc_), _vp(fwd['wsum']), _vp(gs), 1, C, V, A, _vp(gf))
    lib.ref_refine_backward_weight(_vp(ws), _vp(si_), _vp(sp_), _vp(sc_), _vp(fwd['out']), _vp(fwd['aggr']), _vp(fwd['aggr_up']),
                                   _vp(fwd['wsum']), _vp(fwd['wsum_up']), _vp(gs), 1, C, V, A, _vp(gw))
    return gf, gw


def grid_edges(H: int, W: int) -> np.ndarray:
    """MinimumSpanningTree._build_matrix_index (tree_filter.py:15-26): vertical edges first, then horizontal"""
    raw = np.arange(H * W, dtype=np.int32).reshape(H, W)
    rows = np.stack([raw[:-1, :], raw[1:, :]], 2).reshape(-1, 2)
    cols = np.stack([raw[:, :-1], raw[:, 1:]], 2).reshape(-1, 2)
    return np.concatenate([rows, cols], 0)


def grid_weights(fm: np.ndarray) -> np.ndarray:
    """_build_feature_weight (:28-35) with norm2_distance (:76-80): fm [C,H,W] -> [E] f32, +1"""
    f = fm.astype(np.float32)
    dr = ((f[:, :-1, :] - f[:, 1:, :]) ** 2).sum(0, dtype=np.float32).reshape(-1)
    dc = ((f[:, :, :-1] - f[:, :, 1:]) ** 2).sum(0, dtype=np.float32).reshape(-1)
    return (np.concatenate([dr, dc]) + np.float32(1)).astype(np.float32)


def mst_edges(index: np.ndarray, weight: np.ndarray, V: int) -> np.ndarray:
    """the minimum spanning tree under the total order (weight, edge index) -> chosen edge ids, ascending"""
    order = np.lexsort((np.arange(len(weight)), weight.astype(np.float32)))
    parent = list(range(V))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a
    chosen = []
    for e in order:
        a, b = find(int(index[e, 0])), find(int(index[e, 1]))
        if a != b:
            parent[a] = b
            chosen.append(int(e))
    return np.array(sorted(chosen), np.int64)


def bfs_order(edges: np.ndarray, V: int, max_adj: int = 4):
    """bfs.cu:19-90: root = vertex 0 -> sorted_index [V], sorted_parent [V] (position of the parent), sorted_child
    [V,max_adj] (positions, 0 = none)"""
    adj = [[] for _ in range(V)]
    for a, b in edges:
        adj[int(a)].append(int(b)); adj[int(b)].append(int(a))
    si = np.zeros(V, np.int32); sp = np.zeros(V, np.int32); sc = np.zeros((V, max_adj), np.int32)
    par_vertex = np.zeros(V, np.int32)
    n = 1
    for i in range(V):
        cur, par = int(si[i]), int(par_vertex[i])
        k = 0
        for ch in adj[cur]:
            if ch != par or (i == 0 and False):
                if i == 0 or ch != par:
                    si[n] = ch; par_vertex[n] = cur; sp[n] = i; sc[i, k] = n
                    k += 1; n += 1
    assert n == V, 'edges do not form a spanning tree'
    return si, sp, sc


def edge_weights(embed: np.ndarray, si: np.ndarray, sp: np.ndarray, low_tree: bool, sigma: float = 0.02) -> np.ndarray:
    """TreeFilter2D.build_edge_weight (:90-108): embed [C,V] -> w [V] (sorted order; w[0] is never used)"""
    src = embed[:, si].astype(np.float64)
    d = ((src - src[:, sp]) ** 2).sum(0)
    return np.exp(-d / sigma) if low_tree else np.exp(-d)


def _up(x_sorted: np.ndarray, w: np.ndarray, sc: np.ndarray) -> np.ndarray:
    """leaf_root_aggr_kernel (refine.cu:64-121): U_i = x_i + sum_children w_c U_c   (x given in sorted order)"""
    U = x_sorted.astype(np.float64).copy()
    for i in range(len(w) - 1, -1, -1):
        for c in sc[i]:
            if c > 0:
                U[..., i] += U[..., c] * w[c]
    return U


def _down(U: np.ndarray, w: np.ndarray, sp: np.ndarray) -> np.ndarray:
    """root_leaf_prop_kernel (:17-62): D_0 = U_0 ; D_i = U_i (1 - w_i^2) + D_parent w_i   (sorted order)"""
    D = U.copy()
    for i in range(1, len(w)):
        D[..., i] = U[..., i] * (1 - w[i] * w[i]) + D[..., sp[i]] * w[i]
    return D


def refine_forward(x: np.ndarray, w: np.ndarray, si, sp, sc):
    """refine_forward (:186-233): x [C,V] in vertex order -> (out [C,V] vertex order, saved dict)"""
    U = _up(x[:, si], w, sc); D = _down(U, w, sp)
    WU = _up(np.ones(len(w)), w, sc); WD = _down(WU, w, sp)
    out = np.empty_like(D)
    out[:, si] = D / WD
    return out, dict(U=U, D=D, WU=WU, WD=WD)


def refine_backward_feature(g: np.ndarray, w, si, sp, sc, saved) -> np.ndarray:
    """refine_backward_feature (:235-282): the operator is self-adjoint"""
    gn = g[:, si] / saved['WD']
    out = np.empty_like(gn)
    out[:, si] = _down(_up(gn, w, sc), w, sp)
    return out


def refine_backward_weight(x: np.ndarray, g: np.ndarray, w, si, sp, sc, saved) -> np.ndarray:
    """refine_backward_weight (:284-370) + root_leaf_grad_kernel (:123-184) -> d/d w [V] (sorted order, [0] = 0)"""
    V = len(w)
    out = saved['D'] / saved['WD']
    gn = g[:, si] / saved['WD']
    GU = _up(gn, w, sc)                      # grad_out_norm_aggr_sum
    FGU = _up(gn * out, w, sc)               # feature_grad_aggr_sum

    def pass_(in_data, in_grad, out_data):
        grad = np.zeros_like(in_grad)
        og = in_grad.copy()                  # out_grad aliases in_grad in the reference: updated root to leaf
        for i in range(1, V):
            p = sp[i]
            left = in_grad[..., i] * (out_data[..., p] - w[i] * in_data[..., i])
            right = in_data[..., i] * (og[..., p] - w[i] * in_grad[..., i])
            grad[..., i] = left + right
            og[..., i] = in_grad[..., i] * (1 - w[i] * w[i]) + og[..., p] * w[i]
        return grad
    g_all = pass_(saved['U'], GU, saved['D'])
    g_norm = pass_(saved['WU'][None], FGU, saved['WD'][None])
    return (g_all - g_norm).sum(0)


def refine_closed_form(x: np.ndarray, w: np.ndarray, si, sp) -> np.ndarray:
    """out_i = sum_j S(i,j) x_j / sum_j S(i,j), S(i,j) = product of the edge weights on the tree path (O(V^2))"""
    V = len(w)
    S = np.zeros((V, V))
    for i in range(V):            # sorted order: parents precede children
        S[i, i] = 1.0
        if i:
            p = sp[i]
            S[i, :i] = S[p, :i] * w[i]
            S[:i, i] = S[i, :i]
    xs = x[:, si].astype(np.float64)
    o = (S[None] * xs[:, None, :]).sum(2) / S.sum(1)[None]
    out = np.empty_like(o)
    out[:, si] = o
    return out
