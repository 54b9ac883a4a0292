"""The reference's ``tree_filter`` extension on the HIP kernels of ``csrc/tree_filter.hip``:
``mst`` / ``bfs`` / ``refine`` (``mmdet/ops/tree_filter/functions/*.py`` over ``tree_filter_cuda``) and the modules
``MinimumSpanningTree`` / ``TreeFilter2D`` (``mmdet/ops/tree_filter/modules/tree_filter.py:10-150``).
Same names, argument order and return values; the glue between the native calls is torch on the GPU, as in the
reference.  No CPU path.  Graphs of up to 10200 vertices (Box2Mask's 96x96 maps) run LDS-resident kernels, larger ones
(BoxLevelSet's full-resolution mask features) the global-workspace kernels of ``csrc/tree_filter_large.hip``.
"""
from __future__ import annotations

import torch
from torch.autograd.function import once_differentiable

from . import _lib


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def _need_cuda(**tensors):
    for name, t in tensors.items():
        if t is not None and not t.is_cuda:
            raise RuntimeError(f'{name} must be a CUDA (HIP) tensor: boxinstseg_amd has no CPU path')


def mst(edge_index, edge_weight, vertex_count):
    """``mst(edge_index [B,E,2] int32, edge_weight [B,E], vertex_count)`` -> tree edges [B,V-1,2] int32 (functions/mst.py).
    Computed on the GPU (the reference round-trips through the host); the edges are listed in ascending edge order."""
    _need_cuda(edge_index=edge_index, edge_weight=edge_weight)
    dev = edge_index.device
    idx = edge_index.detach().to(torch.int32).contiguous()
    w = edge_weight.detach().to(torch.float32).contiguous()
    B, E = w.shape
    V = int(vertex_count)
    out = torch.empty((B, V - 1, 2), dtype=torch.int32, device=dev)
    lib = _lib.load()
    ws = torch.empty(max(lib.bxi_mst_workspace_bytes(B, E, V), 16), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check('bxi_mst_forward_i32', lib.bxi_mst_forward_i32(idx.data_ptr(), w.data_ptr(), B, E, V, out.data_ptr(), ws.data_ptr(),
                                                                   ws.numel(), _stream(dev)))
    return out


def _bfs_levels(edge_index, max_adj_per_vertex):
    _need_cuda(edge_index=edge_index)
    dev = edge_index.device
    tree = edge_index.detach().to(torch.int32).contiguous()
    B, V = tree.size(0), tree.size(1) + 1
    si = torch.empty((B, V), dtype=torch.int32, device=dev)
    sp = torch.empty((B, V), dtype=torch.int32, device=dev)
    sc = torch.empty((B, V, max_adj_per_vertex), dtype=torch.int32, device=dev)
    lv = torch.empty((B, V + 2), dtype=torch.int32, device=dev)
    lib = _lib.load()
    ws = torch.empty(max(lib.bxi_bfs_workspace_bytes(B, V), 16), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check('bxi_bfs_forward_i32', lib.bxi_bfs_forward_i32(tree.data_ptr(), B, V, int(max_adj_per_vertex), si.data_ptr(),
                                                                   sp.data_ptr(), sc.data_ptr(), lv.data_ptr(), ws.data_ptr(), ws.numel(),
                                                                   _stream(dev)))
    return si, sp, sc, lv


def bfs(edge_index, max_adj_per_vertex):
    """``bfs(tree, 4)`` -> (sorted_index, sorted_parent, sorted_child) (functions/bfs.py).  Deterministic order."""
    si, sp, sc, lv = _bfs_levels(edge_index, max_adj_per_vertex)
    si._bxi_levels = lv
    return si, sp, sc


def _levels_of(sorted_index, sorted_parent):
    lv = getattr(sorted_index, '_bxi_levels', None)
    if lv is not None:
        return lv
    # orderings that did not come from bfs() above (e.g. expanded per group): rebuild the level table from the parents
    B, V = sorted_parent.shape
    depth = torch.zeros((B, V), dtype=torch.int64, device=sorted_parent.device)
    par = sorted_parent.long()
    cur = par.clone()
    for _ in range(V):                       # depth by pointer jumping towards position 0
        step = (cur > 0).long()
        if not bool(step.any()):
            break
        depth += step
        cur = torch.gather(par, 1, cur) * (cur > 0)
    depth[:, 0] = 0
    lv = torch.zeros((B, V + 2), dtype=torch.int32, device=sorted_parent.device)
    for b in range(B):
        d = depth[b]
        D = int(d.max()) + 1
        counts = torch.bincount(d, minlength=D)
        lv[b, 0] = D
        lv[b, 2:2 + D] = torch.cumsum(counts, 0).int()
    return lv


class _Refine(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feature_in, edge_weight, sorted_index, sorted_parent, sorted_child, low_tree, levels):
        _need_cuda(feature_in=feature_in, edge_weight=edge_weight, sorted_index=sorted_index)
        dev = feature_in.device
        x = feature_in.detach().to(torch.float32).contiguous()
        w = edge_weight.detach().to(torch.float32).contiguous()
        B, C, V = x.shape
        A = sorted_child.size(2)
        si, sp, sc = (t.to(torch.int32).contiguous() for t in (sorted_index, sorted_parent, sorted_child))
        out, aggr, aggr_up = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        wsum = torch.empty((B, V), dtype=torch.float32, device=dev)
        wsum_up = torch.empty((B, V), dtype=torch.float32, device=dev)
        lib = _lib.load()
        ws = torch.empty(max(lib.bxi_tree_refine_workspace_bytes(B, C, V), 16), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check('bxi_tree_refine_forward_f32', lib.bxi_tree_refine_forward_f32(
                x.data_ptr(), w.data_ptr(), si.data_ptr(), sc.data_ptr(), levels.data_ptr(), B, C, V, A, out.data_ptr(), aggr.data_ptr(),
                aggr_up.data_ptr(), wsum.data_ptr(), wsum_up.data_ptr(), ws.data_ptr(), ws.numel(), _stream(dev)))
        ctx.save_for_backward(w, si, sp, sc, levels, out, aggr, aggr_up, wsum, wsum_up)
        ctx.low_tree = low_tree
        ctx.dtypes = (feature_in.dtype, edge_weight.dtype)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        w, si, sp, sc, levels, out, aggr, aggr_up, wsum, wsum_up = ctx.saved_tensors
        dev = w.device
        B, C, V = out.shape
        A = sc.size(2)
        g = g.to(torch.float32).contiguous()
        lib = _lib.load()
        gf = torch.empty_like(out)
        gw = None
        if ctx.low_tree:                # functions/refine.py:33-41: the low-level tree passes no gradient to its weights
            ws = torch.empty(max(lib.bxi_tree_refine_workspace_bytes(B, C, V), 16), dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                _lib.check('bxi_tree_refine_backward_feature_f32', lib.bxi_tree_refine_backward_feature_f32(
                    g.data_ptr(), w.data_ptr(), si.data_ptr(), sc.data_ptr(), levels.data_ptr(), wsum.data_ptr(), B, C, V, A, gf.data_ptr(),
                    ws.data_ptr(), ws.numel(), _stream(dev)))
        else:                           # both gradients from one launch (the weight gradient's first traversal is the feature gradient)
            gw = torch.empty_like(w)
            ws = torch.empty(max(lib.bxi_tree_refine_backward_weight_workspace_bytes(B, C, V), 16), dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                _lib.check('bxi_tree_refine_backward_weight_f32', lib.bxi_tree_refine_backward_weight_f32(
                    g.data_ptr(), w.data_ptr(), si.data_ptr(), sp.data_ptr(), sc.data_ptr(), levels.data_ptr(), out.data_ptr(),
                    aggr.data_ptr(), aggr_up.data_ptr(), wsum.data_ptr(), wsum_up.data_ptr(), B, C, V, A, gw.data_ptr(), gf.data_ptr(),
                    ws.data_ptr(), ws.numel(), _stream(dev)))
            gw = gw.to(ctx.dtypes[1])
        return gf.to(ctx.dtypes[0]), gw, None, None, None, None, None


def refine(feature_in, edge_weight, sorted_index, sorted_parent, sorted_child, low_tree):
    """``refine(feature_in [B,C,V], edge_weight [B,V], sorted_index, sorted_parent, sorted_child, low_tree)`` (functions/refine.py)"""
    return _Refine.apply(feature_in, edge_weight, sorted_index, sorted_parent, sorted_child, low_tree,
                         _levels_of(sorted_index, sorted_parent))


class MinimumSpanningTree(torch.nn.Module):
    """``MinimumSpanningTree(distance_func)(guide_in [B,C,H,W], label=None)`` -> tree [B,H*W-1,2] (tree_filter.py:10-63)"""

    def __init__(self, distance_func):
        super().__init__()
        self.distance_func = distance_func

    @staticmethod
    def _grid_edges(fm):
        """vertical neighbours first, then horizontal ones (:15-26)"""
        B, H, W = fm.shape[0], fm.shape[2], fm.shape[3]
        ids = torch.arange(H * W, dtype=torch.int32, device=fm.device).view(H, W)
        down = torch.stack([ids[:-1, :], ids[1:, :]], 2).reshape(-1, 2)
        right = torch.stack([ids[:, :-1], ids[:, 1:]], 2).reshape(-1, 2)
        return torch.cat([down, right], 0).unsqueeze(0).expand(B, -1, -1)

    def _pair_distance(self, fm):
        B = fm.shape[0]
        d_down = self.distance_func(fm[:, :, :-1, :], fm[:, :, 1:, :]).reshape(B, -1)
        d_right = self.distance_func(fm[:, :, :, :-1], fm[:, :, :, 1:]).reshape(B, -1)
        return torch.cat([d_down, d_right], 1)

    def forward(self, guide_in, label=None):
        with torch.no_grad():
            index = self._grid_edges(guide_in)
            weight = self._pair_distance(guide_in) + 1                                   # _build_feature_weight (:28-35)
            if label is not None:                                                        # _build_label_weight (:37-52)
                B = label.shape[0]
                both = torch.cat([(label[:, :, :-1, :] + label[:, :, 1:, :]).sum(1).reshape(B, -1),
                                  (label[:, :, :, :-1] + label[:, :, :, 1:]).sum(1).reshape(B, -1)], 1)
                labelled = (self._pair_distance(label) * both) > 0
                weight[labelled] = torch.sigmoid(weight[labelled])
            return mst(index, weight, guide_in.shape[2] * guide_in.shape[3])


class TreeFilter2D(torch.nn.Module):
    """``TreeFilter2D(groups, sigma, distance_func, enable_log)(feature_in, embed_in, tree, low_tree=True)`` (:66-150)"""

    def __init__(self, groups=1, sigma=0.02, distance_func=None, enable_log=False):
        super().__init__()
        self.groups = groups
        self.enable_log = enable_log
        self.distance_func = self.norm2_distance if distance_func is None else distance_func
        self.sigma = sigma

    @staticmethod
    def norm2_distance(fm_ref, fm_tar):
        diff = fm_ref - fm_tar
        return (diff * diff).sum(dim=1)

    @staticmethod
    def batch_index_opr(data, index):
        idx = index.unsqueeze(1).expand(-1, data.shape[1], -1).long()
        return torch.gather(data, 2, idx)

    def build_edge_weight(self, fm, sorted_index, sorted_parent, low_tree):
        B, C = fm.shape[0], fm.shape[1]
        V = fm.shape[2] * fm.shape[3]
        src = self.batch_index_opr(fm.reshape(B, C, V), sorted_index)             # embedding of the node at every position
        tar = self.batch_index_opr(src, sorted_parent)                            # ... and of its parent
        src = src.reshape(-1, C // self.groups, V)
        tar = tar.reshape(-1, C // self.groups, V)
        d = self.distance_func(src, tar)
        return torch.exp(-d / self.sigma) if low_tree else torch.exp(-d)

    def forward(self, feature_in, embed_in, tree, low_tree=True):
        shape = feature_in.shape
        si, sp, sc, lv = _bfs_levels(tree, 4)
        edge_weight = self.build_edge_weight(embed_in, si, sp, low_tree)
        G = self.groups
        x = feature_in.reshape(shape[0] * G, shape[1] // G, -1).contiguous()
        if G > 1:                                 # split_group (:110-120): every group walks the same tree
            si, sp, sc, lv = (t.unsqueeze(1).expand(t.shape[0], G, *t.shape[1:]).reshape(-1, *t.shape[1:]).contiguous()
                              for t in (si, sp, sc, lv))
        out = _Refine.apply(x, edge_weight, si, sp, sc, low_tree, lv)
        return out.reshape(shape)
