#!/usr/bin/env python3
"""Developer measurement of the tree_filter extension on the GPU box: Box2Mask's call pattern
(box2mask_head.py:269-322): MST of B images at 96x96, two TreeFilter2D passes over N instance maps, forward+backward."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as entry
entry.build()
from boxinstseg_amd import MinimumSpanningTree, TreeFilter2D
from oracle import tree_filter_oracle as tfo

dev = torch.device('cuda:0')


def ev(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


res = {}
g = torch.Generator().manual_seed(0)
H = W = 96
mstm = MinimumSpanningTree(TreeFilter2D.norm2_distance); tf = TreeFilter2D()
for B, N in ((2, 16), (2, 100)):
    img = torch.rand(B, 3, H, W, generator=g).to(dev); lst = torch.rand(B, 8, H, W, generator=g).to(dev)
    rep = torch.arange(N) % B
    pred = torch.rand(N, 1, H, W, generator=g).to(dev).requires_grad_(True)
    def trees(): return mstm(img), mstm(lst)
    t_img, t_lst = trees()
    ti, tl = t_img[rep], t_lst[rep]
    imgs_n, lst_n = img[rep], lst[rep]
    def filt():
        a = tf(feature_in=pred, embed_in=imgs_n, tree=ti)
        b = tf(a, lst_n, tl, low_tree=False)
        (a.sum() + b.sum()).backward(); pred.grad = None
    # the reference's MST runs on the host: time its own boruvka.cpp (oracle/_ref) on one 96x96 graph if it is there
    cpu_ms = None
    if tfo.ref_available():
        idx = tfo.grid_edges(H, W); wt = tfo.grid_weights(img[0].cpu().numpy())
        t0 = time.perf_counter(); tfo.ref_boruvka_mst(idx, wt, H * W); cpu_ms = (time.perf_counter() - t0) * 1e3
    res[f'B{B}_N{N}'] = dict(hip_two_msts_us=ev(trees), hip_two_filters_fwd_bwd_us=ev(filt),
                             reference_boruvka_cpp_one_graph_ms_on_host=cpu_ms)
# BoxLevelSet's size (box_solov2_head.py:354-358): one 200x304 mask-feature map, 5 channels -- the global-workspace kernels
H2, W2 = 200, 304
img2 = torch.rand(1, 3, H2, W2, generator=g).to(dev)
feat2 = torch.rand(1, 5, H2, W2, generator=g).to(dev).requires_grad_(True)
t2 = mstm(img2)
def filt2():
    o = tf(feature_in=feat2, embed_in=img2, tree=t2)
    o.sum().backward(); feat2.grad = None
cpu_ms = None
if tfo.ref_available():
    idx = tfo.grid_edges(H2, W2); wt = tfo.grid_weights(img2[0].cpu().numpy())
    t0 = time.perf_counter(); tfo.ref_boruvka_mst(idx, wt, H2 * W2); cpu_ms = (time.perf_counter() - t0) * 1e3
res['large_200x304_C5'] = dict(hip_mst_us=ev(lambda: mstm(img2), n=5, warm=1), hip_filter_fwd_bwd_us=ev(filt2, n=5, warm=1),
                               reference_boruvka_cpp_one_graph_ms_on_host=cpu_ms)
print(json.dumps(res, indent=1))
