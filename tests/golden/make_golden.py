#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the REFERENCE's own code.

Run in the build container only (it reads /root/reference; the GPU box has no such path):

    python tests/golden/make_golden.py

What is executed is the upstream source itself: the module-level functions
``compute_pairwise_term, dice_coefficient, compute_project_term, unfold_wo_center,
get_image_color_similarity, get_original_image`` and the ``CondInstMaskHead`` methods ``loss``,
``get_targets``, ``get_bitmasks_from_boxes`` are pulled out of
``mmdet/models/dense_heads/condinst_head.py`` by AST (oracle/reference_extract.py) and run on CPU torch.
Nothing of it is copied into this repository; only inputs/outputs are stored.

Third-party hooks the reference calls but which are NOT in its tree and not installable here are
supplied by the restatements in oracle/torch_oracle.py (so those two stay "parity unpinned"):
  tensor2imgs (mmcv)  -> torch_oracle.denormalize_u8       color.rgb2lab (scikit-image) -> torch_oracle.rgb2lab
``pairwise_nlog`` (the CUDA op; no CPU build exists) is bound to the reference's own pure-torch
``compute_pairwise_term``; the CUDA kernels themselves are run separately, on the CPU, through oracle/ref_wrap
(pairwise_refk.npz; likewise bfs.cu / refine.cu for the refk_* keys of tree_filter.npz -- the BFS order stored there
depends on thread arrival, so regenerating that file changes its refk_* arrays, consistently).

Fixtures (all small, float64 where the reference supports it):
  pairwise_f64.npz   logits, size/dilation cases -> compute_pairwise_term and its autograd gradient
  pairwise_refk.npz  the same op by the reference's own pairwise.cu kernels run on the CPU (oracle/ref_wrap), f32 + f64
  project_f64.npz    logits, bitmasks            -> compute_project_term and gradient
  similarity.npz     lab, mask                   -> get_image_color_similarity
  loss_cfg1.npz      BASELINE configs[0] (1x256x256, 4 boxes): full CondInstMaskHead.loss, f32
  loss_ragged.npz    2 ragged images, 2 instances per box, warm-up 0.37
  lab_kat.npz        textbook CIE-Lab known answers for the rgb2lab restatement
  dynamic_head_f64.npz  CondInstMaskHead.forward (+ parse_dynamic_params, aligned_bilinear) and autograd gradients
  training_sample.npz   CondInstMaskHead.training_sample (topk_per_img / max_proposals branches), inputs and selections
  simple_test.npz       CondInstMaskHead.simple_test (test-time masks per image and class)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from boxinstseg_amd import synthetic  # noqa: E402
from oracle import reference_extract as rx  # noqa: E402
from oracle import torch_oracle as to  # noqa: E402


def reference_namespace():
    holder = {}

    def tensor2imgs(tensor, mean, std, to_rgb):
        # returns BGR uint8 HxWx3 like mmcv (the caller flips back with [:, :, ::-1])
        rgb = to.denormalize_u8(tensor[0], tensor.shape[2:], mean, std, to_rgb)      # RGB [3,h,w] float
        return [np.ascontiguousarray(rgb.permute(1, 2, 0).numpy().astype(np.uint8)[:, :, ::-1])]

    ns = rx.load(extra_globals=dict(
        tensor2imgs=tensor2imgs,
        color=type('color', (), {'rgb2lab': staticmethod(to.rgb2lab)}),
        pairwise_nlog=lambda x, s, d: holder['ns'].compute_pairwise_term(x, s, d)))
    holder['ns'] = ns
    return ns


class StubHead:
    """The attributes CondInstMaskHead.loss / get_targets read (condinst_head.py:1100-1106)."""

    def __init__(self, ns, it, warm=10000, size=3, dil=2, thresh=0.3, bottom=10, stride=4):
        self.ns = ns
        self.boxinst_enabled = True
        self.bottom_pixels_removed = bottom
        self.out_stride = stride
        self.pairwise_size, self.pairwise_dilation, self.pairwise_color_thresh = size, dil, thresh
        self._iter = torch.tensor([float(it)])
        self._warmup_iters = warm

    def get_targets(self, *a):
        return self.ns.CondInstMaskHead_get_targets(self, *a)

    def get_bitmasks_from_boxes(self, *a):
        return self.ns.CondInstMaskHead_get_bitmasks_from_boxes(self, *a)


def run_reference_loss(ns, d, it):
    head = StubHead(ns, it)
    imgs = torch.from_numpy(d['imgs'])
    logits = torch.from_numpy(d['mask_logits']).requires_grad_(True)
    boxes = [torch.from_numpy(b) for b in d['gt_bboxes']]
    gi = torch.from_numpy(d['gt_inds'])
    out = ns.CondInstMaskHead_loss(head, imgs, d['img_metas'], logits, gi, boxes, None, None)
    (out['loss_prj'] + out['loss_pairwise']).backward()
    head2 = StubHead(ns, it)
    sims, bms, _ = head2.get_targets(boxes, None, imgs, d['img_metas'])
    return dict(loss_prj=out['loss_prj'].item(), loss_pairwise=out['loss_pairwise'].item(),
                grad=logits.grad.numpy()[:, 0], sim=np.stack([s[0].numpy() for s in sims]),
                bitmask=torch.cat(bms).numpy(), warmup=min((it + 1) / 10000.0, 1.0))


def pack_case(d):
    return dict(imgs=d['imgs'], mask_logits=d['mask_logits'], gt_inds=d['gt_inds'],
                boxes=np.concatenate(d['gt_bboxes']), gt_count=np.array([len(b) for b in d['gt_bboxes']]),
                img_shapes=np.array([m['img_shape'][:2] for m in d['img_metas']]),
                ori_shapes=np.array([m['ori_shape'][:2] for m in d['img_metas']]))


def main():
    assert rx.available(), 'needs /root/reference'
    ns = reference_namespace()
    rng = np.random.default_rng(20240925)

    # ---- pairwise term (reference: compute_pairwise_term == the CUDA op) --------------------------------
    out = {}
    for name, shape, size, dil in [('a', (3, 13, 17), 3, 2), ('b', (2, 10, 12), 3, 1), ('c', (2, 9, 11), 5, 2)]:
        x = torch.tensor(rng.standard_normal(shape) * 4.0, dtype=torch.float64)[:, None].requires_grad_(True)
        y = ns.compute_pairwise_term(x, size, dil)
        gp = torch.tensor(rng.standard_normal(tuple(y.shape)), dtype=torch.float64)
        y.backward(gp)
        out.update({f'{name}_logits': x.detach().numpy()[:, 0], f'{name}_size': size, f'{name}_dil': dil,
                    f'{name}_pairwise': y.detach().numpy(), f'{name}_gp': gp.numpy(),
                    f'{name}_grad': x.grad.numpy()[:, 0]})
    ext = torch.tensor([0, 1e-3, -1e-3, 3, -3, 30, -30, 100, -100], dtype=torch.float64)
    xe = (ext[None, :, None] + 0.5 * ext[None, None, :])[:, None].contiguous()
    out['ext_logits'] = xe.numpy()[:, 0]
    out['ext_pairwise'] = ns.compute_pairwise_term(xe, 3, 1).numpy()
    np.savez_compressed(os.path.join(HERE, 'pairwise_f64.npz'), **out)

    # ---- the same op by the reference's OWN pairwise.cu kernels, run on the CPU (oracle/_ref/libpairwise_ref.so) -------
    # (own generator: the draws of the fixtures around this block are untouched)
    from oracle import pairwise_ref as pr
    assert pr.available(), 'run `make -C oracle ref` first (compiles the reference pairwise.cu kernels where they lie)'
    rk = np.random.default_rng(20240926)
    out = {}
    for name, (dt, size, dil, shape) in {'f32_3_2': (np.float32, 3, 2, (3, 1, 19, 27)), 'f64_3_2': (np.float64, 3, 2, (2, 1, 16, 21)),
                                         'f32_5_1': (np.float32, 5, 1, (2, 1, 13, 11)), 'f64_3_1': (np.float64, 3, 1, (1, 1, 9, 70))}.items():
        lg = (3.0 * rk.standard_normal(shape)).astype(dt)
        lg[0, 0, 0, :6] = [-90.0, 90.0, -21.0, 21.0, -45.0, 0.0]              # both branches of _logsig, saturation
        pw = pr.forward(lg, size, dil)
        gp = rk.standard_normal(pw.shape).astype(dt)
        out.update({f'{name}_logits': lg, f'{name}_size': size, f'{name}_dil': dil, f'{name}_pairwise': pw, f'{name}_gp': gp,
                    f'{name}_grad': pr.backward(lg, pw, gp, size, dil)})
    np.savez_compressed(os.path.join(HERE, 'pairwise_refk.npz'), **out)

    # ---- projection term ---------------------------------------------------------------------------------
    x = torch.tensor(rng.standard_normal((3, 1, 12, 15)) * 2.0, dtype=torch.float64, requires_grad=True)
    bm = torch.zeros((3, 1, 12, 15), dtype=torch.float64)
    bm[0, 0, 2:9, 3:11] = 1
    bm[1, 0, 0:4, 0:15] = 1
    loss = ns.compute_project_term(x.sigmoid(), bm)      # instance 2: empty box
    loss.backward()
    np.savez_compressed(os.path.join(HERE, 'project_f64.npz'), logits=x.detach().numpy()[:, 0], bitmask=bm.numpy()[:, 0],
                        loss=loss.item(), grad=x.grad.numpy()[:, 0])

    # ---- colour similarity ------------------------------------------------------------------------------------
    lab = torch.tensor(rng.uniform(-40, 90, size=(1, 3, 9, 13)), dtype=torch.float32)
    lab[0, :, 4:, 6:] = lab[0, :, 4:5, 6:7]              # a flat region: similarities of exactly 1
    mask = torch.ones((9, 13))
    mask[-2:, :] = 0
    sim = ns.get_image_color_similarity(lab, mask, 3, 2)
    sim5 = ns.get_image_color_similarity(lab, mask, 5, 1)
    np.savez_compressed(os.path.join(HERE, 'similarity.npz'), lab=lab.numpy()[0], mask=mask.numpy(), sim_3_2=sim.numpy()[0],
                        sim_5_1=sim5.numpy()[0])

    # ---- full loss: BASELINE configs[0] and a ragged two-image batch ---------------------------------------------
    d = synthetic.cfg1(0)
    r = run_reference_loss(ns, d, it=10000)
    np.savez_compressed(os.path.join(HERE, 'loss_cfg1.npz'), **pack_case(d), **r)
    d = synthetic.make_batch(B=2, H=96, W=160, boxes_per_img=3, inst_per_box=2, seed=7,
                             img_shapes=[(96, 131), (70, 160)], ori_shapes=[(48, 66), (210, 480)],
                             min_box=16, max_box=80)
    r = run_reference_loss(ns, d, it=3699)
    np.savez_compressed(os.path.join(HERE, 'loss_ragged.npz'), **pack_case(d), **r)

    # ---- producer of the logits: CondInstMaskHead.forward (reference source) + autograd --------------------------------
    out = {}
    for name, C, no_rel, fac, (B, H, W, N) in [('a', 16, False, 2, (2, 9, 13, 5)), ('b', 8, True, 4, (1, 6, 7, 3)),
                                                ('c', 16, False, 1, (2, 5, 34, 4))]:
        class S:
            pass
        st = S()
        st.in_stride, st.out_stride, st.disable_rel_coors, st.dynamic_convs, st.dynamic_channels = 8, 8 // fac, no_rel, 3, 8
        cin = C if no_rel else C + 2
        st.dy_weights, st.dy_biases = [cin * 8, 64, 8], [8, 8, 1]
        st.sizes_of_interest = torch.tensor([64, 128, 256, 512, 1024])
        st.parse_dynamic_params = lambda p, st=st: ns.CondInstMaskHead_parse_dynamic_params(st, p)
        feat = torch.tensor(rng.standard_normal((B, C, H, W)), dtype=torch.float64, requires_grad=True)
        params = torch.tensor(rng.standard_normal((N, sum(st.dy_weights) + 17)) * 0.4, dtype=torch.float64, requires_grad=True)
        coors = torch.tensor(rng.uniform(0, 8 * W, size=(N, 2)), dtype=torch.float64)
        lvl = torch.tensor(rng.integers(0, 5, size=N))
        img = torch.tensor(rng.integers(0, B, size=N))
        y = ns.CondInstMaskHead_forward(st, feat, params, coors, lvl, img)
        g = torch.tensor(rng.standard_normal(tuple(y.shape)), dtype=torch.float64)
        y.backward(g)
        out.update({f'{name}_feat': feat.detach().numpy(), f'{name}_params': params.detach().numpy(), f'{name}_coors': coors.numpy(),
                    f'{name}_level': lvl.numpy(), f'{name}_img': img.numpy(), f'{name}_cfg': np.array([C, int(no_rel), fac]),
                    f'{name}_logits': y.detach().numpy(), f'{name}_g': g.numpy(), f'{name}_gfeat': feat.grad.numpy(),
                    f'{name}_gparams': params.grad.numpy()})
    np.savez_compressed(os.path.join(HERE, 'dynamic_head_f64.npz'), **out)

    # ---- the callers either side of the path: training_sample and simple_test, the reference's own methods -------------
    # (own generator: the draws of the fixtures around this block are untouched)
    cns = rx.load(methods=('training_sample', 'simple_test', 'forward', 'parse_dynamic_params'), functions=('aligned_bilinear',))
    rk = np.random.default_rng(20240927)
    out = {}
    for name, (topk, maxp) in {'topk64': (64, -1), 'topk8': (8, -1), 'topk3': (3, -1), 'maxp': (-1, 20)}.items():
        B, levels, Ccls, P = 2, [(8, 10), (4, 5), (2, 3)], 5, 7
        cls = [rk.standard_normal((B, Ccls, h, w)).astype(np.float32) for h, w in levels]
        ctr = [rk.standard_normal((B, 1, h, w)).astype(np.float32) for h, w in levels]
        par = [rk.standard_normal((B, P, h, w)).astype(np.float32) for h, w in levels]
        n = sum(B * h * w for h, w in levels)
        coors = rk.standard_normal((n, 2)).astype(np.float32)
        lvl = rk.integers(0, 3, size=n)
        img = np.concatenate([np.repeat(np.arange(B), h * w) for h, w in levels])
        gt = rk.integers(0, 6, size=n)
        gt = np.where(rk.random(n) < 0.35, -1, gt + 6 * img)             # indices into the batch-concatenated box list
        class S:
            pass
        st = S()
        st.max_proposals, st.topk_per_img = maxp, topk
        torch.manual_seed(7)                                             # the max_proposals branch draws a randperm
        got = cns.CondInstMaskHead_training_sample(st, [torch.from_numpy(a) for a in cls], [torch.from_numpy(a) for a in ctr],
                                                   [torch.from_numpy(a) for a in par], torch.from_numpy(coors), torch.from_numpy(lvl),
                                                   torch.from_numpy(img), torch.from_numpy(gt))
        out.update({f'{name}_cfg': np.array([topk, maxp]), f'{name}_coors': coors, f'{name}_lvl': lvl, f'{name}_img': img, f'{name}_gt': gt})
        for i, (a, b, c) in enumerate(zip(cls, ctr, par)):
            out.update({f'{name}_cls{i}': a, f'{name}_ctr{i}': b, f'{name}_par{i}': c})
        for key, t in zip(('o_params', 'o_coors', 'o_lvl', 'o_img', 'o_gt'), got):
            out[f'{name}_{key}'] = t.numpy()
    np.savez_compressed(os.path.join(HERE, 'training_sample.npz'), **out)

    out = {}
    for name, C, rescale in (('a', 16, False), ('b', 8, True)):
        class S:
            pass
        st = S()
        st.in_stride, st.out_stride, st.disable_rel_coors, st.dynamic_convs, st.dynamic_channels = 8, 4, False, 3, 8
        st.dy_weights, st.dy_biases = [(C + 2) * 8, 64, 8], [8, 8, 1]
        st.sizes_of_interest = torch.tensor([64, 128, 256, 512, 1024])
        st.parse_dynamic_params = lambda p, st=st: cns.CondInstMaskHead_parse_dynamic_params(st, p)
        st.forward = lambda *a, st=st: cns.CondInstMaskHead_forward(st, *a)
        B, H, W, counts, ncls = 2, 6, 9, [3, 2], 4
        feat = rk.standard_normal((B, C, H, W)).astype(np.float32)
        params = [(rk.standard_normal((c, sum(st.dy_weights) + 17)) * 0.6).astype(np.float32) for c in counts]
        coors = [rk.uniform(0, 8 * W, size=(c, 2)).astype(np.float32) for c in counts]
        lvls = [rk.integers(0, 5, size=c) for c in counts]
        labels = [rk.integers(0, ncls, size=c) for c in counts]
        metas = [dict(img_shape=(41, 70, 3), ori_shape=(60, 101, 3)), dict(img_shape=(48, 72, 3), ori_shape=(33, 50, 3))]
        res = cns.CondInstMaskHead_simple_test(st, torch.from_numpy(feat), [torch.from_numpy(a) for a in labels],
                                               [torch.from_numpy(a) for a in params], [torch.from_numpy(a) for a in coors],
                                               [torch.from_numpy(a) for a in lvls], metas, ncls, rescale=rescale)
        out.update({f'{name}_feat': feat, f'{name}_rescale': np.array(int(rescale)), f'{name}_ncls': np.array(ncls),
                    f'{name}_shapes': np.array([[m['img_shape'][:2], m['ori_shape'][:2]] for m in metas])})
        for i in range(B):
            out.update({f'{name}_params{i}': params[i], f'{name}_coors{i}': coors[i], f'{name}_lvl{i}': lvls[i], f'{name}_labels{i}': labels[i]})
            for c in range(ncls):
                out[f'{name}_masks{i}_{c}'] = np.asarray(res[i][c], np.uint8)
    np.savez_compressed(os.path.join(HERE, 'simple_test.npz'), **out)

    # ---- DiscoBox MeanField / dice_loss / mil_loss, the reference's own class and functions --------
    dns = rx.load_discobox()
    out = {}
    for name, (H, W, n, ks, iters, base, use_inter) in {'a': (24, 40, 4, 3, 10, 0.10, False),
                                                        'b': (33, 70, 5, 3, 6, 0.45, True),
                                                        'c': (20, 131, 3, 5, 4, 0.10, False)}.items():
        yy, xx = np.mgrid[0:H, 0:W]
        feat = np.stack([np.sin(xx / 7.0) + 0.3 * np.cos(yy / 5.0), np.cos(xx / 9.0 + yy / 11.0), 0.5 * np.sin(yy / 4.0)])
        feat = (feat + 0.05 * rng.standard_normal(feat.shape)).astype(np.float32)
        mf = dns.MeanField(torch.from_numpy(feat)[None], alpha0=2.0, theta0=0.5, theta1=30.0, theta2=20.0, iter=iters,
                           kernel_size=ks, base=base)
        x = torch.tensor(rng.uniform(0, 1, size=(n, 1, H, W)), dtype=torch.float32)
        t = torch.zeros(n, 1, H, W)
        for i in range(n):
            r0, c0 = int(rng.integers(0, H // 2)), int(rng.integers(0, W // 2))
            t[i, 0, r0:r0 + int(rng.integers(4, H // 2 + 1)), c0:c0 + int(rng.integers(4, W // 2 + 1))] = 1
        t[n - 1] = 0                                   # an instance without target
        inter = torch.tensor(rng.uniform(0, 30, size=(n, 2, H, W)), dtype=torch.float32) if use_inter else None
        ret, valid = mf(x, t, inter)
        out.update({f'{name}_feat': feat, f'{name}_kernel': mf.kernel[0, 0].reshape(ks * ks, H, W).numpy(),
                    f'{name}_x': x[:, 0].numpy(), f'{name}_t': t[:, 0].numpy().astype(np.uint8),
                    f'{name}_cfg': np.array([ks, iters, base, 2.0, 0.5, 30.0, mf.gamma]), f'{name}_ret': ret[:, 0].numpy(),
                    f'{name}_valid': valid.numpy()})
        if inter is not None:
            out[f'{name}_inter'] = inter.numpy()
        inp = torch.tensor(rng.uniform(0, 1, size=(n, H, W)), dtype=torch.float32, requires_grad=True)
        l = dns.mil_loss(dns.dice_loss, inp, inp, t[:, 0].byte())
        gl = torch.tensor(rng.uniform(0.5, 1.5, size=n), dtype=torch.float32)
        (l * gl).sum().backward()
        inp2 = torch.tensor(rng.uniform(0, 1, size=(n, H, W)), dtype=torch.float32, requires_grad=True)
        d = dns.dice_loss(inp2 * t[:, 0], ret[:, 0])
        (d * gl).sum().backward()
        out.update({f'{name}_mil_in': inp.detach().numpy(), f'{name}_mil_loss': l.detach().numpy(), f'{name}_gl': gl.numpy(),
                    f'{name}_mil_grad': inp.grad.numpy(), f'{name}_dice_in': inp2.detach().numpy(),
                    f'{name}_dice_loss': d.detach().numpy(), f'{name}_dice_grad': inp2.grad.numpy()})
    np.savez_compressed(os.path.join(HERE, 'discobox.npz'), **out)

    # ---- BoxProjectionLoss / LevelsetLoss / LCM, the reference's own classes under autograd -----------
    lns = rx.load_levelset()
    out = {}
    for name, (N, H, W, C) in {'a': (3, 17, 23, 3), 'b': (4, 40, 72, 2), 'c': (1, 96, 96, 3)}.items():
        s = torch.tensor(rng.uniform(0, 1, (N, 1, H, W)), dtype=torch.float64, requires_grad=True)
        box = np.zeros((N, 1, H, W))
        for i in range(N):
            r0, c0 = int(rng.integers(0, H // 2)), int(rng.integers(0, W // 2))
            box[i, 0, r0:r0 + int(rng.integers(3, H // 2 + 1)), c0:c0 + int(rng.integers(3, W // 2 + 1))] = 1
        soft = torch.tensor(box * rng.uniform(0.3, 1.0, box.shape), dtype=torch.float64)      # interpolated masks are not binary
        gl = torch.tensor(rng.uniform(0.5, 1.5, N), dtype=torch.float64)
        l = lns.BoxProjectionLoss(loss_weight=1.3)(s, soft)
        (l * gl).sum().backward()
        out.update({f'{name}_scores': s.detach().numpy(), f'{name}_bitmask': soft.numpy(), f'{name}_gl': gl.numpy(),
                    f'{name}_prj_loss': l.detach().numpy(), f'{name}_prj_grad': s.grad.numpy()})
        ms = torch.tensor(rng.uniform(0, 1, (N, 2, H, W)) * box, dtype=torch.float64, requires_grad=True)
        T = torch.tensor(rng.uniform(-1, 1, (N, C, H, W)), dtype=torch.float64, requires_grad=True)
        if name == 'a':
            with torch.no_grad():
                ms[0, 1] = 0.0                      # a region without score: the clamp of :34-35 is active
        pn = torch.tensor(np.maximum(box.sum((1, 2, 3)), 1.0))
        l = lns.LevelsetLoss(loss_weight=0.7)(ms, T, pn)
        (l * gl).sum().backward()
        out.update({f'{name}_ms': ms.detach().numpy(), f'{name}_T': T.detach().numpy(), f'{name}_pn': pn.numpy(),
                    f'{name}_lst_loss': l.detach().numpy(), f'{name}_lst_gms': ms.grad.numpy(), f'{name}_lst_gT': T.grad.numpy()})
        yy, xx = np.mgrid[0:H, 0:W]
        img = np.stack([np.sin(xx / 5.0 + i) + 0.2 * rng.standard_normal((3, H, W))[0] for i in range(N)])[:, None]
        img = np.concatenate([img, np.cos(yy / 4.0)[None, None].repeat(N, 0) + 0.1 * rng.standard_normal((N, 1, H, W)),
                              0.3 * rng.standard_normal((N, 1, H, W))], 1).astype(np.float32)
        phi = torch.tensor(rng.uniform(0, 1, (N, 1, H, W)), dtype=torch.float32, requires_grad=True)
        bx = torch.tensor(box, dtype=torch.float32)
        lcm = lns.LocalConsistencyModule(num_iter=10, dilations=[2])
        ref = lcm(torch.from_numpy(img), phi)
        l = lns.LCM(torch.from_numpy(img), phi, bx)
        l.backward()
        out.update({f'{name}_img': img, f'{name}_phi': phi.detach().numpy(), f'{name}_refined': ref.detach().numpy(),
                    f'{name}_lcm_loss': np.array(float(l.detach())), f'{name}_lcm_grad': phi.grad.numpy(), f'{name}_box': box.astype(np.float32)})
    np.savez_compressed(os.path.join(HERE, 'levelset.npz'), **out)

    # ---- tree_filter -- minimum spanning trees by the reference's own boruvka.cpp (oracle/_ref) ----------
    from oracle import tree_filter_oracle as tfo
    assert tfo.ref_available(), 'run `make -C oracle ref` first (compiles the reference boruvka.cpp where it lies)'
    out = {}
    for name, (H, W, quant) in {'a': (12, 17, 0.0), 'b': (31, 24, 0.5), 'c': (96, 96, 0.0), 'd': (5, 90, 0.25)}.items():
        yy, xx = np.mgrid[0:H, 0:W]
        fm = np.stack([np.sin(xx / 6.0) + 0.3 * np.cos(yy / 5.0), np.cos(xx / 9.0 + yy / 7.0), 0.4 * np.sin(yy / 3.0)])
        fm = (fm + 0.15 * rng.standard_normal(fm.shape)).astype(np.float32)
        if quant:
            fm = (np.round(fm / quant) * quant).astype(np.float32)      # many equal weights: ties go to the smaller edge index
        idx = tfo.grid_edges(H, W)
        wt = tfo.grid_weights(fm)
        out.update({f'{name}_fm': fm, f'{name}_tree': tfo.ref_boruvka_mst(idx, wt, H * W)})
    # bfs + refine by the reference's own CUDA kernels run on the CPU (oracle/_ref/libtreekernels_ref.so): its BFS order,
    # the five tensors of refine_forward and both gradients, on the trees above (own generator: the draws above are untouched)
    assert tfo.ref_kernels_available(), 'run `make -C oracle ref` first'
    rk = np.random.default_rng(20240925)
    for name, C in (('a', 3), ('b', 2), ('d', 1)):
        tree = out[f'{name}_tree']
        V = tree.shape[0] + 1
        si, sp, sc = tfo.ref_bfs(tree, V, 4)
        x = rk.standard_normal((C, V)).astype(np.float32)
        w = np.exp(-rk.random(V) * (3.0 if name == 'b' else 0.5)).astype(np.float32)
        g = rk.standard_normal((C, V)).astype(np.float32)
        fwd = tfo.ref_refine_forward(x, w, si, sp, sc)
        gf, gw = tfo.ref_refine_backward(g, w, si, sp, sc, fwd)
        out.update({f'refk_{name}_si': si, f'refk_{name}_sp': sp, f'refk_{name}_sc': sc, f'refk_{name}_x': x, f'refk_{name}_w': w,
                    f'refk_{name}_g': g, f'refk_{name}_gf': gf, f'refk_{name}_gw': gw})
        out.update({f'refk_{name}_{k}': v for k, v in fwd.items()})
    np.savez_compressed(os.path.join(HERE, 'tree_filter.npz'), **out)

    # ---- Lab known answers (published CIE values) -------------------------------------------------------
    rgb = np.array([[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 200, 77]], np.uint8)
    want = np.array([[100.0, -0.0025, 0.0047], [0, 0, 0], [53.2406, 80.0923, 67.2028], [87.7351, -86.1830, 83.1797],
                     [32.2957, 79.1856, -107.8573], [70.8063, -66.6255, 48.8599]])
    np.savez_compressed(os.path.join(HERE, 'lab_kat.npz'), rgb=rgb, lab=want)
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print(f'{f:20s} {os.path.getsize(os.path.join(HERE, f)) / 1024:8.1f} KiB')


if __name__ == '__main__':
    main()
