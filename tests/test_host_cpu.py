"""CPU: the host side of the drop-in (no kernel is launched here).

* libboxinst_hip.so loads and exports every symbol include/boxinst_hip.h declares;
* argument validation of the C ABI that needs no device (status codes, workspace sizing);
* the Python mirror of the reference interface: constructor keywords, state-dict keys, registry,
  config loading, and the loud failure on CPU tensors (there is no fallback)."""
import ctypes as C
import os
import sys
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_CFG = '/root/reference/configs/boxinst/boxinst_r50_fpn_1x_coco.py'


@pytest.fixture(scope='module', autouse=True)
def _built(built):
    return built


def header_symbols():
    import glob
    names = set()
    for path in sorted(glob.glob(os.path.join(ROOT, 'include', '*.h'))):       # the production ABI and the developer hooks
        src = open(path).read()
        src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
        names |= set(re.findall(r'\b(bxi_[a-z0-9_]+)\s*\(', src))
    return sorted(names - {'bxi_launch_hook'})


def test_library_exports_every_declared_symbol():
    from boxinstseg_amd import _lib
    lib = _lib.load()
    names = header_symbols()
    assert len(names) >= 16
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/boxinst_hip.h but not exported'
        assert n in _lib.SIGNATURES, f'{n} has no ctypes signature'
    assert sorted(_lib.SIGNATURES) == names
    assert lib.bxi_abi_version() == _lib.BXI_ABI_VERSION == 7
    for code, name in _lib.STATUS.items():
        assert _lib.status_string(code) and 'unknown' not in _lib.status_string(code)
    assert 'unknown' in _lib.status_string(-99)


def test_abi_argument_validation_without_device():
    from boxinstseg_amd import _lib
    lib = _lib.load()
    # shapes / arguments are validated before anything touches a device
    assert lib.bxi_pairwise_nlog_forward_f32(None, -1, 4, 4, 3, 2, None, None) == -2      # BAD_SHAPE
    assert lib.bxi_pairwise_nlog_forward_f32(None, 1, 4, 4, 4, 2, None, None) == -3       # even window
    assert lib.bxi_pairwise_nlog_forward_f32(None, 1, 4, 4, 3, 0, None, None) == -3       # dilation < 1
    assert lib.bxi_pairwise_nlog_forward_f32(None, 1, 4, 4, 3, 2, None, None) == -1       # NULL pointers
    assert lib.bxi_pairwise_nlog_forward_f32(None, 0, 4, 4, 3, 2, None, None) == 0        # N == 0: no-op
    # the targets-ahead entry point (ABI 6): NULL pointers, an even window, a window the fused path is not built for -- before anything touches a device
    assert lib.bxi_boxinst_targets_f32(None, None, None, 4, 3, 2, 0.3, None, 0, None) == -1
    b = _lib.ImageBatch(); b.B, b.Hc, b.Wc = 1, 64, 64
    import ctypes as C
    ptrs, cnt = (C.c_void_p * 1)(0), (C.c_int * 1)(0)
    assert lib.bxi_boxinst_targets_f32(C.byref(b), ptrs, cnt, 4, 4, 2, 0.3, None, 0, None) == -3       # even window
    assert lib.bxi_boxinst_targets_f32(C.byref(b), ptrs, cnt, 4, 5, 2, 0.3, None, 0, None) == -4       # 5 x 5: composed from the op-level kernels instead
    assert lib.bxi_boxinst_targets_f32(C.byref(b), ptrs, cnt, 4, 3, 2, 0.3, None, 0, None) == -5       # no workspace
    # the evaluation's workspace now carries the targets regions (box table + per-box pair counts: ~1 MB) whatever N
    assert lib.bxi_boxinst_eval_workspace_bytes(2, 800, 1024, 4, 32) > 1024 * 8 * 128
    assert lib.bxi_boxinst_loss_workspace_bytes(32, 200, 256) > 32 * 25 * 256 * 5
    assert lib.bxi_boxinst_loss_workspace_bytes(-1, 200, 256) == 0
    assert lib.bxi_boxinst_loss_state_bytes(32, 200, 256) >= 32 * 456 * 8
    assert lib.bxi_boxinst_eval_workspace_bytes(2, 800, 1024, 4, 32) > 2 * 3 * 200 * 256 * 4
    assert lib.bxi_box_bitmasks_f32(None, None, 65, 64, 64, 4, 2, None, None) == -2       # > BXI_MAX_IMAGES
    # the "next" rows (dynamic head, DiscoBox, Box2Mask losses, tree_filter): same contract
    assert lib.bxi_dynamic_mask_forward_f32(None, 1, 12, 4, 4, None, 1, None, None, None, None, 5, 8, 2, 0, None, None) == -4   # C not in {8,16}
    assert lib.bxi_meanfield_kernel_f32(None, 1, 3, 4, 4, 4, 2.0, 0.5, 30.0, None, None) == -4       # even kernel
    assert lib.bxi_meanfield_forward_f32(None, 1, 4, 4, 3, None, None, 0, None, 1, 10, 0.7, None, 0.01, None, None, None, 0, None) == -3   # base >= 0.5
    assert lib.bxi_meanfield_workspace_bytes(2, 10, 130) == 3 * 8 * 2 * 10 * 3
    assert lib.bxi_mil_loss_forward_f32(None, None, 0, 1, 4, 4, None, None, None) == -1
    # tree_filter workspaces: none where the graph fits LDS; beyond, the Euler-tour BFS needs two 64-bit words per arc slot (4 per vertex)
    # up to 2^20 - 1 vertices, the level walk beyond that far less; refine: two record buffers + parents + the doubling pass's arrays
    V = 200 * 304
    assert lib.bxi_bfs_workspace_bytes(2, 96 * 96) == 0 and lib.bxi_tree_refine_workspace_bytes(2, 5, 96 * 96) == 0
    assert lib.bxi_bfs_workspace_bytes(1, V) >= 64 * V and lib.bxi_bfs_workspace_bytes(2, V) == 2 * lib.bxi_bfs_workspace_bytes(1, V)
    assert lib.bxi_bfs_workspace_bytes(1, 1 << 21) < 64 * (1 << 21)
    assert lib.bxi_tree_refine_workspace_bytes(2, 5, V) >= 2 * 5 * V * (16 + 16 + 4 + 12)
    assert lib.bxi_levelset_loss_forward_f32(None, None, None, 1, 5000, 4, 4, 1.0, None, None, None) == -4         # C > 4096
    assert lib.bxi_levelset_state_bytes(3, 2) == 8 * 3 * 10 * 9
    assert lib.bxi_lcm_refine_f32(None, None, 1, 4, 4, 0, 10, 0, None, None, 0, None) == -3                         # dilation < 1
    assert lib.bxi_mst_forward_i32(None, None, 1, 10, 20000, None, None, 0, None) == -1                             # large graphs are served (workspace arrays): NULL pointers
    assert lib.bxi_mst_forward_i32(None, None, 1, 10, (1 << 24) + 1, None, None, 0, None) == -4
    assert lib.bxi_mst_workspace_bytes(2, 18240, 9216) == 16 and lib.bxi_mst_workspace_bytes(1, 121096, 60800) > 16 * 60800
    assert lib.bxi_bfs_workspace_bytes(2, 9216) == 0 and lib.bxi_bfs_workspace_bytes(1, 60800) >= 32 * 60800
    assert lib.bxi_tree_refine_workspace_bytes(2, 8, 9216) == 0 and lib.bxi_tree_refine_workspace_bytes(1, 8, 60800) >= 8 * 36 * 60800
    assert lib.bxi_bfs_forward_i32(None, 1, 1, 4, None, None, None, None, None, 0, None) == -2
    assert lib.bxi_tree_refine_backward_weight_workspace_bytes(2, 3, 100) == 4 * 4 * 2 * 3 * 100
    if not torch.cuda.is_available():
        assert lib.bxi_check_device(0) == -7                                               # NO_DEVICE


def test_head_constructor_and_state_dict_match_reference_contract():
    import boxinstseg_amd as bx
    kw = dict(in_channels=16, in_stride=8, out_stride=4, dynamic_convs=3, dynamic_channels=8, disable_rel_coors=False,
              bbox_head_channels=256, sizes_of_interest=[64, 128, 256, 512, 1024], max_proposals=-1, topk_per_img=64,
              boxinst_enabled=True, bottom_pixels_removed=10, pairwise_size=3, pairwise_dilation=2,
              pairwise_color_thresh=0.3, pairwise_warmup=10000)           # configs/boxinst/boxinst_r50_fpn_1x_coco.py:54-71
    head = bx.build_head(dict(type='CondInstMaskHead', **kw))
    assert isinstance(head, bx.CondInstMaskHead)
    sd = head.state_dict()
    assert set(sd) == {'sizes_of_interest', '_iter', 'param_conv.weight', 'param_conv.bias'}
    assert tuple(sd['param_conv.weight'].shape) == (233, 256, 3, 3) and tuple(sd['_iter'].shape) == (1,)
    assert head.num_gen_params == 233 and head.dy_weights == [144, 64, 8] and head.dy_biases == [8, 8, 1]
    # _iter survives a checkpoint round trip and drives the warm-up (condinst_head.py:1330-1331)
    head._iter.fill_(2500.0)
    other = bx.build_head(dict(type='CondInstMaskHead', **kw))
    other.load_state_dict(head.state_dict())
    assert abs(other._tick() - 0.2501) < 1e-7 and float(other._iter) == 2501.0
    # (the fused evaluation counts and ramps on the device instead -- tests/test_gpu_parity.py; _tick is the reference's own += 1 / .item())
    other._iter.fill_(10.0)
    assert abs(other._tick() - 0.0011) < 1e-9 and float(other._iter) == 11.0
    assert other._counts_in_evaluation(torch.zeros(1)) and not other._counts_in_evaluation(torch.zeros(1, device='meta'))
    with pytest.raises(AssertionError):
        bx.CondInstMaskHead(max_proposals=500, topk_per_img=64)
    with pytest.raises(KeyError):
        bx.build_head(dict(type='NoSuchHead'))


@pytest.mark.parametrize('case', ['topk64', 'topk8', 'topk3', 'maxp'])
def test_training_sample_selects_what_the_reference_selects(case):
    """CondInstMaskHead.training_sample (sorts, no per-box Python loop) == the reference's own method (AST-extracted and
    run on the same inputs by tests/golden/make_golden.py): same instances in the same order, both branches."""
    import boxinstseg_amd as bx
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'training_sample.npz'))
    topk, maxp = (int(v) for v in g[f'{case}_cfg'])
    head = bx.CondInstMaskHead(in_channels=16, max_proposals=maxp, topk_per_img=topk)
    t = lambda k: torch.from_numpy(g[f'{case}_{k}'])
    torch.manual_seed(7)
    got = head.training_sample([t(f'cls{i}') for i in range(3)], [t(f'ctr{i}') for i in range(3)], [t(f'par{i}') for i in range(3)],
                               t('coors'), t('lvl'), t('img'), t('gt'))
    for key, a in zip(('o_params', 'o_coors', 'o_lvl', 'o_img', 'o_gt'), got):
        assert np.array_equal(a.numpy(), g[f'{case}_{key}']), key
    # no positive location at all: empty selections, no exception
    none = head.training_sample([t(f'cls{i}') for i in range(3)], [t(f'ctr{i}') for i in range(3)], [t(f'par{i}') for i in range(3)],
                                t('coors'), t('lvl'), t('img'), torch.full_like(t('gt'), -1))
    assert none[0].shape == (0, 7) and none[4].numel() == 0


@pytest.mark.parametrize('seed', list(range(40)))
def test_topk_per_box_against_loops(seed):
    """the sort-based selection of training_sample against a plain-loop statement of the rule (condinst_head.py:1201-1225):
    per image, per box in ascending index, keep max(int(topk / boxes_in_image), 1) locations -- all of them in location
    order if the box has no more than that, otherwise the best by score, best first."""
    from boxinstseg_amd.mask_head import _topk_per_box
    rng = np.random.default_rng(500 + seed)
    n = int(rng.integers(0, 120)); B = int(rng.integers(1, 4)); G = int(rng.integers(1, 9)); topk = int(rng.choice([1, 2, 5, 16, 64]))
    img = np.sort(rng.integers(0, B, size=n))
    gt = rng.integers(0, G, size=n) + G * img
    score = rng.permutation(n).astype(np.float32) / max(n, 1)           # distinct scores: no ties
    want = []
    for b in range(B):
        idx_b = np.nonzero(img == b)[0]
        boxes = np.unique(gt[idx_b])
        if len(boxes) == 0:
            continue
        quota = max(int(topk / len(boxes)), 1)
        for g_ in boxes:
            idx = idx_b[gt[idx_b] == g_]
            if len(idx) > quota:
                idx = idx[np.argsort(-score[idx], kind='stable')[:quota]]
            want.extend(idx.tolist())
    got = _topk_per_box(torch.from_numpy(img), torch.from_numpy(gt), torch.from_numpy(score), topk, B).tolist()
    assert got == want


def test_cpu_tensors_fail_loudly():
    import boxinstseg_amd as bx
    from boxinstseg_amd import synthetic
    d = synthetic.cfg1(0)
    head = bx.CondInstMaskHead(in_channels=16, boxinst_enabled=True, max_proposals=-1)
    args = (torch.from_numpy(d['imgs']), d['img_metas'], torch.from_numpy(d['mask_logits']),
            torch.from_numpy(d['gt_inds']), [torch.from_numpy(b) for b in d['gt_bboxes']], None, None)
    with pytest.raises(RuntimeError, match='CUDA'):
        head.loss(*args)
    with pytest.raises(RuntimeError, match='CUDA'):
        bx.pairwise_nlog(torch.zeros(1, 1, 8, 8), 3, 2)
    with pytest.raises(RuntimeError, match='CUDA'):
        bx.color_affinity(args[0], d['img_metas'])
    with pytest.raises(RuntimeError, match='CUDA'):
        bx.boxinst_mask_loss(args[2], args[3], args[4], imgs=args[0], img_metas=d['img_metas'])


def test_missing_library_is_an_error_not_a_fallback(monkeypatch):
    from boxinstseg_amd import _lib, build
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(build, 'LIB_PATH', os.path.join(build.LIB_DIR, 'does_not_exist.so'))
    with pytest.raises(RuntimeError, match='no CPU or PyTorch fallback'):
        _lib.load()


@pytest.mark.skipif(not os.path.exists(REF_CFG), reason='reference checkout not present (GPU box)')
def test_reference_boxinst_configs_build_the_head_unchanged():
    import glob
    import boxinstseg_amd as bx
    cfgs = sorted(glob.glob('/root/reference/configs/boxinst/*.py'))
    assert len(cfgs) >= 5
    for path in cfgs:
        cfg = bx.load_config(path)
        mh = cfg['model']['mask_head']
        assert mh['type'] == 'CondInstMaskHead' and mh['boxinst_enabled'] is True
        head = bx.build_head(mh)
        assert head.pairwise_size == 3 and head.pairwise_dilation == 2 and head.out_stride == 4
    cfg = bx.load_config(REF_CFG)
    assert cfg['dist_params'] == dict(backend='nccl')             # RCCL under PyTorch-ROCm: unchanged
    assert cfg['data']['samples_per_gpu'] == 2


def test_config_loader_base_merge(tmp_path):
    import boxinstseg_amd as bx
    (tmp_path / 'base.py').write_text("model = dict(type='A', head=dict(x=1, y=2))\nlr = 0.1\n")
    (tmp_path / 'child.py').write_text("_base_ = ['base.py']\nmodel = dict(head=dict(y=3, z=dict(_delete_=True, q=1)))\n")
    cfg = bx.load_config(str(tmp_path / 'child.py'))
    assert cfg == dict(model=dict(type='A', head=dict(x=1, y=3, z=dict(q=1))), lr=0.1)


def test_synthetic_recipe_is_seeded_and_unambiguous():
    from boxinstseg_amd import synthetic
    a, b = synthetic.cfg1(3), synthetic.cfg1(3)
    assert np.array_equal(a['imgs'], b['imgs']) and np.array_equal(a['mask_logits'], b['mask_logits'])
    d = synthetic.cfg2(0)
    assert d['imgs'].shape == (2, 3, 800, 1024) and d['mask_logits'].shape == (32, 1, 200, 256) and d['G'] == 32
    mean = np.asarray(synthetic.MEAN, np.float64).reshape(1, 3, 1, 1)
    std = np.asarray(synthetic.STD, np.float64).reshape(1, 3, 1, 1)
    v = a['imgs'].astype(np.float64) * std + mean
    frac = v - np.floor(v)
    assert frac.min() > 0.2 and frac.max() < 0.3      # the +0.25 offset: uint8 truncation is unambiguous


def test_rows_removed_matches_reference_arithmetic():
    from boxinstseg_amd.functional import rows_removed
    assert rows_removed(10, (800, 1024, 3), (800, 1024, 3)) == 10
    assert rows_removed(10, (800, 1199, 3), (427, 640, 3)) == int(10 * float(800) / float(427)) == 18
    assert rows_removed(0, (64, 64, 3), (64, 64, 3)) == 0


def test_registers_into_mmdet_when_mmdet_is_importable():
    """The real drop-in path: with mmdet importable, CondInstMaskHead / the Box2Mask losses are registered INTO
    mmdet.models.builder.HEADS / LOSSES with force=True, replacing the stock classes of the same name
    (mmdet/models/builder.py:7-15, mmcv Registry semantics).  mmdet / mmcv cannot be installed here, so a stand-in
    `mmdet.models.builder` with an mmcv-style Registry is injected into sys.modules of a fresh interpreter."""
    import subprocess
    import textwrap
    code = textwrap.dedent('''
        import sys, types
        class Registry:                                   # mmcv.utils.Registry: the surface mmdet's builder uses
            def __init__(self, name): self.name, self._module_dict = name, {}
            @property
            def module_dict(self): return self._module_dict
            def get(self, key): return self._module_dict.get(key)
            def _register(self, cls, name=None, force=False):
                key = name or cls.__name__
                if not force and key in self._module_dict:
                    raise KeyError(f"{key} is already registered in {self.name}")
                self._module_dict[key] = cls
            def register_module(self, name=None, force=False, module=None):
                if module is not None:
                    self._register(module, name, force); return module
                def deco(cls):
                    self._register(cls, name, force); return cls
                return deco
            def build(self, cfg, default_args=None):
                args = dict(cfg); cls = self.get(args.pop("type"))
                for k, v in (default_args or {}).items(): args.setdefault(k, v)
                return cls(**args)
        builder = types.ModuleType("mmdet.models.builder")
        builder.HEADS, builder.LOSSES = Registry("models"), Registry("models")
        class StockHead: pass
        class StockLoss: pass
        builder.HEADS.register_module(name="CondInstMaskHead", module=StockHead)          # the reference's own classes
        builder.LOSSES.register_module(name="BoxProjectionLoss", module=StockLoss)
        for name in ("mmdet", "mmdet.models"):
            sys.modules[name] = types.ModuleType(name)
        sys.modules["mmdet.models.builder"] = builder
        sys.path.insert(0, %r)
        import boxinstseg_amd
        from boxinstseg_amd import registry
        assert isinstance(registry.HEADS, registry._MMDetHeads) and isinstance(registry.LOSSES, registry._MMDetHeads)
        assert builder.HEADS.get("CondInstMaskHead") is boxinstseg_amd.CondInstMaskHead, builder.HEADS.module_dict
        from boxinstseg_amd import levelset
        assert builder.LOSSES.get("BoxProjectionLoss") is levelset.BoxProjectionLoss
        assert builder.LOSSES.get("LevelsetLoss") is levelset.LevelsetLoss
        head = builder.HEADS.build(dict(type="CondInstMaskHead", in_channels=16, boxinst_enabled=True, topk_per_img=64,
                                        max_proposals=-1))               # what CondInst.__init__ does via build_head
        assert type(head) is boxinstseg_amd.CondInstMaskHead and head.param_conv.weight.shape == (233, 256, 3, 3)
        assert type(registry.build_loss(dict(type="BoxProjectionLoss", loss_weight=2.0))) is levelset.BoxProjectionLoss
        print("ok")
    ''') % ROOT
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith('ok'), r.stderr[-2000:]


def test_rgb2lab_oracle_matches_real_scikit_image():
    """tests/golden/lab_skimage.npz holds skimage.color.rgb2lab (scikit-image 0.18.3, run in the build container by
    tests/golden/make_lab_golden.py) of 65 536 random colours + the grey axis + primaries, cast to f32 as the reference does
    (condinst_head.py:1413-1416).  The exhaustive tally over all 2^24 colours is in lab_skimage_exhaustive.json: 9 differ,
    by one f32 ulp each (the last double bit of skimage's BLAS matrix product in rgb2xyz)."""
    import json
    from oracle import c_oracle
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'lab_skimage.npz'))
    rgb, want = g['rgb'], g['lab']
    got = c_oracle.rgb2lab_u8(np.ascontiguousarray(rgb.T).reshape(3, -1, 1))[:, :, 0].T
    bad = (got != want).any(axis=1)
    assert bad.sum() <= 2, f'{int(bad.sum())} of {len(rgb)} colours differ from scikit-image'
    if bad.any():
        ulp = np.abs(got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64))
        assert ulp.max() <= 1
    tally = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'lab_skimage_exhaustive.json')))
    assert tally['inputs'] == 1 << 24 and tally['f32_mismatches'] <= 16 and tally['max_ulp'] <= 1


@pytest.mark.parametrize('convs,ch,cin,no_rel', [(3, 8, 8, False), (2, 4, 6, False), (4, 16, 8, True), (1, 8, 5, False)])
def test_composed_dynamic_head_for_shapes_outside_the_hip_build(convs, ch, cin, no_rel):
    """CondInstMaskHead.forward for layer counts / widths the HIP kernels are not instantiated for (the reference leaves
    dynamic_convs / dynamic_channels / in_channels free, condinst_head.py:1079-1089) is composed of torch ops; it is pure
    torch, so it can be checked here: against the oracle of the shipped shape (pinned to the reference fixture), and
    against the reference's formulation -- grouped 1x1 convolutions over a [1, N*C, H, W] view (:1139-1164) -- in general."""
    import torch.nn.functional as F
    from boxinstseg_amd import CondInstMaskHead
    from oracle import torch_oracle as to
    torch.manual_seed(convs * 100 + ch)
    head = CondInstMaskHead(in_channels=cin, dynamic_convs=convs, dynamic_channels=ch, disable_rel_coors=no_rel,
                            boxinst_enabled=True).double()
    n, B, H, W = 5, 2, 6, 10
    feat = torch.randn(B, cin, H, W, dtype=torch.float64)
    params = torch.randn(n, head.num_gen_params, dtype=torch.float64, requires_grad=True)
    coors = torch.rand(n, 2, dtype=torch.float64) * 60
    lvl = torch.tensor([0, 1, 2, 3, 4]); img = torch.tensor([0, 1, 1, 0, 1])
    got = head._composed_forward(feat, params, coors, lvl, img)
    assert got.shape == (n, 1, 2 * H, 2 * W)
    if (convs, ch) == (3, 8):
        want = to.dynamic_mask_forward(feat, params, coors, lvl, img, head.sizes_of_interest, in_stride=8, out_stride=4,
                                       dynamic_channels=8, disable_rel_coors=no_rel)
        assert torch.allclose(got, want, rtol=0, atol=1e-12)
    x = feat[img]
    if not no_rel:
        xs = torch.arange(0, W * 8, 8, dtype=torch.float64) + 4; ys = torch.arange(0, H * 8, 8, dtype=torch.float64) + 4
        loc = torch.stack([xs[None, :].expand(H, W), ys[:, None].expand(H, W)], 0)
        x = torch.cat([(coors[:, :, None, None] - loc[None]) / head.sizes_of_interest.double()[lvl][:, None, None, None], x], 1)
    weights, biases = head.parse_dynamic_params(params)
    x = x.reshape(1, -1, H, W)
    for i, (w, b) in enumerate(zip(weights, biases)):
        x = F.conv2d(x, w, bias=b, groups=n)
        if i < convs - 1:
            x = F.relu(x)
    want = to.aligned_upsample(x.permute(1, 0, 2, 3), 2)
    assert torch.allclose(got, want, rtol=0, atol=1e-12)
    got.sum().backward()
    assert params.grad is not None and torch.isfinite(params.grad).all()


def test_iter_buffer_is_listed_for_ddp_to_skip():
    """DDP's broadcast_buffers would rewrite `_iter` in place every forward (one host sync per step afterwards): the helper lists it in
    the wrapped model's `_ddp_params_and_buffers_to_ignore`, under its qualified name, without dropping what is already there."""
    import torch.nn as nn
    from boxinstseg_amd import CondInstMaskHead
    from boxinstseg_amd.dist import exclude_iter_from_ddp_broadcast

    class Det(nn.Module):
        def __init__(self):
            super().__init__()
            self.mask_head = CondInstMaskHead(in_channels=16, boxinst_enabled=True)
            self._ddp_params_and_buffers_to_ignore = ['other']

    m = Det()
    assert exclude_iter_from_ddp_broadcast(m) == ['mask_head._iter']
    assert m._ddp_params_and_buffers_to_ignore == ['other', 'mask_head._iter']
    assert 'mask_head._iter' in dict(m.named_buffers())
    exclude_iter_from_ddp_broadcast(m)
    assert m._ddp_params_and_buffers_to_ignore.count('mask_head._iter') == 1


def test_ctypes_structs_match_the_c_header(tmp_path):
    """The two structs that cross the C ABI by pointer: size and every field offset of the ctypes mirrors (boxinstseg_amd/_lib.py) equal
    what a C compiler makes of include/boxinst_hip.h (a field added to one side only would shift everything behind it silently)."""
    import shutil, subprocess
    from boxinstseg_amd import _lib
    gcc = shutil.which('gcc')
    if gcc is None:
        pytest.skip('no C compiler')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mirrors = {'bxi_image_batch': _lib.ImageBatch, 'bxi_instances': _lib.Instances}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "boxinst_hip.h"', 'int main(void) {']
    for cname, cls in mirrors.items():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout'
    subprocess.run([gcc, '-std=c99', '-I', os.path.join(root, 'include'), str(src), '-o', str(exe)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in mirrors.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f'{cname}.{fname}']) == getattr(cls, fname).offset, f'{cname}.{fname}'
