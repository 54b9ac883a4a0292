"""Host side of the BoxInst mask-loss path: thin torch wrappers over the C ABI of libboxinst_hip.so.

Everything here only *marshals*: it reads tensor pointers / shapes / the current HIP stream, lets
torch own the memory, and calls ``include/boxinst_hip.h`` entry points.  No arithmetic of the path
is done in Python or in torch ops, there is no CPU fallback, and nothing synchronises the host.

Reference interfaces mirrored (``condinst_head.py`` = ``mmdet/models/dense_heads/condinst_head.py``):
  color_affinity      get_original_image :170-186 + get_targets :1345-1393 +
                      get_bitmasks_from_boxes :1395-1424 + get_image_color_similarity :220-246
  box_bitmasks        get_bitmasks_from_boxes :1426-1438
  boxinst_mask_loss   CondInstMaskHead.loss :1297-1337 (boxinst branch), forward + backward
"""
from __future__ import annotations

import contextlib
import ctypes as C
import threading
import warnings
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch.autograd.function import once_differentiable

from . import _lib


def _require_cuda(**tensors: Optional[torch.Tensor]) -> None:
    for name, t in tensors.items():
        if t is not None and not t.is_cuda:
            raise RuntimeError(f'{name} must be a CUDA (HIP) tensor: boxinstseg_amd has no CPU path')


_RAW_STREAM = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _current_stream(dev: torch.device) -> int:
    """The current stream's handle (the raw getter skips building a Stream object: 0.3 instead of 1.9 us)."""
    if _RAW_STREAM is not None:
        return _RAW_STREAM(dev.index if dev.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(dev).cuda_stream


def _stream(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def rows_removed(bottom_pixels_removed: int, img_shape: Sequence[int], ori_shape: Sequence[int]) -> int:
    """``int(bottom_pixels_removed * float(img_h) / float(ori_h))`` -- condinst_head.py:1358-1361."""
    return int(bottom_pixels_removed * float(img_shape[0]) / float(ori_shape[0]))


class _Batch:
    """Keeps the ctypes arrays behind a ``bxi_image_batch`` alive for the duration of a call."""

    def __init__(self, imgs: torch.Tensor, img_metas: Sequence[dict], bottom_pixels_removed: int,
                 image_masks: Optional[torch.Tensor] = None, denormalize: bool = True):
        if imgs.dim() != 4 or imgs.size(1) != 3:
            raise RuntimeError(f'imgs must be [B,3,H,W], got {tuple(imgs.shape)}')
        if imgs.dtype != torch.float32:
            raise RuntimeError(f'imgs must be float32, got {imgs.dtype}')
        B, _, Hc, Wc = imgs.shape
        if B > _lib.BXI_MAX_IMAGES:
            raise RuntimeError(f'at most {_lib.BXI_MAX_IMAGES} images per call, got {B}')
        if len(img_metas) != B:
            raise RuntimeError(f'{B} images but {len(img_metas)} img_metas')
        self.imgs = imgs.contiguous()
        self.masks = None if image_masks is None else image_masks.to(torch.float32).contiguous()
        self._h = _lib.int_array(m['img_shape'][0] for m in img_metas)
        self._w = _lib.int_array(m['img_shape'][1] for m in img_metas)
        self._rm = _lib.int_array(rows_removed(bottom_pixels_removed, m['img_shape'], m['ori_shape'])
                                  for m in img_metas)
        s = _lib.ImageBatch()
        s.imgs = self.imgs.data_ptr()
        s.B, s.Hc, s.Wc = B, Hc, Wc
        s.img_h_host = C.cast(self._h, C.POINTER(C.c_int))
        s.img_w_host = C.cast(self._w, C.POINTER(C.c_int))
        s.rows_removed_host = C.cast(self._rm, C.POINTER(C.c_int))
        if denormalize and B > 0:
            cfg = img_metas[0]['img_norm_cfg']
            for m in img_metas[1:]:
                c2 = m['img_norm_cfg']
                if list(c2['mean']) != list(cfg['mean']) or list(c2['std']) != list(cfg['std']) or \
                        bool(c2['to_rgb']) != bool(cfg['to_rgb']):
                    raise RuntimeError('all images of a batch must share img_norm_cfg')
            mean, std, to_rgb = cfg['mean'], cfg['std'], bool(cfg['to_rgb'])
        else:  # already an RGB image in 0..255 (the padded_images argument of get_bitmasks_from_boxes)
            mean, std, to_rgb = (0.0, 0.0, 0.0), (1.0, 1.0, 1.0), True
        for i in range(3):
            s.mean[i] = float(mean[i])
            s.std[i] = float(std[i])
        s.to_rgb = int(to_rgb)
        s.image_masks = 0 if self.masks is None else self.masks.data_ptr()
        self.struct = s
        self.B, self.Hc, self.Wc = B, Hc, Wc


class _Inst:
    """Keeps the ctypes arrays behind a ``bxi_instances`` alive for the duration of a call."""

    def __init__(self, mask_logits: torch.Tensor, gt_inds: torch.Tensor, gt_bboxes: Sequence[torch.Tensor],
                 Hc: int, Wc: int, stride: int):
        if mask_logits.dim() != 4 or mask_logits.size(1) != 1:
            raise RuntimeError(f'mask_logits must be [N,1,h,w], got {tuple(mask_logits.shape)}')
        N, _, h, w = mask_logits.shape
        if h * stride != Hc or w * stride != Wc:
            raise RuntimeError(f'mask_logits {h}x{w} x stride {stride} != image canvas {Hc}x{Wc}')
        if len(gt_bboxes) > _lib.BXI_MAX_IMAGES:
            raise RuntimeError(f'at most {_lib.BXI_MAX_IMAGES} images per call')
        self.logits = mask_logits.detach().to(torch.float32).contiguous()
        self.gt_inds = gt_inds.to(device=mask_logits.device, dtype=torch.int64).contiguous()
        if self.gt_inds.numel() != N:
            raise RuntimeError(f'{N} instances but {self.gt_inds.numel()} gt_inds')
        self.boxes = [b.detach().to(device=mask_logits.device, dtype=torch.float32).contiguous().view(-1, 4)
                      for b in gt_bboxes]
        self._ptrs = _lib.ptr_array(b.data_ptr() if b.numel() else 0 for b in self.boxes)
        self._cnt = _lib.int_array(b.size(0) for b in self.boxes)
        s = _lib.Instances()
        s.logits = self.logits.data_ptr()
        s.N, s.h, s.w = N, h, w
        s.gt_inds = self.gt_inds.data_ptr()
        s.boxes_per_img_host = C.cast(self._ptrs, C.POINTER(C.c_void_p))
        s.gt_count_host = C.cast(self._cnt, C.POINTER(C.c_int))
        s.B = len(self.boxes)
        s.Hc, s.Wc, s.stride = Hc, Wc, stride
        self.struct = s
        self.N, self.h, self.w = N, h, w


# ------------------------------------------------------------------------------------------------
# targets
# ------------------------------------------------------------------------------------------------
def color_affinity(imgs: torch.Tensor, img_metas: Sequence[dict], *, out_stride: int = 4,
                   bottom_pixels_removed: int = 10, pairwise_size: int = 3, pairwise_dilation: int = 2,
                   pairwise_color_thresh: float = 0.3, want_similarity: bool = True, want_bits: bool = True,
                   image_masks: Optional[torch.Tensor] = None, denormalize: bool = True
                   ) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor], torch.Tensor]:
    """-> (sim [B,K,h,w] f32 | None, bits [B,h,w] u8/int32 | None, lab [B,3,h,w] f32)."""
    _require_cuda(imgs=imgs, image_masks=image_masks)
    batch = _Batch(imgs, img_metas, bottom_pixels_removed, image_masks, denormalize)
    B, Hc, Wc = batch.B, batch.Hc, batch.Wc
    if Hc % out_stride or Wc % out_stride:
        raise RuntimeError(f'canvas {Hc}x{Wc} is not a multiple of out_stride {out_stride}')
    h, w = Hc // out_stride, Wc // out_stride
    K = pairwise_size * pairwise_size - 1
    dev = imgs.device
    lab = torch.empty((B, 3, h, w), dtype=torch.float32, device=dev)
    sim = torch.empty((B, K, h, w), dtype=torch.float32, device=dev) if want_similarity else None
    bits = None
    if want_bits:
        bits = torch.empty((B, h, w), dtype=torch.uint8 if K <= 8 else torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check('bxi_color_affinity_f32', _lib.load().bxi_color_affinity_f32(
            C.byref(batch.struct), int(out_stride), int(pairwise_size), int(pairwise_dilation),
            float(pairwise_color_thresh), lab.data_ptr(), 0, 0 if sim is None else sim.data_ptr(),
            0 if bits is None else bits.data_ptr(), _stream(dev)))
    return sim, bits, lab


def box_bitmasks(gt_bboxes: Sequence[torch.Tensor], Hc: int, Wc: int, stride: int, start: int) -> torch.Tensor:
    """Per-box {0,1} masks sampled at ``[start::stride, start::stride]`` -> [G, ., .] f32 (:1426-1432)."""
    if not gt_bboxes:
        raise RuntimeError('gt_bboxes is empty')
    _require_cuda(**{f'gt_bboxes[{i}]': b for i, b in enumerate(gt_bboxes)})
    dev = gt_bboxes[0].device
    boxes = [b.detach().to(torch.float32).contiguous().view(-1, 4) for b in gt_bboxes]
    ptrs = _lib.ptr_array(b.data_ptr() if b.numel() else 0 for b in boxes)
    cnt = _lib.int_array(b.size(0) for b in boxes)
    G = sum(b.size(0) for b in boxes)
    h, w = (Hc - start + stride - 1) // stride, (Wc - start + stride - 1) // stride
    out = torch.empty((G, h, w), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check('bxi_box_bitmasks_f32', _lib.load().bxi_box_bitmasks_f32(
            C.cast(ptrs, C.POINTER(C.c_void_p)), C.cast(cnt, C.POINTER(C.c_int)), len(boxes), int(Hc), int(Wc),
            int(stride), int(start), out.data_ptr(), _stream(dev)))
    return out


# ------------------------------------------------------------------------------------------------
# loss
# ------------------------------------------------------------------------------------------------
class _EvalPlan:
    """The two C structs of an evaluation and their host arrays, marshalled once per SHAPE CLASS -- (device, batch size, canvas,
    stride, number of box lists, img_norm_cfg) -- and patched per call with everything a training iteration changes: image
    shapes, rows removed, box counts, instance count, pointers.  (Real COCO iterations change all of those every step.)"""

    def __init__(self, B, Hc, Wc, stride, n_lists, mean, std, to_rgb):
        lib = _lib.load()
        self._h, self._w, self._rm = _lib.int_array([0] * B), _lib.int_array([0] * B), _lib.int_array([0] * B)
        self._cnt = _lib.int_array([0] * n_lists)
        self._ptrs = _lib.ptr_array([0] * max(n_lists, 1))
        b = _lib.ImageBatch()
        b.B, b.Hc, b.Wc = B, Hc, Wc
        b.img_h_host = C.cast(self._h, C.POINTER(C.c_int))
        b.img_w_host = C.cast(self._w, C.POINTER(C.c_int))
        b.rows_removed_host = C.cast(self._rm, C.POINTER(C.c_int))
        for i in range(3):
            b.mean[i], b.std[i] = mean[i], std[i]
        b.to_rgb = int(to_rgb)
        b.image_masks = 0
        s = _lib.Instances()
        s.h, s.w = Hc // stride, Wc // stride
        s.boxes_per_img_host = C.cast(self._ptrs, C.POINTER(C.c_void_p))
        s.gt_count_host = C.cast(self._cnt, C.POINTER(C.c_int))
        s.B = n_lists
        s.Hc, s.Wc, s.stride = Hc, Wc, stride
        self.batch, self.inst = b, s
        self.batch_ref, self.inst_ref = C.byref(b), C.byref(s)
        self.B, self.stride = B, stride
        self.eval = lib.bxi_boxinst_eval_f32
        self.rescale = lib.bxi_boxinst_grad_rescale_f32
        self.rescale_nhw = lib.bxi_boxinst_grad_rescale_nhw_f32
        self.ws = None                      # the workspace of the last call (tests look at it)

    def patch(self, img_metas, bottom_pixels_removed: int, N: int, boxes) -> None:
        h_, w_, rm_ = self._h, self._w, self._rm
        for i, m in enumerate(img_metas):
            shp = m['img_shape']
            h_[i], w_[i] = shp[0], shp[1]
            rm_[i] = int(bottom_pixels_removed * float(shp[0]) / float(m['ori_shape'][0]))     # condinst_head.py:1358-1361
        cnt, ptrs = self._cnt, self._ptrs
        for i, b in enumerate(boxes):
            n = b.shape[0] if b.dim() == 2 else b.numel() // 4
            cnt[i] = n
            ptrs[i] = b.data_ptr() if n else 0
        self.inst.N = N
        self.inst.iter_counter = 0          # set by the one caller that owns an iteration counter, after this


class _Local(threading.local):
    """Per host thread (one per device in the reference's launcher): plans are mutated per call, workspaces are in flight."""

    def __init__(self):
        self.plans: Dict[tuple, _EvalPlan] = {}
        self.workspaces: Dict[tuple, Tuple[torch.Tensor, int]] = {}
        self.sizes: Dict[tuple, Tuple[int, int]] = {}
        self.norms: Dict[tuple, tuple] = {}
        self.flags = 0                      # BXI_EVAL_* bits every evaluation of this thread is launched with (eval_flags, note_fault)
        self.wait_free = False              # after a fault in the two-launch form: the launches-only path (no in-kernel wait at all)
        self.forced: Optional[int] = None   # tests: the flags instead of what this module would choose
        self.prepared: Dict[tuple, list] = {}   # (device, canvas) -> rotating workspaces of prepare_targets
        self.prepared_next: Dict[tuple, int] = {}


_TLS = _Local()
_MAX_PLANS = 32
DEBUG_KEEP_LAST = False          # tests: keep the last evaluation's buffer so that its status word can be read
_LAST: Dict[str, object] = {}


def last_eval_status() -> Tuple[int, int]:
    """(status, tile rows) of the last evaluation (needs DEBUG_KEEP_LAST; synchronises)."""
    if not _LAST or not _LAST['need_grad']:
        return 0, 0
    plan, buf = _LAST['plan'], _LAST['buf']
    off = 256 + _lib.load().bxi_boxinst_loss_state_status_offset(plan.inst.N, plan.inst.h, plan.inst.w)
    v = buf.view(torch.uint8)[off:off + 8].view(torch.int32).cpu()
    return int(v[0]), int(v[1])


_MAX_WORKSPACES = 64


def _workspace(dev: torch.device, stream: int, canvas: tuple, N: int) -> torch.Tensor:
    """One workspace per (device, stream, canvas = (B, Hc, Wc, stride)): the C ABI's contract (include/boxinst_hip.h, section 3) is one
    LAYOUT per workspace -- the layout is fixed by the canvas and the workspace's size, every word only ever holds one kind of record,
    which is what makes the evaluation's tags safe -- ZEROED when it is allocated and never written by the host again.  Evaluations on
    a stream are serialised, so they share it whatever their instance count, up to the count it was sized for (rounded up to a
    multiple of 32, then grow-only: a larger N replaces it with a fresh zeroed one; the old buffer goes back to the caching allocator,
    which keeps it alive for the work already queued).  Padded training batches come in a few dozen canvases: LRU of 64, ~2 MB each."""
    key = (dev.index, stream) + canvas
    hit = _TLS.workspaces.get(key)
    if hit is None or hit[1] < N:
        n_cap = max((N + 31) // 32 * 32, 32 if hit is None else 2 * hit[1])
        B, Hc, Wc, stride = canvas
        need = max(_lib.load().bxi_boxinst_eval_workspace_bytes(B, Hc, Wc, stride, n_cap), 256)
        if hit is None and len(_TLS.workspaces) >= _MAX_WORKSPACES:
            _TLS.workspaces.pop(next(iter(_TLS.workspaces)))
        # A second stream on this device: evaluations may now run side by side, and the library is told so from here on (sticky;
        # BXI_EVAL_SHARED_DEVICE: no workgroup may hold a slot while it waits for workgroups later in the grid).  The library
        # itself never guesses what else runs on the device.
        if any(k[0] == dev.index and k[1] != stream for k in _TLS.workspaces) and not (_TLS.flags & _lib.EVAL_SHARED_DEVICE):
            _TLS.flags |= _lib.EVAL_SHARED_DEVICE
            warnings.warn('boxinstseg_amd: evaluations on a second stream of device %d seen; this thread launches every evaluation '
                          'with BXI_EVAL_SHARED_DEVICE from here on (reset_eval_state() forgets it)' % dev.index, RuntimeWarning, stacklevel=3)
        hit = _TLS.workspaces[key] = (torch.zeros(need, dtype=torch.uint8, device=dev), n_cap)
    elif len(_TLS.workspaces) > 1:
        _TLS.workspaces[key] = _TLS.workspaces.pop(key)          # most recently used last
    return hit[0]


class PreparedTargets:
    """What :func:`prepare_targets` leaves: a workspace holding Lab, the predicate words and the per-box pair counts of ONE batch, the
    event that says they are there, and what they were computed from (so that ``loss()`` can tell whether they are its batch's)."""
    __slots__ = ('ws', 'n_cap', 'event', 'match', 'slot', 'refs', 'gen')

    def matches(self, imgs: torch.Tensor, gt_bboxes, cfg: Dict) -> bool:
        # (the slot's generation: a later prepare_targets that took the same rotating workspace has overwritten these targets)
        return self.slot.get('gen') == self.gen and self.match == _targets_match_key(imgs, gt_bboxes, cfg)


def _targets_match_key(imgs, gt_bboxes, cfg) -> tuple:
    return (imgs.data_ptr(), tuple(imgs.shape), imgs._version, tuple((b.data_ptr(), tuple(b.shape), b._version) for b in gt_bboxes),
            int(cfg['out_stride']), int(cfg['bottom_pixels_removed']), int(cfg['pairwise_size']), int(cfg['pairwise_dilation']),
            float(cfg['pairwise_color_thresh']))


_PREPARED_SLOTS = 2              # targets of iteration i + 1 may be prepared while the loss of iteration i is still queued
_PREPARED_N_CAP = 256            # instances a prepared workspace is sized for (topk_per_img = 64 x samples_per_gpu <= 4)
_PREPARED_CANVASES = 8           # canvases with prepared workspaces kept per thread (2 x ~50 MB each at 2 x 800 x 1024)


def prepare_targets(imgs: torch.Tensor, img_metas: Sequence[dict], gt_bboxes: Sequence[torch.Tensor], *, out_stride: int = 4,
                    bottom_pixels_removed: int = 10, pairwise_size: int = 3, pairwise_dilation: int = 2,
                    pairwise_color_thresh: float = 0.3, stream: Optional['torch.cuda.Stream'] = None) -> Optional[PreparedTargets]:
    """The image side of ``CondInstMaskHead.loss`` ahead of time (``bxi_boxinst_targets_f32``): ``get_targets`` (condinst_head.py
    :1298-1299, :1345-1448) needs only the network input and the GT boxes, both of which exist before the backbone runs
    (``mmdet/models/detectors/condinst.py:53`` vs ``:73``).  Enqueued on ``stream`` (a side stream, ordered behind the current one; ``None``
    = the current stream), it takes image -> Lab -> colour predicates -> pair counts off the loss's critical path; hand the result to
    :func:`boxinst_mask_loss` (``targets=``).  Returns ``None`` where the library does not build it (more than 1024 GT boxes, a
    threshold <= 0, another window): the loss then computes its targets itself, as without this call."""
    _require_cuda(imgs=imgs)
    if _TLS.wait_free or not fused_supported(pairwise_size, pairwise_dilation):
        return None
    dev = imgs.device
    imgs_c = _f32c(imgs)
    boxes = [b if (b.dtype == torch.float32 and b.device == dev and b.is_contiguous()) else
             b.detach().to(device=dev, dtype=torch.float32).contiguous() for b in gt_bboxes]
    batch = _Batch(imgs_c, img_metas, bottom_pixels_removed)
    if len(boxes) != batch.B:
        raise RuntimeError(f'{batch.B} images but {len(boxes)} box lists')
    if batch.Hc % out_stride or batch.Wc % out_stride:
        raise RuntimeError(f'canvas {batch.Hc}x{batch.Wc} is not a multiple of out_stride {out_stride}')
    canvas = (batch.B, batch.Hc, batch.Wc, int(out_stride))
    key = (dev.index,) + canvas
    slots = _TLS.prepared.get(key)
    lib = _lib.load()
    if slots is None:
        if len(_TLS.prepared) >= _PREPARED_CANVASES:     # (multi-scale training: the oldest canvas goes; its blocks return to the allocator stream-safely)
            _TLS.prepared.pop(next(iter(_TLS.prepared)))
        need = max(lib.bxi_boxinst_eval_workspace_bytes(*canvas, _PREPARED_N_CAP), 256)
        slots = _TLS.prepared[key] = [dict(ws=torch.zeros(need, dtype=torch.uint8, device=dev), free=None, gen=0) for _ in range(_PREPARED_SLOTS)]
        _TLS.prepared_next[key] = 0
    k = _TLS.prepared_next[key]
    _TLS.prepared_next[key] = (k + 1) % len(slots)
    slot = slots[k]
    cur = torch.cuda.current_stream(dev)
    st = cur if stream is None else stream
    ptrs = _lib.ptr_array(b.data_ptr() if b.numel() else 0 for b in boxes)
    cnt = _lib.int_array(b.shape[0] if b.dim() == 2 else b.numel() // 4 for b in boxes)
    with torch.cuda.device(dev):
        if stream is not None:
            st.wait_stream(cur)                     # the images (and boxes) are produced on the current stream
        if slot['free'] is not None:
            st.wait_event(slot['free'])             # the evaluation that last read this workspace
        rc = lib.bxi_boxinst_targets_f32(C.byref(batch.struct), C.cast(ptrs, C.POINTER(C.c_void_p)), C.cast(cnt, C.POINTER(C.c_int)),
                                         int(out_stride), int(pairwise_size), int(pairwise_dilation), float(pairwise_color_thresh),
                                         slot['ws'].data_ptr(), slot['ws'].numel(), st.cuda_stream)
        if rc == _lib.BXI_ERR_UNSUPPORTED:
            return None
        _lib.check('bxi_boxinst_targets_f32', rc)
        ev = torch.cuda.Event()
        ev.record(st)
    if stream is not None:                          # the caching allocator must not hand these to somebody else while the side stream reads them
        imgs_c.record_stream(st)
        for b in boxes:
            if b.numel():
                b.record_stream(st)
        # ... nor the workspace itself, should its slot be dropped (canvas cap, reset_eval_state, a fault) while targets nobody consumed are
        # still being written on the side stream
        slot['ws'].record_stream(st)
    t = PreparedTargets()
    slot['gen'] += 1
    t.ws, t.n_cap, t.event, t.slot, t.gen = slot['ws'], _PREPARED_N_CAP, ev, slot, slot['gen']
    cfg = dict(out_stride=out_stride, bottom_pixels_removed=bottom_pixels_removed, pairwise_size=pairwise_size,
               pairwise_dilation=pairwise_dilation, pairwise_color_thresh=pairwise_color_thresh)
    t.match = _targets_match_key(imgs_c, boxes, cfg)
    t.refs = (imgs_c, boxes)
    return t


def eval_launch_flags() -> int:
    """The ``flags`` argument of the evaluation entry points for this host thread."""
    return _TLS.flags if _TLS.forced is None else _TLS.forced


@contextlib.contextmanager
def eval_flags(flags: int):
    """Tests / benchmarks: launch every evaluation of this thread with exactly these ``BXI_EVAL_*`` flags inside the block."""
    prev, _TLS.forced = _TLS.forced, int(flags)
    try:
        yield
    finally:
        _TLS.forced = prev


def _drop_or_zero_workspaces(drop: bool) -> None:
    """After a fault the ABI asks for a zeroed workspace.  DROPPED, the next evaluation allocates a fresh zeroed one and the old buffer
    goes back to the caching allocator, which keeps it alive for work already queued on whatever stream used it; zeroed in place
    (tests: the same memory again), each workspace is zeroed ON ITS OWN STREAM -- the host never writes a workspace on a stream other
    than the one its evaluations are serialised on."""
    _TLS.prepared.clear()
    _TLS.prepared_next.clear()
    if drop:
        _TLS.workspaces.clear()
        return
    lib = _lib.load()
    for key, (ws, _) in _TLS.workspaces.items():
        with torch.cuda.device(ws.device):
            _lib.check('bxi_boxinst_eval_workspace_init', lib.bxi_boxinst_eval_workspace_init(ws.data_ptr(), ws.numel(), key[1]))


def reset_eval_state(drop_workspaces: bool = True) -> None:
    """Forget this thread's sticky launch flags (and, by default, its workspaces: the next evaluation allocates a zeroed one)."""
    _TLS.flags = 0
    _TLS.wait_free = False
    _drop_or_zero_workspaces(drop_workspaces)


def note_fault(what: str = '') -> None:
    """An evaluation reported a non-zero status / non-finite losses (a bounded in-kernel wait ran out -- never expected on a GPU
    the process has to itself): from now on this thread takes the two-launch form, whose every wait is for a workgroup EARLIER
    in its grid (progress whatever else occupies the device), and its workspaces are zeroed again as the ABI asks after a fault.
    Called where losses reach the host anyway (``dist.parse_losses``); never synchronises by itself.
    A fault while ALREADY in the two-launch form (workgroups are not being dispatched in grid order, or the device is starved for
    longer than the waits' bound) takes the last step down: the path without any in-kernel wait -- ``bxi_color_affinity_f32`` +
    ``bxi_boxinst_loss_fwd_bwd_f32`` + ``bxi_boxinst_loss_backward_f32``, four launches ordered by the stream alone."""
    if _TLS.flags & _lib.EVAL_TWO_LAUNCHES:
        if not _TLS.wait_free:
            warnings.warn('boxinstseg_amd: an evaluation in the two-launch form reported a fault%s; taking the path without in-kernel '
                          'waits from here on' % (f' ({what})' if what else ''), RuntimeWarning, stacklevel=2)
        _TLS.wait_free = True
    else:
        warnings.warn('boxinstseg_amd: an evaluation reported a fault%s; taking the two-launch form from here on'
                      % (f' ({what})' if what else ''), RuntimeWarning, stacklevel=2)
    _TLS.flags = (_TLS.flags | _lib.EVAL_TWO_LAUNCHES) & ~_lib.EVAL_SINGLE_LAUNCH
    _drop_or_zero_workspaces(True)


def _sizes(N: int, h: int, w: int, B: int, Hc: int, Wc: int, stride: int) -> Tuple[int, int]:
    """(state bytes rounded to 256, workspace bytes) -- two C calls, cached per shape."""
    key = (N, h, w, B, Hc, Wc, stride)
    v = _TLS.sizes.get(key)
    if v is None:
        lib = _lib.load()
        if len(_TLS.sizes) > 4096:
            _TLS.sizes.clear()
        v = _TLS.sizes[key] = ((max(lib.bxi_boxinst_loss_state_bytes(N, h, w), 256) + 255) // 256 * 256,
                               max(lib.bxi_boxinst_eval_workspace_bytes(B, Hc, Wc, stride, N), 256))
    return v


def _eval_plan(imgs, img_metas, mask_logits, gt_bboxes, stride: int, bottom_pixels_removed: int, stream: int) -> _EvalPlan:
    if imgs.dim() != 4 or imgs.size(1) != 3:
        raise RuntimeError(f'imgs must be [B,3,H,W], got {tuple(imgs.shape)}')
    if mask_logits.dim() != 4 or mask_logits.size(1) != 1:
        raise RuntimeError(f'mask_logits must be [N,1,h,w], got {tuple(mask_logits.shape)}')
    B, _, Hc, Wc = imgs.shape
    N, _, h, w = mask_logits.shape
    if len(img_metas) != B:
        raise RuntimeError(f'{B} images but {len(img_metas)} img_metas')
    if h * stride != Hc or w * stride != Wc:
        raise RuntimeError(f'mask_logits {h}x{w} x stride {stride} != image canvas {Hc}x{Wc}')
    cfg = img_metas[0]['img_norm_cfg'] if B else None
    for m in img_metas[1:]:
        c2 = m['img_norm_cfg']
        if c2 is not cfg and (c2['mean'] is not cfg['mean'] or c2['std'] is not cfg['std'] or bool(c2['to_rgb']) != bool(cfg['to_rgb'])) and \
                (list(c2['mean']) != list(cfg['mean']) or list(c2['std']) != list(cfg['std']) or bool(c2['to_rgb']) != bool(cfg['to_rgb'])):
            raise RuntimeError('all images of a batch must share img_norm_cfg')
    norm = None
    if cfg is not None:
        # mmdet's Normalize transform hands every sample the SAME mean / std arrays in a fresh dict: remember the conversion per array pair
        mean_o, std_o = cfg['mean'], cfg['std']
        hit = _TLS.norms.get((id(mean_o), id(std_o)))
        if hit is not None and hit[0] is mean_o and hit[1] is std_o:
            norm = (hit[2], hit[3], bool(cfg['to_rgb']))
        else:
            norm = (tuple(float(v) for v in mean_o), tuple(float(v) for v in std_o), bool(cfg['to_rgb']))
            if len(_TLS.norms) > 64:
                _TLS.norms.clear()
            _TLS.norms[(id(mean_o), id(std_o))] = (mean_o, std_o, norm[0], norm[1])
    key = (imgs.device.index, B, Hc, Wc, stride, len(gt_bboxes), norm)
    plans = _TLS.plans
    plan = plans.get(key)
    if plan is None:
        if B > _lib.BXI_MAX_IMAGES or len(gt_bboxes) > _lib.BXI_MAX_IMAGES:
            raise RuntimeError(f'at most {_lib.BXI_MAX_IMAGES} images per call, got {B}')
        mean, std, to_rgb = norm if norm is not None else ((0.0,) * 3, (1.0,) * 3, True)
        if len(plans) >= _MAX_PLANS:
            plans.pop(next(iter(plans)))            # the oldest entry; nothing is in flight on a plan (host arrays only)
        plan = plans[key] = _EvalPlan(B, Hc, Wc, stride, len(gt_bboxes), mean, std, to_rgb)
    plan.patch(img_metas, bottom_pixels_removed, N, gt_bboxes)
    plan.state_bytes, _ = _sizes(N, h, w, B, Hc, Wc, stride)
    plan.grad_elems = N * h * w
    plan.ws = _workspace(imgs.device, stream, (B, Hc, Wc, stride), N)
    plan.ws_ptr, plan.ws_bytes = plan.ws.data_ptr(), plan.ws.numel()
    return plan


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.to(torch.float32)
    return t if t.is_contiguous() else t.contiguous()


class BoxInstMaskLoss(torch.autograd.Function):
    """(loss_prj, loss_pairwise) = f(mask_logits); forward AND backward in ONE pass over the logits.

    forward  : bxi_boxinst_eval_f32 (one launch at the shipped shapes, otherwise two) writes both scalars and the FINISHED gradient for unit upstream
               factors -- what ``loss.backward()`` seeds the two terms with.
               (Precomputed affinity bits: bxi_boxinst_loss_fwd_bwd_f32 + bxi_boxinst_loss_backward_f32.)
    backward : bxi_boxinst_grad_rescale_f32 -- reads the two upstream scalars from device memory (no host sync) and returns
               at once when both are 1; otherwise rewrites the gradient for them.
    A second backward through the same node (``retain_graph=True``) re-evaluates into a fresh buffer, as the first
    buffer then belongs to autograd (the reference's op supports re-entrant backward, pairwise.py:17-26).
    """

    @staticmethod
    def forward(ctx, mask_logits: torch.Tensor, imgs: Optional[torch.Tensor], img_metas, gt_inds: torch.Tensor,
                gt_bboxes, cfg: Dict, affinity_bits: Optional[torch.Tensor]):
        _require_cuda(mask_logits=mask_logits, imgs=imgs, gt_inds=gt_inds, affinity_bits=affinity_bits)
        ctx.in_dtype = mask_logits.dtype
        ctx.cfg = cfg
        ctx.fused = affinity_bits is None
        need_grad = bool(ctx.needs_input_grad[0])
        if affinity_bits is not None:
            return BoxInstMaskLoss._forward_bits(ctx, mask_logits, gt_inds, gt_bboxes, cfg, affinity_bits, need_grad)
        ctx.imgs, ctx.metas, ctx.gt_inds, ctx.boxes = imgs, img_metas, gt_inds, gt_bboxes
        ctx.logits = mask_logits.detach()
        ctx.calls = 0
        losses, ctx.grad, ctx.state, ctx.plan, ctx.keep = BoxInstMaskLoss._evaluate(ctx, need_grad)
        return losses.unbind(0)

    @staticmethod
    def _evaluate(ctx, need_grad: bool):
        cfg, logits = ctx.cfg, ctx.logits
        dev = logits.device
        stream = _current_stream(dev)
        imgs, x = _f32c(ctx.imgs), _f32c(logits)
        gi = ctx.gt_inds
        if gi.dtype != torch.int64 or gi.device != dev or not gi.is_contiguous():
            gi = gi.to(device=dev, dtype=torch.int64).contiguous()
        boxes = [b if (b.dtype == torch.float32 and b.device == dev and b.is_contiguous()) else
                 b.detach().to(device=dev, dtype=torch.float32).contiguous() for b in ctx.boxes]
        plan = _eval_plan(imgs, ctx.metas, x, boxes, int(cfg['out_stride']), int(cfg['bottom_pixels_removed']), stream)
        if gi.numel() != plan.inst.N:
            raise RuntimeError(f'{plan.inst.N} instances but {gi.numel()} gt_inds')
        # targets prepared ahead (prepare_targets): THEIR workspace, behind their event; only for the first evaluation of this node, and
        # only if they were computed from this very batch (else the evaluation computes its own, as without them)
        flags = eval_launch_flags()
        tg = cfg.get('targets') if ctx.calls == 0 else None
        if tg is not None and not _TLS.wait_free and plan.inst.N <= tg.n_cap and tg.ws.device == dev and tg.matches(imgs, boxes, cfg):
            torch.cuda.current_stream(dev).wait_event(tg.event)
            plan.ws = tg.ws
            plan.ws_ptr, plan.ws_bytes = tg.ws.data_ptr(), tg.ws.numel()
            flags |= _lib.EVAL_TARGETS_READY
        else:
            tg = None
        # ONE allocation, as floats: [losses 256 B][state (a multiple of 256 B)][gradient]
        nfl = 64 + (plan.state_bytes // 4 + plan.grad_elems if need_grad else 0)
        buf = torch.empty(nfl, dtype=torch.float32, device=dev)
        base = buf.data_ptr()
        if base & 255:
            raise RuntimeError('allocator returned a buffer that is not 256-byte aligned')
        plan.batch.imgs = imgs.data_ptr()
        plan.inst.logits = x.data_ptr()
        plan.inst.gt_inds = gi.data_ptr()
        it = cfg.get('iter_counter')
        if it is not None and ctx.calls == 0:           # counted once per loss() call, not again by a re-entrant backward
            plan.inst.iter_counter = it.data_ptr()
        grad = None
        if need_grad:
            grad = buf.as_strided(x.shape, x.stride(), 64 + plan.state_bytes // 4)          # one view, not a slice and a reshape
        # the warm-up factor: by value, or (warmup_iters given with the counter) evaluated on the device from the counter -- no host
        # mirror of `_iter`, and right under hipGraph replay.  A re-entrant backward evaluates again with factor 1 and folds the FIRST
        # evaluation's recorded factor into the upstream gradient (backward()).
        warm = float(cfg['warmup_factor'])
        if ctx.calls > 0 and cfg.get('warmup_iters') is not None:
            warm = 1.0
        elif it is not None and cfg.get('warmup_iters') is not None:
            warm = -float(cfg['warmup_iters'])
        args = (plan.batch_ref, plan.inst_ref, int(cfg['pairwise_size']), int(cfg['pairwise_dilation']),
                float(cfg['pairwise_color_thresh']), warm, 0, 0, base,
                base + 256 + plan.state_bytes if need_grad else 0, base + 256 if need_grad else 0,
                plan.ws_ptr, plan.ws_bytes, flags, stream)
        if torch.cuda.current_device() == dev.index:          # the usual case: no device guard to set up and tear down
            rc = plan.eval(*args)
        else:
            with torch.cuda.device(dev):
                rc = plan.eval(*args)
        _lib.check('bxi_boxinst_eval_f32', rc)
        if tg is not None:              # the next prepare_targets that takes this workspace waits for this evaluation
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            tg.slot['free'] = ev
        losses = buf[:2]
        if DEBUG_KEEP_LAST:
            _LAST.clear()
            _LAST.update(buf=buf, plan=plan, need_grad=need_grad)
        # the plan is shared by every evaluation of its shape class: backward() binds it again from what is kept here
        return losses, grad, base + 256, plan, (imgs, x, gi, boxes, buf, ctx.metas, int(cfg['bottom_pixels_removed']))

    @staticmethod
    def _forward_bits(ctx, mask_logits, gt_inds, gt_bboxes, cfg, affinity_bits, need_grad):
        dev = mask_logits.device
        stride, size, dil = int(cfg['out_stride']), int(cfg['pairwise_size']), int(cfg['pairwise_dilation'])
        Hc, Wc = mask_logits.shape[2] * stride, mask_logits.shape[3] * stride
        inst = _Inst(mask_logits, gt_inds, gt_bboxes, Hc, Wc, stride)
        lib = _lib.load()
        losses = torch.empty(2, dtype=torch.float32, device=dev)
        grad = torch.empty_like(inst.logits) if need_grad else None
        state = None
        if need_grad:
            state = torch.empty(max(lib.bxi_boxinst_loss_state_bytes(inst.N, inst.h, inst.w), 256),
                                dtype=torch.uint8, device=dev)
        bits = affinity_bits.contiguous()
        if bits.dtype != torch.uint8 or tuple(bits.shape) != (len(gt_bboxes), inst.h, inst.w):
            raise RuntimeError('affinity_bits must be uint8 [B,h,w]')
        ws = torch.empty(max(lib.bxi_boxinst_loss_workspace_bytes(inst.N, inst.h, inst.w), 256), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check('bxi_boxinst_loss_fwd_bwd_f32', lib.bxi_boxinst_loss_fwd_bwd_f32(
                C.byref(inst.struct), bits.data_ptr(), size, dil, float(cfg['warmup_factor']),
                losses.data_ptr(), 0 if grad is None else grad.data_ptr(),
                0 if state is None else state.data_ptr(), ws.data_ptr(), ws.numel(), _stream(dev)))
        ctx.inst, ctx.dil, ctx.grad, ctx.state_t = inst, dil, grad, state
        return losses[0], losses[1]

    @staticmethod
    @once_differentiable
    def backward(ctx, g_prj: torch.Tensor, g_pw: torch.Tensor):
        if not ctx.fused:
            return BoxInstMaskLoss._backward_bits(ctx, g_prj, g_pw)
        if ctx.grad is None and ctx.calls == 0:
            raise RuntimeError('BoxInstMaskLoss.backward without a gradient request')
        if ctx.calls > 0:            # re-entrant backward: the first buffer now belongs to autograd
            first_buf, first_x = ctx.keep[4], ctx.keep[1]
            _, grad, state, plan, keep = BoxInstMaskLoss._evaluate(ctx, True)
            if ctx.cfg.get('warmup_iters') is not None and ctx.cfg.get('iter_counter') is not None:
                # evaluated with factor 1 (the counter has moved on): the factor the first evaluation applied is in its state
                off = 64 + _lib.load().bxi_boxinst_loss_state_warmup_offset(first_x.size(0), first_x.size(2), first_x.size(3)) // 4
                g_pw = g_pw.to(device=grad.device, dtype=torch.float32) * first_buf[off]
        else:
            grad, state, plan, keep = ctx.grad, ctx.state, ctx.plan, ctx.keep
            ctx.grad = None
        ctx.calls += 1
        dev = grad.device
        if grad.numel() > 0:
            if g_prj.dtype != torch.float32 or g_prj.device != dev:
                g_prj = g_prj.to(device=dev, dtype=torch.float32)
            if g_pw.dtype != torch.float32 or g_pw.device != dev:
                g_pw = g_pw.to(device=dev, dtype=torch.float32)
            x = keep[1]
            # (N, h, w) is all the rescale needs of the instances: nothing of the shape class's shared plan is touched here
            args = (x.size(0), x.size(2), x.size(3), g_prj.data_ptr(), g_pw.data_ptr(), int(ctx.cfg['pairwise_dilation']), state,
                    grad.data_ptr(), _current_stream(dev))
            if torch.cuda.current_device() == dev.index:          # the usual case: no device guard to set up and tear down
                rc = plan.rescale_nhw(*args)
            else:
                with torch.cuda.device(dev):
                    rc = plan.rescale_nhw(*args)
            if rc:
                _lib.check('bxi_boxinst_grad_rescale_nhw_f32', rc)
        if grad.dtype != ctx.in_dtype:
            grad = grad.to(ctx.in_dtype)
        return grad, None, None, None, None, None, None

    @staticmethod
    def _backward_bits(ctx, g_prj, g_pw):
        grad, inst = ctx.grad, ctx.inst
        if grad is None:
            raise RuntimeError('BoxInstMaskLoss.backward (precomputed-bits path) called twice: its gradient buffer is '
                               'finished in place by the first call; evaluate again')
        ctx.grad = None
        dev = grad.device
        if inst.N > 0:
            g_prj = g_prj.to(device=dev, dtype=torch.float32).contiguous()
            g_pw = g_pw.to(device=dev, dtype=torch.float32).contiguous()
            with torch.cuda.device(dev):
                _lib.check('bxi_boxinst_loss_backward_f32', _lib.load().bxi_boxinst_loss_backward_f32(
                    C.byref(inst.struct), g_prj.data_ptr(), g_pw.data_ptr(), ctx.dil, ctx.state_t.data_ptr(),
                    grad.data_ptr(), _stream(dev)))
        if grad.dtype != ctx.in_dtype:
            grad = grad.to(ctx.in_dtype)
        return grad, None, None, None, None, None, None


class HeadBoxInstLoss(torch.autograd.Function):
    """(mask_logits, loss_prj, loss_pairwise) = f(feat, params): ``CondInstMaskHead.forward`` + ``.loss`` with the dynamic
    mask head evaluated inside the loss evaluation's first launch (``bxi_boxinst_head_eval_f32``).

    backward: the finished gradient w.r.t. the logits (rescaled for the upstream factors, plus whatever arrives for the
    logits output itself) goes through ``bxi_dynamic_mask_backward_f32`` to ``feat`` and ``params``."""

    @staticmethod
    def forward(ctx, feat, params, coors, level_inds, img_inds, sizes_of_interest, head_cfg, imgs, img_metas, gt_inds,
                gt_bboxes, cfg):
        _require_cuda(feat=feat, params=params, coors=coors, imgs=imgs, gt_inds=gt_inds)
        dev = feat.device
        in_stride, factor, no_rel = head_cfg
        B, Cf, Hs, Ws = feat.shape
        N = params.size(0)
        f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
        i64 = lambda t: t.detach().to(device=dev, dtype=torch.int64).contiguous()
        feat_c, params_c, coors_c = f32(feat), f32(params), f32(coors).view(-1, 2)
        lvl, img, soi, gi = i64(level_inds), i64(img_inds), f32(sizes_of_interest), i64(gt_inds)
        boxes = [b.detach().to(device=dev, dtype=torch.float32).contiguous() for b in gt_bboxes]
        imgs_c = _f32c(imgs)
        stream = _current_stream(dev)
        logits = torch.empty((N, 1, Hs * factor, Ws * factor), dtype=torch.float32, device=dev)
        plan = _eval_plan(imgs_c, img_metas, logits, boxes, int(cfg['out_stride']), int(cfg['bottom_pixels_removed']), stream)
        need_grad = bool(ctx.needs_input_grad[0] or ctx.needs_input_grad[1])
        buf = torch.empty(256 + plan.state_bytes + 4 * plan.grad_elems, dtype=torch.uint8, device=dev)
        base = buf.data_ptr()
        plan.batch.imgs = imgs_c.data_ptr()
        plan.inst.logits = logits.data_ptr()
        plan.inst.gt_inds = gi.data_ptr()
        if cfg.get('iter_counter') is not None:
            plan.inst.iter_counter = cfg['iter_counter'].data_ptr()
        with torch.cuda.device(dev):
            _lib.check('bxi_boxinst_head_eval_f32', _lib.load().bxi_boxinst_head_eval_f32(
                plan.batch_ref, plan.inst_ref, feat_c.data_ptr(), Cf, Hs, Ws, params_c.data_ptr(), coors_c.data_ptr(),
                lvl.data_ptr(), img.data_ptr(), soi.data_ptr(), soi.numel(), int(in_stride), int(factor), int(bool(no_rel)),
                int(cfg['pairwise_size']), int(cfg['pairwise_dilation']), float(cfg['pairwise_color_thresh']),
                -float(cfg['warmup_iters']) if (cfg.get('iter_counter') is not None and cfg.get('warmup_iters') is not None)
                else float(cfg['warmup_factor']), 0, 0, base, base + 256 + plan.state_bytes, base + 256, plan.ws_ptr,
                plan.ws_bytes, eval_launch_flags(), stream))
        losses = buf[:8].view(torch.float32)
        ctx.save_for_backward(feat_c, params_c, coors_c, lvl, img, soi)
        ctx.grad = buf[256 + plan.state_bytes:].view(torch.float32).view(logits.shape)
        ctx.state, ctx.plan, ctx.keep = base + 256, plan, (imgs_c, gi, boxes, buf, img_metas, int(cfg['bottom_pixels_removed']), logits)
        ctx.head_cfg, ctx.dil, ctx.need_grad = (int(in_stride), int(factor), int(bool(no_rel))), int(cfg['pairwise_dilation']), need_grad
        ctx.cfg, ctx.calls = cfg, 0
        ctx.dtypes = (feat.dtype, params.dtype)
        ctx.mark_non_differentiable(logits) if not need_grad else None
        return logits, losses[0], losses[1]

    @staticmethod
    @once_differentiable
    def backward(ctx, g_logits_in, g_prj, g_pw):
        feat, params, coors, lvl, img, soi = ctx.saved_tensors
        dev = feat.device
        lib = _lib.load()
        stream = _current_stream(dev)
        grad, plan, state = ctx.grad, ctx.plan, ctx.state
        ctx.grad = None
        if ctx.calls > 0:
            # re-entrant backward (retain_graph=True; the reference's composed graph allows it): the first buffer was finished in place
            # for the first call's upstream factors, so the loss is evaluated again from the logits that were kept
            cfg = ctx.cfg
            imgs_k, gi_k, boxes_k, _, metas_k, bpr_k, logits_k = ctx.keep
            plan = _eval_plan(imgs_k, metas_k, logits_k, boxes_k, int(cfg['out_stride']), bpr_k, stream)
            again = torch.empty(64 + plan.state_bytes // 4 + plan.grad_elems, dtype=torch.float32, device=dev)
            base = again.data_ptr()
            plan.batch.imgs, plan.inst.logits, plan.inst.gt_inds = imgs_k.data_ptr(), logits_k.data_ptr(), gi_k.data_ptr()
            with torch.cuda.device(dev):
                dev_warm = cfg.get('warmup_iters') is not None and cfg.get('iter_counter') is not None
                _lib.check('bxi_boxinst_eval_f32', plan.eval(
                    plan.batch_ref, plan.inst_ref, int(cfg['pairwise_size']), int(cfg['pairwise_dilation']),
                    float(cfg['pairwise_color_thresh']), 1.0 if dev_warm else float(cfg['warmup_factor']), 0, 0, base,
                    base + 256 + plan.state_bytes, base + 256, plan.ws_ptr, plan.ws_bytes, eval_launch_flags(), stream))
            grad, state = again[64 + plan.state_bytes // 4:].view(logits_k.shape), base + 256
            if dev_warm:         # the factor the first evaluation applied (recorded in its state), folded into the upstream gradient
                first = ctx.keep[3]
                off = 256 + lib.bxi_boxinst_loss_state_warmup_offset(logits_k.size(0), logits_k.size(2), logits_k.size(3))
                g_pw = g_pw.to(device=dev, dtype=torch.float32) * first[off:off + 4].view(torch.float32)[0]
        elif grad is None:
            raise RuntimeError('HeadBoxInstLoss.backward without a gradient request')
        ctx.calls += 1
        g_prj = g_prj.to(device=dev, dtype=torch.float32)
        g_pw = g_pw.to(device=dev, dtype=torch.float32)
        in_stride, factor, no_rel = ctx.head_cfg
        B, Cf, Hs, Ws = feat.shape
        N = params.size(0)
        g_feat, g_params = torch.empty_like(feat), torch.empty_like(params)
        ws = torch.empty(max(lib.bxi_dynamic_mask_backward_workspace_bytes(B, Cf, Hs, Ws, N, no_rel), 256), dtype=torch.uint8,
                         device=dev)
        _, gi_k, boxes_k, _, metas_k, bpr_k, logits_k = ctx.keep
        plan.patch(metas_k, bpr_k, N, boxes_k)                  # another evaluation of the same shape class may have come in between
        plan.inst.logits, plan.inst.gt_inds = logits_k.data_ptr(), gi_k.data_ptr()
        with torch.cuda.device(dev):
            _lib.check('bxi_boxinst_grad_rescale_f32', plan.rescale(plan.inst_ref, g_prj.data_ptr(), g_pw.data_ptr(), ctx.dil,
                                                                     state, grad.data_ptr(), stream))
            if g_logits_in is not None:
                grad = grad + g_logits_in.to(torch.float32)
            _lib.check('bxi_dynamic_mask_backward_f32', lib.bxi_dynamic_mask_backward_f32(
                feat.data_ptr(), B, Cf, Hs, Ws, params.data_ptr(), N, coors.data_ptr(), lvl.data_ptr(), img.data_ptr(),
                soi.data_ptr(), soi.numel(), in_stride, factor, no_rel, grad.data_ptr(), g_feat.data_ptr(), g_params.data_ptr(),
                ws.data_ptr(), ws.numel(), stream))
        return (g_feat.to(ctx.dtypes[0]), g_params.to(ctx.dtypes[1])) + (None,) * 10


def boxinst_mask_loss(mask_logits: torch.Tensor, gt_inds: torch.Tensor, gt_bboxes: Sequence[torch.Tensor], *,
                      imgs: Optional[torch.Tensor] = None, img_metas: Optional[Sequence[dict]] = None,
                      affinity_bits: Optional[torch.Tensor] = None, out_stride: int = 4,
                      bottom_pixels_removed: int = 10, pairwise_size: int = 3, pairwise_dilation: int = 2,
                      pairwise_color_thresh: float = 0.3, warmup_factor: float = 1.0,
                      iter_counter: Optional[torch.Tensor] = None, warmup_iters: Optional[float] = None,
                      targets: Optional[PreparedTargets] = None) -> Dict[str, torch.Tensor]:
    """The BoxInst branch of ``CondInstMaskHead.loss`` (condinst_head.py:1297-1337) as one call.

    Either ``imgs`` + ``img_metas`` (targets are computed on the device from the network input) or
    precomputed ``affinity_bits`` (from :func:`color_affinity`) must be given.
    ``iter_counter`` (images path only): a float32 device scalar the evaluation adds 1 to inside its last launch -- the module's
    ``self._iter += 1`` (condinst_head.py:1297) without a launch of its own.  With ``warmup_iters`` (``pairwise_warmup``) the warm-up
    factor ``min(_iter / warmup_iters, 1)`` (:1330-1331) is evaluated ON THE DEVICE from that counter and ``warmup_factor`` is ignored:
    no host copy of the counter exists, and a captured hipGraph ramps as the eager loop does.
    ``targets``: what :func:`prepare_targets` returned for THIS batch (same ``imgs`` tensor, same boxes, same parameters; anything else
    is ignored and the evaluation computes its targets itself): the evaluation then launches only the logit stream, the leaders, the
    tiles and the finisher (``BXI_EVAL_TARGETS_READY``) -- bit-equal results.
    Returns ``{'loss_prj', 'loss_pairwise'}`` attached to the autograd graph of ``mask_logits``.
    Built for ``pairwise_size == 3`` and ``pairwise_dilation <= 4`` (``fused_supported``); other windows are
    composed from the op-level kernels by ``CondInstMaskHead._composed_loss``.
    """
    if affinity_bits is None and (imgs is None or img_metas is None):
        raise RuntimeError('need imgs + img_metas or affinity_bits')
    if not fused_supported(pairwise_size, pairwise_dilation, affinity_bits is not None):
        raise RuntimeError('the fused path is built for pairwise_size == 3 (dilation <= 4 from images, <= 8 from bits); '
                           'use CondInstMaskHead.loss, which composes the op-level kernels for other windows')
    cfg = dict(out_stride=out_stride, bottom_pixels_removed=bottom_pixels_removed, pairwise_size=pairwise_size,
               pairwise_dilation=pairwise_dilation, pairwise_color_thresh=pairwise_color_thresh,
               warmup_factor=warmup_factor)
    if targets is not None and affinity_bits is None:
        cfg['targets'] = targets
    if iter_counter is not None:
        if affinity_bits is not None:
            raise RuntimeError('iter_counter is counted by the evaluation from images; with affinity_bits add to it yourself')
        if iter_counter.dtype != torch.float32 or iter_counter.device != mask_logits.device or iter_counter.numel() != 1:
            raise RuntimeError('iter_counter must be one float32 on the device of mask_logits')
        cfg['iter_counter'] = iter_counter
        if warmup_iters is not None:
            if not float(warmup_iters) > 0:
                raise RuntimeError('warmup_iters must be positive')
            cfg['warmup_iters'] = float(warmup_iters)
    elif warmup_iters is not None:
        raise RuntimeError('warmup_iters needs iter_counter (the factor is evaluated from the device counter)')
    if affinity_bits is None and _TLS.wait_free and _TLS.forced is None and pairwise_dilation <= 8:
        # the last step of the fall-back ladder (note_fault): targets and loss as separate launches, no in-kernel wait anywhere.
        # The counter is advanced and read on the host, as the reference does (condinst_head.py:1297,1330-1331): one sync per call
        # on a path that exists for a device in trouble.
        if iter_counter is not None:
            iter_counter += 1
            if warmup_iters is not None:
                cfg['warmup_factor'] = min(float(iter_counter.item()) / float(warmup_iters), 1.0)
            cfg.pop('iter_counter', None)
            cfg.pop('warmup_iters', None)
        _, affinity_bits, _ = color_affinity(imgs, img_metas, out_stride=out_stride, bottom_pixels_removed=bottom_pixels_removed,
                                             pairwise_size=pairwise_size, pairwise_dilation=pairwise_dilation,
                                             pairwise_color_thresh=pairwise_color_thresh, want_similarity=False, want_bits=True)
    loss_prj, loss_pw = BoxInstMaskLoss.apply(mask_logits, imgs, img_metas, gt_inds, list(gt_bboxes), cfg,
                                              affinity_bits)
    return {'loss_prj': loss_prj, 'loss_pairwise': loss_pw}


def fused_supported(pairwise_size: int, pairwise_dilation: int, from_bits: bool = False) -> bool:
    return pairwise_size == 3 and 1 <= pairwise_dilation <= (8 if from_bits else 4)
