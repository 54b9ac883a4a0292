import os, sys, time, ctypes as C, math
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/bench.py') else os.environ['GRAFT_REPO_ROOT'])
from boxinstseg_amd import build as hb
lib_path = sys.argv[1] if len(sys.argv) > 1 else None
if lib_path:
    hb.LIB_PATH = os.path.abspath(lib_path); hb.is_stale = lambda: False
import numpy as np, torch
from boxinstseg_amd import _lib, functional as Fh, synthetic
from tests.test_gpu_parity import _abi_eval_setup
lib = _lib.load(); dev = torch.device('cuda:0')
d = synthetic.make_batch(B=2, H=128, W=192, boxes_per_img=4, inst_per_box=2, seed=77, min_box=16, max_box=120)
b = _abi_eval_setup(d, dev, lib, Fh, 0); b['inst'].struct.iter_counter = 0
st = torch.cuda.current_stream(dev).cuda_stream
off = lib.bxi_boxinst_loss_state_status_offset(b['inst'].N, b['inst'].h, b['inst'].w)
has_log = hasattr(lib, 'bxi_debug_waitlog')
def wl():
    if not has_log: return None
    buf = (C.c_uint * 16)(); lib.bxi_debug_waitlog(buf, 1); return {i: int(buf[i]) for i in range(16) if buf[i]}
def ev(flags, thresh=0.3):
    t0 = time.perf_counter()
    rc = lib.bxi_boxinst_eval_f32(C.byref(b['batch'].struct), C.byref(b['inst'].struct), 3, 2, thresh, 1.0, None, None, b['losses'].data_ptr(),
                                  b['grad'].data_ptr(), b['state'].data_ptr(), b['ws'].data_ptr(), b['ws'].numel(), flags, st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    return rc, b['losses'].cpu().numpy().copy(), int(b['state'][off:off + 4].view(torch.int32).item()), round(dt, 2), wl()
def targets(thresh=0.3):
    rc = lib.bxi_boxinst_targets_f32(C.byref(b['batch'].struct), b['inst'].struct.boxes_per_img_host, b['inst'].struct.gt_count_host, d['stride'], 3, 2, thresh, b['ws'].data_ptr(), b['ws'].numel(), st)
    assert rc == 0
print('plain', ev(0))
for form in (0, _lib.EVAL_TWO_LAUNCHES):
    print('form', form)
    print(' never computed', ev(_lib.EVAL_TARGETS_READY | form)); b['ws'].zero_()
    targets(0.5); print(' other threshold', ev(_lib.EVAL_TARGETS_READY | form)); b['ws'].zero_()
    targets(); print(' right', ev(_lib.EVAL_TARGETS_READY | form))
    for bx in b['t']['gt_bboxes']: bx[0, 0] += 24.0; bx[0, 2] += 24.0
    print(' other boxes', ev(_lib.EVAL_TARGETS_READY | form))
    for bx in b['t']['gt_bboxes']: bx[0, 0] -= 24.0; bx[0, 2] -= 24.0
    b['ws'].zero_(); targets(); print(' right again', ev(_lib.EVAL_TARGETS_READY | form))
    print(' own targets', ev(form)); print(' after own', ev(_lib.EVAL_TARGETS_READY | form)); b['ws'].zero_()
