#!/usr/bin/env python3
"""Developer tool (GPU box): step time of the evaluation with parts of it switched OFF (-DBXI_ABLATE build; results are wrong by design) --
what the step is sensitive to, without the distortion of the per-wave trace.  Build first:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DBXI_ABLATE -mllvm -amdgpu-kernarg-preload-count=16 \
        -o boxinstseg_amd/lib/libboxinst_hip_ablate.so boxinstseg_amd/csrc/*.hip"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from boxinstseg_amd import _lib, build as hb
hb.LIB_PATH = os.path.join(hb.LIB_DIR, 'libboxinst_hip_ablate.so'); hb.is_stale = lambda: False
from boxinstseg_amd import functional as Fh, synthetic
lib = _lib.load()
dev = torch.device('cuda:0')
ones = torch.ones(2, device=dev)
sets = []
for seed in range(8):
    d = synthetic.cfg2(seed)
    imgs = torch.from_numpy(d['imgs']).to(dev); logits = torch.from_numpy(d['mask_logits']).to(dev)
    gi = torch.from_numpy(d['gt_inds']).to(dev); boxes = [torch.from_numpy(b).to(dev) for b in d['gt_bboxes']]
    batch = Fh._Batch(imgs, d['img_metas'], 10); inst = Fh._Inst(logits, gi, boxes, d['H'], d['W'], 4)
    losses = torch.zeros(2, device=dev); grad = torch.empty_like(inst.logits)
    state = torch.empty(lib.bxi_boxinst_loss_state_bytes(inst.N, inst.h, inst.w), dtype=torch.uint8, device=dev)
    ws = torch.zeros(lib.bxi_boxinst_eval_workspace_bytes(2, 800, 1024, 4, inst.N), dtype=torch.uint8, device=dev)
    sets.append((batch, inst, losses, grad, state, ws, imgs, logits, gi, boxes))
st = torch.cuda.current_stream().cuda_stream
flags = int(os.environ.get('BXI_FLAGS', '0'))
def ev(s):
    batch, inst, losses, grad, state, ws = s[:6]
    rc = lib.bxi_boxinst_eval_f32(C.byref(batch.struct), C.byref(inst.struct), 3, 2, 0.3, 1.0, ones.data_ptr(), ones.data_ptr() + 4,
                                  losses.data_ptr(), grad.data_ptr(), state.data_ptr(), ws.data_ptr(), ws.numel(), flags, st)
    assert rc == 0, rc
NAMES = {1: 'no Lab arithmetic', 2: 'no pair loop', 4: 'no wait for sum W / band flags', 8: 'no wait for predicate words', 16: 'no zero-fill',
         32: 'no image loads', 64: 'no logits stream loads', 128: 'finisher does not wait for arrivals', 256: 'an eighth of the tiles skipped'}
def run(bits, n=3000):
    assert lib.bxi_debug_set_ablate(bits) == 0
    for i in range(200): ev(sets[i % 8])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): ev(sets[i % 8])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
base = run(0)
print('baseline %.2f us' % base)
for bits in (1, 2, 4, 8, 16, 32, 64, 128, 2 | 8, 4 | 8, 2 | 4 | 8, 2 | 4 | 8 | 128, 32 | 1, 16 | 64, 16 | 64 | 2 | 4 | 8, 32 | 1 | 16 | 64, 1 | 2 | 4 | 8 | 16 | 32 | 64 | 128):
    t = run(bits)
    print('%-90s %.2f us (%+.2f)' % (' + '.join(NAMES[b] for b in NAMES if bits & b), t, t - base))
run(0)
