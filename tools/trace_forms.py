#!/usr/bin/env python3
"""Developer tool (GPU box): per-wave phase stamps (100 MHz wall clock) of ONE evaluation in a given form, any instance count.
Needs the -DBXI_TRACE build (full stamps; add -DBXI_TRACE_LIGHT for first / last only):
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DBXI_TRACE -mllvm -amdgpu-kernarg-preload-count=16 \
        -o boxinstseg_amd/lib/libboxinst_hip_trace.so boxinstseg_amd/csrc/*.hip
  IPB=4 BXI_FLAGS=34 python tools/trace_forms.py         (flags: include/boxinst_hip.h BXI_EVAL_*; 32 = targets ready)
Phases of a tile wave: 0 start, 1 tile located, 2 logits in + per-pixel done, 3 predicate words + masks, 5 pair loop done, 4 sum W / bands seen,
6 adds issued, 7 arrived.  Stream / pool waves: 0 start, 1 loads in, 2..4 reductions / barriers, 7 end."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from boxinstseg_amd import _lib, build as hb
hb.LIB_PATH = os.path.join(hb.LIB_DIR, os.environ.get('TRACE_LIB', 'libboxinst_hip_trace.so'))
from boxinstseg_amd import functional as Fh, synthetic
lib = _lib.load()
lib.bxi_debug_set_trace2.argtypes = [C.c_void_p]
if os.environ.get('BXI_ABLATE_BITS'):      # a -DBXI_TRACE -DBXI_ABLATE build: the trace of the launch with parts switched off (tools/ablate.py's bits)
    assert lib.bxi_debug_set_ablate(int(os.environ['BXI_ABLATE_BITS'])) == 0
dev = torch.device('cuda:0')
ones = torch.ones(2, device=dev)
flags = int(os.environ.get('BXI_FLAGS', '0'))
sets = []
for seed in range(6):
    d = synthetic.cfg2(seed, inst_per_box=int(os.environ.get('IPB', '1')))
    imgs = torch.from_numpy(d['imgs']).to(dev); logits = torch.from_numpy(d['mask_logits']).to(dev)
    gi = torch.from_numpy(d['gt_inds']).to(dev); boxes = [torch.from_numpy(b).to(dev) for b in d['gt_bboxes']]
    batch = Fh._Batch(imgs, d['img_metas'], 10); inst = Fh._Inst(logits, gi, boxes, d['H'], d['W'], 4)
    losses = torch.zeros(2, device=dev); grad = torch.empty_like(inst.logits)
    state = torch.empty(lib.bxi_boxinst_loss_state_bytes(inst.N, inst.h, inst.w), dtype=torch.uint8, device=dev)
    ws = torch.zeros(lib.bxi_boxinst_eval_workspace_bytes(2, 800, 1024, 4, inst.N), dtype=torch.uint8, device=dev)
    sets.append((batch, inst, losses, grad, state, ws, imgs, logits, gi, boxes))
st = torch.cuda.current_stream().cuda_stream
def ev(s):
    batch, inst, losses, grad, state, ws = s[:6]
    if flags & _lib.EVAL_TARGETS_READY:
        rc = lib.bxi_boxinst_targets_f32(C.byref(batch.struct), inst.struct.boxes_per_img_host, inst.struct.gt_count_host, 4, 3, 2, 0.3, ws.data_ptr(), ws.numel(), st)
        assert rc == 0, rc
    rc = lib.bxi_boxinst_eval_f32(C.byref(batch.struct), C.byref(inst.struct), 3, 2, 0.3, 1.0, ones.data_ptr(), ones.data_ptr() + 4,
                                  losses.data_ptr(), grad.data_ptr(), state.data_ptr(), ws.data_ptr(), ws.numel(), flags, st)
    assert rc == 0, rc
for i in range(60): ev(sets[i % 6])
torch.cuda.synchronize()
trace = torch.zeros((4, 8192, 8), dtype=torch.int64, device=dev)
assert lib.bxi_debug_set_trace2(trace.data_ptr()) == 0
torch.cuda._sleep(int(0.02 * 2e9)); ev(sets[3]); torch.cuda.synchronize()
t = trace.cpu().numpy().astype(np.float64)
us = lambda x: x * 0.01
live = t[t > 0]
t0 = live.min()
def q(x): return np.round(np.quantile(x, [0, .1, .5, .9, 1]), 2).tolist() if len(x) else []
N = sets[0][1].N
print('N', N, 'flags', flags, '| us from the first stamp of the evaluation; quantiles [min, 10%, median, 90%, max]')
names = {0: 'stream+pool+table waves (kid 0)', 1: 'tile waves (kid 1)', 2: 'predicate waves (kid 2)', 3: 'finisher [0] / leaders (kid 3)'}
for kid in range(4):
    p = t[kid]
    rows = p[(p > 0).any(axis=1)]
    print(names[kid], 'waves with a stamp:', len(rows))
    for ph in range(8):
        col = rows[:, ph]; col = col[col > 0]
        if len(col):
            print('   phase %d: n %5d  %s' % (ph, len(col), q(us(col - t0))))
fw = t[3][0]
print('finisher: start %.2f end %.2f' % (us(fw[0] - t0), us(fw[1] - t0)))
mw = t[1]; mw = mw[(mw[:, 0] > 0) & (mw[:, 7] > 0)]
if len(mw):
    d = mw[:, 7] - mw[:, 0]
    print('tile wave lifetime', q(us(d)), ' with a tile (phase 1 stamped):', int((mw[:, 1] > 0).sum()))
    full = mw[(mw[:, 1] > 0) & (mw[:, 2] > 0) & (mw[:, 5] > 0)]
    if len(full):
        print('  0->1 locate', q(us(full[:, 1] - full[:, 0])), ' 1->2 logits+pixel', q(us(full[:, 2] - full[:, 1])), ' 2->3 pred+masks', q(us(full[:, 3] - full[:, 2])) if (full[:, 3] > 0).all() else '-',
              ' 3->5 pairs', q(us(full[:, 5] - full[:, 3])) if (full[:, 3] > 0).all() else '-', ' 5->4 sumw', q(us(full[:, 4] - full[:, 5])), ' 4->6 adds', q(us(full[:, 6] - full[:, 4])), ' 6->7 arrive', q(us(full[:, 7] - full[:, 6])))
