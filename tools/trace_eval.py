#!/usr/bin/env python3
"""Developer tool (GPU box): run one cfg-2 evaluation with the -DBXI_TRACE library and print per-wave phase timings
(100 MHz wall clock) of prep_kernel / pair_kernel (csrc/fused_eval.hip).  Build first:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DBXI_TRACE -mllvm -amdgpu-kernarg-preload-count=16 \
        -o boxinstseg_amd/lib/libboxinst_hip_trace.so boxinstseg_amd/csrc/*.hip"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from boxinstseg_amd import _lib, build as hb
hb.LIB_PATH = os.path.join(hb.LIB_DIR, os.environ.get('TRACE_LIB', 'libboxinst_hip_trace.so'))
from boxinstseg_amd import functional as Fh, synthetic
lib = _lib.load()
lib.bxi_debug_set_trace2.argtypes = [C.c_void_p]
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
def ev(s):
    batch, inst, losses, grad, state, ws = s[:6]
    rc = lib.bxi_boxinst_eval_f32(C.byref(batch.struct), C.byref(inst.struct), 3, 2, 0.3, 1.0, ones.data_ptr(), ones.data_ptr() + 4,
                                  losses.data_ptr(), grad.data_ptr(), state.data_ptr(), ws.data_ptr(), ws.numel(), int(os.environ.get('BXI_FLAGS', '0')), st)
    assert rc == 0, rc
for i in range(60): ev(sets[i % 8])
torch.cuda.synchronize()
trace = torch.zeros((4, 8192, 8), dtype=torch.int64, device=dev)
assert lib.bxi_debug_set_trace2(trace.data_ptr()) == 0
torch.cuda._sleep(int(0.02 * 2e9)); ev(sets[3]); torch.cuda.synchronize()
t = trace.cpu().numpy().astype(np.float64)
us = lambda x: x * 0.01
N, Sn = 32, 7
pool_first = int(os.environ.get('BXI_POOL_FIRST', '0'))
n_tab = 1 * 4
n_stream = N * Sn * 4
p = t[0]; live = p[:, 0] > 0
n_live = int(np.nonzero(live)[0].max()) + 1
n_pool = n_live - n_tab - n_stream
t0 = p[live, 0].min()
def q(x): return np.round(np.quantile(x, [0, .25, .5, .75, 1]), 2).tolist() if len(x) else []
def d(a, b): return np.where((a > 0) & (b > 0), a - b, np.nan)
def qd(a, b):
    x = d(a, b); x = x[~np.isnan(x)]
    return q(us(x))
print('prep: waves', int(live.sum()), 'start', q(us(p[live, 0] - t0)), 'end', q(us(p[live, 7] - t0)))
tab = p[:n_tab]; tab = tab[tab[:, 7] > 0]
print('  table waves end', q(us(tab[:, 7] - t0)))
if pool_first:
    pw = p[n_tab:n_tab + n_pool]; sw = p[n_tab + n_pool:n_tab + n_pool + n_stream]
else:
    sw = p[n_tab:n_tab + n_stream]; pw = p[n_tab + n_stream:n_live]
sw = sw[sw[:, 7] > 0]; pw = pw[pw[:, 7] > 0]
print('  stream waves:', len(sw), 'start', q(us(sw[:, 0] - t0)), '| loads+zero-fill issued', qd(sw[:, 1], sw[:, 0]), '| data + column max', qd(sw[:, 2], sw[:, 1]),
      '| butterflies', qd(sw[:, 3], sw[:, 2]), '| barrier', qd(sw[:, 4], sw[:, 3]), '| end', q(us(sw[:, 7] - t0)))
print('  pool waves (last item of each):', len(pw), 'start', q(us(pw[:, 0] - t0)), '| loads + denorm at', q(us(pw[:, 1] - t0)), '| barrier 1', qd(pw[:, 2], pw[:, 1]),
      '| Lab f', qd(pw[:, 3], pw[:, 2]), '| barrier 2', qd(pw[:, 4], pw[:, 3]), '| windows (pool-side predicates) done after', qd(pw[:, 5], pw[:, 4]),
      '| count arrival after', qd(pw[:, 6], pw[:, 5]), '| end', q(us(pw[:, 7] - t0)))
prep_end = p[live, 7].max()
one = os.environ.get('BXI_ONE_LAUNCH', '1') != '0'
mw = t[1]; allm = mw[mw[:, 0] > 0]; mw = allm[allm[:, 7] > 0]
cw = t[2]; cw = cw[cw[:, 0] > 0]
ld = t[3][1:]; ld = ld[ld[:, 0] > 0]
if one:
    k0 = t0
    print('single launch: every time below is relative to the first wave of the launch; last front-half wave (table/stream/pool) ends at %.2f' % us(prep_end - t0))
else:
    k0 = min(allm[:, 0].min(), ld[:, 0].min(), cw[:, 0].min())
    print('pair: first wave starts %.2f us after the last prep wave ended' % us(k0 - prep_end))
print('  leaders', len(ld), 'start', q(us(ld[:, 0] - k0)), '| wait + loads + maxima', qd(ld[:, 1], ld[:, 0]), '| sums + dice', qd(ld[:, 2], ld[:, 1]),
      '| coefficients + adds', qd(ld[:, 3], ld[:, 2]), '| dice at', q(us(ld[:, 2] - k0)), '| end', q(us(ld[:, 3] - k0)))
print('  predicate waves', len(cw), 'start', q(us(cw[:, 0] - k0)), '| segment(s) done', qd(cw[:, 1], cw[:, 0]), '| end', q(us(cw[:, 1] - k0)))
print('  tile waves with a tile', len(mw), 'of', len(allm), 'start', q(us(mw[:, 0] - k0)), '| table -> tile', qd(mw[:, 1], mw[:, 0]), '| logits + per-pixel', qd(mw[:, 2], mw[:, 1]),
      '| predicate words + masks', qd(mw[:, 3], mw[:, 2]), '| pair math', qd(mw[:, 5], mw[:, 3]), '| sum W (+ band flags)', qd(mw[:, 4], mw[:, 5]),
      '| adds issued', qd(mw[:, 6], mw[:, 4]), '| arrival issued', qd(mw[:, 7], mw[:, 6]), '| words asked for at', q(us(mw[:, 2] - k0)), '| math done at', q(us(mw[:, 5] - k0)),
      '| end', q(us(mw[:, 7] - k0)))
fw = t[3][0]
print('  finisher: start %.2f end %.2f' % (us(fw[0] - k0), us(fw[1] - k0)))
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, 'gpurun_out', 'trace.npz'), trace=trace.cpu().numpy())
print('losses', float(sets[3][2][0]), float(sets[3][2][1]))
