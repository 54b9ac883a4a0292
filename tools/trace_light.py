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
    rc = lib.bxi_boxinst_eval_f32(C.byref(batch.struct), C.byref(inst.struct), 3, 2, 0.3, 1.0, ones.data_ptr(), ones.data_ptr() + 4,
                                  losses.data_ptr(), grad.data_ptr(), state.data_ptr(), ws.data_ptr(), ws.numel(), int(os.environ.get('BXI_FLAGS', '0')), st)
    assert rc == 0, rc
for i in range(60): ev(sets[i % 8])
torch.cuda.synchronize()
trace = torch.zeros((4, 8192, 8), dtype=torch.int64, device=dev)
assert lib.bxi_debug_set_trace2(trace.data_ptr()) == 0
torch.cuda._sleep(int(0.02 * 2e9)); ev(sets[3]); torch.cuda.synchronize()
# ---- light trace (-DBXI_TRACE -DBXI_TRACE_LIGHT): first / last stamp of every wave only
t = trace.cpu().numpy().astype(np.float64)
us = lambda x: x * 0.01
N = sets[0][1].N; Sn = 7
two = (int(os.environ.get('BXI_FLAGS', '0')) & 2) != 0 or N > 70
n_tab = (((N + 64) // 64 + 3) // 4) * 4 if two else 0
n_stream = N * Sn * 4
p = t[0]; live = p[:, 0] > 0
t0 = p[live, 0].min()
def q(x): return np.round(np.quantile(x, [0, .1, .5, .9, 1]), 2).tolist() if len(x) else []
sw = p[n_tab:n_tab + n_stream]; pw = p[n_tab + n_stream:]; pw = pw[pw[:, 0] > 0]
print('N', N, 'two launches' if two else 'single launch', '| times in us from the first wave of the launch; quantiles [min, 10%, median, 90%, max]')
print('stream waves', len(sw), 'start', q(us(sw[:, 0] - t0)), 'stream role ends', q(us(sw[sw[:, 7] > 0, 7] - t0)))
print('pool waves  ', len(pw), 'start', q(us(pw[:, 0] - t0)), 'end', q(us(pw[pw[:, 7] > 0, 7] - t0)))
cw = t[2]; cw = cw[cw[:, 0] > 0]
print('pred waves  ', len(cw), 'start', q(us(cw[:, 0] - t0)), 'end', q(us(cw[cw[:, 1] > 0, 1] - t0)))
mw = t[1]; allm = mw[mw[:, 0] > 0]; mw = allm[allm[:, 7] > 0]
print('tile waves  ', len(allm), 'start', q(us(allm[:, 0] - t0)), 'end', q(us(mw[:, 7] - t0)))
ld = t[3][1:]; ld = ld[ld[:, 0] > 0]
print('leaders     ', len(ld), 'start', q(us(ld[:, 0] - t0)), 'dice at', q(us(ld[:, 2] - t0)), 'end', q(us(ld[:, 3] - t0)))
fw = t[3][0]
print('finisher: start %.2f end %.2f' % (us(fw[0] - t0), us(fw[1] - t0)))
