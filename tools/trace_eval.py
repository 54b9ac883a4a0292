#!/usr/bin/env python3
"""Developer tool (GPU box): run one cfg-2 evaluation with the -DBXI_TRACE library and dump per-block
phase timestamps (100 MHz wall clock) to gpurun_out/trace.npz."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from boxinstseg_amd import _lib, build as hb
hb.LIB_PATH = os.path.join(hb.LIB_DIR, 'libboxinst_hip_trace.so')
from boxinstseg_amd import functional as Fh, synthetic
lib = _lib.load()
lib.bxi_debug_set_trace.argtypes = [C.c_void_p]
dev = torch.device('cuda:0')
sets = []
for seed in range(8):
    d = synthetic.cfg2(seed)
    imgs = torch.from_numpy(d['imgs']).to(dev); logits = torch.from_numpy(d['mask_logits']).to(dev)
    gi = torch.from_numpy(d['gt_inds']).to(dev); boxes = [torch.from_numpy(b).to(dev) for b in d['gt_bboxes']]
    batch = Fh._Batch(imgs, d['img_metas'], 10); inst = Fh._Inst(logits, gi, boxes, d['H'], d['W'], 4)
    losses = torch.zeros(2, device=dev); grad = torch.empty_like(inst.logits)
    state = torch.empty(lib.bxi_boxinst_loss_state_bytes(inst.N, inst.h, inst.w), dtype=torch.uint8, device=dev)
    ws = torch.empty(lib.bxi_boxinst_eval_workspace_bytes(2, 800, 1024, 4, inst.N), dtype=torch.uint8, device=dev)
    sets.append((batch, inst, losses, grad, state, ws, imgs, logits, gi, boxes))
st = torch.cuda.current_stream().cuda_stream
def ev(s):
    batch, inst, losses, grad, state, ws = s[:6]
    rc = lib.bxi_boxinst_eval_f32(C.byref(batch.struct), C.byref(inst.struct), 3, 2, 0.3, 1.0, losses.data_ptr(), grad.data_ptr(), state.data_ptr(), ws.data_ptr(), ws.numel(), st)
    assert rc == 0, rc
for i in range(40): ev(sets[i % 8])
torch.cuda.synchronize()
trace = torch.zeros((3, 8192, 8), dtype=torch.int64, device=dev)
assert lib.bxi_debug_set_trace(trace.data_ptr()) == 0
torch.cuda._sleep(int(0.02*2e9)); ev(sets[0]); torch.cuda.synchronize()
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, 'gpurun_out', 'trace.npz'), trace=trace.cpu().numpy())
print('saved', float(sets[0][2][0]), float(sets[0][2][1]))
