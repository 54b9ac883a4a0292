#!/usr/bin/env python3
"""Developer diagnostic (GPU box): per-stage errors vs the CPU oracle and crude event timings."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as entry
entry.build()
from boxinstseg_amd import synthetic, color_affinity, boxinst_mask_loss, pairwise_nlog
from tests.helpers import oracle_path, hip_loss, grad_report, to_dev

dev = torch.device('cuda:0')
print(torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).gcnArchName)
for name, d in [('cfg1', synthetic.cfg1(0)), ('cfg2', synthetic.cfg2(0))]:
    t0 = time.time(); ref = oracle_path(d); t_or = time.time() - t0
    sim, bits, rgb = color_affinity(torch.from_numpy(d['imgs']).to(dev), d['img_metas'])
    torch.cuda.synchronize()
    sim = sim.cpu().numpy(); bits = bits.cpu().numpy()
    want_bits = np.zeros(bits.shape, np.uint8)
    for k in range(8): want_bits |= ((ref['sim'][:, k] >= 0.3).astype(np.uint8) << k)
    print(f'[{name}] oracle {t_or:.2f}s  sim maxerr {np.abs(sim-ref["sim"]).max():.3e}  bit flips '
          f'{int(np.unpackbits((bits ^ want_bits)[..., None], axis=-1).sum())} of {bits.size*8}')
    try:
        lp, lw, g = hip_loss(d, dev)
        err, ties = grad_report(g, ref['grad'], d['mask_logits'][:, 0])
        raw = np.abs(g - ref['grad']).max() / np.abs(ref['grad']).max()
        print(f'[{name}] loss_prj {lp:.7f} / {ref["loss_prj"]:.7f}   loss_pw {lw:.7f} / {ref["loss_pairwise"]:.7f}   '
              f'grad err {err:.3e} (raw {raw:.3e}, ties {ties})  max|g| {np.abs(ref["grad"]).max():.3e}')
        bad = np.argwhere(np.abs(g - ref['grad']) > 1e-4 * np.abs(ref['grad']).max())
        print(f'[{name}] #bad {len(bad)}', bad[:8].tolist())
        for n, r, c in bad[:5]:
            print('   ', n, r, c, g[n, r, c], ref['grad'][n, r, c])
    except Exception as e:
        import traceback; traceback.print_exc()

# crude timings (events on the current stream), cfg2
d = synthetic.cfg2(0); t = to_dev(d, dev)
import ctypes as C
from boxinstseg_amd import _lib, functional as Fh
lib = _lib.load()
batch = Fh._Batch(t['imgs'], d['img_metas'], 10)
inst = Fh._Inst(t['logits'], t['gt_inds'], t['gt_bboxes'], d['H'], d['W'], 4)
losses = torch.empty(2, device=dev); grad = torch.empty_like(inst.logits)
state = torch.empty(lib.bxi_boxinst_loss_state_bytes(inst.N, inst.h, inst.w), dtype=torch.uint8, device=dev)
ws = torch.empty(lib.bxi_boxinst_eval_workspace_bytes(batch.B, d['H'], d['W'], 4, inst.N), dtype=torch.uint8, device=dev)
rgb = torch.empty((2, 3, 200, 256), dtype=torch.float32, device=dev); aff = torch.empty((2, 200, 256), dtype=torch.uint8, device=dev)
lws = torch.empty(lib.bxi_boxinst_loss_workspace_bytes(inst.N, inst.h, inst.w), dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
def ev_time(fn, n=200, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
f_eval = lambda: lib.bxi_boxinst_eval_f32(C.byref(batch.struct), C.byref(inst.struct), 3, 2, 0.3, 1.0, losses.data_ptr(), grad.data_ptr(), state.data_ptr(), ws.data_ptr(), ws.numel(), st)
f_aff = lambda: lib.bxi_color_affinity_f32(C.byref(batch.struct), 4, 3, 2, 0.3, rgb.data_ptr(), 0, 0, aff.data_ptr(), st)
f_loss = lambda: lib.bxi_boxinst_loss_fwd_bwd_f32(C.byref(inst.struct), aff.data_ptr(), 3, 2, 1.0, losses.data_ptr(), grad.data_ptr(), state.data_ptr(), lws.data_ptr(), lws.numel(), st)
print('status', f_eval(), f_aff(), f_loss())
t0 = time.perf_counter(); 
for _ in range(1000): f_eval()
host = (time.perf_counter() - t0) / 1000 * 1e6; torch.cuda.synchronize()
print(f'eval: {ev_time(f_eval):.2f} us/eval (events, back-to-back)  host enqueue {host:.2f} us')
print(f'color_affinity (pool+affinity): {ev_time(f_aff):.2f} us   loss (main+finalize): {ev_time(f_loss):.2f} us')
x = t['logits'].clone().requires_grad_(True)
def f_mod():
    out = boxinst_mask_loss(x, t['gt_inds'], t['gt_bboxes'], imgs=t['imgs'], img_metas=d['img_metas'])
    (out['loss_prj'] + out['loss_pairwise']).backward(); x.grad = None
print(f'python module path fwd+bwd: {ev_time(f_mod, n=100):.2f} us')
