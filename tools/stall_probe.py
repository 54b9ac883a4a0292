#!/usr/bin/env python3
"""Developer tool (GPU box): look for the multi-second stall of NOTES R5-7 / R6-3.  Batches of 60 back-to-back evaluations of ONE form at one
instance count; every batch slower than `--slow` x the median is printed with its duration, the status words and (a -DBXI_WAITLOG build) the longest
wait of every bounded in-grid wait by site, in polls:
    1 table entry, 2 Lab records (predicate wave), 3 count words (reducer), 4 predicate words (tile wave), 5 sum W / band flags (tile wave),
    6 / 7 band flags (leader), 8 dice + sum W (finisher), 10 arrivals (finisher)
  python tools/stall_probe.py --lib boxinstseg_amd/lib/libboxinst_hip_occ4.so --ipb 4 --form ready_long --secs 20
"""
import argparse, ctypes as C, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument('--lib', default=None)
ap.add_argument('--ipb', type=int, default=4)
ap.add_argument('--form', default='ready_long')
ap.add_argument('--secs', type=float, default=20.0)
ap.add_argument('--slow', type=float, default=10.0)
args = ap.parse_args()
from boxinstseg_amd import build as hb
if args.lib:
    hb.LIB_PATH = os.path.abspath(args.lib); hb.is_stale = lambda: False
import torch
import bench
from boxinstseg_amd import _lib, functional as Fh, synthetic
lib = _lib.load(); L = _lib
dev = torch.device('cuda', 0); torch.cuda.set_device(0)
ones = torch.ones(2, device=dev)
stream = torch.cuda.Stream(device=dev); st = stream.cuda_stream
forms = {'auto': 0, 'ready': L.EVAL_TARGETS_READY, 'two': L.EVAL_TWO_LAUNCHES,
         'ready_long': L.EVAL_TARGETS_READY | L.EVAL_SINGLE_LAUNCH | L.EVAL_TILE_ROWS_8, 'ready_two': L.EVAL_TARGETS_READY | L.EVAL_TWO_LAUNCHES}
form = forms[args.form]
has_log = hasattr(lib, 'bxi_debug_waitlog')
def waitlog(reset=True):
    if not has_log: return None
    buf = (C.c_uint * 16)()
    lib.bxi_debug_waitlog(buf, 1 if reset else 0)
    return {i: int(buf[i]) for i in range(16) if buf[i]}
sets = [bench.EvalSet(lib, Fh, synthetic, dev, seed=7000 + i, inst_per_box=args.ipb, ones=ones, flags=0) for i in range(6)]
N = sets[0].inst.N
off = lib.bxi_boxinst_loss_state_status_offset(N, sets[0].inst.h, sets[0].inst.w)
with torch.cuda.stream(stream):
    if form & L.EVAL_TARGETS_READY:
        for s in sets:
            rc = lib.bxi_boxinst_targets_f32(C.byref(s.batch.struct), s.inst.struct.boxes_per_img_host, s.inst.struct.gt_count_host, 4, 3, 2, 0.3, s.ws.data_ptr(), s.ws.numel(), st)
            assert rc == 0, rc
    for k in range(120):
        rc = lib.bxi_boxinst_eval_f32(*sets[k % 6].eval_args[:-1], C.c_uint(form), st); assert rc == 0, rc
    torch.cuda.synchronize(); waitlog()
    times, slow, it = [], [], 0
    t_end = time.time() + args.secs
    while time.time() < t_end:
        t0 = time.perf_counter()
        for k in range(60):
            rc = lib.bxi_boxinst_eval_f32(*sets[(it + k) % 6].eval_args[:-1], C.c_uint(form), st); assert rc == 0, rc
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        it += 60
        med = statistics.median(times) if len(times) > 5 else None
        if med is not None and dt > args.slow * med:
            sts = [int(s.state[off:off + 4].view(torch.int32).item()) for s in sets]
            rec = {'batch': len(times), 'ms': round(dt, 2), 'median_ms': round(med, 3), 'status': sts, 'waitlog': waitlog()}
            slow.append(rec); print('SLOW', rec, flush=True)
        else:
            times.append(dt)
print({'form': args.form, 'instances': N, 'batches': len(times) + len(slow), 'evaluations': it, 'median_us_per_eval': round(statistics.median(times) / 60 * 1e3, 2),
       'max_ms': round(max(times + [r['ms'] for r in slow]), 2), 'slow_batches': len(slow), 'waitlog_rest': waitlog()})
