#!/usr/bin/env python3
"""A/B of the evaluation's forms on ONE box, interleaved: us per evaluation at 32 / 64 / 96 / 128 instances (2 x 800 x 1024 images) for
  auto | single launch | two launches, predicates in the second (round 4) | folded (image-only chain at the tail of the first) |
  folded with 4- / 8-row tiles | targets ready (bxi_boxinst_targets_f32 ahead, BXI_EVAL_TARGETS_READY) in both forms | the targets call alone
on rotating cold input sets, timed like bench.py's `value` (back-to-back calls on one stream, completion polled).

    python tools/ab_forms.py [--ipb 1 2 3 4] [--sets 6] [--steps 400] [--reps 3]
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ipb', type=int, nargs='+', default=[1, 2, 3, 4])
    ap.add_argument('--sets', type=int, default=6)
    ap.add_argument('--steps', type=int, default=400)
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--forms', type=str, default='')
    args = ap.parse_args()
    import __graft_entry__ as entry
    entry.build()
    import bench
    from boxinstseg_amd import _lib, functional as Fh, synthetic
    lib = _lib.load()
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    ones = torch.ones(2, device=dev)
    stream = torch.cuda.Stream(device=dev)
    st = stream.cuda_stream
    L = _lib
    forms = {'auto': 0, 'single': L.EVAL_SINGLE_LAUNCH, 'pair': L.EVAL_TWO_LAUNCHES,
             'pair_r4': L.EVAL_TWO_LAUNCHES | L.EVAL_TILE_ROWS_4, 'pair_r8': L.EVAL_TWO_LAUNCHES | L.EVAL_TILE_ROWS_8,
             'ready': L.EVAL_TARGETS_READY, 'ready_two': L.EVAL_TARGETS_READY | L.EVAL_TWO_LAUNCHES,
             'ready_two_r4': L.EVAL_TARGETS_READY | L.EVAL_TWO_LAUNCHES | L.EVAL_TILE_ROWS_4, 'targets_only': -1,
             'ready_single': L.EVAL_TARGETS_READY | L.EVAL_SINGLE_LAUNCH, 'ready_single_nostay': L.EVAL_TARGETS_READY | L.EVAL_SINGLE_LAUNCH | L.EVAL_SHARED_DEVICE,
             'single_any': L.EVAL_SINGLE_LAUNCH, 'nostay': L.EVAL_SHARED_DEVICE,
             'ready_long': L.EVAL_TARGETS_READY | L.EVAL_SINGLE_LAUNCH | L.EVAL_TILE_ROWS_8}
    if args.forms:
        forms = {k: v for k, v in forms.items() if k in args.forms.split(',')}
    out = {}
    for ipb in args.ipb:
        sets = [bench.EvalSet(lib, Fh, synthetic, dev, seed=7000 + i, inst_per_box=ipb, ones=ones, flags=0) for i in range(args.sets)]
        N = sets[0].inst.N

        def targets(s):
            rc = lib.bxi_boxinst_targets_f32(C.byref(s.batch.struct), s.inst.struct.boxes_per_img_host, s.inst.struct.gt_count_host, 4, 3, 2, 0.3,
                                             s.ws.data_ptr(), s.ws.numel(), st)
            assert rc == 0, rc

        def run(form, n):
            for i in range(n):
                s = sets[i % len(sets)]
                if form < 0:
                    targets(s)
                else:
                    rc = lib.bxi_boxinst_eval_f32(*s.eval_args[:-1], C.c_uint(form), st)
                    assert rc == 0, (form, rc)

        def timed(form):
            with torch.cuda.stream(stream):
                if form >= 0 and (form & L.EVAL_TARGETS_READY):
                    for s in sets:
                        targets(s)
                run(form, 40)
                torch.cuda.synchronize()
                done = torch.cuda.Event()
                t0 = time.perf_counter()
                run(form, args.steps)
                done.record(stream)
                while not done.query():
                    pass
                el = time.perf_counter() - t0
                torch.cuda.synchronize()
            return el / args.steps * 1e6

        res = {k: [] for k in forms}
        status = {}
        for rep in range(args.reps):
            for name, form in forms.items():
                if (form >= 0) and name == 'single' and N > 70:
                    continue
                res[name].append(round(timed(form), 2))
                if form >= 0:
                    off = lib.bxi_boxinst_loss_state_status_offset(N, sets[0].inst.h, sets[0].inst.w)
                    status[name] = sets[0].state[off:off + 8].view(torch.int32).cpu().tolist()
                    if form & L.EVAL_TARGETS_READY:          # leave the workspaces as a fused evaluation expects them
                        pass
        out[f'n{N}'] = {k: {'us': v, 'status': status.get(k)} for k, v in res.items() if v}
        print(f'n{N}', json.dumps(out[f'n{N}']), flush=True)
        del sets
        torch.cuda.empty_cache()
    print('AB_FORMS', json.dumps(out))


if __name__ == '__main__':
    main()
