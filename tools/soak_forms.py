#!/usr/bin/env python3
"""Developer tool (GPU box): stall soak of the evaluation at the C ABI.  Per (instances per box, form): batches of 60 back-to-back evaluations on
rotating cold sets for SECS seconds; a batch that takes more than ten times the median, or any non-zero status word, is reported.
(Written after a build that ran the 8-row kernels at four workgroups per CU showed 4-second batches with correct results: NOTES R5-7.)

    python tools/soak_forms.py [--ipb 1 2 3 4] [--forms auto,ready] [--secs 8]
"""
import argparse, ctypes as C, json, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ipb', type=int, nargs='+', default=[1, 2, 3, 4])
    ap.add_argument('--forms', type=str, default='auto,ready')
    ap.add_argument('--secs', type=float, default=8.0)
    args = ap.parse_args()
    import __graft_entry__ as entry
    entry.build()
    import bench
    from boxinstseg_amd import _lib, functional as Fh, synthetic
    lib = _lib.load(); L = _lib
    dev = torch.device('cuda', 0); torch.cuda.set_device(0)
    ones = torch.ones(2, device=dev)
    stream = torch.cuda.Stream(device=dev); st = stream.cuda_stream
    forms = {'auto': 0, 'ready': L.EVAL_TARGETS_READY, 'two': L.EVAL_TWO_LAUNCHES,
             'ready_long': L.EVAL_TARGETS_READY | L.EVAL_SINGLE_LAUNCH | L.EVAL_TILE_ROWS_8}
    out = {}
    for ipb in args.ipb:
        sets = [bench.EvalSet(lib, Fh, synthetic, dev, seed=7000 + i, inst_per_box=ipb, ones=ones, flags=0) for i in range(6)]
        N = sets[0].inst.N
        off = lib.bxi_boxinst_loss_state_status_offset(N, sets[0].inst.h, sets[0].inst.w)
        for name in args.forms.split(','):
            form = forms[name]
            with torch.cuda.stream(stream):
                if form & L.EVAL_TARGETS_READY:
                    for s in sets:
                        rc = lib.bxi_boxinst_targets_f32(C.byref(s.batch.struct), s.inst.struct.boxes_per_img_host, s.inst.struct.gt_count_host, 4, 3, 2, 0.3,
                                                         s.ws.data_ptr(), s.ws.numel(), st)
                        assert rc == 0, rc
                else:
                    for s in sets:
                        s.ws.zero_()              # (a targets-ready soak before this one left its records in the workspace)
                for k in range(60):                 # (the first launches of a form load its kernels: tens of milliseconds, not a stall)
                    rc = lib.bxi_boxinst_eval_f32(*sets[k % 6].eval_args[:-1], C.c_uint(form), st)
                    assert rc == 0, rc
                torch.cuda.synchronize()
                times, bad, it = [], [], 0
                t_end = time.time() + args.secs
                while time.time() < t_end:
                    t0 = time.perf_counter()
                    for k in range(60):
                        s = sets[(it + k) % 6]
                        rc = lib.bxi_boxinst_eval_f32(*s.eval_args[:-1], C.c_uint(form), st)
                        assert rc == 0, rc
                    torch.cuda.synchronize()
                    dt = (time.perf_counter() - t0) / 60 * 1e6
                    times.append(dt)
                    sts = [int(s.state[off:off + 4].view(torch.int32).item()) for s in sets]
                    if any(sts):
                        bad.append(dict(it=it, status=sts))
                    it += 60
            med = statistics.median(times)
            slow = [round(t, 1) for t in times if t > 10 * med]
            out[f'n{N}_{name}'] = dict(evaluations=it, median_us=round(med, 2), max_us=round(max(times), 1), slow_batches=slow[:8], nonzero_status=bad[:4])
            print(f'n{N}_{name}', json.dumps(out[f'n{N}_{name}']), flush=True)
        del sets
        torch.cuda.empty_cache()
    print('SOAK_FORMS', json.dumps(out))


if __name__ == '__main__':
    main()
