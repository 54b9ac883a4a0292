#!/usr/bin/env python3
"""bench.py -- BoxInst mask-loss path (projection + colour-similarity pairwise loss), fwd+bwd.

Metric (BASELINE.json): pairwise+projection loss fwd+bwd images/sec @ 2x800x1024 x 32 instances.
One *step* = one loss evaluation on a 2-image batch, through the C ABI of libboxinst_hip.so:
  bxi_boxinst_eval_f32           stage1 (image pool+Lab || logit streaming) -> box (pairwise tiles + projection leaders + loss scalars)
  bxi_boxinst_loss_backward_f32  loss_apply: normalise, add the projection gradient, fold upstream grads
i.e. everything CondInstMaskHead.loss + .backward() do for mask_logits, from the normalised images,
boxes and logits already resident in HBM to loss_prj, loss_pairwise and d(loss)/d(mask_logits).
Inputs rotate over `--sets` independent batches (default 8 x ~36 MB > the 256 MB Infinity Cache)
so that every step reads cold data.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode graph|eager]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  The path has no exchange step (every unit is rank-local):
ranks are weak-scaled replicas; RCCL is used for the barriers, the MAX of the elapsed times and one
all-reduce of the summed loss scalars after the timed region.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=200)
    ap.add_argument('--mode', choices=['graph', 'eager'], default="eager",
                    help='graph: one hipGraph per input set, replayed; eager: direct C-ABI calls')
    ap.add_argument('--sets', type=int, default=8, help='independent input sets rotated through')
    ap.add_argument('--inst-per-box', type=int, default=1)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--no-pipelined', action='store_true', help='skip the multi-stream extra measurement')
    ap.add_argument('--cpu-seconds', type=float, default=15.0)
    return ap.parse_args()


class EvalSet:
    """One synthetic 2x800x1024 / 32-instance batch resident on the device + its output buffers."""

    def __init__(self, lib, Fh, synthetic, dev, seed, inst_per_box):
        d = synthetic.cfg2(seed=seed, inst_per_box=inst_per_box)
        self.d = d
        self.imgs = torch.from_numpy(d['imgs']).to(dev)
        self.logits = torch.from_numpy(d['mask_logits']).to(dev)
        self.gt_inds = torch.from_numpy(d['gt_inds']).to(dev)
        self.boxes = [torch.from_numpy(b).to(dev) for b in d['gt_bboxes']]
        self.batch = Fh._Batch(self.imgs, d['img_metas'], 10)
        self.inst = Fh._Inst(self.logits, self.gt_inds, self.boxes, d['H'], d['W'], d['stride'])
        N, h, w = self.inst.N, self.inst.h, self.inst.w
        self.losses = torch.zeros(2, device=dev)
        self.grad = torch.empty_like(self.inst.logits)
        self.state = torch.empty(lib.bxi_boxinst_loss_state_bytes(N, h, w), dtype=torch.uint8, device=dev)
        self.ws = torch.empty(lib.bxi_boxinst_eval_workspace_bytes(d['B'], d['H'], d['W'], d['stride'], N),
                              dtype=torch.uint8, device=dev)
        self.ones = torch.ones(2, device=dev)       # upstream gradients of loss_prj / loss_pairwise
        # the two C-ABI argument lists, marshalled once (what a training loop that keeps its buffers would do): only the
        # stream is appended per call
        vp = C.c_void_p
        self.eval_args = (C.byref(self.batch.struct), C.byref(self.inst.struct), C.c_int(3), C.c_int(2), C.c_float(0.3),
                          C.c_float(1.0), vp(self.losses.data_ptr()), vp(self.grad.data_ptr()), vp(self.state.data_ptr()),
                          vp(self.ws.data_ptr()), C.c_size_t(self.ws.numel()))
        self.bwd_args = (C.byref(self.inst.struct), vp(self.ones.data_ptr()), vp(self.ones.data_ptr() + 4), C.c_int(2),
                         vp(self.state.data_ptr()), vp(self.grad.data_ptr()))


def main():
    args = parse_args()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 or 'RANK' in os.environ:      # launched by torch.distributed.run (also with one process)
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    else:
        dist = None
        torch.cuda.set_device(0)
    dev = torch.device('cuda', local_rank if dist is not None else 0)

    import __graft_entry__ as entry
    if rank == 0:
        entry.build()
    if dist is not None:
        dist.barrier()
    from boxinstseg_amd import _lib, functional as Fh, synthetic
    lib = _lib.load()
    assert lib.bxi_check_device(dev.index) == 0, 'not a gfx950 device'

    sets = [EvalSet(lib, Fh, synthetic, dev, seed=1000 * rank + i, inst_per_box=args.inst_per_box)
            for i in range(args.sets)]
    stream = torch.cuda.Stream(device=dev)
    SIZE, DIL, THRESH, WARM = 3, 2, 0.3, 1.0

    assert (SIZE, DIL, THRESH, WARM) == (3, 2, 0.3, 1.0)          # what EvalSet marshalled
    f_eval, f_bwd = lib.bxi_boxinst_eval_f32, lib.bxi_boxinst_loss_backward_f32

    def enqueue(s: EvalSet, st: int) -> None:
        rc = f_eval(*s.eval_args, st) or f_bwd(*s.bwd_args, st)
        if rc != 0:
            raise RuntimeError(f'C ABI status {rc}: {_lib.status_string(rc)}')

    # ---- the CPU-baseline leg (rank 0, N == 1), run BEFORE anything is timed: it is the only place that touches oracle/.
    # The oracle is timed as the reported CPU baseline and, as the checker, gates the run: a HIP result that disagrees
    # with it is not timed.
    with torch.cuda.stream(stream):
        enqueue(sets[0], stream.cuda_stream)
    stream.synchronize()
    parity = None
    cpu_leg = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_leg, ref = cpu_baseline(sets[0].d, args.cpu_seconds)
        got = sets[0].losses.cpu().numpy()
        g = sets[0].grad.cpu().numpy()[:, 0]
        from tests.helpers import grad_report
        # rows / columns whose two largest sigmoids coincide in fp32 have no defined arg-max position (DESIGN section 1:
        # the value of the gradient is the same, where it lands is not); they are excluded and counted
        g_err, n_ties = grad_report(g, ref['grad'], sets[0].d['mask_logits'][:, 0])
        parity = dict(loss_prj_rel=float(abs(got[0] - ref['loss_prj']) / abs(ref['loss_prj'])),
                      loss_pairwise_rel=float(abs(got[1] - ref['loss_pairwise']) / abs(ref['loss_pairwise'])),
                      grad_rel_max=float(g_err))
        if max(parity.values()) > 1e-4:
            raise SystemExit(f'parity gate failed, refusing to time: {parity}')
        # Hamming distance of the colour weight mask (sim >= thresh, 8 bits per pooled pixel) vs the oracle's
        from boxinstseg_amd import color_affinity
        _, bits, _ = color_affinity(torch.from_numpy(sets[0].d['imgs']).to(dev), sets[0].d['img_metas'], out_stride=4)
        want_bits = np.zeros(bits.shape, np.uint8)
        for k in range(8):
            want_bits |= ((ref['sim'][:, k] >= THRESH).astype(np.uint8) << k)
        parity['weight_mask_hamming'] = int(np.unpackbits((bits.cpu().numpy() ^ want_bits)[..., None], axis=-1).sum())
        parity['weight_mask_bits'] = int(want_bits.size * 8)
        parity['ambiguous_argmax_lines_excluded'] = int(n_ties)
        parity['grad_rel_max_raw'] = float(np.abs(g - ref['grad']).max() / np.abs(ref['grad']).max())

    # ---- step function -------------------------------------------------------------------------------
    graphs = None
    if args.mode == 'graph':
        graphs = []
        with torch.cuda.stream(stream):
            for s in sets:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=stream):
                    enqueue(s, torch.cuda.current_stream().cuda_stream)
                graphs.append(g)
        stream.synchronize()

    def run(n_steps: int, first: int = 0) -> None:
        with torch.cuda.stream(stream):
            if graphs is not None:
                for i in range(first, first + n_steps):
                    graphs[i % len(graphs)].replay()
            else:
                st = stream.cuda_stream
                for i in range(first, first + n_steps):
                    enqueue(sets[i % len(sets)], st)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    run(args.warmup)
    barrier()
    t0 = time.perf_counter()
    run(args.steps, first=args.warmup)
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    barrier()
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # the two logged scalars, summed over ranks (what mmdet's _parse_losses all-reduces)
        tot = torch.stack([s.losses for s in sets]).sum(0)
        dist.all_reduce(tot)
    images = 2 * args.steps * world
    value = images / elapsed

    result = {
        'metric': 'pairwise+projection loss fwd+bwd images/sec @2x800x1024x32inst',
        'value': value, 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'BoxInst R-50 FPN loss path, 2x800x1024 synthetic batch, '
                               f'{sets[0].inst.N} instances, 1xMI355X per rank (BASELINE configs[1])',
                   'images_per_step': 2, 'instances': sets[0].inst.N, 'map': [sets[0].inst.h, sets[0].inst.w],
                   'launch': args.mode, 'input_sets': args.sets, 'parallelism': f'replicas x{world} (no exchange step)'},
    }
    if parity is not None:
        result['parity'] = parity

    # ---- extra (not `value`): independent evaluations pipelined over several HIP streams ------------------------
    # `value` above is one evaluation at a time (how a training iteration uses the path).  The kernels are
    # latency-bound, so independent batches overlap well; reported for reference only.
    if rank == 0 and world == 1 and args.mode == 'eager' and not args.no_pipelined:
        extra = {}
        for ns in (2, 4):
            streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
            def run_multi(n_steps):
                for i in range(n_steps):
                    st = streams[i % ns]
                    enqueue(sets[i % len(sets)], st.cuda_stream)
            run_multi(64)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            run_multi(args.steps)
            torch.cuda.synchronize(dev)
            el = time.perf_counter() - t1
            extra[f'{ns}_streams'] = {'images_per_s': 2 * args.steps / el, 'us_per_step': el / args.steps * 1e6}
        t1 = time.perf_counter()
        for i in range(args.steps):
            enqueue(sets[i % len(sets)], stream.cuda_stream)
        extra['host_enqueue_us_per_step'] = (time.perf_counter() - t1) / args.steps * 1e6
        torch.cuda.synchronize(dev)
        result['pipelined_throughput_extra'] = extra

    # ---- extra (not `value`): the warm-cache figure SURVEY 8(d) asks to see beside the cold one -------------------
    if rank == 0 and world == 1 and not args.no_pipelined:
        st = stream.cuda_stream
        with torch.cuda.stream(stream):
            for _ in range(64):
                enqueue(sets[0], st)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(args.steps):
                enqueue(sets[0], st)
            torch.cuda.synchronize(dev)
        el = time.perf_counter() - t1
        result['warm_cache_extra'] = {'images_per_s': 2 * args.steps / el, 'us_per_step': el / args.steps * 1e6,
                                      'note': 'ONE input set re-used (inputs L2 / Infinity-Cache resident); `value` rotates over '
                                              f'{args.sets} sets so that every read is cold'}

    # ---- per-kernel durations with HIP events on the launching stream (rank 0, N == 1) ------------
    if rank == 0 and world == 1 and not args.no_kernel_timing:
        result.update(kernel_timing(lib, _lib, sets, stream, enqueue, min(args.steps, 200), elapsed / args.steps * 1e6))
        d0 = sets[0].d
        # compulsory bytes of a whole evaluation given the API's inputs and outputs -- read imgs, write the
        # similarity map, read it again, read the logits, write their gradient (the fused path never materialises the map)
        whole = 12 * d0['B'] * d0['H'] * d0['W'] + 2 * 4 * 8 * d0['B'] * d0['h'] * d0['w'] + 8 * sets[0].inst.N * d0['h'] * d0['w'] + 640
        result['whole_evaluation'] = {
            'algorithmic_bytes_survey_8d': whole, 'achieved_GBps': whole / (elapsed / args.steps) / 1e9,
            'frac_of_hbm_peak': whole / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBPS,
            'wall_time_images_per_s': value,
            'note': 'three dependent launches; the step is latency-bound (DESIGN section 5), the stream kernel is `roofline`'}
    if cpu_leg is not None:
        result['cpu_baseline'] = cpu_leg
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


# algorithmic (compulsory) HBM bytes per launch -- DESIGN.md section 4.  Per unit: 12 B per input pixel
# (3 x f32 read) + 12 B per pooled pixel (Lab written) for the image half of stage1; 4 B per
# instance-pixel read (logits) + 4 B per instance-pixel written (gradient) for the loss, the write
# split between stage1 (zero-fill outside the box tiles) and box_kernel (the box tiles).
def box_tile_fraction(d, dil=2, br=8, bc=64):
    """fraction of the N x h x w gradient written by box_kernel (tiles meeting the dilated box)."""
    h, w, stride = d['h'], d['w'], d['stride']
    boxes = np.concatenate(d['gt_bboxes'], axis=0)
    hit = 0
    for g in d['gt_inds']:
        x1, y1, x2, y2 = [int(v) for v in boxes[g]]
        rr = [r for r in range(h) if y1 <= r * stride + stride // 2 <= y2]
        cc = [c for c in range(w) if x1 <= c * stride + stride // 2 <= x2]
        if not rr or not cc:
            continue
        r0, r1 = max(rr[0] - dil, 0), min(rr[-1] + 1 + dil, h)
        c0, c1 = max(cc[0] - dil, 0), min(cc[-1] + 1 + dil, w)
        tr = range(r0 // br, (r1 - 1) // br + 1)
        tc = range(c0 // bc, (c1 - 1) // bc + 1)
        for a in tr:
            for b in tc:
                hit += (min(h, a * br + br) - a * br) * (min(w, b * bc + bc) - b * bc)
    return hit / float(len(d['gt_inds']) * h * w)


def algorithmic_bytes(d, N):
    """Compulsory HBM bytes per launch: per-unit figure x units of one launch."""
    px_in = d['B'] * d['H'] * d['W']          # input pixels:     12 B read each (3 x f32)
    px_small = d['B'] * d['h'] * d['w']       # pooled pixels:    12 B written each (Lab, 3 x f32)
    ipx = N * d['h'] * d['w']                 # instance-pixels:  4 B read (logit) + 4 B written (gradient)
    f = box_tile_fraction(d)
    return {
        'stage1': 12 * px_in + 12 * px_small + 4 * ipx + 4 * ipx,
        'box': int(4 * ipx * f) * 4 + int(4 * ipx * f),   # box tiles: logits + 3 Lab planes re-read, gradient rewritten
        'loss_apply': int(8 * ipx * f),                   # read-modify-write of the box tiles
    }


def measured_traffic(kernel):
    """HBM bytes per launch from the committed PMC summary (profiles/*_hbm_traffic.json, written by
    tools/summarize_profiles.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_hbm_traffic.json')))
    if not files:
        return None
    try:
        k = json.load(open(files[-1]))['kernels']
        key = {'stage1': 'stage1_kernel', 'box': 'box_kernel', 'loss_apply': 'loss_apply_kernel'}[kernel]
        return float(k[key]['hbm_bytes'])
    except Exception:
        return None


def kernel_timing(lib, _lib, sets, stream, enqueue, steps, step_us):
    """Bracket every kernel launch with HIP events (bxi_set_launch_hook) over `steps` eager steps."""
    events = {}

    def hook(name, phase, st, user):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(stream)
        events.setdefault(name.decode(), []).append(ev)

    cb = _lib.LAUNCH_HOOK(hook)
    run_stream = stream
    with torch.cuda.stream(run_stream):
        for i in range(20):
            enqueue(sets[i % len(sets)], run_stream.cuda_stream)
        run_stream.synchronize()
        # park the stream behind a ~0.1 s spin kernel so that every launch + event below is already queued
        # when the GPU gets to it: the event pairs then bracket kernels that run back to back, exactly as
        # in the timed region (otherwise they would also measure the host's enqueue latency)
        torch.cuda._sleep(int(0.1 * 2.0e9))
        lib.bxi_set_launch_hook(C.cast(cb, C.c_void_p), None)
        try:
            for i in range(steps):
                enqueue(sets[i % len(sets)], run_stream.cuda_stream)
        finally:
            lib.bxi_set_launch_hook(None, None)
    run_stream.synchronize()
    raws = {name: np.array([evs[j].elapsed_time(evs[j + 1]) * 1e3 for j in range(0, len(evs) - 1, 2)])
            for name, evs in events.items()}                                                      # us
    # An event pair adds queue packets of its own.  Their cost is calibrated against the un-instrumented
    # timed region of this same run: (sum of bracketed durations per step - measured step time) / launches
    # per step.  With it the per-kernel figures tile the step exactly as rocprofv3's durations do.
    bracket_us = max(0.0, (sum(float(np.mean(r)) for r in raws.values()) - step_us) / max(len(raws), 1))
    per_kernel = {}
    for name, raw in raws.items():
        durs = raw - bracket_us
        per_kernel[name] = {'avg_us': float(np.mean(durs)), 'median_us': float(np.median(durs)),
                            'min_us': float(np.min(durs)), 'launches': len(durs), 'raw_event_avg_us': float(np.mean(raw))}
    alg = algorithmic_bytes(sets[0].d, sets[0].inst.N)
    for name, v in per_kernel.items():
        b = alg.get(name, 0)
        v['algorithmic_bytes'] = b
        v['achieved_GBps'] = b / (v['avg_us'] * 1e-6) / 1e9 if b else None
    # the roofline kernel = the one that carries the HBM stream (largest algorithmic byte count); the other
    # kernels are latency-bound on ~1-10 MB and are listed with their own figures under "kernels"
    dom = max((k for k in per_kernel if alg.get(k, 0) > 0), key=lambda k: alg[k])
    a = per_kernel[dom]['achieved_GBps']
    return {
        'roofline': {'kernel': dom, 'selection': 'largest HBM byte count of the step', 'bound': 'hbm', 'achieved': a,
                     'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': a / HBM_PEAK_GBPS, 'traffic': measured_traffic(dom),
                     'avg_launch_us': per_kernel[dom]['avg_us'], 'algorithmic_bytes': alg[dom],
                     'timing': 'hipEvent pairs around each launch on the launching stream, minus the per-bracket cost '
                               'calibrated against the un-instrumented step time (event_bracket_us); launches queued '
                               'behind a parked stream, cold input sets; rocprofv3 durations of the same command in profiles/'},
        'kernels': per_kernel,
        'event_bracket_us': bracket_us,
    }


def cpu_baseline(d, budget_s):
    """The reference's CPU loss path (torch CPU ops, oracle/torch_oracle.py) on the host cores, fwd+bwd,
    same workload; bounded to ~budget_s seconds.  Also the C oracle (OpenMP).  Returns (report, the C oracle's
    result for the parity gate)."""
    from oracle import torch_oracle as to
    from tests.helpers import oracle_path
    host_cores = os.cpu_count() or 1
    cores = min(host_cores, 32)            # torch CPU ops stop scaling (and thrash) far below a 256-thread host
    torch.set_num_threads(cores)
    imgs = torch.from_numpy(d['imgs'])
    gi = torch.from_numpy(d['gt_inds'])
    boxes = [torch.from_numpy(b) for b in d['gt_bboxes']]

    def once():
        x = torch.from_numpy(d['mask_logits']).requires_grad_(True)
        out = to.mask_loss(imgs, d['img_metas'], x, gi, boxes)
        (out['loss_prj'] + out['loss_pairwise']).backward()

    once()                                              # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        once()
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 10:
            break
    # (i) the loss given the similarity map and (ii) the colour-affinity precompute, timed separately; and
    # the same evaluation on ONE core (a single run each: bounded)
    t1 = time.perf_counter()
    targets = to.get_targets(imgs, d['img_metas'], boxes)
    t_targets = time.perf_counter() - t1

    def loss_only():
        x = torch.from_numpy(d['mask_logits']).requires_grad_(True)
        out = to.mask_loss(imgs, d['img_metas'], x, gi, boxes, targets=targets)
        (out['loss_prj'] + out['loss_pairwise']).backward()

    loss_only()
    t1 = time.perf_counter()
    loss_only()
    t_loss = time.perf_counter() - t1
    torch.set_num_threads(1)
    t1 = time.perf_counter()
    once()
    t_one = time.perf_counter() - t1
    torch.set_num_threads(cores)
    t_c0 = time.perf_counter()
    ref = oracle_path(d, want_targets=False)
    t_c = time.perf_counter() - t_c0
    ref['sim'] = oracle_path(d)['sim']                  # the similarity map, for the weight-mask comparison of the parity gate
    split = {'loss_given_similarity_ms': t_loss * 1e3, 'colour_affinity_targets_ms': t_targets * 1e3,
             'one_core': {'value': 2 / t_one, 'unit': 'images/s', 'ms_per_eval': t_one * 1e3, 'cores': 1}}
    return {'value': 2 * n / el, 'unit': 'images/s', 'cores': cores, 'kind': 'port',
            'sample': f'{n} full evaluations (targets + loss fwd+bwd) of the same 2x800x1024x{len(d["gt_inds"])} '
                      f'workload with the torch-CPU restatement of the reference path, {cores} threads',
            'ms_per_eval': el / n * 1e3, 'host_cores': host_cores, **split,
            'c_oracle_openmp': {'value': 2 / t_c, 'unit': 'images/s', 'ms_per_eval': t_c * 1e3}}, ref


if __name__ == '__main__':
    main()
