#!/usr/bin/env python3
"""bench.py -- BoxInst mask-loss path (projection + colour-similarity pairwise loss), fwd+bwd.

Metric (BASELINE.json): pairwise+projection loss fwd+bwd images/sec @ 2x800x1024 x 32 instances.
One *step* = one loss evaluation on a 2-image batch, through the C ABI of libboxinst_hip.so:
  bxi_boxinst_eval_f32   ONE launch (eval1) at this size: front half (image pool + Lab || logit streaming || table) -> back half (colour
                         predicates + pair-weight sum per image, projection leaders, pairwise term on the box tiles, both loss scalars,
                         FINISHED gradient); other shapes: the same roles as two launches (prep, pair).  `config.kernels_per_step` says which
i.e. everything CondInstMaskHead.loss + .backward() do for mask_logits, from the normalised images, boxes and logits
already resident in HBM to loss_prj, loss_pairwise and d(loss_prj + loss_pairwise)/d(mask_logits); the two upstream
factors are read from device memory inside the kernels (ones here, as `loss.backward()` seeds them).
Inputs rotate over `--sets` independent batches (default 8 x ~36 MB > the 256 MB Infinity Cache) so that every step
reads cold data.  Before the timed region the GPU is kept busy for >= 0.15 s whatever --warmup says, so that a
20-step run sees the same clocks as a 2000-step run.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode eager|graph]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

`python bench.py --gpus N` without RANK in the environment spawns the N ranks itself (one process per GPU, backend nccl
= RCCL).  The path has no exchange step (every unit is rank-local): ranks are weak-scaled replicas and the timed region
holds no collective (RCCL carries the barriers around it and the MAX over the ranks' clocks).  What a training job all-reduces AROUND
the path is measured beside it, as an extra (`multi_gpu.*_extra`): per evaluation ONE RCCL all-reduce of a buffer the size of the mask
head's `param_conv` parameters (537 065 f32 = 2.15 MB, mmdet/models/detectors/base.py:201-217 + DDP) carrying the two loss scalars in
its tail, issued asynchronously so that it overlaps the next evaluations.
Rank 0 prints ONE JSON line on stdout (whatever libraries print there -- RCCL's version banner -- is sent to stderr).
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
PARAM_CONV_FLOATS = 233 * 256 * 9 + 233      # param_conv.weight [233,256,3,3] + bias [233]: 2.15 MB of f32
PREROLL_S = 0.15


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=200)
    ap.add_argument('--mode', choices=['graph', 'eager'], default='eager',
                    help='graph: one hipGraph per input set, replayed; eager: direct C-ABI calls')
    ap.add_argument('--sets', type=int, default=8, help='independent input sets rotated through')
    ap.add_argument('--inst-per-box', type=int, default=1)
    ap.add_argument('--flags', type=int, default=0, help='BXI_EVAL_* flags of every timed evaluation (0 = the library chooses)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--no-extras', '--no-pipelined', dest='no_extras', action='store_true',
                    help='skip the extra measurements (multi-stream, warm cache, autograd / module API)')
    ap.add_argument('--cpu-seconds', type=float, default=15.0)
    ap.add_argument('--spawn', action='store_true', help='spawn the ranks even for --gpus 1 (exercises the launcher)')
    return ap.parse_args(argv)


class EvalSet:
    """One synthetic 2x800x1024 / 32-instance batch resident on the device + its output buffers."""

    def __init__(self, lib, Fh, synthetic, dev, seed, inst_per_box, ones, flags=0):
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
        # zeroed ONCE, as the C ABI asks (the workspace carries the evaluation's tag counter); never touched by the host again
        self.ws = torch.zeros(lib.bxi_boxinst_eval_workspace_bytes(d['B'], d['H'], d['W'], d['stride'], N),
                              dtype=torch.uint8, device=dev)
        # the C-ABI argument lists, marshalled once (what a training loop that keeps its buffers does): only the
        # stream is appended per call
        vp = C.c_void_p
        self.eval_args = (C.byref(self.batch.struct), C.byref(self.inst.struct), C.c_int(3), C.c_int(2), C.c_float(0.3),
                          C.c_float(1.0), vp(ones.data_ptr()), vp(ones.data_ptr() + 4), vp(self.losses.data_ptr()),
                          vp(self.grad.data_ptr()), vp(self.state.data_ptr()), vp(self.ws.data_ptr()),
                          C.c_size_t(self.ws.numel()), C.c_uint(flags))
        self.rescale_args = (C.byref(self.inst.struct), vp(ones.data_ptr()), vp(ones.data_ptr() + 4), C.c_int(2),
                             vp(self.state.data_ptr()), vp(self.grad.data_ptr()))


def free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _spawned(rank: int, argv, world: int, port: int):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    worker(parse_args(argv))


def main():
    args = parse_args()
    if 'RANK' not in os.environ and (args.gpus > 1 or args.spawn):
        # the launcher of the reference is tools/dist_train.sh:10-20 (torch.distributed.launch, one process per GPU)
        if torch.cuda.device_count() < args.gpus:
            raise SystemExit(f'--gpus {args.gpus} but only {torch.cuda.device_count()} visible device(s)')
        import __graft_entry__ as entry
        entry.build()                          # once, before the ranks start
        import torch.multiprocessing as mp
        mp.spawn(_spawned, args=(sys.argv[1:], args.gpus, free_port()), nprocs=args.gpus, join=True)
        return
    worker(args)


def worker(args):
    # stdout carries ONE JSON line: everything else a library prints there (RCCL's version banner at communicator creation) goes to
    # stderr -- file descriptor 1 is pointed at stderr for the duration, the JSON is written to a duplicate of the original
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if 'RANK' in os.environ:                   # one process per GPU: torch.distributed.run, or spawned above
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f'--gpus {args.gpus} but the process group has {dist.get_world_size()} ranks')
    else:
        dist = None
        torch.cuda.set_device(0)
        if args.gpus != 1:
            raise SystemExit('--gpus > 1 needs one process per GPU')
    dev = torch.device('cuda', local_rank if dist is not None else 0)

    import __graft_entry__ as entry
    if rank == 0:
        entry.build()
    if dist is not None:
        dist.barrier()
    from boxinstseg_amd import _lib, functional as Fh, synthetic
    lib = _lib.load()
    assert lib.bxi_check_device(dev.index) == 0, 'not a gfx950 device'

    ones = torch.ones(2, device=dev)            # upstream gradients of loss_prj / loss_pairwise
    sets = [EvalSet(lib, Fh, synthetic, dev, seed=1000 * rank + i, inst_per_box=args.inst_per_box, ones=ones, flags=args.flags)
            for i in range(args.sets)]
    stream = torch.cuda.Stream(device=dev)
    f_eval, f_rescale = lib.bxi_boxinst_eval_f32, lib.bxi_boxinst_grad_rescale_f32

    def enqueue(s: EvalSet, st: int, shared: bool = False) -> None:
        # shared: evaluations in flight side by side on several streams -- the caller says so (BXI_EVAL_SHARED_DEVICE), the library
        # does not guess
        rc = f_eval(*s.eval_args, st) if not shared else f_eval(*s.eval_args[:-1], C.c_uint(args.flags | _lib.EVAL_SHARED_DEVICE), st)
        if rc != 0:
            raise RuntimeError(f'C ABI status {rc}: {_lib.status_string(rc)}')

    # ---- the CPU-baseline leg (rank 0, N == 1), run BEFORE anything is timed: it is the only place that touches oracle/.
    # The oracle is timed as the reported CPU baseline and, as the checker, gates the run: a HIP result that disagrees
    # with it is not timed.
    with torch.cuda.stream(stream):
        enqueue(sets[0], stream.cuda_stream)
    stream.synchronize()
    off = lib.bxi_boxinst_loss_state_status_offset(sets[0].inst.N, sets[0].inst.h, sets[0].inst.w)
    status = sets[0].state[off:off + 8].view(torch.int32).cpu().tolist()
    if status[0] != 0:
        raise SystemExit(f'in-kernel wait timed out (status {status[0]}): refusing to time')
    parity = None
    rank_parity = None
    cpu_leg = None
    if dist is not None and not args.no_cpu_baseline:
        # one process per GPU (any N): EVERY rank checks its own first set against the C oracle before anything is timed (the checker
        # only: the timed CPU baseline stays a rank-0, N == 1 figure), and one rank's mismatch fails the whole run
        rank_parity = parity = parity_gate_only(sets[0])
        bad = torch.tensor([0.0 if parity['ok'] else 1.0], device=dev)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if float(bad.item()) != 0.0:
            raise SystemExit(f'parity gate failed on some rank (this rank {rank}: {parity}), refusing to time')
    c_oracle_leg = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the checker first (the C oracle, ~0.1 s): nothing is timed before the HIP result agrees with it.  The TIMED CPU baseline (torch-CPU
        # restatement, tens of seconds on up to 64 threads) runs at the very end: its worker threads otherwise slow every host-bound
        # figure measured after it (module_api: 77 -> 145 us per call behind a 64-thread sample).
        ref, c_oracle_leg = oracle_gate(sets[0].d)
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
            want_bits |= ((ref['sim'][:, k] >= 0.3).astype(np.uint8) << k)
        parity['weight_mask_hamming'] = int(np.unpackbits((bits.cpu().numpy() ^ want_bits)[..., None], axis=-1).sum())
        parity['weight_mask_bits'] = int(want_bits.size * 8)
        parity['ambiguous_argmax_lines_excluded'] = int(n_ties)
        parity['grad_rel_max_raw'] = float(np.abs(g - ref['grad']).max() / np.abs(ref['grad']).max())
        parity['tile_rows'] = status[1]

    # ---- what a training job all-reduces around the path (N > 1): one bucket per evaluation -------------------------
    bucket = None
    if dist is not None:
        bucket = [torch.zeros(PARAM_CONV_FLOATS + 2, device=dev) for _ in range(args.sets)]

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

    st = stream.cuda_stream

    def run(n_steps: int, first: int = 0, comm: bool = True) -> None:
        # (called with `stream` current: the measurement below runs inside ONE `with torch.cuda.stream(stream)` block, so that the
        # timed region holds the steps and not the harness's context switch)
        works = []
        for i in range(first, first + n_steps):
            k = i % len(sets)
            if graphs is not None:
                graphs[k].replay()
            else:
                enqueue(sets[k], st)
            if dist is not None and comm:
                bucket[k][-2:].copy_(sets[k].losses, non_blocking=True)      # the two logged scalars ride in the bucket
                works.append(dist.all_reduce(bucket[k], async_op=True))
                if len(works) > 2 * len(sets):                                # bound the work queue, keep buffers safe
                    works.pop(0).wait()
        for wk in works:
            wk.wait()

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def preroll():                       # >= PREROLL_S of back-to-back evaluations: clocks up, caches in steady state
        t_end = time.perf_counter() + PREROLL_S
        n = 0
        while time.perf_counter() < t_end:
            run(256, first=n, comm=False)
            torch.cuda.synchronize(dev)
            n += 256

    prev_stream = torch.cuda.current_stream(dev)
    torch.cuda.set_stream(stream)          # `stream` stays current from here to the end of the timed region
    run(args.warmup)
    # launches per step, counted: the launch hook is called before and after every kernel launch the library makes
    calls = []
    cb_count = _lib.LAUNCH_HOOK(lambda name, phase, st, user: calls.append(name))
    lib.bxi_dev_set_launch_hook(C.cast(cb_count, C.c_void_p), None)
    try:
        enqueue(sets[0], stream.cuda_stream)
    finally:
        lib.bxi_dev_set_launch_hook(None, None)
    torch.cuda.synchronize(dev)
    launches_per_step = len(calls) // 2
    launched = sorted({c.decode() for c in calls})
    preroll()
    barrier()
    done = torch.cuda.Event()
    t0 = time.perf_counter()
    run(args.steps, first=args.warmup, comm=False)   # the path has no exchange step: no collective in the timed region
    done.record(stream)
    while not done.query():              # completion observed by polling: a blocking synchronize adds its wake-up latency
        pass                             # (10-30 us: 5 % of a 20-step run) to the interval; the synchronize still follows
    elapsed = time.perf_counter() - t0
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.set_stream(prev_stream)
    local_elapsed = elapsed
    allreduce_us = None
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # the same all-reduce alone, back to back (what one of them costs when nothing hides it)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        with torch.cuda.stream(stream):
            for i in range(100):
                dist.all_reduce(bucket[i % len(bucket)])
        torch.cuda.synchronize(dev)
        allreduce_us = (time.perf_counter() - t1) / 100 * 1e6
        # extra (not `value`): the same steps with what a training job all-reduces AROUND the path riding along -- per evaluation one
        # asynchronous RCCL all-reduce of a param_conv-sized bucket carrying the two loss scalars, overlapping the next evaluations
        preroll()
        barrier()
        torch.cuda.set_stream(stream)
        t1 = time.perf_counter()
        run(args.steps, comm=True)
        torch.cuda.synchronize(dev)
        with_comm = time.perf_counter() - t1
        torch.cuda.set_stream(prev_stream)
    images = 2 * args.steps * world
    value = images / elapsed
    step_us = elapsed / args.steps * 1e6
    sol = speed_of_light(lib, sets, dev, stream, args.steps, args.warmup) if rank == 0 else None

    result = {
        'metric': 'pairwise+projection loss fwd+bwd images/sec @2x800x1024x32inst',
        'value': value, 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'BoxInst R-50 FPN loss path, 2x800x1024 synthetic batch, '
                               f'{sets[0].inst.N} instances, 1xMI355X per rank (BASELINE configs[1])',
                   'images_per_step': 2, 'instances': sets[0].inst.N, 'map': [sets[0].inst.h, sets[0].inst.w],
                   'launch': args.mode, 'launches_per_step': launches_per_step, 'kernels_per_step': launched, 'input_sets': args.sets, 'preroll_s': PREROLL_S,
                   'parallelism': f'replicas x{world} (no exchange step inside the path)'},
    }
    if dist is not None:
        props = torch.cuda.get_device_properties(dev)
        mine = {'rank': rank, 'device': dev.index, 'name': props.name, 'arch': getattr(props, 'gcnArchName', None),
                'pci_bus_id': '%04x:%02x:%02x' % (getattr(props, 'pci_domain_id', 0), getattr(props, 'pci_bus_id', 0), getattr(props, 'pci_device_id', 0)),
                'uuid': str(getattr(props, 'uuid', ''))}
        if rank_parity is not None:
            mine['parity'] = rank_parity
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)
        result['multi_gpu'] = {
            'world_size': dist.get_world_size(), 'backend': 'nccl (RCCL)', 'ranks': ranks,
            'distinct_devices': len({(r['pci_bus_id'], r['uuid']) for r in ranks}),
            'per_rank_images_per_s': 2 * args.steps / local_elapsed,
            'per_rank_images_per_s_with_bucket_allreduce_extra': 2 * args.steps / with_comm,
            'allreduce_per_evaluation_extra': f'{(PARAM_CONV_FLOATS + 2) * 4} B (param_conv-sized gradient bucket + the 2 loss scalars), '
                                              'async, overlapping the following evaluations; NOT part of `value`: the path has no '
                                              'exchange step, this is what DDP does around it',
            'allreduce_alone_us': allreduce_us}
    if parity is not None:
        result['parity'] = parity

    extras = rank == 0 and world == 1 and not args.no_extras
    # the extras are not `value`: each runs its own count of steps (the driver's 20-step invocation is too short a sample for them)
    n_extra = max(args.steps, 300)
    # ---- extra (not `value`): independent evaluations pipelined over several HIP streams ------------------------
    if extras and args.mode == 'eager':
        extra = {}
        for ns in (2, 4):
            streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
            def run_multi(n_steps):
                for i in range(n_steps):
                    enqueue(sets[i % len(sets)], streams[i % ns].cuda_stream, shared=True)     # a set (and its workspace) stays on one stream
            run_multi(64)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            run_multi(n_extra)
            torch.cuda.synchronize(dev)
            el = time.perf_counter() - t1
            extra[f'{ns}_streams'] = {'images_per_s': 2 * n_extra / el, 'us_per_step': el / n_extra * 1e6}
        torch.cuda.synchronize(dev)
        n_host = 200              # few enough that the queue never fills: the host's own cost per call
        t1 = time.perf_counter()
        for i in range(n_host):
            enqueue(sets[i % len(sets)], stream.cuda_stream)
        extra['host_enqueue_us_per_step'] = (time.perf_counter() - t1) / n_host * 1e6
        torch.cuda.synchronize(dev)
        result['pipelined_throughput_extra'] = extra

    # ---- extra: the warm-cache figure SURVEY 8(d) asks to see beside the cold one ---------------------------------
    if extras:
        st = stream.cuda_stream
        with torch.cuda.stream(stream):
            for _ in range(64):
                enqueue(sets[0], st)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(n_extra):
                enqueue(sets[0], st)
            torch.cuda.synchronize(dev)
        el = time.perf_counter() - t1
        result['warm_cache_extra'] = {'images_per_s': 2 * n_extra / el, 'us_per_step': el / n_extra * 1e6,
                                      'note': 'ONE input set re-used (inputs L2 / Infinity-Cache resident); `value` rotates over '
                                              f'{args.sets} sets so that every read is cold'}
        # ---- extra: evaluation + the rescale launch autograd's backward() adds (returns at once for unit factors) ----
        with torch.cuda.stream(stream):
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for i in range(n_extra):
                s = sets[i % len(sets)]
                enqueue(s, st)
                f_rescale(*s.rescale_args, st)
            torch.cuda.synchronize(dev)
        el = time.perf_counter() - t1
        result['autograd_backward_extra'] = {'images_per_s': 2 * n_extra / el, 'us_per_step': el / n_extra * 1e6,
                                             'note': 'bxi_boxinst_eval_f32 + bxi_boxinst_grad_rescale_f32 (one launch more than `value`): the C-ABI '
                                                     'sequence behind loss() + backward() when the upstream factors are only '
                                                     'known at backward time'}
        result['extras'] = {}
        try:
            result['extras'].update(targets_ahead(lib, _lib, sets, dev, stream, n_extra))
        except Exception as e:
            result['extras']['targets_ahead_error'] = f'{type(e).__name__}: {e}'[:300]
        for ipb in (2, 4):           # the shape real training runs: topk_per_img=64 x samples_per_gpu=2 -> up to 128 instances per evaluation
            result['extras'][f'n{32 * ipb}'] = instance_count_extra(lib, _lib, Fh, synthetic, dev, stream, ones, ipb, args.flags, gate=not args.no_cpu_baseline)
        result['extras'].update(rows_extra())
        result['head_fused_extra'] = head_fused(lib, Fh, sets, dev, stream, n_extra)
        result['module_api'] = module_api(sets, dev, 300)

    # ---- per-kernel durations with HIP events on the launching stream (rank 0, N == 1) ------------
    if rank == 0 and world == 1 and not args.no_kernel_timing:
        result.update(kernel_timing(lib, _lib, sets, stream, enqueue, min(max(args.steps, 50), 200), step_us, status[1]))
    if sol is not None:
        # the floor of THIS launch shape on this part, measured in the same run on the same cold sets: the evaluation's bytes moved by
        # the evaluation's grid with no arithmetic and no workgroup waiting for another (csrc/sol_eval.hip)
        rf = result.setdefault('roofline', {'bound': 'hbm', 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'achieved': survey_bytes(sets[0].d, sets[0].inst.N) / (step_us * 1e-6) / 1e9,
                                            'frac': survey_bytes(sets[0].d, sets[0].inst.N) / (step_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 'traffic': None,
                                            'algorithmic_bytes': survey_bytes(sets[0].d, sets[0].inst.N), 'time_us': step_us})
        rf['sol_us'] = sol['us_per_step']
        rf['frac_of_sol'] = sol['us_per_step'] / step_us
        rf['sol'] = sol

    # The driver's record keeps `config`, `roofline` and `cpu_baseline` of this line (and the last few KB of its text): the figures the
    # documents quote besides the headline -- the shapes training runs (64 / 128 instances), the loss with the targets made ahead, the op-level
    # pairwise_nlog kernels -- go INTO `roofline` as short scalars (us = microseconds per call, frac = algorithmic bytes / time / HBM peak,
    # sol_us = the same bytes moved by a copy in the same run, ok = the oracle gate of that shape).
    if 'roofline' in result and 'extras' in result:
        ex, shapes = result['extras'], {}
        def r4(v):
            return None if v is None else round(float(v), 4)
        for key in ('n64', 'n128'):
            e = ex.get(key)
            if isinstance(e, dict) and 'us_per_step' in e:
                par = e.get('parity') or {}
                shapes[key] = {'us': r4(e['us_per_step']), 'frac': r4(e['frac']), 'sol_us': r4(e.get('sol_us')),
                               'ok': bool(par.get('ok')) if par else None,
                               'given_targets_us': r4((e.get('loss_given_targets') or {}).get('us_per_step'))}
        lg = ex.get('loss_given_targets')
        if isinstance(lg, dict) and 'us_per_step' in lg:
            shapes['n32_given_targets'] = {'us': r4(lg['us_per_step']), 'frac': r4(lg.get('frac'))}
        po = ex.get('pairwise_op')
        if isinstance(po, dict) and 'bwd_us' in po:
            both = (po['fwd_bytes'] + po['bwd_bytes']) / ((po['fwd_us'] + po['bwd_us']) * 1e-6) / 1e9 / HBM_PEAK_GBPS
            shapes['pairwise_op'] = {'fwd_us': r4(po['fwd_us']), 'bwd_us': r4(po['bwd_us']), 'fwd_frac': r4(po['fwd_frac']), 'bwd_frac': r4(po['bwd_frac']),
                                     'fwd_sol_us': r4(po.get('fwd_sol_us')), 'bwd_sol_us': r4(po.get('bwd_sol_us')),
                                     'bwd_frac_of_sol': r4(po.get('bwd_frac_of_sol')), 'fwd_bwd_frac': r4(both)}
        ma = result.get('module_api')
        if isinstance(ma, dict) and 'us_per_call' in ma:
            shapes['module_api_us'] = r4(ma['us_per_call'])
        result['roofline']['shapes'] = shapes

    if c_oracle_leg is not None:
        cpu_leg = cpu_baseline(sets[0].d, args.cpu_seconds)
        cpu_leg['c_oracle_openmp'] = c_oracle_leg
        result['cpu_baseline'] = cpu_leg
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        for key in ('roofline', 'cpu_baseline'):          # last in the line: the driver keeps the line's tail
            if key in result:
                result[key] = result.pop(key)
        json_out.write(json.dumps(result) + '\n')
        json_out.flush()


def speed_of_light(lib, sets, dev, stream, steps, warmup):
    """The byte-only stand-in of the single-launch evaluation (bxi_dev_sol_eval_f32, csrc/sol_eval.hip): the same grid -- workgroups per
    role, four per CU -- performing the same loads and stores on the same rotating cold input sets, with no arithmetic and no workgroup
    waiting for another; timed exactly like `value` (K back-to-back launches on one stream, un-instrumented wall time)."""
    vp = C.c_void_p
    packs = []
    for s in sets:
        d = s.d
        scratch = torch.empty(20 * d['B'] * d['h'] * d['w'] + 256, dtype=torch.uint8, device=dev)
        g = torch.empty_like(s.inst.logits)
        packs.append((vp(s.imgs.data_ptr()), d['B'], d['H'], d['W'], vp(s.inst.logits.data_ptr()), s.inst.N, s.inst.h, s.inst.w, vp(g.data_ptr()),
                      vp(scratch.data_ptr()), C.c_size_t(scratch.numel()), scratch, g))
    st = stream.cuda_stream
    f = lib.bxi_dev_sol_eval_f32

    def run(n):
        for i in range(n):
            pk = packs[i % len(packs)]
            rc = f(*pk[:11], st)
            if rc != 0:
                raise RuntimeError(f'bxi_dev_sol_eval_f32: status {rc}')
    n = max(steps, 300)
    with torch.cuda.stream(stream):
        run(max(warmup, 50))
        torch.cuda.synchronize(dev)
        done = torch.cuda.Event()
        t0 = time.perf_counter()
        run(n)
        done.record(stream)
        while not done.query():
            pass
        el = time.perf_counter() - t0
        torch.cuda.synchronize(dev)
    d0 = sets[0].d
    moved = algorithmic_bytes(d0, sets[0].inst.N, 4)['eval1']
    return {'us_per_step': el / n * 1e6, 'steps': n, 'bytes_moved_per_launch': moved, 'GBps': moved / (el / n) / 1e9,
            'what': 'ONE launch with eval1_kernel\'s grid (same workgroups per role, 4 per CU) doing the evaluation\'s loads and stores, no arithmetic, '
                    'no in-grid dependency; same cold sets, timed like `value` (launch boundary included)'}


def instance_count_extra(lib, _lib, Fh, synthetic, dev, stream, ones, inst_per_box, flags, n_sets=6, steps=400, gate=True):
    """Extra (not `value`): the same evaluation with 2 / 4 instances per box -- N = 64 / 128, what configs/boxinst/boxinst_r50_fpn_1x_coco.py:65,125
    (topk_per_img=64, samples_per_gpu=2) gives condinst_head.py:1190-1225 -- timed like `value` on rotating cold sets, with its own
    roofline: SURVEY 8(d)'s compulsory bytes at that N / step time / HBM peak."""
    sets = [EvalSet(lib, Fh, synthetic, dev, seed=7000 + i, inst_per_box=inst_per_box, ones=ones, flags=flags) for i in range(n_sets)]
    st = stream.cuda_stream
    f_eval = lib.bxi_boxinst_eval_f32
    calls = []
    cb = _lib.LAUNCH_HOOK(lambda name, phase, s_, user: calls.append(name.decode()))

    def run(n):
        for i in range(n):
            rc = f_eval(*sets[i % n_sets].eval_args, st)
            if rc != 0:
                raise RuntimeError(f'C ABI status {rc}')
    with torch.cuda.stream(stream):
        lib.bxi_dev_set_launch_hook(C.cast(cb, C.c_void_p), None)
        try:
            run(1)
        finally:
            lib.bxi_dev_set_launch_hook(None, None)
        run(60)
        torch.cuda.synchronize(dev)
        done = torch.cuda.Event()
        t0 = time.perf_counter()
        run(steps)
        done.record(stream)
        while not done.query():
            pass
        el = time.perf_counter() - t0
        torch.cuda.synchronize(dev)
    s0 = sets[0]
    off = lib.bxi_boxinst_loss_state_status_offset(s0.inst.N, s0.inst.h, s0.inst.w)
    status = s0.state[off:off + 8].view(torch.int32).cpu().tolist()
    nbytes = survey_bytes(s0.d, s0.inst.N)
    us = el / steps * 1e6
    out = {'instances': s0.inst.N, 'us_per_step': us, 'images_per_s': 2 * steps / el, 'algorithmic_bytes': nbytes,
           'achieved_GBps': nbytes / (us * 1e-6) / 1e9, 'frac': nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
           'kernels_per_step': sorted(set(calls)), 'status': status[0], 'tile_rows': status[1], 'steps': steps, 'input_sets': n_sets}
    if gate:
        # (the checker, as in the cpu_baseline leg: the C oracle on the set the last timed step of set 0 left its results for)
        with torch.cuda.stream(stream):
            run(1)
            torch.cuda.synchronize(dev)
        out['parity'] = parity_gate_only(s0)
    # the floor of this shape: its bytes in one launch with nothing else in it (as roofline.sol_us at 32 instances)
    sol = speed_of_light(lib, sets, dev, stream, 300, 40)
    out['sol_us'] = sol['us_per_step']
    out['frac_of_sol'] = sol['us_per_step'] / us
    # ... and with the targets ahead (bxi_boxinst_targets_f32 once per set, then BXI_EVAL_TARGETS_READY): what a training iteration
    # that calls prepare_targets at the top of forward_train pays at loss time
    ta = targets_ahead(lib, _lib, sets, dev, stream, steps, with_pipeline=s0.inst.N >= 128)
    out['loss_given_targets'] = ta['loss_given_targets']
    for k_ in ('two_stage_pipelined', 'under_a_backbone_standin'):
        if k_ in ta:
            out[k_] = ta[k_]
    return out


def targets_ahead(lib, _lib, sets, dev, stream, steps, with_pipeline=True):
    """Extras (not `value`): the evaluation split where the reference's data flow allows it -- get_targets needs only imgs + gt_bboxes
    (condinst_head.py:1298-1299; both exist before the backbone runs, condinst.py:53 vs :73):
      targets_only        bxi_boxinst_targets_f32 alone (two launches): SURVEY 8(d)'s K1 bytes, 22 937 600 at 2 x 800 x 1024
      loss_given_targets  bxi_boxinst_eval_f32 with BXI_EVAL_TARGETS_READY (logit stream, leaders, tiles, finisher): K2 + K3 bytes,
                          16 384 640 at 32 instances -- the GPU leg beside cpu_baseline.loss_given_similarity_ms
      two_stage_pipelined targets of set i + 1 on a second stream under the loss of set i (eager, events between the streams), whole-evaluation
                          bytes per step; host-bound at 32 instances (four launches + four event calls per step from one thread)
    each timed like `value` on the rotating cold sets; the split evaluation is checked bit for bit against the un-split one."""
    vp = C.c_void_p
    st = stream.cuda_stream
    N = sets[0].inst.N
    d0 = sets[0].d
    f_eval, f_tgt = lib.bxi_boxinst_eval_f32, lib.bxi_boxinst_targets_f32
    ready = _lib.EVAL_TARGETS_READY

    def targets(s, on):
        rc = f_tgt(C.byref(s.batch.struct), s.inst.struct.boxes_per_img_host, s.inst.struct.gt_count_host, 4, 3, 2, 0.3, s.ws.data_ptr(), s.ws.numel(), on)
        if rc != 0:
            raise RuntimeError(f'bxi_boxinst_targets_f32: status {rc}')

    def ev(s, flags, on):
        rc = f_eval(*s.eval_args[:-1], C.c_uint(flags), on)
        if rc != 0:
            raise RuntimeError(f'bxi_boxinst_eval_f32: status {rc}')

    def timed(fn, n):
        with torch.cuda.stream(stream):
            for i in range(40):
                fn(i)
            torch.cuda.synchronize(dev)
            done = torch.cuda.Event()
            t0 = time.perf_counter()
            for i in range(n):
                fn(i)
            done.record(stream)
            while not done.query():
                pass
            el = time.perf_counter() - t0
            torch.cuda.synchronize(dev)
        return el / n * 1e6

    out = {}
    # bit-for-bit: the un-split evaluation of set 0, then targets + the split one
    with torch.cuda.stream(stream):
        ev(sets[0], 0, st)
        torch.cuda.synchronize(dev)
        want = (sets[0].losses.clone(), sets[0].grad.clone())
        targets(sets[0], st)
        ev(sets[0], ready, st)
        torch.cuda.synchronize(dev)
    off = lib.bxi_boxinst_loss_state_status_offset(N, sets[0].inst.h, sets[0].inst.w)
    status = sets[0].state[off:off + 8].view(torch.int32).cpu().tolist()
    same = bool(torch.equal(want[0], sets[0].losses) and torch.equal(want[1], sets[0].grad))
    us_t = timed(lambda i: targets(sets[i % len(sets)], st), steps)
    k1 = 12 * d0['B'] * d0['H'] * d0['W'] + 4 * 8 * d0['B'] * d0['h'] * d0['w']
    out['targets_only'] = {'us_per_step': us_t, 'algorithmic_bytes': k1, 'frac': k1 / (us_t * 1e-6) / 1e9 / HBM_PEAK_GBPS, 'launches': 2,
                           'bytes_model': 'SURVEY 8(d) K1: imgs read + the similarity map written (here: Lab records + predicate words + per-box counts)'}
    with torch.cuda.stream(stream):
        for s in sets:
            targets(s, st)
    calls = []
    cb = _lib.LAUNCH_HOOK(lambda name, phase, s_, user: calls.append(name.decode()))
    lib.bxi_dev_set_launch_hook(C.cast(cb, C.c_void_p), None)
    try:
        with torch.cuda.stream(stream):
            ev(sets[0], ready, st)
    finally:
        lib.bxi_dev_set_launch_hook(None, None)
    us_l = timed(lambda i: ev(sets[i % len(sets)], ready, st), steps)
    k23 = 8 * N * d0['h'] * d0['w'] + 4 * 8 * d0['B'] * d0['h'] * d0['w'] + 640
    lg = {'us_per_step': us_l, 'instances': N, 'algorithmic_bytes': k23, 'frac': k23 / (us_l * 1e-6) / 1e9 / HBM_PEAK_GBPS,
          'kernels_per_step': sorted(set(calls)), 'status': status[0], 'bit_equal_to_unsplit_evaluation': same,
          'bytes_model': 'SURVEY 8(d) K2 + K3: logits read + gradient written + the similarity map read'}
    # its floor: the same bytes in one launch with the evaluation's stream / tile grid and nothing else (no image roles)
    packs = []
    for s in sets:
        scratch = torch.empty(20 * d0['B'] * d0['h'] * d0['w'] + 256, dtype=torch.uint8, device=dev)
        g = torch.empty_like(s.inst.logits)
        packs.append((vp(0), d0['B'], d0['H'], d0['W'], vp(s.inst.logits.data_ptr()), N, s.inst.h, s.inst.w, vp(g.data_ptr()), vp(scratch.data_ptr()),
                      C.c_size_t(scratch.numel()), scratch, g))

    def sol(i):
        rc = lib.bxi_dev_sol_eval_f32(*packs[i % len(packs)][:11], st)
        if rc != 0:
            raise RuntimeError(f'bxi_dev_sol_eval_f32: status {rc}')
    lg['sol_us'] = timed(sol, max(steps, 300))
    lg['frac_of_sol'] = lg['sol_us'] / us_l
    out['loss_given_targets'] = lg
    if not with_pipeline:
        return out
    # ---- two stages on two streams (eager; events order the two uses of a workspace): the targets of step i + 1 on the side stream
    # while the loss of step i runs on the main one
    try:
        side = torch.cuda.Stream(device=dev)
        n = len(sets)
        t_done = [torch.cuda.Event() for _ in range(n)]
        e_done = [torch.cuda.Event() for _ in range(n)]
        side_st = side.cuda_stream

        def pipelined(steps_):
            used = [False] * n
            side.wait_stream(stream)
            targets(sets[0], side_st)
            t_done[0].record(side)
            for i in range(steps_):
                k, kn = i % n, (i + 1) % n
                if used[kn]:
                    side.wait_event(e_done[kn])          # the evaluation that last read this workspace
                targets(sets[kn], side_st)
                t_done[kn].record(side)
                stream.wait_event(t_done[k])
                ev(sets[k], ready | _lib.EVAL_SHARED_DEVICE, st)
                e_done[k].record(stream)
                used[k] = True
            stream.wait_stream(side)
        with torch.cuda.stream(stream):
            pipelined(64)
            torch.cuda.synchronize(dev)
            done = torch.cuda.Event()
            t0 = time.perf_counter()
            pipelined(steps)
            done.record(stream)
            while not done.query():
                pass
            el = time.perf_counter() - t0
            torch.cuda.synchronize(dev)
            # the host's share: the same calls with nothing to wait for on the device would take this long to ENQUEUE
            t1 = time.perf_counter()
            pipelined(100)
            host_us = (time.perf_counter() - t1) / 100 * 1e6
            torch.cuda.synchronize(dev)
        us = el / steps * 1e6
        whole = survey_bytes(d0, N)
        status = sets[0].state[off:off + 8].view(torch.int32).cpu().tolist()
        out['two_stage_pipelined'] = {'us_per_step': us, 'images_per_s': 2e6 / us, 'algorithmic_bytes': whole, 'frac': whole / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                                      'status': status[0], 'serial_us': us_t + us_l, 'host_enqueue_us_per_step': host_us,
                                      'note': 'eager, two streams: per step four launches, two event records and two event waits from ONE host thread -- '
                                              'where host_enqueue_us_per_step is not below us_per_step the figure is bound by the host, not by the device'}
    except Exception as e:                                       # an extra never takes the headline down
        out['two_stage_pipelined'] = {'error': f'{type(e).__name__}: {e}'[:300]}
        torch.cuda.synchronize(dev)
    # ---- what the split is FOR: the targets under the backbone.  A training iteration is [backbone ... head][loss]; get_targets needs
    # nothing the backbone produces (condinst.py:53 vs :73).  Stand-in for the backbone: a chain of bf16 matrix products on the main
    # stream (compute-bound, every CU busy), the loss behind it.  (a) the un-split evaluation behind the stand-in; (b) the targets on a
    # side stream WHILE the stand-in runs, then the evaluation with BXI_EVAL_TARGETS_READY.  GPU time per iteration, events on the main stream.
    try:
        side = torch.cuda.Stream(device=dev)
        side_st = side.cuda_stream
        hog_a = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
        hog_b = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
        hog_c = torch.empty_like(hog_a)

        def backbone():
            torch.mm(hog_a, hog_b, out=hog_c)
            torch.mm(hog_c, hog_b, out=hog_a)

        def iteration(split, i):
            s = sets[i % len(sets)]
            if split:
                side.wait_stream(stream)                  # (the images of this iteration exist; the previous loss is done with the workspace)
                targets(s, side_st)
            backbone()
            if split:
                stream.wait_stream(side)
                ev(s, ready | _lib.EVAL_SHARED_DEVICE, st)
            else:
                ev(s, 0, st)

        # the three variants ALTERNATE, one event pair per iteration, medians: the matrix products' own duration drifts by several us over a
        # run (clocks), more than what is being measured
        variants = (('standin_alone', None), ('unsplit', False), ('targets_under_the_standin', True))
        pairs = {name: [] for name, _ in variants}
        with torch.cuda.stream(stream):
            for i in range(30):
                iteration(bool(i & 1), i)
            torch.cuda.synchronize(dev)
            for i in range(3 * 150):
                name, split = variants[i % 3]
                a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a_.record(stream)
                if split is None:
                    backbone()
                else:
                    iteration(split, i // 3)
                b_.record(stream)
                pairs[name].append((a_, b_))
            torch.cuda.synchronize(dev)
        res = {name: float(np.median([a_.elapsed_time(b_) * 1e3 for a_, b_ in evs])) for name, evs in pairs.items()}
        out['under_a_backbone_standin'] = {
            'us_per_iteration': res, 'loss_cost_unsplit_us': res['unsplit'] - res['standin_alone'],
            'loss_cost_with_targets_ahead_us': res['targets_under_the_standin'] - res['standin_alone'],
            'standin': 'two 4096^3 bf16 matrix products per iteration on the main stream (torch.mm)',
            'note': 'medians over 150 iterations per variant, the variants alternating (one event pair per iteration); the second figure contains '
                    'whatever the concurrent targets kernels cost the stand-in'}
    except Exception as e:
        out['under_a_backbone_standin'] = {'error': f'{type(e).__name__}: {e}'[:300]}
        torch.cuda.synchronize(dev)
    return out


def rows_extra():
    """Extras (not `value`): one entry per row of SURVEY 8(a)/(f) that the timed step does not contain -- the op-level pairwise_nlog
    (a-9..a-11) and the "next" rows f-2..f-4 -- so that the driver's record holds them too.  Each is the developer bench of that row
    (tools/bench_*.py: event-timed batches of calls through the Python API, no profiler), run in this process; `frac` is on the row's
    compulsory bytes (stated) against the HBM peak where the row is a streaming kernel, `bound` says what limits it where it is not."""
    import contextlib
    import io
    import runpy

    def run_tool(name):
        buf = io.StringIO()
        try:
            with contextlib.redirect_stdout(buf):
                runpy.run_path(os.path.join(ROOT, 'tools', name), run_name='__main__')
            return json.loads(buf.getvalue()[buf.getvalue().index('{'):])
        except Exception as e:                                   # an extra never takes the headline down
            return {'error': f'{type(e).__name__}: {e}'[:300]}

    def frac(nbytes, us):
        return nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS

    out = {}
    r = run_tool('bench_pairwise_op.py')
    if 'error' not in r:
        f32 = r['torch.float32']
        nb = 32 * 200 * 256 * 4
        out['pairwise_op'] = {'shape': '[32,1,200,256] f32, size 3, dilation 2 (pairwise.cu:68-202); COLD inputs (6 rotating sets > Infinity Cache)',
                              'fwd_us': f32['fwd_us'], 'bwd_us': f32['bwd_us'], 'fwd_warm_us': f32.get('fwd_warm_us'), 'bwd_warm_us': f32.get('bwd_warm_us'),
                              'fwd_bytes': 9 * nb, 'bwd_bytes': 10 * nb, 'fwd_frac': frac(9 * nb, f32['fwd_us']), 'bwd_frac': frac(10 * nb, f32['bwd_us']),
                              'bytes_model': 'forward: logits read + 8 planes written; backward: logits + 8 upstream planes read + gradient written',
                              'fwd_sol_us': f32.get('fwd_sol_us'), 'bwd_sol_us': f32.get('bwd_sol_us'),
                              'fwd_frac_of_sol': f32['fwd_sol_us'] / f32['fwd_us'] if f32.get('fwd_sol_us') else None,
                              'bwd_frac_of_sol': f32['bwd_sol_us'] / f32['bwd_us'] if f32.get('bwd_sol_us') else None,
                              'sol': 'the same bytes moved by a copy kernel (bxi_dev_sol_pairwise_f32: 16-byte accesses, nothing computed), timed the same way',
                              'f64_fwd_us': r['torch.float64']['fwd_us'], 'f64_bwd_us': r['torch.float64']['bwd_us']}
    else:
        out['pairwise_op'] = r
    # (the dynamic head and the tree filter are timed here rather than through their tools/ scripts: those also time comparison paths
    # that live under oracle/, which bench.py touches in its cpu_baseline leg only)
    dev = torch.device('cuda', torch.cuda.current_device())

    def ev(fn, n=60, warm=8):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a_.record()
        for _ in range(n):
            fn()
        b_.record()
        torch.cuda.synchronize()
        return a_.elapsed_time(b_) / n * 1e3
    try:
        from boxinstseg_amd import dynamic_mask_forward
        g = torch.Generator(device='cpu').manual_seed(0)
        N, B, Cf, H, W = 32, 2, 16, 100, 128
        feat = torch.randn(B, Cf, H, W, generator=g).to(dev).requires_grad_(True)
        params = (torch.randn(N, 233, generator=g) * 0.3).to(dev).requires_grad_(True)
        coors = (torch.rand(N, 2, generator=g) * 1000).to(dev)
        lvl, img = torch.randint(0, 5, (N,), generator=g).to(dev), torch.randint(0, B, (N,), generator=g).to(dev)
        soi = torch.tensor([64, 128, 256, 512, 1024], device=dev)
        gout = torch.randn(N, 1, 2 * H, 2 * W, generator=g).to(dev)

        def fwd():
            return dynamic_mask_forward(feat, params, coors, lvl, img, soi)

        def fwd_bwd():
            fwd().backward(gout)
            feat.grad = None
            params.grad = None
        with torch.no_grad():
            t_f = ev(fwd)
        feat_b, logit_b = B * Cf * H * W * 4, N * 4 * H * W * 4
        out['f2_dynamic_head'] = {'shape': '32 instances, feat [2,16,100,128] -> logits [32,1,200,256] (condinst_head.py:1139-1164)', 'fwd_us': t_f,
                                  'fwd_bwd_us': ev(fwd_bwd), 'fwd_bytes': feat_b + logit_b, 'fwd_frac': frac(feat_b + logit_b, t_f),
                                  'bound': 'issue (a workgroup\'s load -> 3-layer MLP -> store chain, DESIGN_NEXT_ROWS 3.5), not HBM; '
                                           'wall time through the Python API'}
    except Exception as e:
        out['f2_dynamic_head'] = {'error': f'{type(e).__name__}: {e}'[:300]}
    try:
        from boxinstseg_amd import MinimumSpanningTree, TreeFilter2D
        g = torch.Generator().manual_seed(0)
        mstm, tf = MinimumSpanningTree(TreeFilter2D.norm2_distance), TreeFilter2D()
        Bt, Nt, Ht = 2, 16, 96
        img_t, lst = torch.rand(Bt, 3, Ht, Ht, generator=g).to(dev), torch.rand(Bt, 8, Ht, Ht, generator=g).to(dev)
        rep_i = torch.arange(Nt) % Bt
        pred = torch.rand(Nt, 1, Ht, Ht, generator=g).to(dev).requires_grad_(True)
        ti, tl = mstm(img_t)[rep_i], mstm(lst)[rep_i]
        imgs_n, lst_n = img_t[rep_i], lst[rep_i]

        def filt():
            a_ = tf(feature_in=pred, embed_in=imgs_n, tree=ti)
            b_ = tf(a_, lst_n, tl, low_tree=False)
            (a_.sum() + b_.sum()).backward()
            pred.grad = None
        img2 = torch.rand(1, 3, 200, 304, generator=g).to(dev)
        feat2 = torch.rand(1, 5, 200, 304, generator=g).to(dev).requires_grad_(True)
        t2 = mstm(img2)

        def filt2():
            tf(feature_in=feat2, embed_in=img2, tree=t2).sum().backward()
            feat2.grad = None
        out['f4_tree_filter'] = {'small_96x96_B2_N16': {'two_msts_us': ev(lambda: (mstm(img_t), mstm(lst)), n=20, warm=3), 'two_filters_fwd_bwd_us': ev(filt, n=20, warm=3)},
                                 'large_200x304_C5': {'mst_us': ev(lambda: mstm(img2), n=5, warm=1), 'filter_fwd_bwd_us': ev(filt2, n=5, warm=1)},
                                 'bound': 'latency: graph-serial (MST phases, BFS ranking, leaf->root / root->leaf passes are dependent launches / LDS walks); '
                                          'mmdet/ops/tree_filter/**'}
    except Exception as e:
        out['f4_tree_filter'] = {'error': f'{type(e).__name__}: {e}'[:300]}
    r = run_tool('bench_discobox.py')
    if 'error' not in r:
        n16 = r['n16']
        out['f3_discobox'] = {'shape': '16 instances at 200x304, 10 mean-field iterations (discobox_head.py:585-655)', 'meanfield_us': n16['hip_meanfield_us'],
                              'mil_fwd_bwd_us': n16['hip_mil_fwd_bwd_us'], 'bound': 'latency: a chain of 10 dependent stencil launches over 3.9 MB (no HBM roofline applies)',
                              'torch_rocm_meanfield_us': n16['torch_rocm_meanfield_us'], 'label_mismatch_vs_torch_rocm': n16['label_mismatch_vs_torch_rocm']}
    else:
        out['f3_discobox'] = r
    r = run_tool('bench_levelset.py')
    if 'error' not in r:
        n16 = r['N16']
        out['f4_box2mask_losses'] = {'shape': '16 instances at 200x304 (projection, level set), 96x96 (LCM, 10 iterations)',
                                     'projection_fwd_bwd_us': n16['hip_projection_fwd_bwd_us'], 'levelset_fwd_bwd_us': n16['hip_levelset_fwd_bwd_us'],
                                     'lcm_fwd_bwd_us': n16['hip_lcm_fwd_bwd_us'], 'bound': 'latency / host launch (a few small launches each; wall time through the Python API)',
                                     'torch_rocm_projection_fwd_bwd_us': n16['torch_rocm_projection_fwd_bwd_us'],
                                     'torch_rocm_levelset_fwd_bwd_us': n16['torch_rocm_levelset_fwd_bwd_us'], 'torch_rocm_lcm_fwd_bwd_us': n16['torch_rocm_lcm_fwd_bwd_us']}
    else:
        out['f4_box2mask_losses'] = r
    # ---- the streaming kernels of the (f) rows against the HBM peak, each on its stated compulsory bytes (C ABI calls, event-timed batches)
    try:
        from boxinstseg_amd import _lib
        lib = _lib.load()
        st = torch.cuda.current_stream(dev).cuda_stream
        g = torch.Generator().manual_seed(1)
        fr = {}
        N, Cc, H, W = 100, 3, 200, 304                    # Box2Mask: 100 queries (box2mask_head.py:269-335)
        ms, T = torch.rand(N, 2, H, W, generator=g).to(dev), torch.rand(N, Cc, H, W, generator=g).to(dev)
        pn, loss = torch.full((N,), float(H * W), device=dev), torch.empty(N, device=dev)
        state = torch.empty(max(lib.bxi_levelset_state_bytes(N, Cc), 256), dtype=torch.uint8, device=dev)
        t_ = ev(lambda: lib.bxi_levelset_loss_forward_f32(ms.data_ptr(), T.data_ptr(), pn.data_ptr(), N, Cc, H, W, 1.0, loss.data_ptr(), state.data_ptr(), st), n=100)
        nb = N * (2 + Cc) * H * W * 4
        fr['levelset_partial_kernel'] = {'call': 'bxi_levelset_loss_forward_f32, 100 x (2 + 3) x 200 x 304', 'us': t_, 'bytes': nb, 'frac': frac(nb, t_),
                                         'bytes_model': 'mask scores (2 planes) + targets (C planes) read once (warm: 122 MB < the Infinity Cache)'}
        img96, aff = torch.rand(N, 3, 96, 96, generator=g).to(dev), torch.empty(N, 8, 96, 96, device=dev)
        t_ = ev(lambda: lib.bxi_lcm_affinity_f32(img96.data_ptr(), N, 3, 96, 96, 2, 0.3, aff.data_ptr(), st), n=100)
        nb = N * (3 + 8) * 96 * 96 * 4
        fr['lcm_affinity_kernel'] = {'call': 'bxi_lcm_affinity_f32, 100 x 3 x 96 x 96 -> 100 x 8 x 96 x 96', 'us': t_, 'bytes': nb, 'frac': frac(nb, t_),
                                     'bytes_model': 'images read + 8 affinity planes written; bound by its double-precision statistics (a square root and a division per pixel and channel), not by HBM'}
        from boxinstseg_amd.discobox import meanfield_forward, meanfield_kernel
        Bm, Nm = 2, 16
        ker = meanfield_kernel(torch.rand(Bm, 3, H, W, generator=g).to(dev))
        xm, tm = torch.rand(Nm, H, W, generator=g).to(dev), (torch.rand(Nm, H, W, generator=g) > 0.5).float().to(dev)
        ii = (torch.arange(Nm) % Bm).to(dev)
        with torch.no_grad():
            t10 = ev(lambda: meanfield_forward(ker, xm, tm, 20, 0.45, img_inds=ii), n=40)
            t0_ = ev(lambda: meanfield_forward(ker, xm, tm, 10, 0.45, img_inds=ii), n=40)
        per = (t10 - t0_) / 10
        nb = Bm * 9 * H * W * 4 + 2 * Nm * H * W // 8
        fr['mf_step_kernel'] = {'call': 'one mean-field update of 16 instances at 200 x 304 ((20 iterations - 10 iterations) / 10)', 'us': per, 'bytes': nb,
                                'frac': frac(nb, per), 'bytes_model': 'the 9 kernel planes of each image read once + one bit plane per instance in and out '
                                '(4.5 MB per update: launch-bound, ~3 us per dependent launch)'}
        out['f_streaming_kernels'] = fr
    except Exception as e:
        out['f_streaming_kernels'] = {'error': f'{type(e).__name__}: {e}'[:300]}
    return out


def head_fused(lib, Fh, sets, dev, stream, steps):
    """Extra (not `value`): the step one level out -- CondInstMaskHead.forward + .loss at the C ABI, i.e. the
    dynamic mask head (16 mask-feature channels at stride 8, random parameters) producing the logits the evaluation consumes:
    bxi_dynamic_mask_forward_f32 + bxi_boxinst_eval_f32 (2 launches: the head, the single-launch evaluation) against
    bxi_boxinst_head_eval_f32 (2 launches: the head's tiles run inside the two-launch evaluation's first launch)."""
    vp = C.c_void_p
    g = torch.Generator(device='cpu').manual_seed(0)
    packs = []
    for s in sets:
        d, N = s.d, s.inst.N
        Hs, Ws = d['H'] // 8, d['W'] // 8
        feat = torch.randn(d['B'], 16, Hs, Ws, generator=g).to(dev)
        params = (0.3 * torch.randn(N, 18 * 8 + 64 + 8 + 8 + 8 + 1, generator=g)).to(dev)
        coors = (torch.rand(N, 2, generator=g) * torch.tensor([float(d['W']), float(d['H'])])).to(dev)
        lvl = torch.randint(0, 5, (N,), generator=g).to(dev)
        counts = np.cumsum([0] + [b.shape[0] for b in d['gt_bboxes']])
        img = torch.tensor([int(np.searchsorted(counts, int(x), side='right') - 1) for x in d['gt_inds']], dtype=torch.int64).to(dev)
        soi = torch.tensor([64., 128., 256., 512., 1024.], device=dev)
        head = (vp(feat.data_ptr()), C.c_int(16), C.c_int(Hs), C.c_int(Ws), vp(params.data_ptr()), vp(coors.data_ptr()),
                vp(lvl.data_ptr()), vp(img.data_ptr()), vp(soi.data_ptr()), C.c_int(5), C.c_int(8), C.c_int(2), C.c_int(0))
        logits2 = torch.empty_like(s.inst.logits)               # the head writes here: the timed sets keep their logits
        inst2 = Fh._Inst(logits2, s.gt_inds, s.boxes, d['H'], d['W'], d['stride'])
        ev = (s.eval_args[0], C.byref(inst2.struct)) + tuple(s.eval_args[2:])
        fwd = (vp(feat.data_ptr()), C.c_int(d['B']), C.c_int(16), C.c_int(Hs), C.c_int(Ws), vp(params.data_ptr()), C.c_int(N),
               vp(coors.data_ptr()), vp(lvl.data_ptr()), vp(img.data_ptr()), vp(soi.data_ptr()), C.c_int(5), C.c_int(8), C.c_int(2),
               C.c_int(0), vp(logits2.data_ptr()))
        packs.append((ev, head, fwd, (feat, params, coors, lvl, img, soi, logits2, inst2)))
    st = stream.cuda_stream

    def fused(pk):
        ev, head, _, _ = pk
        rc = lib.bxi_boxinst_head_eval_f32(ev[0], ev[1], *head, *ev[2:], st)
        assert rc == 0, rc

    def separate(pk):
        ev, _, fwd, _ = pk
        rc = lib.bxi_dynamic_mask_forward_f32(*fwd, st)
        assert rc == 0, rc
        rc = lib.bxi_boxinst_eval_f32(*ev, st)
        assert rc == 0, rc

    out = {}
    with torch.cuda.stream(stream):
        for name, fn in (('head_then_eval_two_calls', separate), ('head_eval_fused_one_call', fused)):
            for i in range(50):
                fn(packs[i % len(packs)])
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for i in range(steps):
                fn(packs[i % len(packs)])
            torch.cuda.synchronize(dev)
            el = time.perf_counter() - t1
            out[name] = {'us_per_step': el / steps * 1e6, 'images_per_s': 2 * steps / el}
    out['note'] = ('dynamic mask head forward + loss evaluation (forward and finished backward w.r.t. the logits) per step, C ABI, '
                   'cold input sets, its own logits buffers')
    return out


def module_api(sets, dev, n):
    """The same evaluation through the drop-in module: CondInstMaskHead.loss(...) + (loss_prj + loss_pairwise).backward()."""
    import gc
    from boxinstseg_amd import CondInstMaskHead
    head = CondInstMaskHead(in_channels=16, boxinst_enabled=True, topk_per_img=64, max_proposals=-1).to(dev)
    head._iter.fill_(20000.0)
    xs = [s.logits.clone().requires_grad_(True) for s in sets]

    def once(i):
        s, x = sets[i % len(sets)], xs[i % len(sets)]
        out = head.loss(s.imgs, s.d['img_metas'], x, s.gt_inds, s.boxes, None, None)
        (out['loss_prj'] + out['loss_pairwise']).backward()
        x.grad = None

    # the first ~second of eager autograd work in a process runs at up to THREE times the steady-state host time (measured:
    # 239 -> 123 -> 75 us per call over the first 7000 calls, a plain torch graph alongside 155 -> 78 -> 62; profiles/NOTES.md R4-5):
    # warm until the figure has settled, bounded
    t_w, k_w, last = time.perf_counter(), 0, None
    while time.perf_counter() - t_w < 4.0:
        t1 = time.perf_counter()
        for i in range(500):
            once(k_w + i)
        k_w += 500
        cur = (time.perf_counter() - t1) / 500
        if last is not None and cur > 0.97 * last and k_w >= 3000:
            break
        last = cur
    torch.cuda.synchronize(dev)
    gc.collect()
    gc.freeze()                      # the collector otherwise walks the whole application heap every few hundred allocations
    t0 = time.perf_counter()
    for i in range(n):
        once(i)
    host = time.perf_counter() - t0
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    gc.unfreeze()
    # the library's share of the host time: the same call with the autograd engine out of the picture
    from boxinstseg_amd import functional as Fh
    t0 = time.perf_counter()
    with torch.no_grad():
        for i in range(n):
            s = sets[i % len(sets)]
            head.loss(s.imgs, s.d['img_metas'], s.logits, s.gt_inds, s.boxes, None, None)
    fwd_only = time.perf_counter() - t0
    torch.cuda.synchronize(dev)
    varying = module_api_varying(head, sets, dev, n)
    # for scale: a plain PyTorch graph of the same shape (two reductions of the logits, their sum, backward) -- what the autograd
    # engine and three eager launches cost on this host whatever the nodes do
    w = sets[0].logits.clone().requires_grad_(True)

    def torch_only():
        (w.sum() + w.mean()).backward()
        w.grad = None
    for _ in range(100):
        torch_only()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(n):
        torch_only()
    torch_ref = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize(dev)
    return {'images_per_s': 2 * n / el, 'us_per_call': el / n * 1e6, 'host_us_per_call': host / n * 1e6,
            'loss_call_host_us_no_autograd': fwd_only / n * 1e6, 'varying_shapes': varying,
            'torch_only_same_graph_shape_host_us': torch_ref, 'forward_loss': module_forward_loss(sets, dev, min(n, 200)),
            'note': 'CondInstMaskHead.loss + backward() of the two scalars, eager PyTorch (gc frozen); host-bound: '
                    'the GPU work is `value`.  torch_only_same_graph_shape_host_us = (w.sum() + w.mean()).backward() on the same '
                    'logits: the autograd engine\'s round and three eager launches, i.e. the part of us_per_call that is not this library'}


def module_forward_loss(sets, dev, n):
    """CondInstMaskHead.forward_loss -- forward() + loss() (+ backward to the mask features and the dynamic parameters) as ONE module
    call -- with the head fused into the evaluation's first launch and as the two calls (the default), mmdet/models/detectors/condinst.py:69-75."""
    from boxinstseg_amd import CondInstMaskHead
    head = CondInstMaskHead(in_channels=16, boxinst_enabled=True, topk_per_img=64, max_proposals=-1).to(dev)
    head._iter.fill_(20000.0)
    g = torch.Generator(device='cpu').manual_seed(0)
    packs = []
    for s in sets:
        d, N = s.d, s.inst.N
        feat = torch.randn(d['B'], 16, d['H'] // 8, d['W'] // 8, generator=g).to(dev).requires_grad_(True)
        params = (0.3 * torch.randn(N, 233, generator=g)).to(dev).requires_grad_(True)
        coors = (torch.rand(N, 2, generator=g) * torch.tensor([float(d['W']), float(d['H'])])).to(dev)
        lvl = torch.randint(0, 5, (N,), generator=g).to(dev)
        counts = np.cumsum([0] + [b.shape[0] for b in d['gt_bboxes']])
        img = torch.tensor([int(np.searchsorted(counts, int(x), side='right') - 1) for x in d['gt_inds']], dtype=torch.int64).to(dev)
        packs.append((feat, params, coors, lvl, img))
    out = {}
    def once(i, fused):
        s, (feat, params, coors, lvl, img) = sets[i % len(sets)], packs[i % len(sets)]
        _, losses = head.forward_loss(feat, params, coors, lvl, img, s.imgs, s.d['img_metas'], s.gt_inds, s.boxes, fuse_head=fused)
        (losses['loss_prj'] + losses['loss_pairwise']).backward()
        feat.grad = None
        params.grad = None
    for i in range(300):
        once(i, bool(i & 1))
    torch.cuda.synchronize(dev)
    acc = {False: [], True: []}
    for rnd in range(4):                         # A / B / A / B: whichever runs first in a process is otherwise the slower one
        for fused in (False, True):
            t0 = time.perf_counter()
            for i in range(n):
                once(i, fused)
            torch.cuda.synchronize(dev)
            acc[fused].append((time.perf_counter() - t0) / n * 1e6)
    out['two_calls_us'], out['fused_head_us'] = float(np.median(acc[False])), float(np.median(acc[True]))
    out['rounds'] = {'two_calls_us': acc[False], 'fused_head_us': acc[True]}
    out['default'] = 'fused (fuse_head=True)' if CondInstMaskHead.forward_loss.__defaults__[-1] else 'two calls (fuse_head=False)'
    return out


def module_api_varying(head, sets, dev, n):
    """The module call when every iteration differs, as real COCO iterations do: img_shape / ori_shape, the number of GT boxes per
    image and the number of instances change on every call (canvas fixed, as with a padded batch).  Host time of loss() alone."""
    import gc
    rng = np.random.default_rng(7)
    calls = []
    for i in range(16):
        s = sets[i % len(sets)]
        d = s.d
        metas = []
        for m in d['img_metas']:
            ih, iw = int(rng.integers(d['H'] // 2, d['H'] + 1)), int(rng.integers(d['W'] // 2, d['W'] + 1))
            metas.append(dict(m, img_shape=(ih, iw, 3), ori_shape=(int(ih * rng.uniform(0.4, 1.5)), int(iw * rng.uniform(0.4, 1.5)), 3)))
        boxes = [b[:int(rng.integers(1, b.shape[0] + 1))].contiguous() for b in s.boxes]
        G = sum(b.shape[0] for b in boxes)
        N = int(rng.integers(max(G // 2, 1), G + 1))
        gi = torch.from_numpy(rng.integers(0, G, N).astype(np.int64)).to(dev)
        calls.append((s.imgs, metas, s.logits[:N].contiguous(), gi, boxes))
    def once(i):
        imgs, metas, x, gi, boxes = calls[i % len(calls)]
        with torch.no_grad():
            head.loss(imgs, metas, x, gi, boxes, None, None)
    for i in range(32):
        once(i)
    torch.cuda.synchronize(dev)
    gc.collect()
    gc.freeze()
    t0 = time.perf_counter()
    for i in range(n):
        once(i)
    host = time.perf_counter() - t0
    torch.cuda.synchronize(dev)
    gc.unfreeze()
    return {'loss_call_host_us_no_autograd': host / n * 1e6,
            'note': 'img_shape, ori_shape, boxes per image and instance count differ on every call (16 distinct calls, rotated)'}


# algorithmic (compulsory) HBM bytes -- DESIGN.md section 4 / SURVEY 8(d).
def hull_fraction(d, dil=2, rows=8):
    """fraction of the N x h x w gradient the pair kernel's tiles own (R-aligned rows x the dilated box's columns)."""
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
        hit += (min(h, -(-r1 // rows) * rows) - r0 // rows * rows) * (c1 - c0)
    return hit / float(len(d['gt_inds']) * h * w)


def algorithmic_bytes(d, N, rows):
    """Compulsory HBM bytes per launch: per-unit figure x units of one launch."""
    px_in = d['B'] * d['H'] * d['W']          # input pixels:     12 B read each (3 x f32)
    px_small = d['B'] * d['h'] * d['w']       # pooled pixels:    16 B written (Lab as a tagged float4) by the pool workgroups;
    ipx = N * d['h'] * d['w']                 #                   16 B read + 4 B written (predicate word) by the predicate waves
    f = hull_fraction(d, rows=rows)           # instance-pixels:  4 B read (logit) + 4 B written (zero-filled gradient) by the stream workgroups
    front = 12 * px_in + 16 * px_small + 4 * ipx + 4 * ipx
    # box tiles: logits read (4 B), predicate word read (4 B), gradient added at the memory side (4 B read + 4 B written)
    back = 20 * px_small + int(ipx * f) * (4 + 4 + 8)
    return {'prep': front, 'pair': back, 'eval1': front + back}     # two-launch form: prep + pair; single-launch form: eval1


def survey_bytes(d, N):
    """compulsory bytes of a whole evaluation given the API's inputs and outputs -- read imgs, write the
    similarity map, read it again, read the logits, write their gradient (the fused path never materialises the map)."""
    return 12 * d['B'] * d['H'] * d['W'] + 2 * 4 * 8 * d['B'] * d['h'] * d['w'] + 8 * N * d['h'] * d['w'] + 640


def measured_traffic():
    """HBM bytes per launch from the committed PMC summary of this command (profiles/*_hbm_traffic.json, written by
    tools/summarize_profiles.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes) -> (dict, file) or (None, None)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r0*_hbm_traffic.json')))
    if not files:
        return None, None
    try:
        k = json.load(open(files[-1]))['kernels']
        return {name: float(v['hbm_bytes']) for name, v in k.items()}, os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


def kernel_timing(lib, _lib, sets, stream, enqueue, steps, step_us, rows):
    """Bracket every kernel launch with HIP events (bxi_dev_set_launch_hook) over `steps` eager steps."""
    events = {}

    def hook(name, phase, st, user):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(stream)
        events.setdefault(name.decode(), []).append(ev)

    cb = _lib.LAUNCH_HOOK(hook)
    with torch.cuda.stream(stream):
        for i in range(20):
            enqueue(sets[i % len(sets)], stream.cuda_stream)
        stream.synchronize()
        # park the stream behind a ~0.1 s spin kernel so that every launch + event below is already queued
        # when the GPU gets to it: the event pairs then bracket kernels that run back to back, exactly as
        # in the timed region (otherwise they would also measure the host's enqueue latency)
        torch.cuda._sleep(int(0.1 * 2.0e9))
        lib.bxi_dev_set_launch_hook(C.cast(cb, C.c_void_p), None)
        try:
            for i in range(steps):
                enqueue(sets[i % len(sets)], stream.cuda_stream)
        finally:
            lib.bxi_dev_set_launch_hook(None, None)
    stream.synchronize()
    raws = {name: np.array([evs[j].elapsed_time(evs[j + 1]) * 1e3 for j in range(0, len(evs) - 1, 2)])
            for name, evs in events.items()}                                                      # us
    # What an event pair costs by itself, measured on its own: pairs of events with NOTHING between them, queued behind the same
    # parked stream.  (Round 2 defined this cost as whatever made the kernels tile the step, which made the per-kernel split a ratio,
    # not a measurement.)  The per-kernel figures below are raw - this; rocprofv3's durations of the same command are in profiles/.
    with torch.cuda.stream(stream):
        torch.cuda._sleep(int(0.05 * 2.0e9))
        empties = []
        for _ in range(200):
            a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a_.record(stream); b_.record(stream)
            empties.append((a_, b_))
    stream.synchronize()
    bracket_us = float(np.median([a_.elapsed_time(b_) * 1e3 for a_, b_ in empties]))
    d0, N = sets[0].d, sets[0].inst.N
    alg = algorithmic_bytes(d0, N, rows)
    traffic, traffic_file = measured_traffic()
    key = {'prep': 'prep_kernel', 'pair': 'pair_kernel', 'eval1': 'eval1_kernel', 'prep_fold': 'prep_kernel', 'prep_ready': 'prep_kernel',
           'pair_tiles': 'pair_kernel', 'eval1_ready': 'eval1_kernel'}
    per_kernel = {}
    for name, raw in raws.items():
        durs = raw - bracket_us
        b = alg.get(name, 0)
        avg = float(np.mean(durs))
        per_kernel[name] = {'avg_us': avg, 'median_us': float(np.median(durs)), 'min_us': float(np.min(durs)),
                            'launches': len(durs), 'raw_event_avg_us': float(np.mean(raw)), 'algorithmic_bytes': b,
                            'achieved_GBps': b / (avg * 1e-6) / 1e9 if b else None,
                            'frac': b / (avg * 1e-6) / 1e9 / HBM_PEAK_GBPS if b else None,
                            'traffic': traffic.get(key.get(name, name)) if traffic else None}
    whole = survey_bytes(d0, N)
    ksum = sum(v['avg_us'] for v in per_kernel.values())
    # the figure of merit is computed on the UN-INSTRUMENTED step time (kernels + their boundaries: what a training loop sees)
    a = whole / (step_us * 1e-6) / 1e9
    return {
        'roofline': {'kernel': ' + '.join(sorted(per_kernel)) + ' = the whole evaluation', 'bound': 'hbm', 'achieved': a, 'peak': HBM_PEAK_GBPS,
                     'unit': 'GB/s', 'frac': a / HBM_PEAK_GBPS,
                     'traffic': sum(traffic[key[k]] for k in per_kernel if key.get(k) in traffic) if traffic and all(key.get(k) in traffic for k in per_kernel) else None,
                     'traffic_source': traffic_file,
                     'algorithmic_bytes': whole, 'algorithmic_bytes_source': 'SURVEY 8(d): 39 322 240 B at 2x800x1024x32',
                     'time_us': step_us, 'time_source': 'ms_per_step of this run (un-instrumented: every launch of a step and its boundary)',
                     'event_kernel_time_us': ksum, 'frac_on_event_kernel_time': whole / (ksum * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                     'timing': 'per kernel: hipEvent pairs around each launch on the launching stream (bxi_dev_set_launch_hook), minus the '
                               'cost of an empty pair measured separately (event_bracket_us); launches queued behind a parked stream, '
                               'cold input sets.  rocprofv3 durations of the same command: profiles/',
                     'per_kernel': {k: {'avg_us': v['avg_us'], 'raw_event_avg_us': v['raw_event_avg_us'],
                                        'algorithmic_bytes': v['algorithmic_bytes'], 'frac': v['frac']}
                                    for k, v in per_kernel.items()}},
        'kernels': per_kernel,
        'event_bracket_us': bracket_us,
    }


def parity_gate_only(s):
    """The C oracle (OpenMP) as the checker of one evaluated set: losses <= 1e-4 rel, gradient <= 1e-4 of its max (arg-max ties excluded
    and counted).  Used on every rank of an N > 1 run; the N == 1 run's gate is part of cpu_baseline()'s leg below."""
    from tests.helpers import grad_report, oracle_path
    ref = oracle_path(s.d, want_targets=False)
    got = s.losses.cpu().numpy()
    g = s.grad.cpu().numpy()[:, 0]
    g_err, n_ties = grad_report(g, ref['grad'], s.d['mask_logits'][:, 0])
    out = dict(loss_prj_rel=float(abs(got[0] - ref['loss_prj']) / abs(ref['loss_prj'])),
               loss_pairwise_rel=float(abs(got[1] - ref['loss_pairwise']) / abs(ref['loss_pairwise'])), grad_rel_max=float(g_err),
               ambiguous_argmax_lines_excluded=int(n_ties))
    out['ok'] = bool(max(out['loss_prj_rel'], out['loss_pairwise_rel'], out['grad_rel_max']) <= 1e-4)
    return out


def oracle_gate(d):
    """The C oracle (OpenMP) on the workload: its result for the parity gate (and the similarity map for the weight-mask comparison), and how
    long it took -- part of the cpu_baseline leg, the only place bench.py touches oracle/."""
    from tests.helpers import oracle_path
    t_c0 = time.perf_counter()
    ref = oracle_path(d, want_targets=False)
    t_c = time.perf_counter() - t_c0
    ref['sim'] = oracle_path(d)['sim']
    return ref, {'value': 2 / t_c, 'unit': 'images/s', 'ms_per_eval': t_c * 1e3}


def cpu_baseline(d, budget_s):
    """The reference's CPU loss path (torch CPU ops, oracle/torch_oracle.py) on the host cores, fwd+bwd,
    same workload; bounded to ~budget_s seconds."""
    from oracle import torch_oracle as to
    host_cores = os.cpu_count() or 1
    cores = min(host_cores, 32)            # torch CPU ops stop scaling (and thrash) far below a 256-thread host: both are measured below
    imgs = torch.from_numpy(d['imgs'])
    gi = torch.from_numpy(d['gt_inds'])
    boxes = [torch.from_numpy(b) for b in d['gt_bboxes']]

    def once():
        x = torch.from_numpy(d['mask_logits']).requires_grad_(True)
        out = to.mask_loss(imgs, d['img_metas'], x, gi, boxes)
        (out['loss_prj'] + out['loss_pairwise']).backward()

    def sample(threads, budget, most):
        torch.set_num_threads(threads)
        once()                                          # warm-up
        k, t_ = 0, time.perf_counter()
        while True:
            once()
            k += 1
            e_ = time.perf_counter() - t_
            if e_ > budget or k >= most:
                return k, e_

    n, el = sample(cores, budget_s, 10)
    # BASELINE.md section 3 asks for ALL host cores as well: a second, shorter sample at os.cpu_count() threads; the better of the two is `value`
    by_threads = {str(cores): {'value': 2 * n / el, 'ms_per_eval': el / n * 1e3, 'evaluations': n}}
    more = min(host_cores, 64)              # (all 256 threads of the driver's host, measured once -- profiles/r05_cpu_threads.json: 34.4 s per evaluation,
    if more > cores:                        # 0.058 images/s, and the 256 spinning OpenMP workers slowed every host-bound figure measured after it)
        n2, el2 = sample(more, min(budget_s, 6.0), 4)
        by_threads[str(more)] = {'value': 2 * n2 / el2, 'ms_per_eval': el2 / n2 * 1e3, 'evaluations': n2}
        if 2 * n2 / el2 > 2 * n / el:
            n, el, cores = n2, el2, more
        torch.set_num_threads(cores)
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
    split = {'loss_given_similarity_ms': t_loss * 1e3, 'colour_affinity_targets_ms': t_targets * 1e3,
             'one_core': {'value': 2 / t_one, 'unit': 'images/s', 'ms_per_eval': t_one * 1e3, 'cores': 1}}
    return {'value': 2 * n / el, 'unit': 'images/s', 'cores': cores, 'kind': 'port',
            'sample': f'{n} full evaluations (targets + loss fwd+bwd) of the same 2x800x1024x{len(d["gt_inds"])} '
                      f'workload with the torch-CPU restatement of the reference path, {cores} threads',
            'ms_per_eval': el / n * 1e3, 'host_cores': host_cores, 'by_threads': by_threads, **split}


if __name__ == '__main__':
    main()
