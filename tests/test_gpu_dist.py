"""The N > 1 path on real GPUs: one process per GPU, backend nccl (= RCCL), the HIP loss path on every rank.

Two ranks when the box has two GPUs; on a one-GPU box the same code runs with world size 1 (process group, RCCL
communicator and collectives are still created and issued, over a single rank).  The CPU-only counterpart
(tests/test_dist_gloo.py) covers the rank arithmetic of the logging collective with two gloo ranks.
Reference: mmdet/models/detectors/base.py:176-219 (_parse_losses), tools/dist_train.sh:10-20 (one process per GPU).
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    import torch.distributed as dist
    from boxinstseg_amd import boxinst_mask_loss, dist as bdist, synthetic
    from tests.helpers import oracle_path, to_dev
    torch.cuda.set_device(rank)
    dist.init_process_group('nccl', device_id=torch.device('cuda', rank))
    assert dist.get_world_size() == world and dist.get_backend() == 'nccl'
    dev = torch.device('cuda', rank)
    # every rank evaluates ITS OWN batch with the HIP path (weak scaling, no exchange step inside the path)
    d = synthetic.make_batch(B=2, H=96, W=160, boxes_per_img=3, seed=300 + rank, min_box=16, max_box=80)
    ref = oracle_path(d, want_targets=False)
    t = to_dev(d, dev)
    x = t['logits'].clone().requires_grad_(True)
    losses = boxinst_mask_loss(x, t['gt_inds'], t['gt_bboxes'], imgs=t['imgs'], img_metas=d['img_metas'])
    loss, log_vars = bdist.parse_losses(losses)                     # key guard + ONE all-reduce of the logged scalars
    loss.backward()
    # what DDP does next with the mask head's parameter gradients: a bucket all-reduce (here a stand-in buffer)
    bucket = torch.full((537065,), float(rank + 1), device=dev)
    dist.all_reduce(bucket)
    torch.cuda.synchronize()
    g = x.grad.cpu().numpy()[:, 0]
    gerr = float(np.abs(g - ref['grad']).max() / np.abs(ref['grad']).max())
    q.put((rank, float(losses['loss_prj']), float(losses['loss_pairwise']), ref['loss_prj'], ref['loss_pairwise'], gerr,
           {k: float(v) for k, v in log_vars.items()}, float(bucket[0]), float(bucket[-1])))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_hip_path_on_every_rank_over_nccl(dev):
    world = min(2, torch.cuda.device_count())
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=400) for _ in procs)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    mean_prj = sum(o[1] for o in out) / world
    mean_pw = sum(o[2] for o in out) / world
    for rank, lp, lw, rp, rw, gerr, lv, b0, b1 in out:
        assert abs(lp - rp) <= 1e-4 * abs(rp) and abs(lw - rw) <= 1e-4 * abs(rw) and gerr <= 1e-4      # HIP path == oracle on this rank
        assert abs(lv['loss_prj'] - mean_prj) < 1e-6 and abs(lv['loss_pairwise'] - mean_pw) < 1e-6       # rank mean, all ranks
        assert b0 == b1 == world * (world + 1) / 2                                                        # the bucket all-reduce
    if world == 2:
        assert out[0][1] != out[1][1]                                                                     # different batches


@pytest.mark.timeout(900)
def test_bench_spawns_its_own_ranks(dev):
    """`python bench.py --gpus N` without RANK in the environment starts the N ranks itself and reports n_gpus = N."""
    n = min(2, torch.cuda.device_count())
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--spawn', '--steps', '40', '--warmup', '10',
           '--no-cpu-baseline', '--no-extras', '--no-kernel-timing']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=800, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.strip().splitlines() if ln.startswith('{')][-1]
    res = json.loads(line)
    assert res['n_gpus'] == n and res['multi_gpu']['world_size'] == n and res['value'] > 0
    assert res['multi_gpu']['allreduce_alone_us'] > 0
    # with the CPU leg allowed, EVERY rank gates itself against the C oracle before timing and reports it
    cmd2 = [c for c in cmd if c != '--no-cpu-baseline'] + ['--cpu-seconds', '1']
    r = subprocess.run(cmd2, capture_output=True, text=True, timeout=800, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith('{')][-1])
    assert len(res['multi_gpu']['ranks']) == n
    for rk in res['multi_gpu']['ranks']:
        assert rk['parity']['ok'] and rk['parity']['grad_rel_max'] <= 1e-4
    # asking for more GPUs than there are must fail loudly, not report a smaller world
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(torch.cuda.device_count() + 1), '--steps', '5'],
                        capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r2.returncode != 0 and 'visible device' in (r2.stderr + r2.stdout)
