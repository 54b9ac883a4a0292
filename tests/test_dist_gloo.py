"""CPU, 2 processes over gloo: the N > 1 path.  The loss path has no exchange step;
what is collective is the logging of the loss scalars (boxinstseg_amd/dist.py <-> base.py:176-219)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mismatch, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from boxinstseg_amd import dist as bdist
    from oracle import torch_oracle as to
    from boxinstseg_amd import synthetic
    r, w, _ = bdist.init_distributed('gloo')
    assert (r, w) == (rank, world) and dist.get_backend() == 'gloo'
    # every rank evaluates ITS OWN batch (weak scaling): the CPU oracle stands in for the HIP path here
    d = synthetic.make_batch(B=1, H=64, W=64, boxes_per_img=2, seed=100 + rank, min_box=16, max_box=40)
    x = torch.from_numpy(d['mask_logits']).requires_grad_(True)
    losses = to.mask_loss(torch.from_numpy(d['imgs']), d['img_metas'], x, torch.from_numpy(d['gt_inds']),
                          [torch.from_numpy(b) for b in d['gt_bboxes']])
    if mismatch and rank == 1:
        losses['loss_extra'] = losses['loss_prj'] * 0
    loss, log_vars = bdist.parse_losses(losses)
    loss.backward()                                    # gradients stay rank-local: nothing to reduce on the path
    q.put((rank, float(losses['loss_prj']), float(losses['loss_pairwise']), {k: float(v) for k, v in log_vars.items()},
           float(x.grad.abs().sum())))
    dist.barrier()
    dist.destroy_process_group()


def _run(mismatch):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mismatch, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return out


@pytest.mark.timeout(300)
def test_two_ranks_single_allreduce_logging():
    (r0, p0, w0, lv0, g0), (r1, p1, w1, lv1, g1) = _run(False)
    assert p0 != p1 and g0 > 0 and g1 > 0                       # different batches per rank, local gradients
    for lv in (lv0, lv1):                                       # every rank logs the rank-mean of every key
        assert abs(lv['loss_prj'] - 0.5 * (p0 + p1)) < 1e-6
        assert abs(lv['loss_pairwise'] - 0.5 * (w0 + w1)) < 1e-6
        assert abs(lv['loss'] - 0.5 * (p0 + p1 + w0 + w1)) < 1e-6
    assert lv0 == lv1


def _mismatch_worker(rank, world, port, same_count, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from boxinstseg_amd import dist as bdist
    bdist.init_distributed('gloo')
    losses = {'loss_prj': torch.tensor(1.0 + rank), 'loss_pairwise': torch.tensor(2.0)}
    if rank == 1:
        if same_count:
            losses = {'loss_prj': losses['loss_prj'], 'loss_other': losses['loss_pairwise']}     # same count, other names
        else:
            losses['loss_extra'] = torch.tensor(0.0)
    try:
        bdist.parse_losses(losses)
        q.put((rank, 'no error'))
    except AssertionError as e:
        q.put((rank, 'assert: ' + str(e)[:60]))
    dist.barrier()                               # nobody hangs: the guard has a fixed size on every rank
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize('same_count', [False, True])
def test_two_ranks_key_mismatch_raises_on_every_rank(same_count):
    """base.py:201-210: ranks that log different keys must fail loudly BEFORE a collective of differing size is issued
    (NCCL / RCCL would hang or reduce garbage; only gloo rejects it by itself)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mismatch_worker, args=(r, 2, port, same_count, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert out[0].startswith('assert') and out[1].startswith('assert'), out
