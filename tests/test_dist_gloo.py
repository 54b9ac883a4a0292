"""CPU, 2 processes over gloo: the N > 1 path.  The loss path has no exchange step;
what is collective is the logging of the loss scalars (boxinstseg_amd/dist.py <-> base.py:176-219).  The per-rank losses are the
product's own -- a recorded run of the HIP path (tests/golden/make_hip_run.py) -- not the oracle's."""
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


GOLD = os.path.join(ROOT, 'tests', 'golden', 'hip_run_2ranks.npz')


def _worker(rank, world, port, nan_rank, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    import warnings
    import numpy as np
    from boxinstseg_amd import _lib, dist as bdist, functional as Fh
    r, w, _ = bdist.init_distributed('gloo')
    assert (r, w) == (rank, world) and dist.get_backend() == 'gloo'
    # every rank holds ITS OWN batch's losses (weak scaling) -- what the PRODUCT (the HIP path through CondInstMaskHead.loss) produced on a
    # GPU box for batch `100 + rank`, recorded by tests/golden/make_hip_run.py: there is no GPU here and the product has no CPU path
    g = np.load(GOLD)
    lp = torch.tensor(float(g[f'rank{rank}_loss_prj']), requires_grad=True)
    lw = torch.tensor(float(g[f'rank{rank}_loss_pairwise']), requires_grad=True)
    losses = {'loss_prj': lp * 1.0, 'loss_pairwise': lw * (float('nan') if nan_rank == rank else 1.0)}
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter('always')
        loss, log_vars = bdist.parse_losses(losses)
    loss.backward()                                    # gradients stay rank-local: nothing to reduce on the path
    host = bdist.to_host({k: v for k, v in log_vars.items()}) if nan_rank is None else {k: float(v) for k, v in log_vars.items()}
    q.put((rank, float(lp), float(lw), dict(host), float(lp.grad), bool(Fh.eval_launch_flags() & _lib.EVAL_TWO_LAUNCHES),
           sum('two-launch' in str(c.message) for c in caught)))
    dist.barrier()
    dist.destroy_process_group()


def _run(nan_rank):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, nan_rank, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return out


@pytest.mark.timeout(300)
def test_two_ranks_single_allreduce_logging():
    """parse_losses over two gloo ranks on the losses a recorded run of the HIP path produced (tests/golden/hip_run_2ranks.npz)."""
    (r0, p0, w0, lv0, g0, two0, _), (r1, p1, w1, lv1, g1, two1, _) = _run(None)
    assert p0 != p1 and g0 == 1.0 and g1 == 1.0                 # different batches per rank, local gradients
    for lv in (lv0, lv1):                                       # every rank logs the rank-mean of every key
        assert abs(lv['loss_prj'] - 0.5 * (p0 + p1)) < 1e-6
        assert abs(lv['loss_pairwise'] - 0.5 * (w0 + w1)) < 1e-6
        assert abs(lv['loss'] - 0.5 * (p0 + p1 + w0 + w1)) < 1e-6
    assert lv0 == lv1
    assert not two0 and not two1                                # finite losses: nobody falls back


@pytest.mark.timeout(300)
def test_a_faulted_evaluation_on_one_rank_is_noticed_on_every_rank():
    """An evaluation whose bounded in-kernel wait ran out hands back NaN losses (include/boxinst_hip.h section 3).  The key guard's one
    host read carries a non-finite flag under MAX, so EVERY rank -- not only the one that faulted -- switches its following evaluations to
    the two-launch form (functional.note_fault), in step."""
    res = _run(1)
    for rank, _, _, lv, _, two, warned in res:
        assert two and warned == 1, (rank, two, warned)
        assert lv['loss_pairwise'] != lv['loss_pairwise']       # the logged mean is NaN, as the reference's would be


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
