"""Data-parallel glue for the loss path: one process per GPU, ``torch.distributed`` over RCCL/xGMI.

The BoxInst mask loss has no exchange step -- ``loss_prj`` is a rank-local mean and ``loss_pairwise`` is
normalised by the rank-local weight sum (condinst_head.py:143, :1327-1328) -- so ranks are independent
replicas and the path itself issues no collective.  What the reference does all-reduce around it is the
*logging* of the loss scalars in ``BaseDetector._parse_losses`` (mmdet/models/detectors/base.py:176-219):
one ``all_reduce`` per logged key plus a key-count check, each followed by ``.item()``.  On xGMI a
collective is latency-, not bandwidth-bound at this size, so :func:`parse_losses` stacks every logged
scalar into ONE tensor and issues ONE all-reduce for the values (after the reference's fixed-size key guard);
values stay on the device until the caller asks for them.
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise from the torchrun environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).
    backend: 'nccl' (= RCCL on ROCm) when a GPU is present, else 'gloo'.  Returns (rank, world, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC only on this driver
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)
    return rank, world, local_rank


def _key_digest(keys) -> int:
    """order-sensitive 31-bit digest of the logged key names (same on every rank iff the key lists agree)."""
    import zlib
    return zlib.crc32('\x00'.join(keys).encode()) & 0x7fffffff


def parse_losses(losses: Dict[str, torch.Tensor], sync: bool = True, check_keys: bool = True
                 ) -> Tuple[torch.Tensor, 'OrderedDict[str, torch.Tensor]']:
    """``BaseDetector._parse_losses`` (base.py:176-219) with one collective for the values.

    Returns ``(loss, log_vars)``: ``loss`` = sum of every entry whose key contains 'loss' (rank-local,
    attached to the autograd graph); ``log_vars`` = every entry (plus 'loss') averaged over ranks,
    as 0-dim device tensors (the reference calls ``.item()`` on each; do that only when logging).

    The key guard of the reference (base.py:201-210: ranks that disagree on the logged keys would otherwise issue
    collectives of different sizes -- "GPUs will wait infinitely") is a FIXED-size all-reduce issued first, like the
    reference's: [count, -count, digest, -digest] under MAX; a mismatch raises ``AssertionError`` on every rank before
    the value collective is issued.  It costs one host read (the reference pays one per key); ``check_keys=False`` skips
    it for callers whose key set cannot vary."""
    log_vars: 'OrderedDict[str, torch.Tensor]' = OrderedDict()
    for name, value in losses.items():
        if isinstance(value, torch.Tensor):
            log_vars[name] = value.mean()
        elif isinstance(value, (list, tuple)):
            log_vars[name] = sum(v.mean() for v in value)
        else:
            raise TypeError(f'{name} is not a tensor or list of tensors')
    loss = sum(v for k, v in log_vars.items() if 'loss' in k)
    log_vars['loss'] = loss
    if sync and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        world = dist.get_world_size()
        keys = list(log_vars.keys())
        if check_keys:
            n, dg = float(len(keys)), float(_key_digest(keys))
            guard = torch.tensor([n, -n, dg, -dg, 0.0], device=loss.device, dtype=torch.float64)
            # the host read below is the one sync point of the call: a non-finite total rides along (an evaluation whose bounded
            # in-kernel wait ran out hands back NaN losses, include/boxinst_hip.h section 3) and is seen by EVERY rank
            guard[4] = (~torch.isfinite(loss.detach())).to(torch.float64)
            dist.all_reduce(guard, op=dist.ReduceOp.MAX)
            g = guard.tolist()
            assert g[0] == -g[1] == n and g[2] == -g[3] == dg, (
                f'loss log variables differ across ranks: this rank logs {len(keys)} keys {",".join(keys)}; '
                f'count range [{-g[1]:.0f}, {g[0]:.0f}]')
            if g[4] != 0.0:
                _note_fault('a non-finite loss on some rank')
        packed = torch.stack([log_vars[k].detach().float() for k in keys])
        dist.all_reduce(packed)                                   # one RCCL all-reduce for every logged value
        packed = packed / world
        for i, k in enumerate(keys):
            log_vars[k] = packed[i]
    else:
        log_vars = OrderedDict((k, v.detach()) for k, v in log_vars.items())
    return loss, log_vars


def _note_fault(what: str) -> None:
    from . import functional
    functional.note_fault(what)


def to_host(log_vars: Dict[str, torch.Tensor]) -> 'OrderedDict[str, float]':
    """The ``.item()`` of every logged value (what base.py:216-217 does per key) as ONE device-to-host copy.  This is where losses
    reach the host anyway, so it is also where a faulted evaluation (NaN losses) is noticed: from then on this thread's evaluations
    take the two-launch form (``functional.note_fault``)."""
    keys = list(log_vars.keys())
    if not keys:
        return OrderedDict()
    vals = torch.stack([log_vars[k].detach().float().reshape(()) for k in keys]).tolist()
    out = OrderedDict(zip(keys, vals))
    if any(v != v or v in (float('inf'), float('-inf')) for k, v in out.items() if k in ('loss_prj', 'loss_pairwise')):
        _note_fault('non-finite loss_prj / loss_pairwise')
    return out


def exclude_iter_from_ddp_broadcast(model) -> list:
    """Call on the model BEFORE wrapping it in ``DistributedDataParallel``: lists every ``CondInstMaskHead._iter`` buffer in
    ``model._ddp_params_and_buffers_to_ignore``.  DDP's default ``broadcast_buffers=True`` rewrites each buffer in place at the
    start of every forward; ``_iter`` is identical on all ranks by construction (each rank counts its own calls,
    condinst_head.py:1297), so the broadcast only costs a small collective per iteration (the head keeps no host copy of the
    counter any more, so a rewrite is harmless).  Optional.  Returns the names added."""
    from .mask_head import CondInstMaskHead
    names = [f'{prefix}._iter' if prefix else '_iter' for prefix, m in model.named_modules() if isinstance(m, CondInstMaskHead)]
    have = list(getattr(model, '_ddp_params_and_buffers_to_ignore', []))
    model._ddp_params_and_buffers_to_ignore = have + [n for n in names if n not in have]
    return names

