"""Data-parallel glue for the loss path: one process per GPU, ``torch.distributed`` over RCCL/xGMI.

The BoxInst mask loss has no exchange step -- ``loss_prj`` is a rank-local mean and ``loss_pairwise`` is
normalised by the rank-local weight sum (condinst_head.py:143, :1327-1328) -- so ranks are independent
replicas and the path itself issues no collective.  What the reference does all-reduce around it is the
*logging* of the loss scalars in ``BaseDetector._parse_losses`` (mmdet/models/detectors/base.py:176-219):
one ``all_reduce`` per logged key plus a key-count check, each followed by ``.item()``.  On xGMI a
collective is latency-, not bandwidth-bound at this size, so :func:`parse_losses` stacks every logged
scalar (and the key count) into ONE tensor and issues ONE all-reduce; values stay on the device until
the caller asks for them.
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


def parse_losses(losses: Dict[str, torch.Tensor], sync: bool = True) -> Tuple[torch.Tensor, 'OrderedDict[str, torch.Tensor]']:
    """``BaseDetector._parse_losses`` (base.py:176-219) with a single collective.

    Returns ``(loss, log_vars)``: ``loss`` = sum of every entry whose key contains 'loss' (rank-local,
    attached to the autograd graph); ``log_vars`` = every entry (plus 'loss') averaged over ranks,
    as 0-dim device tensors (the reference calls ``.item()`` on each; do that only when logging).
    Raises like the reference (base.py:201-210) when ranks disagree on the number of logged keys.
    """
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
        packed = torch.stack([log_vars[k].detach().float() for k in keys] +
                             [torch.tensor(float(len(keys)), device=loss.device)])
        dist.all_reduce(packed)                                   # one RCCL all-reduce for everything
        # the key-count guard: every rank contributed len(keys) -> the sum must be world * len(keys).
        # Checked without a sync: the flag rides along and poisons the logged values with NaN on mismatch.
        ok = packed[-1] == float(world * len(keys))
        packed = torch.where(ok, packed / world, torch.full_like(packed, float('nan')))
        for i, k in enumerate(keys):
            log_vars[k] = packed[i]
    else:
        log_vars = OrderedDict((k, v.detach()) for k, v in log_vars.items())
    return loss, log_vars
