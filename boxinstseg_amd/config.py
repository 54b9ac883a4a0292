"""mmcv-free loader for mmdet python-dict config files (``configs/boxinst/*.py``).

The reference parses configs with ``mmcv.Config.fromfile`` (``tools/train.py:111-120``): a config is
a python file of assignments whose ``_base_`` names files to merge underneath it (dict keys merge
recursively; ``_delete_=True`` replaces).  mmcv is not installable in the build environment, and the
loss path needs nothing else of it, so this is the ~40 lines of that behaviour the drop-in needs to
show that the reference's config files build the new head unchanged.
"""
from __future__ import annotations

import os
import types
from typing import Any, Dict


def _merge(base: Dict[str, Any], top: Dict[str, Any]) -> Dict[str, Any]:
    out = dict(base)
    for k, v in top.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get('_delete_', False):
            out[k] = _merge(out[k], v)
        elif isinstance(v, dict):
            out[k] = {kk: vv for kk, vv in v.items() if kk != '_delete_'}
        else:
            out[k] = v
    return out


def load_config(path: str) -> Dict[str, Any]:
    """Return the merged config dict of an mmdet config file."""
    path = os.path.abspath(path)
    scope: Dict[str, Any] = {}
    with open(path) as fh:
        exec(compile(fh.read(), path, 'exec'), scope)
    cfg = {k: v for k, v in scope.items()
           if not k.startswith('__') and not isinstance(v, (types.ModuleType, types.FunctionType, type))}
    bases = cfg.pop('_base_', [])
    if isinstance(bases, str):
        bases = [bases]
    merged: Dict[str, Any] = {}
    for b in bases:
        merged = _merge(merged, load_config(os.path.join(os.path.dirname(path), b)))
    return _merge(merged, cfg)
