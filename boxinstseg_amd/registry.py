"""``HEADS`` registry for the drop-in head.

With mmdet/mmcv installed, ``CondInstMaskHead`` is registered into mmdet's own ``HEADS`` registry
(``mmdet/models/builder.py:7-15``; ``force=True`` replaces the stock class), so
``configs/boxinst/*.py`` resolve ``type='CondInstMaskHead'`` to the MI355X implementation with no
edit.  Without mmcv (this build environment) a minimal registry with the same
``register_module`` / ``build`` surface stands in, enough to build the head from a config dict.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional


class Registry:
    def __init__(self, name: str):
        self.name = name
        self._modules: Dict[str, type] = {}

    @property
    def module_dict(self) -> Dict[str, type]:
        return self._modules

    def get(self, key: str) -> Optional[type]:
        return self._modules.get(key)

    def register_module(self, name: Optional[str] = None, force: bool = False, module: Optional[type] = None):
        def _do(cls: type) -> type:
            key = name or cls.__name__
            if key in self._modules and not force:
                raise KeyError(f'{key} is already registered in {self.name}')
            self._modules[key] = cls
            return cls
        if module is not None:
            return _do(module)
        return _do

    def build(self, cfg: dict, default_args: Optional[dict] = None):
        if not isinstance(cfg, dict) or 'type' not in cfg:
            raise TypeError('cfg must be a dict with a "type" key')
        args = dict(cfg)
        typ = args.pop('type')
        cls = typ if isinstance(typ, type) else self.get(typ)
        if cls is None:
            raise KeyError(f'{typ} is not in the {self.name} registry')
        for k, v in (default_args or {}).items():
            args.setdefault(k, v)
        return cls(**args)


class _MMDetHeads:
    """Adapter: register into mmdet's registry, replacing the stock class of the same name."""

    def __init__(self, reg):
        self._reg = reg
        self.name = getattr(reg, 'name', 'models')

    def register_module(self, name=None, force=True, module=None):
        return self._reg.register_module(name=name, force=True, module=module)

    def get(self, key):
        return self._reg.get(key)

    def build(self, cfg, default_args=None):
        return self._reg.build(cfg, default_args=default_args)


def _make_heads():
    try:  # mmdet is not installable in the build container: exercised against a stand-in module, tests/test_host_cpu.py
        from mmdet.models.builder import HEADS as mm_heads
        return _MMDetHeads(mm_heads)
    except Exception:
        return Registry('head')


HEADS = _make_heads()


def _make_losses():
    try:
        from mmdet.models.builder import LOSSES as mm_losses
        return _MMDetHeads(mm_losses)
    except Exception:
        return Registry('loss')


LOSSES = _make_losses()      # BoxProjectionLoss, LevelsetLoss (mmdet/models/builder.py: LOSSES)


def build_head(cfg: dict, default_args: Optional[dict] = None):
    """``mmdet.models.builder.build_head`` for the heads this package provides."""
    return HEADS.build(cfg, default_args=default_args)


def build_loss(cfg: dict, default_args: Optional[dict] = None):
    """``mmdet.models.builder.build_loss`` for the losses this package provides."""
    return LOSSES.build(cfg, default_args=default_args)
