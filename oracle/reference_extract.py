"""Pull the reference's torch-only functions out of the upstream checkout by AST and execute them.

TEST INFRASTRUCTURE ONLY, build container only: ``/root/reference`` does not exist on the GPU box.
Nothing is copied into this repository -- the source text is read from the reference tree where it
lies, compiled in memory and run.  ``mmdet`` itself cannot be imported here (mmcv, scikit-image
and cv2 are not installed), so the third-party hooks the methods call (``tensor2imgs``,
``color.rgb2lab``, ``pairwise_nlog``) are supplied by the caller.
"""
from __future__ import annotations

import ast
import os
import types
from typing import Dict, Iterable, Optional

REFERENCE_ROOT = os.environ.get('BOXINST_REFERENCE_ROOT', '/root/reference')
HEAD_FILE = 'mmdet/models/dense_heads/condinst_head.py'
PROJ_LOSS_FILE = 'mmdet/models/losses/box_projection_loss.py'

FUNCTIONS = ('compute_pairwise_term', 'dice_coefficient', 'compute_project_term', 'unfold_wo_center',
             'get_image_color_similarity', 'get_original_image', 'aligned_bilinear')
METHODS = ('loss', 'get_targets', 'get_bitmasks_from_boxes', 'forward', 'parse_dynamic_params')


def available() -> bool:
    return os.path.exists(os.path.join(REFERENCE_ROOT, HEAD_FILE))


def _strip_decorators(node):
    node.decorator_list = []
    return node


def load(extra_globals: Optional[Dict] = None, functions: Iterable[str] = FUNCTIONS,
         methods: Iterable[str] = METHODS) -> types.SimpleNamespace:
    """Return a namespace holding the reference's module-level functions and, as plain functions
    taking ``self``, the ``CondInstMaskHead`` methods on the path (decorators removed:
    ``force_fp32`` is an mmcv no-op for this head)."""
    import numpy as np
    import torch
    import torch.nn as nn
    import torch.nn.functional as F

    path = os.path.join(REFERENCE_ROOT, HEAD_FILE)
    with open(path) as fh:
        tree = ast.parse(fh.read(), filename=path)
    body = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in functions:
            body.append(_strip_decorators(node))
        elif isinstance(node, ast.ClassDef) and node.name == 'CondInstMaskHead':
            for sub in node.body:
                if isinstance(sub, ast.FunctionDef) and sub.name in methods:
                    sub.name = 'CondInstMaskHead_' + sub.name
                    body.append(_strip_decorators(sub))
    mod = ast.Module(body=body, type_ignores=[])
    ast.fix_missing_locations(mod)
    env = {'torch': torch, 'nn': nn, 'F': F, 'np': np}
    if extra_globals:
        env.update(extra_globals)
    exec(compile(mod, path, 'exec'), env)
    names = list(functions) + ['CondInstMaskHead_' + m for m in methods]
    return types.SimpleNamespace(**{n: env[n] for n in names if n in env}, _env=env)


DISCOBOX_FILE = 'mmdet/models/dense_heads/discobox_head.py'
DISCOBOX_FUNCTIONS = ('dice_loss', 'mil_loss')
DISCOBOX_CLASSES = ('MeanField',)


def load_discobox(extra_globals: Optional[Dict] = None) -> types.SimpleNamespace:
    """the reference's ``MeanField`` module (discobox_head.py:585-651) and its
    ``dice_loss`` / ``mil_loss`` (:542-562), compiled in memory from the reference tree."""
    import numpy as np
    import torch
    import torch.nn as nn
    import torch.nn.functional as F

    path = os.path.join(REFERENCE_ROOT, DISCOBOX_FILE)
    with open(path) as fh:
        tree = ast.parse(fh.read(), filename=path)
    body = [n for n in tree.body
            if (isinstance(n, ast.FunctionDef) and n.name in DISCOBOX_FUNCTIONS)
            or (isinstance(n, ast.ClassDef) and n.name in DISCOBOX_CLASSES)]
    mod = ast.Module(body=body, type_ignores=[])
    ast.fix_missing_locations(mod)
    env = {'torch': torch, 'nn': nn, 'F': F, 'np': np}
    if extra_globals:
        env.update(extra_globals)
    exec(compile(mod, path, 'exec'), env)
    return types.SimpleNamespace(**{n: env[n] for n in DISCOBOX_FUNCTIONS + DISCOBOX_CLASSES}, _env=env)


LEVELSET_FILE = 'mmdet/models/losses/levelset_loss.py'
LEVELSET_NAMES = ('LevelsetLoss', 'region_levelset', 'length_regularization', 'LCM', 'LocalConsistencyModule')


def load_levelset(extra_globals: Optional[Dict] = None) -> types.SimpleNamespace:
    """the reference's ``LevelsetLoss`` / ``region_levelset`` / ``LCM`` / ``LocalConsistencyModule``
    (levelset_loss.py:7-126) and ``BoxProjectionLoss`` (box_projection_loss.py:5-43); registry decorators removed."""
    import numpy as np
    import torch
    import torch.nn as nn
    import torch.nn.functional as F

    env = {'torch': torch, 'nn': nn, 'F': F, 'np': np}
    if extra_globals:
        env.update(extra_globals)
    names = []
    for rel, wanted in ((LEVELSET_FILE, LEVELSET_NAMES), (PROJ_LOSS_FILE, ('BoxProjectionLoss',))):
        path = os.path.join(REFERENCE_ROOT, rel)
        with open(path) as fh:
            tree = ast.parse(fh.read(), filename=path)
        body = [_strip_decorators(n) for n in tree.body
                if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in wanted]
        mod = ast.Module(body=body, type_ignores=[])
        ast.fix_missing_locations(mod)
        exec(compile(mod, path, 'exec'), env)
        names += list(wanted)
    return types.SimpleNamespace(**{n: env[n] for n in names}, _env=env)
