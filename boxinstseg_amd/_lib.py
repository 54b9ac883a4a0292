"""ctypes binding of libboxinst_hip.so -- the C ABI declared in include/boxinst_hip.h.

There is no fallback: if the shared library is missing (or a status is non-zero) this raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

from . import build as _build

c_void_p, c_int, c_float, c_size_t = C.c_void_p, C.c_int, C.c_float, C.c_size_t

BXI_MAX_IMAGES = 64
BXI_ABI_VERSION = 7
# `flags` of bxi_boxinst_eval_f32 / bxi_boxinst_head_eval_f32 (include/boxinst_hip.h)
EVAL_SINGLE_LAUNCH, EVAL_TWO_LAUNCHES, EVAL_TILE_ROWS_8, EVAL_TILE_ROWS_4, EVAL_SHARED_DEVICE, EVAL_TARGETS_READY, EVAL_WAITS_GIVE_UP = 1, 2, 4, 8, 16, 32, 64

STATUS = {0: 'BXI_OK', -1: 'BXI_ERR_NULL_POINTER', -2: 'BXI_ERR_BAD_SHAPE', -3: 'BXI_ERR_BAD_ARGUMENT',
          -4: 'BXI_ERR_UNSUPPORTED', -5: 'BXI_ERR_WORKSPACE', -6: 'BXI_ERR_LAUNCH', -7: 'BXI_ERR_NO_DEVICE'}
BXI_ERR_UNSUPPORTED = -4


class ImageBatch(C.Structure):
    """struct bxi_image_batch"""
    _fields_ = [('imgs', c_void_p), ('B', c_int), ('Hc', c_int), ('Wc', c_int),
                ('img_h_host', C.POINTER(c_int)), ('img_w_host', C.POINTER(c_int)),
                ('rows_removed_host', C.POINTER(c_int)),
                ('mean', C.c_double * 3), ('std', C.c_double * 3), ('to_rgb', c_int),
                ('image_masks', c_void_p)]


class Instances(C.Structure):
    """struct bxi_instances"""
    _fields_ = [('logits', c_void_p), ('N', c_int), ('h', c_int), ('w', c_int),
                ('gt_inds', c_void_p), ('boxes_per_img_host', C.POINTER(c_void_p)),
                ('gt_count_host', C.POINTER(c_int)), ('B', c_int), ('Hc', c_int), ('Wc', c_int),
                ('stride', c_int), ('iter_counter', c_void_p)]


# name -> (restype, argtypes); must list every symbol of include/boxinst_hip.h and include/boxinst_hip_dev.h (tests check this)
SIGNATURES = {
    'bxi_abi_version': (c_int, []),
    'bxi_status_string': (C.c_char_p, [c_int]),
    'bxi_last_hip_error': (c_int, []),
    'bxi_check_device': (c_int, [c_int]),
    'bxi_dev_set_launch_hook': (None, [c_void_p, c_void_p]),           # boxinst_hip_dev.h (bench / tests only)
    'bxi_dev_set_tree_level_walk': (None, [c_int]),                    # boxinst_hip_dev.h (tests only)
    'bxi_dev_sol_eval_f32': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t,
                                     c_void_p]),                        # boxinst_hip_dev.h (bench only)
    'bxi_dev_sol_pairwise_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),   # boxinst_hip_dev.h (bench only)
    'bxi_pairwise_nlog_forward_f32': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'bxi_pairwise_nlog_forward_f64': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'bxi_pairwise_nlog_backward_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                               c_void_p, c_void_p]),
    'bxi_pairwise_nlog_backward_f64': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                               c_void_p, c_void_p]),
    'bxi_color_affinity_f32': (c_int, [C.POINTER(ImageBatch), c_int, c_int, c_int, c_float, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p]),
    'bxi_box_bitmasks_f32': (c_int, [C.POINTER(c_void_p), C.POINTER(c_int), c_int, c_int, c_int, c_int, c_int,
                                     c_void_p, c_void_p]),
    'bxi_boxinst_loss_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'bxi_boxinst_loss_state_bytes': (c_size_t, [c_int, c_int, c_int]),
    'bxi_boxinst_loss_state_status_offset': (c_size_t, [c_int, c_int, c_int]),
    'bxi_boxinst_loss_state_warmup_offset': (c_size_t, [c_int, c_int, c_int]),
    'bxi_boxinst_loss_fwd_bwd_f32': (c_int, [C.POINTER(Instances), c_void_p, c_int, c_int, c_float, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'bxi_boxinst_loss_backward_f32': (c_int, [C.POINTER(Instances), c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                             c_void_p]),
    'bxi_boxinst_eval_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    'bxi_boxinst_eval_workspace_lab_offset': (c_size_t, []),
    'bxi_boxinst_eval_workspace_init': (c_int, [c_void_p, c_size_t, c_void_p]),
    'bxi_boxinst_targets_f32': (c_int, [C.POINTER(ImageBatch), C.POINTER(c_void_p), C.POINTER(c_int), c_int, c_int, c_int, c_float, c_void_p,
                                        c_size_t, c_void_p]),
    'bxi_boxinst_eval_f32': (c_int, [C.POINTER(ImageBatch), C.POINTER(Instances), c_int, c_int, c_float, c_float,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, C.c_uint, c_void_p]),
    'bxi_boxinst_head_eval_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, C.c_uint, c_void_p]),
    'bxi_boxinst_grad_rescale_f32': (c_int, [C.POINTER(Instances), c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                             c_void_p]),
    'bxi_boxinst_grad_rescale_nhw_f32': (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    'bxi_dynamic_mask_forward_f32': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p,
                                             c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'bxi_dynamic_mask_backward_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    'bxi_dynamic_mask_generic_forward_f32': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                                     c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'bxi_dynamic_mask_generic_backward_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    'bxi_dynamic_mask_generic_backward_f32': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                                      c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                                      c_void_p, c_size_t, c_void_p]),
    'bxi_dynamic_mask_backward_f32': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p,
                                              c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_size_t, c_void_p]),
    'bxi_meanfield_kernel_f32': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, C.c_float, C.c_float, C.c_float,
                                         c_void_p, c_void_p]),
    'bxi_meanfield_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'bxi_meanfield_forward_f32': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                          c_int, C.c_float, c_void_p, C.c_float, c_void_p, c_void_p, c_void_p, c_size_t,
                                          c_void_p]),
    'bxi_dice_loss_forward_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, C.c_int64, c_void_p, c_void_p, c_void_p]),
    'bxi_dice_loss_backward_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, C.c_int64, c_void_p, c_void_p, c_void_p,
                                           c_void_p]),
    'bxi_mil_loss_state_bytes': (c_size_t, [c_int, c_int, c_int]),
    'bxi_mil_loss_forward_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'bxi_mil_loss_backward_f32': (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'bxi_projection_loss_forward_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, C.c_float, c_void_p, c_void_p, c_void_p]),
    'bxi_levelset_state_bytes': (c_size_t, [c_int, c_int]),
    'bxi_levelset_loss_forward_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, C.c_float, c_void_p,
                                              c_void_p, c_void_p]),
    'bxi_levelset_loss_backward_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, C.c_float, c_void_p,
                                               c_void_p, c_void_p, c_void_p, c_void_p]),
    'bxi_lcm_affinity_f32': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, C.c_float, c_void_p, c_void_p]),
    'bxi_lcm_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'bxi_lcm_refine_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t,
                                   c_void_p]),
    'bxi_mst_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'bxi_mst_forward_i32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    'bxi_bfs_workspace_bytes': (c_size_t, [c_int, c_int]),
    'bxi_bfs_forward_i32': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'bxi_tree_refine_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'bxi_tree_refine_forward_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                            c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'bxi_tree_refine_backward_feature_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                                     c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    'bxi_tree_refine_backward_weight_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'bxi_tree_refine_backward_weight_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                    c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                                    c_void_p, c_size_t, c_void_p]),
}

LAUNCH_HOOK = C.CFUNCTYPE(None, C.c_char_p, c_int, c_void_p, c_void_p)

_lib: Optional[C.CDLL] = None


def lib_path() -> str:
    return _build.LIB_PATH


def load() -> C.CDLL:
    """dlopen the in-tree shared library (never a fallback: raises if it is not there)."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise RuntimeError(
                f'{path} is missing: the BoxInst HIP extension has not been built '
                '(run `python -c "import __graft_entry__ as g; g.build()"` or `python -m boxinstseg_amd.build`). '
                'boxinstseg_amd has no CPU or PyTorch fallback for this path.')
        lib = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.bxi_abi_version() != BXI_ABI_VERSION:
            raise RuntimeError(f'{path}: ABI version {lib.bxi_abi_version()} != {BXI_ABI_VERSION}')
        _lib = lib
    return _lib


def status_string(status: int) -> str:
    return load().bxi_status_string(status).decode()


class BoxInstHipError(RuntimeError):
    def __init__(self, fn: str, status: int):
        self.status = status
        detail = status_string(status)
        if status == -6:
            detail += f' [hipError_t {load().bxi_last_hip_error()}]'
        super().__init__(f'{fn}: {STATUS.get(status, status)} -- {detail}')


def check(fn: str, status: int) -> None:
    if status != 0:
        raise BoxInstHipError(fn, status)


def int_array(values) -> C.Array:
    values = [int(v) for v in values]
    return (c_int * max(len(values), 1))(*values)


def ptr_array(values) -> C.Array:
    values = [int(v) for v in values]
    return (c_void_p * max(len(values), 1))(*values)
