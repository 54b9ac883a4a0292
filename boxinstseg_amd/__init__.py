"""boxinstseg_amd -- the BoxInst box-supervised mask-loss path of LiWentomng/BoxInstSeg, rebuilt
MI355X-native (gfx950 HIP kernels behind a C ABI; see include/boxinst_hip.h and DESIGN.md).

Public surface (mirrors the reference's for this path):
    pairwise_nlog                      <-> mmdet.ops.pairwise.pairwise_nlog
    pairwise_nlog_forward / _backward  <-> mmdet.ops.pairwise.pairwise_ext
    CondInstMaskHead                   <-> mmdet.models.dense_heads.CondInstMaskHead (loss path)
    boxinst_mask_loss, color_affinity, box_bitmasks : functional form of the same kernels
    MeanField, dice_loss, mil_loss     <-> mmdet.models.dense_heads.discobox_head
    BoxProjectionLoss, LevelsetLoss, LocalConsistencyModule, LCM <-> mmdet.models.losses
    MinimumSpanningTree, TreeFilter2D, mst, bfs, refine <-> mmdet.ops.tree_filter
"""
from .pairwise import PairwiseNLog, pairwise_nlog, pairwise_nlog_backward, pairwise_nlog_forward
from .functional import BoxInstMaskLoss, box_bitmasks, boxinst_mask_loss, color_affinity
from .dynamic import DynamicMaskHead, dynamic_mask_forward
from .mask_head import CondInstMaskHead
from .discobox import MeanField, dice_loss, meanfield_forward, meanfield_kernel, mil_loss
from .levelset import LCM, BoxProjectionLoss, LevelsetLoss, LocalConsistencyModule, region_levelset
from .registry import HEADS, LOSSES, build_head, build_loss
from .tree_filter import MinimumSpanningTree, TreeFilter2D, bfs, mst, refine
from .config import load_config

__all__ = ['pairwise_nlog', 'pairwise_nlog_forward', 'pairwise_nlog_backward', 'PairwiseNLog',
           'boxinst_mask_loss', 'BoxInstMaskLoss', 'dynamic_mask_forward', 'DynamicMaskHead', 'color_affinity', 'box_bitmasks',
           'CondInstMaskHead', 'HEADS', 'build_head', 'load_config',
           'MeanField', 'meanfield_kernel', 'meanfield_forward', 'dice_loss', 'mil_loss',
           'BoxProjectionLoss', 'LevelsetLoss', 'region_levelset', 'LocalConsistencyModule', 'LCM', 'LOSSES', 'build_loss',
           'MinimumSpanningTree', 'TreeFilter2D', 'mst', 'bfs', 'refine']
__version__ = '0.1.0'
