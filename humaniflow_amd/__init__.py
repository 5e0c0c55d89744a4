"""humaniflow_amd -- MI355X-native (gfx950) implementation of HuManiFlow's per-image inference hot path.

Public API mirrors the reference: ``HumaniflowModel`` (models/humaniflow_model.py), ``SMPL`` (models/smpl.py),
``get_humaniflow_cfg_defaults`` (configs/humaniflow_config.py) and the SO(3) helpers of
utils/rigid_transform_utils.py; all arithmetic of the path runs in libhumaniflow_amd.so (include/humaniflow_amd.h).
"""
from .config import get_humaniflow_cfg_defaults, get_benchmark_cfg, CfgNode
from .humaniflow_model import HumaniflowModel, immediate_parent_to_all_ancestors
from .losses import HumaniflowLoss
from .smpl import SMPL, SMPLOutput, synthetic_smpl_model, compute_vertex_variance_from_samples, SMPL_PARENTS

__all__ = ['HumaniflowModel', 'HumaniflowLoss', 'SMPL', 'SMPLOutput', 'synthetic_smpl_model', 'compute_vertex_variance_from_samples',
           'get_humaniflow_cfg_defaults', 'get_benchmark_cfg', 'CfgNode', 'immediate_parent_to_all_ancestors', 'SMPL_PARENTS']
