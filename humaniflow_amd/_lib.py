"""ctypes binding of libhumaniflow_amd.so (the C ABI declared in include/humaniflow_amd.h).

There is NO fallback: if the shared library is missing or stale it is (re)built with hipcc (in-tree); if
that fails, or a compute entry point is reached without a HIP device, a RuntimeError is raised.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libhumaniflow_amd.so')

c_float_p = C.POINTER(C.c_float)
c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int32)


class SmplHost(C.Structure):
    _fields_ = [('num_betas', C.c_int),
                ('v_template_host', c_float_p), ('shapedirs_host', c_float_p), ('posedirs_host', c_float_p),
                ('J_regressor_host', c_float_p), ('lbs_weights_host', c_float_p), ('parents_host', c_int_p),
                ('extra_vertex_ids_host', c_int_p), ('J_regressor_more_host', c_float_p)]


class FlowHost(C.Structure):
    _fields_ = [('feat_dim', C.c_int), ('num_betas', C.c_int), ('isgc_dim', C.c_int), ('ctx_dim', C.c_int),
                ('num_transforms', C.c_int), ('hidden', C.c_int * 3), ('count_bins', C.c_int),
                ('radius', C.c_float), ('base_std', C.c_float),
                ('parents_host', c_int_p), ('w_isgc_host', c_float_p), ('b_isgc_host', c_float_p),
                ('w_ctx_host', C.POINTER(c_float_p)), ('b_ctx_host', C.POINTER(c_float_p)),
                ('w_nn_host', C.POINTER(c_float_p)), ('b_nn_host', C.POINTER(c_float_p)),
                ('transform_type', C.c_int), ('permute_type', C.c_int), ('num_hidden', C.c_int), ('hidden_dims', C.c_int * 4),
                ('plu_num_hidden', C.c_int), ('plu_hidden_dims', C.c_int * 4),
                ('w_plu_host', C.POINTER(c_float_p)), ('b_plu_host', C.POINTER(c_float_p)), ('lu_host', C.POINTER(c_float_p)),
                ('force_generic', C.c_int)]


class FlowGradRecords(C.Structure):
    """hmf_flow_grad_records_t (include/humaniflow_amd.h): device pointers of the per-(part, row) deltas / activations."""
    _fields_ = [(k, C.c_void_p) for k in ('D1', 'D2', 'D3', 'D4', 'A1', 'A2', 'A3', 'XIN', 'ACTX', 'DCTX')] + [('rows', C.c_int)]


class FlowVariantGradRecords(C.Structure):
    """hmf_flow_generic_grad_records_t: records of the variants' reverse sweeps (per part, transform, row); zero-filled by the caller."""
    _fields_ = [(k, C.c_void_p) for k in ('D', 'A', 'XIN', 'DP', 'AP', 'GLU', 'ACTX', 'DCTX')] + [('rows', C.c_int), ('L', C.c_int), ('LP', C.c_int)]


class Rng(C.Structure):
    """hmf_rng_t: where hmf_flow_forward's in-kernel base noise sits in the device generator's Philox stream."""
    _fields_ = [('seed', C.c_uint64), ('offset', C.c_uint64), ('state_dev', C.c_void_p), ('mirror_dev', C.c_void_p), ('grid_cap_blocks', C.c_int)]


class EvalBatch(C.Structure):
    """hmf_eval_batch_t: one batch of the metric tracker after the point-set kernels (include/humaniflow_amd.h)."""
    _fields_ = ([(k, C.c_int) for k in ('B', 'N', 'num_verts', 'num_joints', 'chunks_v', 'chunks_j')]
                + [(k, C.c_void_p) for k in ('e_verts', 'e_tpose', 'e_joints', 'es_verts', 'es_tpose', 'es_joints', 'div_verts', 'div_joints',
                                             'j2d_pred', 'j2d_samples', 'j2d_target', 'j2d_vis', 'scratch')]
                + [('slot', C.c_int * 23)])


_VP = C.c_void_p
HMF_FLOW_POINT_EST, HMF_FLOW_PACKED_ROWS, HMF_FLOW_SHAPE_MODE, HMF_FLOW_CHAIN32, HMF_FLOW_CHAIN16 = 1, 2, 4, 8, 16          # flags of hmf_flow_forward (include/humaniflow_amd.h)
_SIGS = {
    'hmf_abi_version': (C.c_int, []),
    'hmf_last_error': (C.c_char_p, []),
    'hmf_device_count': (C.c_int, []),
    'hmf_rot6d_to_rotmat': (C.c_int, [_VP, C.c_int, _VP, _VP]),
    'hmf_rot6d_to_rotmat_backward': (C.c_int, [_VP, _VP, C.c_int, _VP, _VP]),
    'hmf_so3_exp': (C.c_int, [_VP, C.c_int, _VP, _VP]),
    'hmf_so3_log': (C.c_int, [_VP, C.c_int, _VP, _VP]),
    'hmf_so3_log_abs_det_jacobian': (C.c_int, [_VP, C.c_int, _VP, _VP]),
    'hmf_batch_rodrigues': (C.c_int, [_VP, C.c_int, _VP, _VP]),
    'hmf_batch_rodrigues_pose': (C.c_int, [_VP, C.c_int, _VP, _VP, _VP]),
    'hmf_smpl_create': (C.c_int, [C.POINTER(_VP), C.POINTER(SmplHost)]),
    'hmf_smpl_destroy': (None, [_VP]),
    'hmf_smpl_lbs_workspace_bytes': (C.c_size_t, [C.c_int]),
    'hmf_smpl_lbs': (C.c_int, [_VP, _VP, C.c_int, _VP, _VP, C.c_int, C.c_int, _VP, _VP, _VP, C.c_size_t, C.c_int, _VP]),
    'hmf_smpl_tpose': (C.c_int, [_VP, _VP, C.c_int, _VP, _VP, _VP]),
    'hmf_vertex_variance': (C.c_int, [_VP, C.c_int, C.c_int, _VP, _VP, _VP]),
    'hmf_flow_create': (C.c_int, [C.POINTER(_VP), C.POINTER(FlowHost)]),
    'hmf_flow_create_updatable': (C.c_int, [C.POINTER(_VP), C.POINTER(FlowHost), _VP, C.c_size_t]),
    'hmf_flow_update_from_device': (C.c_int, [_VP, _VP, C.c_size_t, _VP]),
    'hmf_flow_destroy': (None, [_VP]),
    'hmf_flow_forward_workspace_bytes': (C.c_size_t, [_VP, C.c_int, C.c_int, C.c_int]),
    'hmf_flow_forward': (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, C.POINTER(Rng), C.c_int, C.c_int, C.c_int,
                                   _VP, _VP, _VP, _VP, _VP, C.c_size_t, _VP]),
    'hmf_flow_draw_noise': (C.c_int, [C.POINTER(Rng), C.c_int, C.c_int, C.c_int, _VP, _VP, _VP, C.c_size_t, _VP]),
    'hmf_flow_rng_increment': (C.c_ulonglong, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint]),
    'hmf_smpl_lbs_packed': (C.c_int, [_VP, _VP, C.c_int, C.c_int, _VP, _VP, _VP, _VP, _VP, _VP, C.c_size_t, C.c_int, _VP]),
    'hmf_conv1x1_gemm_avgpool': (C.c_int, [_VP] * 6 + [C.c_int] * 7 + [_VP]),
    'hmf_avgpool_nhwc': (C.c_int, [_VP, C.c_int, C.c_int, C.c_int, _VP, _VP]),
    'hmf_heads_workspace_bytes': (C.c_size_t, [C.c_int] * 3),
    'hmf_heads': (C.c_int, [_VP, C.c_int, C.c_int, _VP, _VP, C.c_int] + [_VP] * 8 + [C.c_int] + [_VP] * 6 + [C.c_size_t, _VP]),
    'hmf_heads_train': (C.c_int, [_VP, C.c_int, C.c_int, _VP, _VP, C.c_int] + [_VP] * 8 + [C.c_int] + [_VP] * 8 + [C.c_size_t, _VP]),
    'hmf_heads_backward': (C.c_int, [_VP, _VP, C.c_int, C.c_int, C.c_int] + [_VP] * 16),
    'hmf_flow_isgc': (C.c_int, [_VP, _VP, _VP, _VP, _VP, C.c_int, _VP, _VP]),
    'hmf_flow_isgc_train': (C.c_int, [_VP] * 7 + [C.c_int, C.c_int, _VP, _VP, _VP]),
    'hmf_flow_isgc_train_backward': (C.c_int, [_VP] * 4 + [C.c_int, C.c_int] + [_VP] * 6),
    'hmf_flow_context': (C.c_int, [_VP, _VP, _VP, C.c_int, _VP, _VP]),
    'hmf_flow_log_prob_SO3': (C.c_int, [_VP, _VP, c_int_p, C.c_int, _VP, C.c_int, _VP, _VP]),
    'hmf_flow_log_prob_so3': (C.c_int, [_VP, _VP, c_int_p, C.c_int, _VP, C.c_int, _VP, _VP]),
    'hmf_flow_log_prob_SO3_backward': (C.c_int, [_VP, _VP, c_int_p, C.c_int, _VP, C.c_int, _VP, _VP, C.POINTER(FlowGradRecords), _VP]),
    'hmf_flow_log_prob_so3_vjp': (C.c_int, [_VP, _VP, c_int_p, C.c_int, _VP, C.c_int, _VP, _VP, _VP, _VP]),
    'hmf_flow_sample_backward': (C.c_int, [_VP, _VP, _VP, _VP, C.c_int, C.c_int, _VP, C.POINTER(FlowGradRecords), _VP]),
    'hmf_flow_generic_log_prob_SO3_backward': (C.c_int, [_VP, _VP, c_int_p, C.c_int, _VP, C.c_int, _VP, _VP, C.POINTER(FlowVariantGradRecords), _VP]),
    'hmf_flow_generic_sample_backward': (C.c_int, [_VP, _VP, _VP, _VP, C.c_int, C.c_int, _VP, C.POINTER(FlowVariantGradRecords), _VP]),
    'hmf_rows_atb_batched': (C.c_int, [_VP] * 4 + [C.c_int] * 6 + [C.c_longlong] * 2 + [_VP]),
    'hmf_smpl_joints_backward': (C.c_int, [_VP, _VP, _VP, _VP, C.c_int, _VP, _VP, _VP, _VP, _VP]),
    'hmf_smpl_joints': (C.c_int, [_VP] * 4 + [C.c_int, _VP, _VP]),
    'hmf_bn_stats': (C.c_int, [_VP, C.c_int, C.c_int, _VP, _VP]),
    'hmf_bn_finalize': (C.c_int, [_VP, C.c_int, C.c_int, _VP, _VP, C.c_float, C.c_float] + [_VP] * 7),
    'hmf_bn_apply': (C.c_int, [_VP] * 5 + [C.c_int] * 3 + [_VP]),
    'hmf_bn_bwd_reduce': (C.c_int, [_VP] * 5 + [C.c_int] * 3 + [_VP, _VP]),
    'hmf_bn_bwd_finalize': (C.c_int, [_VP, C.c_int, C.c_int] + [_VP] * 9),
    'hmf_bn_bwd_apply': (C.c_int, [_VP] * 8 + [C.c_int] * 3 + [_VP]),
    'hmf_maxpool3x3s2_nhwc_backward': (C.c_int, [_VP] * 3 + [C.c_int] * 4 + [_VP]),
    'hmf_bn_relu_maxpool3x3s2_nhwc': (C.c_int, [_VP] * 5 + [C.c_int] * 4 + [_VP]),
    'hmf_bn_relu_maxpool3x3s2_nhwc_bwd_reduce': (C.c_int, [_VP] * 6 + [C.c_int] * 4 + [_VP, _VP]),
    'hmf_bn_relu_maxpool3x3s2_nhwc_bwd_apply': (C.c_int, [_VP] * 8 + [C.c_int] * 4 + [_VP]),
    'hmf_conv_wgrad': (C.c_int, [_VP] * 3 + [C.c_int] * 8 + [_VP]),
    'hmf_conv_wgrad_winograd_workspace_floats': (C.c_size_t, [C.c_int, C.c_int]),
    'hmf_conv_wgrad_winograd_supported': (C.c_int, [C.c_int] * 4),
    'hmf_conv_wgrad_winograd': (C.c_int, [_VP] * 4 + [C.c_int] * 5 + [_VP]),
    'hmf_proxy_rep': (C.c_int, [_VP, _VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_float,
                                _VP, _VP, _VP, _VP, _VP]),
    'hmf_stem_packed_floats': (C.c_longlong, []),
    'hmf_stem_pack_host': (C.c_int, [_VP, _VP]),
    'hmf_stem_conv': (C.c_int, [_VP] * 4 + [C.c_int] * 5 + [_VP]),
    'hmf_stem_conv_pool': (C.c_int, [_VP] * 4 + [C.c_int] * 4 + [_VP]),
    'hmf_train_loss_workspace_bytes': (C.c_longlong, [C.c_int]),
    'hmf_train_loss': (C.c_int, [_VP, C.c_int, _VP, _VP, _VP, C.c_int, _VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_int, C.c_int] + [C.c_float] * 5 + [C.c_int] + [_VP] * 8 + [C.c_longlong, _VP]),
    'hmf_stem_wino_packed_floats': (C.c_longlong, []),
    'hmf_stem_wino_pack_host': (C.c_int, [_VP, _VP]),
    'hmf_stem_wino_pack_device': (C.c_int, [_VP, _VP, _VP]),
    'hmf_stem_wino': (C.c_int, [_VP] * 4 + [C.c_int] * 5 + [_VP]),
    'hmf_stem_wino_pool_workspace_bytes': (C.c_longlong, [C.c_int] * 3),
    'hmf_stem_wino_pool': (C.c_int, [_VP] * 4 + [C.c_int] * 4 + [_VP, C.c_longlong, _VP]),
    'hmf_maxpool3x3s2_nhwc': (C.c_int, [_VP, _VP] + [C.c_int] * 4 + [_VP]),
    'hmf_conv_patch_packed_floats': (C.c_longlong, [C.c_int] * 3),
    'hmf_conv_patch_pack_host': (C.c_int, [_VP, _VP] + [C.c_int] * 4),
    'hmf_conv_patch': (C.c_int, [_VP] * 5 + [C.c_int] * 9 + [_VP]),
    'hmf_conv_patch_workspace_bytes': (C.c_longlong, [C.c_int] * 8),
    'hmf_conv_patch_ws': (C.c_int, [_VP] * 5 + [C.c_int] * 9 + [_VP, C.c_longlong, _VP]),
    'hmf_pointset_errors': (C.c_int, [_VP, _VP] + [C.c_int] * 4 + [_VP, _VP]),
    'hmf_sample_diversity': (C.c_int, [_VP] + [C.c_int] * 3 + [_VP, _VP]),
    'hmf_sample_diversity_chunks': (C.c_int, [C.c_int]),
    'hmf_eval_joints': (C.c_int, [_VP, _VP, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_float, _VP, _VP, _VP, _VP]),
    'hmf_eval_tracker_scratch_bytes': (C.c_longlong, [C.c_int]),
    'hmf_eval_tracker_update': (C.c_int, [C.POINTER(EvalBatch), _VP, _VP, _VP]),
    'hmf_smpl_tpose_errors_workspace_bytes': (C.c_longlong, [C.c_int]),
    'hmf_smpl_tpose_errors': (C.c_int, [_VP, _VP, _VP, C.c_int, C.c_int, _VP, _VP, C.c_longlong, _VP]),
    'hmf_smpl_tpose_errors_packed': (C.c_int, [_VP, _VP, _VP, C.c_int, C.c_int, _VP, _VP, C.c_longlong, _VP]),
    'hmf_conv_pair_pack_host': (C.c_int, [_VP] * 4 + [C.c_int] * 3),
    'hmf_conv_pair': (C.c_int, [_VP] * 9 + [C.c_int] * 11 + [_VP]),
    'hmf_conv_winograd_packed_floats': (C.c_longlong, [C.c_int] * 2),
    'hmf_conv_winograd_pack_host': (C.c_int, [_VP, _VP] + [C.c_int] * 2),
    'hmf_conv_winograd': (C.c_int, [_VP] * 5 + [C.c_int] * 6 + [_VP]),
    'hmf_conv_winograd_workspace_bytes': (C.c_longlong, [C.c_int] * 5),
    'hmf_conv_winograd_ws': (C.c_int, [_VP] * 5 + [C.c_int] * 6 + [_VP, C.c_longlong, _VP]),
    'hmf_conv_winograd_pack_device': (C.c_int, [_VP] * 2 + [C.c_int] * 3 + [_VP]),
    'hmf_conv_winograd_pack_device_batch': (C.c_int, [_VP] * 5 + [C.c_int, _VP]),
    'hmf_conv1x1_dual_nhwc': (C.c_int, [_VP] * 5 + [C.c_int] * 10 + [_VP]),
    'hmf_conv2d_nhwc': (C.c_int, [_VP] * 5 + [C.c_int] * 9 + [_VP]),
    'hmf_conv1x1_gemm': (C.c_int, [_VP] * 6 + [C.c_int] * 11 + [_VP]),
    'hmf_i64_increment_batch': (C.c_int, [_VP, C.c_int, _VP]),
    'hmf_conv_dgrad_s2': (C.c_int, [_VP] * 6 + [C.c_int] * 6 + [_VP]),
    'hmf_conv1x1_split_packed_bytes': (C.c_longlong, [C.c_int] * 3),
    'hmf_conv1x1_split_pack': (C.c_int, [_VP, _VP, C.c_int, C.c_int, C.c_int, _VP]),
    'hmf_conv1x1_gemm_split': (C.c_int, [_VP] * 6 + [C.c_int] * 12 + [_VP]),
    'hmf_profile_enable': (C.c_int, [C.c_int]),
    'hmf_profile_collect': (C.c_int, [c_double_p, c_int_p, c_int_p, C.c_int]),
    'hmf_profile_clock_probe': (C.c_int, [C.c_int, _VP, _VP]),
    'hmf_smpl_packed_blend_floats': (C.c_size_t, [C.c_int]),
    'hmf_smpl_pack_blend_host': (C.c_int, [C.POINTER(SmplHost), c_float_p]),
    'hmf_smpl_packed_blend_index': (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    'hmf_flow_packed_floats': (C.c_size_t, [C.POINTER(FlowHost)]),
    'hmf_flow_pack_host': (C.c_int, [C.POINTER(FlowHost), c_float_p]),
}
EXPORTED_SYMBOLS = sorted(_SIGS)
# include/humaniflow_amd_experiments.h: only in tools/libhumaniflow_amd_experiments.so (the same sources built with -DHMF_EXPERIMENTS)
_EXPERIMENT_SIGS = {
    'hmf_conv_winograd_variant': (C.c_int, [_VP] * 5 + [C.c_int] * 7 + [_VP]),      # tools/winograd_ablate.py
    'hmf_debug_set_wino_nw_mode': (C.c_int, [C.c_int]),                             # tools/wino_nw_probe.py
    'hmf_conv_winograd_nw': (C.c_int, [_VP] * 5 + [C.c_int] * 6 + [_VP]),
    'hmf_debug_set_timeline': (C.c_int, [_VP]),                                     # tools/flow_timeline.py
    'hmf_debug_set_lbs_cycles': (C.c_int, [_VP]),
    'hmf_debug_lbs16_phases': (C.c_int, [_VP]),                                   # tools/lbs_tune.py
    'hmf_debug_gemm_stamps': (C.c_int, [_VP, C.c_int]),
    'hmf_debug_set_wgw_target_wgs': (C.c_int, [C.c_int]),
    'hmf_debug_set_wgw_single_stage': (C.c_int, [C.c_int]),                            # tools/wgw_probe.py
    'hmf_debug_stem_stamps': (C.c_int, [_VP, C.c_int]),
    'hmf_debug_stem_wino_stamps': (C.c_int, [_VP, C.c_int]),                             # tools/stem_probe.py
    'hmf_debug_set_gemm_whatif': (C.c_int, [C.c_int]),                              # tools/whatif_1x1.py
    'hmf_debug_set_pair_whatif': (C.c_int, [C.c_int]),
    'hmf_debug_set_flow_whatif': (C.c_int, [C.c_int]),
    'hmf_debug_wino_stamps': (C.c_int, [_VP, C.c_int]),                             # tools/wino_probe.py                             # tools/gemm1x1_probe.py
}

_lib = None


def _bind(path, sigs, hint):
    try:
        lib = C.CDLL(path)
    except OSError as e:
        raise RuntimeError('humaniflow_amd: cannot load %s (%s). Build it with `%s`; there is no CPU fallback.' % (path, e, hint))
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.hmf_abi_version() != 4:
        raise RuntimeError('humaniflow_amd: ABI version mismatch')
    return lib


def load():
    """Load (building first if needed) the shared library; raises RuntimeError when impossible."""
    global _lib
    if _lib is not None:
        return _lib
    from .build import build_library
    build_library()          # no-op unless the library is missing or older than its sources (content hash); serialised across ranks
    _lib = _bind(LIB_PATH, _SIGS, 'python -m humaniflow_amd.build')
    return _lib


def use_experiments_library():
    """tools/ only: make load() return tools/libhumaniflow_amd_experiments.so -- the same sources compiled with -DHMF_EXPERIMENTS, i.e. the product
    entry points plus the timing variants of include/humaniflow_amd_experiments.h.  Call before anything else loads the library."""
    global _lib
    from .build import build_library
    path = build_library(experiments=True)
    sigs = dict(_SIGS)
    sigs.update(_EXPERIMENT_SIGS)
    _lib = _bind(path, sigs, 'python -m humaniflow_amd.build --experiments')
    return _lib


def check(rc, what=''):
    if rc != 0:
        msg = load().hmf_last_error()
        raise RuntimeError('humaniflow_amd %s failed (code %d): %s' % (what, rc, (msg or b'').decode(errors='replace')))


def fptr(a):
    """float32 C-contiguous numpy array -> POINTER(c_float) (array must stay alive)."""
    assert a.dtype == np.float32 and a.flags['C_CONTIGUOUS']
    return a.ctypes.data_as(c_float_p)


def iptr(a):
    assert a.dtype == np.int32 and a.flags['C_CONTIGUOUS']
    return a.ctypes.data_as(c_int_p)


def require_cuda(t, name):
    if not t.is_cuda:
        raise RuntimeError('humaniflow_amd: `%s` must live on a HIP device (got %s); the hot path has no CPU fallback '
                           '-- use the oracle/ package for CPU checks.' % (name, t.device))


def dptr(t):
    """device pointer of a contiguous torch tensor (or None)."""
    if t is None:
        return None
    assert t.is_contiguous()
    return t.data_ptr()


def current_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream
