"""HumaniflowModel with the reference's constructor, forward signature, return dict and checkpoint layout,
running its flow on the HIP kernels of libhumaniflow_amd.

Mirror of models/humaniflow_model.py:33-340.  The ResNet encoder runs on the library's hand-written convolution
kernels (encoder.py); what stays PyTorch are the four small head GEMMs fc1/fc_shape/fc_glob/fc_cam (:232-252).  Everything from
`rot6d_to_rotmat` on -- shape sampling, input/shape/glob/cam features (:116-150), the 23 per-part context
layers (:152-186), the ancestor-conditioned flows in kinematic order (:286-311) and the teacher-forced
log-likelihood objects (:314-320) -- is one C-ABI call sequence (hmf_flow_forward / hmf_flow_context /
hmf_flow_log_prob_*), with no Python loop over body parts.

state_dict keys: init_glob, init_cam, image_encoder.*, fc1, fc_shape, fc_glob, fc_cam,
fc_input_shape_glob_cam_feats, fc_flow_context.{0..22}, pose_so3flow_transform_modules.{0..45}.nn.layers.{0..3}.

Flow configurations (create_conditional_norm_flow): the default (spline_coupling + permute, hidden [64,32,32], 2 transforms) runs on the
MFMA kernels; TRANSFORM_TYPE 'additive_coupling', PERMUTE_TYPE 'linear_plu' / 'conditional_linear_plu', other NUM_TRANSFORMS / hidden
widths run on the generic kernels of csrc/flow_generic.hip (inference only).  With a PLU permutation the transform-module list interleaves
[permutation module, coupling module] per transform, exactly as the reference builds it (pyro_conditional_norm_flow.py:115), so its
checkpoint keys (`...LU`, `...permutation`, `...cond_plu_nn.layers.*`) line up.

Training (BASELINE configs[4]): forward under autograd returns differentiable samples / log-likelihood objects; the backward
passes of the flow live in csrc/flow_grad.hip and are wired up in autograd.py.
"""
import ctypes as C
import os
from collections import defaultdict

import numpy as np
import torch
from torch import nn
from torch.distributions import Normal

from . import _lib
from .encoder import resnet18, resnet50
from .rigid_transform_utils import rot6d_to_rotmat, rotmat_to_rot6d


def immediate_parent_to_all_ancestors(immediate_parents):
    """models/humaniflow_model.py:16-30: dict[joint] = [parent, grand-parent, ...], root excluded,
    body part j = SMPL joint j+1."""
    ancestors_dict = defaultdict(list)
    for i in range(1, len(immediate_parents)):
        joint = i - 1
        parent = immediate_parents[i] - 1
        if parent >= 0:
            ancestors_dict[joint] += [parent] + ancestors_dict[parent]
        else:
            ancestors_dict[joint] += []
    return ancestors_dict


class ConditionalDenseNN(nn.Module):
    """Parameter container with pyro.nn.ConditionalDenseNN's layout: `layers` = [Linear(in+ctx, h0), ..., Linear(h_last,
    sum(param_dims))]; the forward pass runs inside the flow kernels (input order [context, x], ReLU)."""

    def __init__(self, input_dim, context_dim, hidden_dims, param_dims):
        super().__init__()
        self.input_dim, self.context_dim, self.hidden_dims, self.param_dims = input_dim, context_dim, hidden_dims, param_dims
        dims = [input_dim + context_dim] + list(hidden_dims) + [sum(param_dims)]
        self.layers = nn.ModuleList([nn.Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 1)])


class ConditionalSplineCoupling(nn.Module):
    """models/norm_flows/transforms/conditional_spline_coupling_transform.py:10-78 (split_dim = 1, identity=True):
    holds the hyper-network as `.nn`; evaluation happens in hmf_flow_forward / hmf_flow_log_prob_*."""

    def __init__(self, input_dim, context_dim, hidden_dims, count_bins, bound):
        super().__init__()
        split_dim = input_dim // 2
        up = input_dim - split_dim
        self.input_dim, self.split_dim, self.count_bins, self.bound = input_dim, split_dim, count_bins, bound
        self.nn = ConditionalDenseNN(split_dim, context_dim, hidden_dims,
                                     param_dims=[up * count_bins, up * count_bins, up * (count_bins - 1), up * count_bins])


class ConditionalAdditiveCoupling(nn.Module):
    """models/norm_flows/transforms/additive_coupling_transform.py:69-85,119-137 (split_dim = 1): y1 = x1, y2 = x2 + nn([context, x1]),
    log-det 0.  `.nn` = ConditionalDenseNN(1, context_dim, hidden_dims, [2]).  (In the reference this option fails with a TypeError at the
    first `.condition()` because `bound` / `count_bins` are forwarded into AdditiveCoupling.__init__; the transform's own arithmetic is built.)"""

    def __init__(self, input_dim, context_dim, hidden_dims):
        super().__init__()
        self.split_dim = input_dim // 2
        self.nn = ConditionalDenseNN(self.split_dim, context_dim, hidden_dims, param_dims=[input_dim - self.split_dim])


class DenseNN(nn.Module):
    """pyro.nn.DenseNN parameter container: `layers` = [Linear(input_dim, h0), ..., Linear(h_last, sum(param_dims))]."""

    def __init__(self, input_dim, hidden_dims, param_dims):
        super().__init__()
        dims = [input_dim] + list(hidden_dims) + [sum(param_dims)]
        self.layers = nn.ModuleList([nn.Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 1)])


def _permutation_matrix(permutation):
    return torch.eye(len(permutation), len(permutation))[torch.as_tensor(permutation, dtype=torch.int64)]


class LinearPLUTransform(nn.Module):
    """models/norm_flows/transforms/linearPLU_transform.py:85-120: y = P L U x with a dense `LU` parameter (L strictly below the diagonal,
    unit diagonal; U on and above), initialised from the LU factors of a random orthogonal matrix; buffer `permutation` = eye[perm]."""

    def __init__(self, input_dim, permutation):
        super().__init__()
        W, _ = torch.linalg.qr(torch.randn(input_dim, input_dim))
        LU, _ = torch.linalg.lu_factor(W)
        self.register_buffer('permutation', _permutation_matrix(permutation))
        self.LU = nn.Parameter(LU)


class ConditionalLinearPLUTransform(nn.Module):
    """linearPLU_transform.py:123-150: LU = cond_plu_nn(context) (DenseNN -> 9), diagonal through softplus(beta=0.75); buffer `permutation`."""

    def __init__(self, input_dim, context_dim, hidden_dims, permutation):
        super().__init__()
        self.input_dim = input_dim
        self.cond_plu_nn = DenseNN(context_dim, hidden_dims, param_dims=[input_dim * input_dim])
        self.register_buffer('permutation', _permutation_matrix(permutation))


class _ConditionedFlowDist:
    """What `.condition(ctx)` returns in the reference: an object exposing log_prob().  group=True mirrors
    LocalDiffeoTransformedDistribution (R (B,3,3) fp64 -> (B,) fp32, local_diffeo_transformed_distribution.py:84-142),
    group=False the algebra density (v (B,3) fp32).  log_prob on the group is differentiable w.r.t. the context and the
    hyper-network weights (training, losses/humaniflow_loss.py:30-31).  `parent` = (ctx_all (B,23,64), part): lets a loss that
    receives the whole list evaluate all 23 densities in one launch (humaniflow_amd.losses)."""

    def __init__(self, model, part, ctx, group, check_finite=False, parent=None):
        self.model, self.part, self.ctx, self.group, self.check_finite, self.parent = model, part, ctx, group, check_finite, parent

    def log_prob(self, value):
        m = self.model
        _lib.require_cuda(value, 'value')
        B = self.ctx.shape[0]
        if self.check_finite:                      # the reference's NaN/Inf asserts (:94-95), opt-in: they force a sync
            assert torch.isnan(value).sum() == 0 and not torch.isinf(value).any()
        if self.group:
            out = m.log_prob_parts(self.ctx[:, None, :], value.reshape(B, 1, 3, 3), self.part, 1)[:, 0]
        else:
            if torch.is_grad_enabled() and (value.requires_grad or self.ctx.requires_grad):
                # round 6: differentiable w.r.t. the algebra vector and the context (hmf_flow_log_prob_so3_vjp); flow weights are constants on this path
                from .autograd import FlowLogProbAlgebra, _is_variant
                if _is_variant(m):
                    raise NotImplementedError('humaniflow_amd: gradients of the algebra density are built for the default flow configuration only')
                return FlowLogProbAlgebra.apply(m, self.part, 1, self.ctx[:, None, :], value.reshape(B, 1, 3))[:, 0]
            out = m._log_prob_kernel(self.ctx[:, None, :], value.reshape(B, 1, 3), self.part, 1, group=False)[:, 0]
        if self.check_finite:
            assert torch.isnan(out).sum() == 0
        return out


class _FlowDistHandle:
    """Stand-in for the reference's per-part `pose_SO3flow_dists[i]` / `pose_so3flow_dists[i]` objects: callers only
    use `.condition(ctx)` (inside forward) and `.clear_cache()` (train/train_humaniflow.py:353-354)."""

    def __init__(self, model, part, group):
        self.model, self.part, self.group = model, part, group

    def condition(self, context, parent=None):
        return _ConditionedFlowDist(self.model, self.part, context.to(torch.float32), self.group, parent=parent)

    def clear_cache(self):
        pass


class HumaniflowModel(nn.Module):
    def __init__(self, device, model_cfg, smpl_parents):
        super().__init__()
        nf = model_cfg.NORM_FLOW
        # create_conditional_norm_flow (models/norm_flows/pyro_conditional_norm_flow.py:21-117): the reference's own transforms are built;
        # pyro's stock affine_coupling / affine_masked / spline_masked options are not
        if nf.TRANSFORM_TYPE not in ('spline_coupling', 'additive_coupling'):
            raise NotImplementedError("humaniflow_amd builds TRANSFORM_TYPE 'spline_coupling' and 'additive_coupling'; got %r" % (nf.TRANSFORM_TYPE,))
        if nf.PERMUTE_TYPE not in ('permute', 'linear_plu', 'conditional_linear_plu'):
            raise NotImplementedError("humaniflow_amd builds PERMUTE_TYPE 'permute', 'linear_plu' and 'conditional_linear_plu'; got %r" % (nf.PERMUTE_TYPE,))
        if nf.PERMUTE_TYPE == 'conditional_linear_plu':
            assert nf.PERMUTE_NN_HIDDEN_DIMS is not None                                  # pyro_conditional_norm_flow.py:41-42
        self.default_flow = (nf.TRANSFORM_TYPE == 'spline_coupling' and nf.PERMUTE_TYPE == 'permute' and nf.NUM_TRANSFORMS == 2
                             and list(nf.TRANSFORM_NN_HIDDEN_DIMS) == [64, 32, 32])
        self.force_generic_flow = False           # tests: run the default configuration through the variant kernels
        self.parents = list(smpl_parents)
        self.ancestors_dict = immediate_parent_to_all_ancestors(self.parents)
        self.num_bodyparts = len(self.ancestors_dict)
        self.model_cfg = model_cfg
        self.num_shape_params = model_cfg.NUM_SMPL_BETAS
        self.num_glob_params = 6
        self.register_buffer('init_glob', rotmat_to_rot6d(torch.eye(3)[None, :].float()))
        self.num_cam_params = 3
        self.register_buffer('init_cam', torch.tensor([0.9, 0.0, 0.0]).float())

        if model_cfg.NUM_RESNET_LAYERS == 18:
            self.image_encoder = resnet18(in_channels=model_cfg.NUM_IN_CHANNELS)
            input_feats_dim, fc1_dim = 512, 512
        elif model_cfg.NUM_RESNET_LAYERS == 50:
            self.image_encoder = resnet50(in_channels=model_cfg.NUM_IN_CHANNELS)
            input_feats_dim, fc1_dim = 2048, 1024
        else:
            raise ValueError('NUM_RESNET_LAYERS must be 18 or 50')
        self.input_feats_dim = input_feats_dim

        self.activation = nn.ELU()
        self.fc1 = nn.Linear(input_feats_dim, fc1_dim)
        self.fc_shape = nn.Linear(fc1_dim, self.num_shape_params * 2)
        self.fc_glob = nn.Linear(fc1_dim, self.num_glob_params)
        self.fc_cam = nn.Linear(fc1_dim, self.num_cam_params)
        self.fc_input_shape_glob_cam_feats = nn.Linear(input_feats_dim + self.num_shape_params + 9 + self.num_cam_params,
                                                       model_cfg.INPUT_SHAPE_GLOB_CAM_FEATS_DIM)
        self.fc_flow_context = nn.ModuleList()
        self.pose_so3flow_transform_modules = nn.ModuleList()
        self.pose_so3flow_dists = []
        self.pose_SO3flow_dists = []
        for bodypart in range(self.num_bodyparts):
            num_ancestors = len(self.ancestors_dict[bodypart])
            self.fc_flow_context.append(nn.Linear(model_cfg.INPUT_SHAPE_GLOB_CAM_FEATS_DIM + num_ancestors * 9, nf.CONTEXT_DIM))
            for t in range(nf.NUM_TRANSFORMS):        # default flow: module index = NUM_TRANSFORMS * part + t
                perm = [(m + t % 3) % 3 for m in range(3)]                                    # :46-47 cyclic shifts
                if nf.PERMUTE_TYPE == 'linear_plu':                                           # :65-67 (an nn.Module: listed before its coupling)
                    self.pose_so3flow_transform_modules.append(LinearPLUTransform(3, perm))
                elif nf.PERMUTE_TYPE == 'conditional_linear_plu':                             # :68-72
                    self.pose_so3flow_transform_modules.append(
                        ConditionalLinearPLUTransform(3, nf.CONTEXT_DIM, list(nf.PERMUTE_NN_HIDDEN_DIMS), perm))
                if nf.TRANSFORM_TYPE == 'spline_coupling':
                    self.pose_so3flow_transform_modules.append(
                        ConditionalSplineCoupling(3, nf.CONTEXT_DIM, list(nf.TRANSFORM_NN_HIDDEN_DIMS), nf.NUM_SPLINE_SEGMENTS,
                                                  nf.COMPACT_SUPPORT_RADIUS))
                else:
                    self.pose_so3flow_transform_modules.append(ConditionalAdditiveCoupling(3, nf.CONTEXT_DIM, list(nf.TRANSFORM_NN_HIDDEN_DIMS)))
            self.pose_so3flow_dists.append(_FlowDistHandle(self, bodypart, group=False))
            self.pose_SO3flow_dists.append(_FlowDistHandle(self, bodypart, group=True))
        self.__dict__.update(_handles={}, _ws={}, _flow_tensors=None, _flow_key=None)
        self.to(device)

    # ---- packed-weight handle: a cache of the live parameters ------------------------------------------------------
    # Dropped by train()/eval(), .to() and load_state_dict; re-validated on every use against the parameters' version counters, so
    # in-place updates (optimizer.step(), p.data.copy_(), load_state_dict on a sub-module, in-place EMA swaps) re-pack.  Re-binding
    # `p.data = other` does not bump a version counter: call invalidate_flow_weights() after that.
    # A captured HIP graph of a step (bench.py --train, torch.cuda.graph around forward/backward) bakes in the device pointers of the handle's
    # packed-weight arena: the handle is therefore KEPT across train()/eval()/load_state_dict and re-packed IN PLACE on its next use
    # (hmf_flow_update_from_device keeps every pointer); only a device / dtype move (_apply) or destruction frees it.  Graph replays update the
    # parameters without bumping their version counters: call mark_params_changed() after replaying a captured training step and before an
    # eager forward, or the eager forward runs on the weights packed before the replays.
    def mark_params_changed(self):
        """The parameters changed behind autograd's back (HIP-graph replay of an optimiser step, `p.data = ...`): re-pack on the next use."""
        self.__dict__['_flow_key'] = None
        self.__dict__['_flow_tensors'] = None
        enc = getattr(self, 'image_encoder', None)
        if enc is not None and hasattr(enc, 'invalidate_plan'):
            enc.invalidate_plan()

    def invalidate_flow_weights(self, destroy=True):
        if destroy and self.__dict__.get('_handles'):
            lib = _lib.load()
            for h in self._handles.values():
                lib.hmf_flow_destroy(h)
            self.__dict__['_handles'] = {}
        self.__dict__['_flow_tensors'] = None
        self.__dict__['_flow_key'] = None

    def _flow_params(self):
        ts = self.__dict__.get('_flow_tensors')
        if ts is None:
            mods = [self.fc_input_shape_glob_cam_feats] + list(self.fc_flow_context) + list(self.pose_so3flow_transform_modules)
            ts = self.__dict__['_flow_tensors'] = [p for m in mods for p in m.parameters()]
        return ts

    def __getstate__(self):                       # handles and workspaces are device-side caches: never pickled / deep-copied
        st = self.__dict__.copy()
        st['_handles'], st['_ws'], st['_flow_tensors'], st['_flow_key'] = {}, {}, None, None
        st.pop('_rng_dev', None)
        return st

    def train(self, mode=True):
        self.invalidate_flow_weights(destroy=False)
        return super().train(mode)

    def _apply(self, fn, *a, **k):
        self.invalidate_flow_weights()
        self.__dict__.pop('_rng_dev', None)
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        self.invalidate_flow_weights(destroy=False)
        return super()._load_from_state_dict(*a, **k)

    def __del__(self):
        try:
            self.invalidate_flow_weights()
        except Exception:
            pass

    def _flow_flat(self):
        """All flow parameters concatenated on their device, in _flow_params() order (the layout hmf_flow_create_updatable indexes)."""
        return torch.cat([t.detach().reshape(-1).float() for t in self._flow_params()])

    def flow_host_struct(self):
        """(hmf_flow_host_t, keep-alive list) from the live parameters."""
        keep = []
        # one device -> host copy for all ~420 tensors (a training step re-packs after every optimiser step)
        ts = self._flow_params()
        flat = np.ascontiguousarray(self._flow_flat().cpu().numpy())
        keep.append(flat)
        offs, o = {}, 0
        for t in ts:
            offs[id(t)] = (o, t.numel())
            o += t.numel()

        def arr(t):
            a, n = offs[id(t)]
            return flat[a:a + n]
        nf = self.model_cfg.NORM_FLOW
        P, T = self.num_bodyparts, nf.NUM_TRANSFORMS
        hs = _lib.FlowHost()
        hs.feat_dim, hs.num_betas = self.input_feats_dim, self.num_shape_params
        hs.isgc_dim, hs.ctx_dim, hs.num_transforms = self.model_cfg.INPUT_SHAPE_GLOB_CAM_FEATS_DIM, nf.CONTEXT_DIM, T
        hd = list(nf.TRANSFORM_NN_HIDDEN_DIMS)
        if not 1 <= len(hd) <= 4 or max(hd) > 64:
            raise NotImplementedError('TRANSFORM_NN_HIDDEN_DIMS: 1..4 layers of <= 64 channels are built, got %s' % (hd,))
        hs.hidden = (C.c_int * 3)(*(hd + [0, 0, 0])[:3])
        hs.num_hidden = len(hd)
        hs.hidden_dims = (C.c_int * 4)(*(hd + [0, 0, 0, 0])[:4])
        hs.transform_type = {'spline_coupling': 0, 'additive_coupling': 1}[nf.TRANSFORM_TYPE]
        hs.permute_type = {'permute': 0, 'linear_plu': 1, 'conditional_linear_plu': 2}[nf.PERMUTE_TYPE]
        hs.force_generic = int(self.force_generic_flow)
        hs.count_bins, hs.radius, hs.base_std = nf.NUM_SPLINE_SEGMENTS, nf.COMPACT_SUPPORT_RADIUS, nf.BASE_DIST_STD
        par = np.ascontiguousarray(np.array(self.parents, np.int32))
        keep.append(par)
        hs.parents_host = _lib.iptr(par)
        hs.w_isgc_host = _lib.fptr(arr(self.fc_input_shape_glob_cam_feats.weight))
        hs.b_isgc_host = _lib.fptr(arr(self.fc_input_shape_glob_cam_feats.bias))
        wc = (_lib.c_float_p * P)(*[_lib.fptr(arr(m.weight)) for m in self.fc_flow_context])
        bc = (_lib.c_float_p * P)(*[_lib.fptr(arr(m.bias)) for m in self.fc_flow_context])
        wn, bn, wp, bp, lu = [], [], [], [], []
        for mod in self.pose_so3flow_transform_modules:
            if isinstance(mod, LinearPLUTransform):
                lu.append(_lib.fptr(arr(mod.LU)))
            elif isinstance(mod, ConditionalLinearPLUTransform):
                for lin in mod.cond_plu_nn.layers:
                    wp.append(_lib.fptr(arr(lin.weight)))
                    bp.append(_lib.fptr(arr(lin.bias)))
            else:
                for lin in mod.nn.layers:
                    wn.append(_lib.fptr(arr(lin.weight)))
                    bn.append(_lib.fptr(arr(lin.bias)))
        wn_a = (_lib.c_float_p * len(wn))(*wn)
        bn_a = (_lib.c_float_p * len(bn))(*bn)
        keep += [wc, bc, wn_a, bn_a]
        hs.w_ctx_host, hs.b_ctx_host, hs.w_nn_host, hs.b_nn_host = wc, bc, wn_a, bn_a
        if wp:
            ph = list(nf.PERMUTE_NN_HIDDEN_DIMS)
            if not 1 <= len(ph) <= 4 or max(ph) > 64:
                raise NotImplementedError('PERMUTE_NN_HIDDEN_DIMS: 1..4 layers of <= 64 channels are built, got %s' % (ph,))
            hs.plu_num_hidden = len(ph)
            hs.plu_hidden_dims = (C.c_int * 4)(*(ph + [0, 0, 0, 0])[:4])
            wp_a, bp_a = (_lib.c_float_p * len(wp))(*wp), (_lib.c_float_p * len(bp))(*bp)
            keep += [wp_a, bp_a]
            hs.w_plu_host, hs.b_plu_host = wp_a, bp_a
        if lu:
            lu_a = (_lib.c_float_p * len(lu))(*lu)
            keep.append(lu_a)
            hs.lu_host = lu_a
        return hs, keep

    def _flow_handle(self, device):
        """Packed-weight handle of `device`, current with the live parameters.  A change of the parameters' version counters (optimiser
        step, in-place copy) re-packs ON THE DEVICE: one concatenation + one gather kernel through the slot->parameter index map the
        handle recorded at creation (hmf_flow_update_from_device) -- no host round trip, no reallocation."""
        lib = _lib.load()
        # staleness contract: the arena is re-packed when a flow parameter's (version counter, storage address) changes -- optimiser steps, in-place
        # copies, load_state_dict and `p.data = other` are all seen.  Substituting parameter OBJECTS (torch.func.functional_call) is not: the kernels
        # read this module's own parameters (autograd.IsgcTrain checks that the tensor autograd hands it is the one the arena was packed from).
        vkey = tuple([(t._version, t.data_ptr()) for t in self._flow_params()])
        key = (device.type, device.index)
        h = self._handles.get(key)
        if h is not None and self.__dict__.get('_flow_key') != vkey:
            flat = self._flow_flat()
            if flat.device == device:
                with torch.cuda.device(device):
                    _lib.check(lib.hmf_flow_update_from_device(h, _lib.dptr(flat), flat.numel(), _lib.current_stream()), 'hmf_flow_update_from_device')
                for k2, h2 in list(self._handles.items()):           # handles of other devices (rare): rebuilt on their next use
                    if k2 != key:
                        lib.hmf_flow_destroy(h2)
                        del self._handles[k2]
            else:
                self.invalidate_flow_weights()
                h = None
            self.__dict__['_flow_key'] = vkey
        if h is None:
            hs, keep = self.flow_host_struct()
            flat = keep[0]
            h = C.c_void_p()
            with torch.cuda.device(device):
                _lib.check(lib.hmf_flow_create_updatable(C.byref(h), C.byref(hs), flat.ctypes.data, flat.size), 'hmf_flow_create_updatable')
            self._handles[key] = h
            self.__dict__['_flow_key'] = vkey
        return h

    def _workspace(self, device, nbytes, tag='flow'):
        # one scratch buffer per (device, stream, use): steps issued on different streams may be in flight together
        key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream, tag)
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
            self._ws[key] = ws
        return ws

    # ---- heads + base noise ------------------------------------------------------------------------------------------
    flow_chain32 = None      # None: by size (k_flow_chain16 up to 1,024 rows, the 32-row k_flow_chain beyond); True / False force one: cross-checks, tools
    isgc_kernel = os.environ.get('HMF_ISGC_TRAIN', '1') != '0'      # training: both uses of the isgc layer on autograd.IsgcTrain (False: torch operators)
    heads_kernel = True      # inference: fc1 + ELU + fc_cam / fc_glob / fc_shape + init adds + 6D->R + exp(log_std) on hmf_heads; False: torch ops

    def _heads_kernel_supports(self, input_feats):
        """Shapes / alignment hmf_heads is built for (csrc/heads_kernels.hip: F and fc1.out_features multiples of 32, fc1.out_features <= 1024,
        <= 16 shape parameters, 16-byte aligned float32 operands).  Anything else the reference accepts takes the torch operators of the
        differentiable path (still on the device), announced once."""
        F, N1 = input_feats.shape[1], self.fc1.out_features
        ok = F % 32 == 0 and N1 % 32 == 0 and N1 <= 1024 and 1 <= self.num_shape_params <= 16 and input_feats.dim() == 2
        if ok:
            ts = [input_feats, self.fc1.weight, self.fc_cam.weight, self.fc_glob.weight, self.fc_shape.weight]
            ok = all((not t.is_contiguous()) or t.dtype != torch.float32 or t.data_ptr() % 16 == 0 for t in ts)   # copies made by the wrapper are aligned
        if not ok and not self.__dict__.get('_heads_fallback_warned'):
            self.__dict__['_heads_fallback_warned'] = True
            import warnings
            warnings.warn('humaniflow_amd: hmf_heads is not built for F=%d, fc1.out_features=%d, num_shape_params=%d or unaligned operands; '
                          'the heads run as torch operators on the device' % (F, N1, self.num_shape_params))
        return ok

    def _heads_kernel(self, input_feats):
        """models/humaniflow_model.py:232-253 on hmf_heads (csrc/heads_kernels.hip): reads the live parameters, no packed copy."""
        lib = _lib.load()
        f = input_feats.detach().to(torch.float32).contiguous()
        B, F = f.shape
        dev = f.device
        nb = self.num_shape_params
        N1 = self.fc1.out_features
        new = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)
        cam, glob_R, mode, lstd, std = new(B, 3), new(B, 3, 3), new(B, nb), new(B, nb), new(B, nb)
        p32 = lambda t: t.detach().to(torch.float32).contiguous()
        with torch.cuda.device(dev):
            ws = self._workspace(dev, int(lib.hmf_heads_workspace_bytes(B, F, N1)), tag='heads')
            _lib.check(lib.hmf_heads(_lib.dptr(f), B, F, _lib.dptr(p32(self.fc1.weight)), _lib.dptr(p32(self.fc1.bias)), N1,
                                     _lib.dptr(p32(self.fc_cam.weight)), _lib.dptr(p32(self.fc_cam.bias)), _lib.dptr(p32(self.init_cam)),
                                     _lib.dptr(p32(self.fc_glob.weight)), _lib.dptr(p32(self.fc_glob.bias)), _lib.dptr(p32(self.init_glob)),
                                     _lib.dptr(p32(self.fc_shape.weight)), _lib.dptr(p32(self.fc_shape.bias)), nb,
                                     _lib.dptr(cam), _lib.dptr(glob_R), _lib.dptr(mode), _lib.dptr(lstd), _lib.dptr(std),
                                     _lib.dptr(ws), ws.numel(), _lib.current_stream()), 'hmf_heads')
        return cam, glob_R, mode, lstd, std

    def _rng_for_pass(self, device, B, N, draw_shape):
        """hmf_rng_t of one sampling pass: where the pass' 1 + 23 normal_() calls sit in the device generator's Philox stream.
        Eager: (seed, offset) read from torch.cuda.default_generators[device] and the generator advanced by what the reference's calls would
        consume -- torch.randn after the forward continues exactly where it would after the reference's forward.  Under HIP-graph capture the
        host cannot touch the generator: the pass reads a device-resident (seed, offset) that every replay advances atomically.  EVERY eager
        pass leaves the generator's state after itself in that device-resident state (hmf_rng_t.mirror_dev, written by the pass' first kernel,
        in stream order), so a pass captured after any number of eager warm-up passes continues the stream instead of repeating their noise.
        Replays cannot advance the HOST generator: a caller who mixes replays with later eager passes or torch.randn calls on the same
        generator calls sync_generator_with_captured_noise() after the replays (one 16-byte read-back), else those draws overlap the replays'."""
        lib = _lib.load()
        props = torch.cuda.get_device_properties(device)
        cap = props.multi_processor_count * (props.max_threads_per_multi_processor // 256)
        idx = device.index if device.index is not None else torch.cuda.current_device()
        r = _lib.Rng()
        r.grid_cap_blocks = cap
        st = self.__dict__.setdefault('_rng_dev', {})
        if torch.cuda.is_current_stream_capturing():
            if idx not in st:
                raise RuntimeError('humaniflow_amd: run one eager forward(num_samples > 0) on this device before capturing it into a HIP graph '
                                   '(the captured pass draws its noise from a device-resident generator state created by the first eager pass)')
            r.state_dev = st[idx].data_ptr()
            return r
        gen = torch.cuda.default_generators[idx]
        seed, off = gen.initial_seed(), gen.get_offset()
        inc = int(lib.hmf_flow_rng_increment(B, N, self.num_shape_params, int(draw_shape), cap))
        gen.set_offset(off + inc)
        r.seed, r.offset = seed & 0xFFFFFFFFFFFFFFFF, off
        if idx not in st:
            st[idx] = torch.zeros(2, dtype=torch.int64, device=device)
        r.mirror_dev = st[idx].data_ptr()           # the pass' first kernel stores {seed, off + inc} there
        return r

    def sync_generator_with_captured_noise(self, device=None):
        """Advance torch's device generator past the noise that replays of captured passes have consumed (they advance only the device-resident
        state), so that the next eager pass / torch.randn continues after them.  One 16-byte device-to-host read (synchronises the
        current stream).  No-op when no pass ran on the device or the generator carries another seed (the next eager pass re-bases the
        device state).  Not called implicitly: a generator re-seeded to the SAME seed must restart at its own offset."""
        device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        idx = device.index if device.index is not None else torch.cuda.current_device()
        st = self.__dict__.get('_rng_dev', {})
        if idx not in st:
            return
        seed_d, off_d = st[idx].tolist()
        gen = torch.cuda.default_generators[idx]
        if (seed_d & 0xFFFFFFFFFFFFFFFF) == (gen.initial_seed() & 0xFFFFFFFFFFFFFFFF) and off_d > gen.get_offset():
            gen.set_offset(off_d)

    def draw_noise(self, device, B, N, draw_shape=True):
        """The base noise of one forward(num_samples = N) as tensors, consuming the device generator exactly as the reference's draws do (and as the
        inference path does inside its kernels): eps_shape (N,B,nb) = the pass' first normal_() call (None with draw_shape=False), eps_pose
        (23,B,N,3) = one (B,N,3) call per body part.  `forward(..., noise={'shape': eps_shape, 'pose': eps_pose})` replays them."""
        lib = _lib.load()
        device = torch.device(device)
        nb, P = self.num_shape_params, self.num_bodyparts
        rng = self._rng_for_pass(device, B, N, draw_shape)
        eps_shape = torch.empty(N, B, nb, dtype=torch.float32, device=device) if draw_shape else None
        eps_pose = torch.empty(P, B, N, 3, dtype=torch.float32, device=device)
        with torch.cuda.device(device):
            ws = self._workspace(device, 256, tag='rng')
            _lib.check(lib.hmf_flow_draw_noise(C.byref(rng), B, N, nb, _lib.dptr(eps_shape), _lib.dptr(eps_pose), _lib.dptr(ws), ws.numel(),
                                               _lib.current_stream()), 'hmf_flow_draw_noise')
        return eps_shape, eps_pose

    def reseed_captured_noise(self, device=None):
        """Re-base the device-resident generator state that HIP-graph-captured passes draw from on the device generator's current state."""
        device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        idx = device.index if device.index is not None else torch.cuda.current_device()
        gen = torch.cuda.default_generators[idx]
        seed = gen.initial_seed()
        t = torch.tensor([seed - (1 << 64) if seed >= (1 << 63) else seed, gen.get_offset()], dtype=torch.int64).to(device)
        st = self.__dict__.setdefault('_rng_dev', {})
        if idx in st:
            st[idx].copy_(t)
        else:
            st[idx] = t

    # ---- reference helper methods (kept for API parity; single calls into the library) -----------------------------
    def _needs_grad(self, *tensors):
        need = torch.is_grad_enabled() and (any(t is not None and t.requires_grad for t in tensors)
                                            or any(p.requires_grad for p in self._flow_params()))
        return need

    def _isgc_kernel(self, input_feats, shape, glob_R, cam):
        lib = _lib.load()
        B = shape.shape[0]
        f32c = lambda t: t.detach().to(torch.float32).contiguous()
        f = f32c(input_feats)
        out = torch.empty(B, self.model_cfg.INPUT_SHAPE_GLOB_CAM_FEATS_DIM, dtype=torch.float32, device=f.device)
        with torch.cuda.device(f.device):
            _lib.check(lib.hmf_flow_isgc(self._flow_handle(f.device), _lib.dptr(f), _lib.dptr(f32c(shape)), _lib.dptr(f32c(glob_R)),
                                         _lib.dptr(f32c(cam)), B, _lib.dptr(out), _lib.current_stream()), 'hmf_flow_isgc')
        return out

    def _isgc_feat_part(self, input_feats):
        """W[:, :F] . feats + b of the layer below: the 2048-wide product that the sample path and the teacher-forced path of one training pass
        share (same image features; round 3: computed once per pass, and on the aligned K = 2048 instead of K = 2060 of the concatenated input)."""
        lin = self.fc_input_shape_glob_cam_feats
        return torch.nn.functional.linear(input_feats, lin.weight[:, :self.input_feats_dim], lin.bias)

    def _isgc_autograd(self, input_feats, shape, glob_R, cam, feat_part=None):
        """The same layer in torch ops (training: W.[f, shape, vec(R), cam] split so that per-sample shapes (B,N,nb) do not replicate the
        2048-wide feature product).  shape (B,nb) -> (B,256); shape (B,N,nb) -> (B,N,256)."""
        lin = self.fc_input_shape_glob_cam_feats
        Fd, nb = self.input_feats_dim, self.num_shape_params
        W = lin.weight
        img = (feat_part if feat_part is not None else self._isgc_feat_part(input_feats)) \
            + torch.nn.functional.linear(torch.cat([glob_R.reshape(glob_R.shape[0], 9), cam], dim=1), W[:, Fd + nb:])
        sh = torch.nn.functional.linear(shape, W[:, Fd:Fd + nb])
        return torch.nn.functional.elu(img[:, None] + sh if shape.dim() == 3 else img + sh)

    def compute_input_shape_glob_cam_feats(self, input_feats, shape, glob_R, cam, feat_part=None):
        """models/humaniflow_model.py:116-150.  (B,F),(B,nb),(B,3,3),(B,3) -> (B,256); the per-sample variant (B,N,nb) of the reference is
        fused into hmf_flow_forward on the inference path and only exists here in its differentiable (training) form."""
        if self._needs_grad(input_feats, shape, glob_R, cam):
            return self._isgc_autograd(input_feats, shape, glob_R, cam, feat_part)
        if shape.dim() != 2:
            raise NotImplementedError('per-sample features are computed inside hmf_flow_forward')
        return self._isgc_kernel(input_feats, shape, glob_R, cam)

    def _flow_context_kernel(self, isgc, pose_R):
        lib = _lib.load()
        B = pose_R.shape[0]
        x = isgc.detach().float().contiguous()
        R = pose_R.detach().float().contiguous()
        ctx = torch.empty(B, self.num_bodyparts, self.model_cfg.NORM_FLOW.CONTEXT_DIM, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.hmf_flow_context(self._flow_handle(x.device), _lib.dptr(x), _lib.dptr(R), B, _lib.dptr(ctx),
                                            _lib.current_stream()), 'hmf_flow_context')
        return ctx

    def compute_flow_context_all(self, input_shape_glob_cam_feats, pose_SO3, anchor=None):
        """models/humaniflow_model.py:152-186 for all 23 parts at once from GIVEN rotations: (B,256),(B,23,3,3) -> (B,23,64).
        `anchor`: the (token, stash) of autograd.new_flow_anchor shared by the consumers of one forward pass (made here when absent); it rides
        on the returned tensor so that log_prob_parts(ctx, ...) joins the same pass."""
        if self._needs_grad(input_shape_glob_cam_feats, pose_SO3):
            from .autograd import FlowContext, new_flow_anchor
            anchor = anchor or new_flow_anchor(self)
            out = FlowContext.apply(self, input_shape_glob_cam_feats, pose_SO3, *anchor)
            out._hmf_flow_anchor = anchor
            return out
        return self._flow_context_kernel(input_shape_glob_cam_feats, pose_SO3)

    def _log_prob_kernel(self, ctx, value, part0, P, group):
        """ctx (B,P,64) fp32; value (B,P,3,3) fp64 rotations (group) or (B,P,3) fp32 algebra vectors -> (B,P) fp32."""
        lib = _lib.load()
        B = ctx.shape[0]
        ids = (C.c_int32 * P)(*range(part0, part0 + P)) if P != self.num_bodyparts else None
        out = torch.empty(B, P, dtype=torch.float32, device=ctx.device)
        cc = ctx.detach().to(torch.float32).contiguous()
        with torch.cuda.device(ctx.device):
            h = self._flow_handle(ctx.device)
            if group:
                R = value.detach().to(torch.float64).contiguous()
                _lib.check(lib.hmf_flow_log_prob_SO3(h, _lib.dptr(cc), ids, P, _lib.dptr(R), B, _lib.dptr(out), _lib.current_stream()),
                           'hmf_flow_log_prob_SO3')
            else:
                v = value.detach().to(torch.float32).contiguous()
                _lib.check(lib.hmf_flow_log_prob_so3(h, _lib.dptr(cc), ids, P, _lib.dptr(v), B, _lib.dptr(out), _lib.current_stream()),
                           'hmf_flow_log_prob_so3')
        return out

    def log_prob_parts(self, ctx, R, part0, P):
        """log p(R_j | ctx_j), j = part0 .. part0+P-1; differentiable w.r.t. ctx and the hyper-network weights -- or, when R requires a gradient, w.r.t. R and ctx."""
        _lib.require_cuda(R, 'value')
        if torch.is_grad_enabled() and R.requires_grad:
            # round 6, the flow-as-prior call pattern (optimise/optimise_humaniflow.py:96-109): differentiable w.r.t. R and the context, weights constant
            from .autograd import _is_variant, log_prob_SO3_wrt_value
            if _is_variant(self):
                raise NotImplementedError('humaniflow_amd: gradients w.r.t. the rotation a flow density is evaluated AT are built for the default flow '
                                          '(spline coupling, fixed permutations) only')
            if any(p.requires_grad for p in self._flow_params()) and not self.__dict__.get('_warned_value_grad'):
                import warnings
                warnings.warn('humaniflow_amd: log_prob(R) with R.requires_grad treats the flow weights as constants (gradients reach R and the context only); '
                              'freeze the flow parameters to silence this')
                self.__dict__['_warned_value_grad'] = True
            return log_prob_SO3_wrt_value(self, ctx.to(torch.float32), R, part0, P)
        if self._needs_grad(ctx):
            from .autograd import FlowLogProbSO3, new_flow_anchor
            anchor = getattr(ctx, '_hmf_flow_anchor', None) or new_flow_anchor(self)
            return FlowLogProbSO3.apply(self, part0, P, ctx, R.detach().to(torch.float64).contiguous(), *anchor)
        return self._log_prob_kernel(ctx, R, part0, P, group=True)

    def log_prob_all_parts(self, ctx, R):
        """Batched form of `[d.log_prob(R[:, j]) for j, d in enumerate(conditioned_pose_SO3flow_dists_for_loglik)]`
        (losses/humaniflow_loss.py:30-31): ctx (B,23,64), R (B,23,3,3) -> (B,23) fp32, one launch (and one backward launch)."""
        return self.log_prob_parts(ctx, R, 0, self.num_bodyparts)

    # ---- forward ----------------------------------------------------------------------------------------------------
    def forward(self, input, compute_point_est=True, num_samples=0, use_shape_mode_for_samples=False, compute_for_loglik=False,
                shape_for_loglik=None, pose_R_for_loglik=None, glob_R_for_loglik=None, input_feats=None,
                grad_for_pose_point_est=False, return_input_feats=False, return_input_feats_only=False, noise=None):
        """models/humaniflow_model.py:188-340.  Extra keyword `noise` = {'shape': (N,B,nb), 'pose': (23,B,N,3)} standard-normal
        tensors injects the base noise (exact-parity tests); otherwise it is drawn from the device generator in the reference's
        order (shape first, then the body parts).  Under autograd (training, train/train_humaniflow.py:248-256) the returned samples,
        shape samples and log-likelihood objects are differentiable; the point estimate is not, exactly as in the reference
        (`torch.set_grad_enabled(grad_for_pose_point_est)`, :294) unless grad_for_pose_point_est=True (round 6: its rotations join the graph through the sample
        chain's reverse sweep at zero base noise; `pose_axisangle_point_est` stays detached)."""
        if input_feats is None:
            _lib.require_cuda(input, 'input')
            input_feats = self.image_encoder(input)
        if return_input_feats_only:
            return {'input_feats': input_feats}
        _lib.require_cuda(input_feats, 'input_feats')
        batch_size = input_feats.shape[0]
        device = input_feats.device
        if compute_for_loglik:
            assert pose_R_for_loglik is not None and pose_R_for_loglik.shape[0] == batch_size
            assert shape_for_loglik is not None and shape_for_loglik.shape[0] == batch_size
            assert glob_R_for_loglik is not None and glob_R_for_loglik.shape[0] == batch_size
        grad = torch.is_grad_enabled() and (input_feats.requires_grad or any(p.requires_grad for p in self.parameters()))
        isgc_feat_part = None                   # W[:, :F] . feats of the isgc layer, once per differentiable pass
        isgc_ll = None                          # the teacher-forced isgc rows when autograd.IsgcTrain produced both uses at once
        flow_anchor = None                      # autograd.new_flow_anchor of this pass (made by the first differentiable flow consumer)
        lib = _lib.load()
        nb, P, N = self.num_shape_params, self.num_bodyparts, int(num_samples)

        if (grad and self.heads_kernel and batch_size <= 128 and isinstance(self.activation, torch.nn.ELU) and self._heads_kernel_supports(input_feats)):
            from .autograd import HeadsTrain                                             # :232-253 under autograd: library kernels forward and backward
            cam, glob_R, shape_mode, shape_log_std = HeadsTrain.apply(input_feats, self.fc1.weight, self.fc1.bias, self.fc_cam.weight, self.fc_cam.bias,
                                                                      self.fc_glob.weight, self.fc_glob.bias, self.fc_shape.weight, self.fc_shape.bias,
                                                                      self.init_cam, self.init_glob)
            shape_std = torch.exp(shape_log_std)
        elif grad or not self.heads_kernel or not self._heads_kernel_supports(input_feats):
            x = self.activation(self.fc1(input_feats))
            cam = self.fc_cam(x) + self.init_cam                                         # :237-238
            glob = self.fc_glob(x) + self.init_glob                                      # :243-244
            glob_R = rot6d_to_rotmat(glob)                                               # :245  (hmf_rot6d_to_rotmat)
            shape_params = self.fc_shape(x)
            shape_mode = shape_params[:, :nb]
            shape_log_std = shape_params[:, nb:]
            shape_std = torch.exp(shape_log_std)
        else:
            cam, glob_R, shape_mode, shape_log_std, shape_std = self._heads_kernel(input_feats)       # :232-253, two launches (hmf_heads)
        shape_dist = Normal(loc=shape_mode, scale=shape_std, validate_args=False)

        return_dict = {'cam_wp': cam, 'glob_rotmat': glob_R, 'shape_mode': shape_mode, 'shape_log_std': shape_log_std,
                       'shape_dist_for_loglik': shape_dist}

        if compute_point_est or N > 0:
            f32c = lambda t: t.detach().to(torch.float32).contiguous()
            feats_c, glob_c, cam_c, mode_c, lstd_c = f32c(input_feats), f32c(glob_R), f32c(cam), f32c(shape_mode), f32c(shape_log_std)
            eps_shape = eps_pose = rng = None
            flags0 = 0
            if N > 0:
                if use_shape_mode_for_samples:
                    flags0 = _lib.HMF_FLOW_SHAPE_MODE
                if noise is not None:
                    eps_pose = f32c(noise['pose'])
                    assert tuple(eps_pose.shape) == (P, batch_size, N, 3)
                    if not use_shape_mode_for_samples:
                        eps_shape = f32c(noise['shape'])
                        assert tuple(eps_shape.shape) == (N, batch_size, nb)
                elif grad:
                    # training: the reparameterised-sample graph below needs the noise as tensors; same draw order as the reference (:258, :309)
                    eps_shape, eps_pose = self.draw_noise(device, batch_size, N, draw_shape=not use_shape_mode_for_samples)
                else:
                    # inference: drawn inside the flow kernels from the device generator's Philox stream, call for call what the reference
                    # draws -- (N,B,nb) for the shapes (:258), then one (B,N,3) per body part (:309); csrc/rng_philox.h
                    rng = self._rng_for_pass(device, batch_size, N, draw_shape=not use_shape_mode_for_samples)
            # Sample rows (b*N + n) and point-estimate rows (B*N + b) share one buffer each, so that ONE SMPL launch of M = B*N + B
            # rows can consume them (SMPL.forward_samples_and_point_est); the reference's separate tensors are views of it.
            BN = batch_size * N
            packed = compute_point_est and N > 0
            flags = ((_lib.HMF_FLOW_POINT_EST if compute_point_est else 0) | (_lib.HMF_FLOW_PACKED_ROWS if packed else 0) | flags0
                     | (_lib.HMF_FLOW_CHAIN32 if self.flow_chain32 is True else (_lib.HMF_FLOW_CHAIN16 if self.flow_chain32 is False else 0)))
            rot_all = torch.empty(BN + (batch_size if compute_point_est else 0), P, 3, 3, dtype=torch.float32, device=device)
            shape_all = torch.empty(BN + (batch_size if packed else 0), nb, dtype=torch.float32, device=device) if N > 0 else None
            rot_s = rot_all[:BN].view(batch_size, N, P, 3, 3) if N > 0 else None
            rot_p = rot_all[BN:] if compute_point_est else None
            shape_samples = shape_all[:BN].view(batch_size, N, nb) if N > 0 else None
            so3_p = torch.empty(batch_size, P, 3, dtype=torch.float32, device=device) if compute_point_est else None
            with torch.cuda.device(device):
                h = self._flow_handle(device)
                nbytes = lib.hmf_flow_forward_workspace_bytes(h, batch_size, N, flags)
                ws = self._workspace(device, nbytes)
                _lib.check(lib.hmf_flow_forward(h, _lib.dptr(feats_c), _lib.dptr(glob_c), _lib.dptr(cam_c), _lib.dptr(mode_c),
                                                _lib.dptr(lstd_c), _lib.dptr(eps_shape), _lib.dptr(eps_pose),
                                                None if rng is None else C.byref(rng), batch_size, N,
                                                flags, _lib.dptr(shape_all), _lib.dptr(rot_s), _lib.dptr(rot_p),
                                                _lib.dptr(so3_p), _lib.dptr(ws), ws.numel(), _lib.current_stream()),
                           'hmf_flow_forward')
            if grad and N > 0:
                # Training: attach the reparameterised-sample graph to the kernel's results (:255-258, :302-311).  The shape samples and the
                # per-row isgc features are re-expressed in torch ops (tiny), the kinematic chain's backward is the HIP reverse sweep.
                from .autograd import FlowSampleChain, new_flow_anchor
                flow_anchor = new_flow_anchor(self)
                if use_shape_mode_for_samples:
                    shape_samples = shape_mode[:, None, :].expand(-1, N, -1)
                else:
                    shape_samples = (shape_mode + eps_shape * torch.exp(shape_log_std)).transpose(0, 1)        # (B,N,nb), = the kernel's values
                if (compute_for_loglik and self.heads_kernel and self.isgc_kernel and input_feats.shape[1] % 256 == 0 and shape_for_loglik.shape == shape_mode.shape
                        and self.fc_input_shape_glob_cam_feats.out_features == 256 and self.fc_input_shape_glob_cam_feats.weight.is_contiguous()):
                    # both uses of the isgc layer in this pass (per-sample rows here, teacher-forced rows below) on the library: autograd.IsgcTrain
                    from .autograd import IsgcTrain
                    lin = self.fc_input_shape_glob_cam_feats
                    H, isgc_ll = IsgcTrain.apply(self, input_feats, lin.weight, lin.bias, glob_R, cam, shape_samples, glob_R_for_loglik, shape_for_loglik)
                    H = H.reshape(BN, -1)
                else:
                    isgc_feat_part = self._isgc_feat_part(input_feats)          # shared with the teacher-forced path below
                    H = self._isgc_autograd(input_feats, shape_samples, glob_R, cam, isgc_feat_part).reshape(BN, -1)
                rot_s = FlowSampleChain.apply(self, H, rot_s, eps_pose, *flow_anchor)
                if packed:
                    rot_all = torch.cat([rot_s.reshape(BN, P, 3, 3), rot_p], dim=0)
                    shape_all = torch.cat([shape_samples.reshape(BN, nb), shape_mode], dim=0)
            if grad and grad_for_pose_point_est and compute_point_est:
                # round 6 (models/humaniflow_model.py:291-299 with grad_for_pose_point_est=True): the z = 0 point estimate is the kinematic chain at zero base noise,
                # so its rotations join the graph through the SAME reverse sweep as the samples (one row per image, eps = 0; the exp-map derivative at the end is
                # that of so3_exp where the forward applied smplx' batch_rodrigues: the same function up to its 1e-8 quirk).  pose_axisangle_point_est stays detached.
                from .autograd import FlowSampleChain, new_flow_anchor
                flow_anchor = flow_anchor or new_flow_anchor(self)
                if isgc_feat_part is None:
                    isgc_feat_part = self._isgc_feat_part(input_feats)
                H_pe = self._isgc_autograd(input_feats, shape_mode, glob_R, cam, isgc_feat_part)
                rot_p_g = FlowSampleChain.apply(self, H_pe, rot_p.view(batch_size, 1, P, 3, 3), torch.zeros(P, batch_size, 1, 3, dtype=torch.float32, device=device),
                                                *flow_anchor).view(batch_size, P, 3, 3)
                if packed:
                    rot_all = torch.cat([rot_all[:BN], rot_p_g], dim=0)
                rot_p = rot_p_g
            if packed:                # extension key (not in the reference dict): inputs of the single fused SMPL launch
                return_dict['smpl_packed_inputs'] = {'body_pose': rot_all, 'betas': shape_all, 'num_samples': N}
            if compute_point_est:
                return_dict['pose_axisangle_point_est'] = so3_p
                return_dict['pose_rotmats_point_est'] = rot_p
            if N > 0:
                return_dict['pose_rotmats_samples'] = rot_s
                return_dict['shape_samples'] = (shape_mode[:, None, :].expand(-1, N, -1) if use_shape_mode_for_samples
                                                else shape_samples)

        if compute_for_loglik:
            isgc = isgc_ll if isgc_ll is not None else \
                self.compute_input_shape_glob_cam_feats(input_feats, shape_for_loglik, glob_R_for_loglik, cam, isgc_feat_part)    # :279-283
            ctx = self.compute_flow_context_all(isgc, pose_R_for_loglik, anchor=flow_anchor)                          # :315-318
            return_dict['conditioned_pose_so3flow_dists_for_loglik'] = [
                self.pose_so3flow_dists[j].condition(ctx[:, j], parent=(ctx, j)) for j in range(P)]
            return_dict['conditioned_pose_SO3flow_dists_for_loglik'] = [
                self.pose_SO3flow_dists[j].condition(ctx[:, j], parent=(ctx, j)) for j in range(P)]
            return_dict['flow_context_for_loglik'] = ctx          # extension: feed log_prob_all_parts() for one-launch NLL

        if return_input_feats:
            return_dict['input_feats'] = input_feats
        return return_dict
