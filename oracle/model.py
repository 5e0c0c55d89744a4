"""Oracle: HumaniflowModel.forward restated on CPU from a state_dict. Test infrastructure only.

Follows models/humaniflow_model.py:188-340 literally ("reference mode"): Python loop over the
23 body parts, per-part gather-cat-Linear-ELU context (:152-186), per-call hyper-network,
fp64 Rodrigues for samples with the fp32 round on store (:311), smplx batch_rodrigues for the
point estimate (:299).  RNG draw order as in the reference (:258, :309): one standard normal
of shape (N,B,10) for the shape samples, then one (B,N,3) per body part -- or taken from the
`noise` dict {'shape': (N,B,10), 'pose': (23,B,N,3)} for exact parity tests.
This is also what bench.py times as the CPU baseline (kind "port").
"""
from collections import defaultdict

import torch
import torch.nn.functional as F

from . import so3
from .flow import PartFlow
from .resnet import resnet_forward

SMPL_PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]


def immediate_parent_to_all_ancestors(parents):
    """models/humaniflow_model.py:16-30 (root excluded, [parent, grand-parent, ...] order)."""
    anc = defaultdict(list)
    for i in range(1, len(parents)):
        p = parents[i] - 1
        if p >= 0:
            anc[i - 1] += [p] + anc[p]
        else:
            anc[i - 1] += []
    return anc


class OracleCfg:
    def __init__(self, num_resnet_layers=50, num_betas=10, context_dim=64, num_transforms=2,
                 radius=1.5 * 3.141592653589793, base_std=0.6, count_bins=8, transform_type='spline_coupling', permute_type='permute'):
        self.transform_type = transform_type
        self.permute_type = permute_type
        self.num_resnet_layers = num_resnet_layers
        self.num_betas = num_betas
        self.context_dim = context_dim
        self.num_transforms = num_transforms
        self.radius = radius
        self.base_std = base_std
        self.count_bins = count_bins


def _lin(sd, name, x):
    return F.linear(x, sd[name + '.weight'], sd[name + '.bias'])


def _layers(sd, prefix):
    n = len([key for key in sd if key.startswith(prefix) and key.endswith('.weight')])
    return [(sd[prefix + '%d.weight' % i], sd[prefix + '%d.bias' % i]) for i in range(n)]


def part_flow(sd, cfg, j):
    """Transforms of body part j.  Default flow: coupling modules at index T*j + t (SURVEY 8b checkpoint layout); with a PLU
    permutation the reference's module list interleaves [PLU module, coupling] per transform (pyro_conditional_norm_flow.py:115)."""
    ptype = getattr(cfg, 'permute_type', 'permute')
    per = 1 if ptype == 'permute' else 2
    couplings, plu = [], []
    for t in range(cfg.num_transforms):
        k = (cfg.num_transforms * j + t) * per
        if ptype == 'linear_plu':
            plu.append(sd['pose_so3flow_transform_modules.%d.LU' % k])
        elif ptype == 'conditional_linear_plu':
            plu.append(_layers(sd, 'pose_so3flow_transform_modules.%d.cond_plu_nn.layers.' % k))
        couplings.append(_layers(sd, 'pose_so3flow_transform_modules.%d.nn.layers.' % (k + per - 1)))
    return PartFlow(couplings, cfg.radius, cfg.base_std, cfg.count_bins, getattr(cfg, 'transform_type', 'spline_coupling'), plu or None)


def isgc_feats(sd, feats, shape, glob_R, cam):
    """models/humaniflow_model.py:116-150."""
    if shape.dim() == 3:
        B, N = shape.shape[:2]
        x = torch.cat([feats[:, None, :].expand(-1, N, -1), shape,
                       glob_R.reshape(B, 1, -1).expand(-1, N, -1), cam[:, None, :].expand(-1, N, -1)], dim=-1)
    else:
        x = torch.cat([feats, shape, glob_R.reshape(shape.shape[0], -1), cam], dim=-1)
    return F.elu(_lin(sd, 'fc_input_shape_glob_cam_feats', x))


def flow_context(sd, j, ancestors, isgc, pose_SO3):
    """models/humaniflow_model.py:152-186."""
    if len(ancestors) > 0:
        if pose_SO3.dim() == 5:
            B, N = pose_SO3.shape[:2]
            x = torch.cat([isgc, pose_SO3[:, :, ancestors].reshape(B, N, -1)], dim=-1)
        else:
            x = torch.cat([isgc, pose_SO3[:, ancestors].reshape(pose_SO3.shape[0], -1)], dim=-1)
    else:
        x = isgc
    return F.elu(_lin(sd, 'fc_flow_context.%d' % j, x))


def heads(sd, cfg, feats):
    """models/humaniflow_model.py:232-252."""
    x = F.elu(_lin(sd, 'fc1', feats))
    cam = _lin(sd, 'fc_cam', x) + sd['init_cam']
    glob = _lin(sd, 'fc_glob', x) + sd['init_glob']
    glob_R = so3.rot6d_to_rotmat(glob)
    sp = _lin(sd, 'fc_shape', x)
    return cam, glob_R, sp[:, :cfg.num_betas], sp[:, cfg.num_betas:]


@torch.no_grad()
def forward(sd, cfg, input=None, num_samples=0, compute_point_est=True, use_shape_mode_for_samples=False,
            input_feats=None, noise=None, parents=SMPL_PARENTS,
            compute_for_loglik=False, shape_for_loglik=None, pose_R_for_loglik=None, glob_R_for_loglik=None):
    anc = immediate_parent_to_all_ancestors(parents)
    P = len(parents) - 1
    feats = input_feats if input_feats is not None else resnet_forward(sd, input, cfg.num_resnet_layers, 'image_encoder.')
    B = feats.shape[0]
    cam, glob_R, shape_mode, shape_log_std = heads(sd, cfg, feats)
    out = {'cam_wp': cam, 'glob_rotmat': glob_R, 'shape_mode': shape_mode, 'shape_log_std': shape_log_std,
           'input_feats': feats}
    N = num_samples
    if N > 0:
        if use_shape_mode_for_samples:
            shape_samples = shape_mode[:, None, :].expand(-1, N, -1)
        else:
            eps = noise['shape'] if noise is not None else torch.randn(N, B, cfg.num_betas)
            shape_samples = (shape_mode + eps * torch.exp(shape_log_std)).transpose(0, 1)
        isgc_s = isgc_feats(sd, feats, shape_samples, glob_R, cam)
        R_s = torch.zeros(B, N, P, 3, 3)
    if compute_point_est:
        isgc_p = isgc_feats(sd, feats, shape_mode, glob_R, cam)
        v_p = torch.zeros(B, P, 3)
        R_p = torch.zeros(B, P, 3, 3)
    if compute_for_loglik:
        isgc_l = isgc_feats(sd, feats, shape_for_loglik, glob_R_for_loglik, cam)
        ctx_l = torch.zeros(B, P, cfg.context_dim)
    for j in range(P):
        pf = part_flow(sd, cfg, j)
        if compute_point_est:
            ctx = flow_context(sd, j, anc[j], isgc_p, R_p)
            v = pf.forward(torch.zeros(B, 3), ctx)
            v_p[:, j] = v
            R_p[:, j] = so3.batch_rodrigues(v)
        if N > 0:
            ctx = flow_context(sd, j, anc[j], isgc_s, R_s)
            eps = noise['pose'][j] if noise is not None else torch.randn(B, N, 3)
            R, _ = pf.sample_SO3(eps, ctx)
            R_s[:, :, j] = R                                                   # fp64 -> fp32 on store
        if compute_for_loglik:
            ctx_l[:, j] = flow_context(sd, j, anc[j], isgc_l, pose_R_for_loglik)
    if compute_point_est:
        out['pose_axisangle_point_est'] = v_p
        out['pose_rotmats_point_est'] = R_p
    if N > 0:
        out['pose_rotmats_samples'] = R_s
        out['shape_samples'] = shape_samples
    if compute_for_loglik:
        out['flow_context_for_loglik'] = ctx_l
    return out


@torch.no_grad()
def log_prob_SO3(sd, cfg, ctx, R):
    """log p(R_j | ctx_j) for every part: ctx (B,23,64) fp32, R (B,23,3,3) fp64 -> (B,23) fp32
    (what losses/humaniflow_loss.py:30-31 evaluates on the conditioned SO3 dists)."""
    P = ctx.shape[1]
    return torch.stack([part_flow(sd, cfg, j).log_prob_SO3(R[:, j], ctx[:, j]) for j in range(P)], dim=1)


@torch.no_grad()
def log_prob_so3(sd, cfg, ctx, v):
    """Density on the algebra, v (B,23,3) fp32 -> (B,23)."""
    P = ctx.shape[1]
    return torch.stack([part_flow(sd, cfg, j).log_prob_so3(v[:, j], ctx[:, j]) for j in range(P)], dim=1)
