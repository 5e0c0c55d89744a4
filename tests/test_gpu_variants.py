"""GPU parity of the non-default flow variants (models/norm_flows/pyro_conditional_norm_flow.py:60-106): generic kernels
(csrc/flow_generic.hip) vs the oracle -- whose PLU arithmetic is pinned by the reference's own transform (linear_plu_golden.npz) --
for sampling, the point estimate and both log-densities; and the DEFAULT configuration forced through the generic kernels vs the MFMA
kernels (two independent implementations of the same flow)."""
import pytest
import torch

import humaniflow_amd as H
from helpers import init_model_for_tests, oracle_cfg_from, random_rotmats
from oracle import model as omodel
from oracle import train as ot

pytestmark = pytest.mark.gpu
RTOL = 5e-5
GREL = 2e-4           # gradients: fraction of each tensor's largest reference entry (tests/test_gpu_train.py)


def _variant(permute, transform, T=2, hidden=(64, 32, 32), plu_hidden=None, seed=3):
    cfg = H.get_humaniflow_cfg_defaults()
    nf = cfg.MODEL.NORM_FLOW
    nf.PERMUTE_TYPE, nf.TRANSFORM_TYPE, nf.NUM_TRANSFORMS = permute, transform, T
    nf.TRANSFORM_NN_HIDDEN_DIMS = list(hidden)
    nf.PERMUTE_NN_HIDDEN_DIMS = None if plu_hidden is None else list(plu_hidden)
    m = init_model_for_tests(H.HumaniflowModel(torch.device('cpu'), cfg.MODEL, H.SMPL_PARENTS), seed=seed).eval()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    return m.to('cuda:0'), sd, cfg


CASES = [('linear_plu', 'spline_coupling', 2, (64, 32, 32), None),
         ('conditional_linear_plu', 'spline_coupling', 2, (64, 32, 32), (30, 30)),
         ('permute', 'additive_coupling', 2, (64, 32, 32), None),
         ('conditional_linear_plu', 'additive_coupling', 3, (40, 20), (16,)),
         ('permute', 'spline_coupling', 3, (48, 48, 24, 24), None)]


@pytest.mark.parametrize('permute,transform,T,hidden,plu_hidden', CASES)
def test_variant_matches_oracle(permute, transform, T, hidden, plu_hidden):
    model, sd, cfg = _variant(permute, transform, T, hidden, plu_hidden)
    ocfg = oracle_cfg_from(cfg.MODEL)
    B, N = 3, 7
    g = torch.Generator().manual_seed(T * 10 + len(hidden))
    feats = torch.randn(B, 512, generator=g) * 0.5
    noise = {'shape': torch.randn(N, B, 10, generator=g), 'pose': torch.randn(23, B, N, 3, generator=g)}
    ref = omodel.forward(sd, ocfg, input_feats=feats, num_samples=N, noise=noise)
    with torch.no_grad():
        out = model(None, input_feats=feats.cuda(), num_samples=N, noise={k: v.cuda() for k, v in noise.items()})
    assert (out['shape_samples'].cpu() - ref['shape_samples']).abs().max() < 1e-5
    assert (out['pose_rotmats_samples'].cpu() - ref['pose_rotmats_samples']).abs().max() < RTOL
    assert (out['pose_rotmats_point_est'].cpu() - ref['pose_rotmats_point_est']).abs().max() < RTOL
    assert (out['pose_axisangle_point_est'].cpu() - ref['pose_axisangle_point_est']).abs().max() < RTOL
    assert (ref['pose_rotmats_samples'] - torch.eye(3)).abs().max() > 0.3
    # teacher-forced densities on the group and on the algebra
    R_gt = random_rotmats(B * 23, 0.7, seed=1).view(B, 23, 3, 3)
    shape_gt, glob_gt = torch.randn(B, 10, generator=g), random_rotmats(B, 1.0, seed=2)
    refl = omodel.forward(sd, ocfg, input_feats=feats, compute_for_loglik=True, shape_for_loglik=shape_gt, pose_R_for_loglik=R_gt,
                          glob_R_for_loglik=glob_gt, compute_point_est=False)
    lp_ref = omodel.log_prob_SO3(sd, ocfg, refl['flow_context_for_loglik'], R_gt.double())
    v = torch.randn(B, 23, 3, generator=g) * 0.8
    lpv_ref = omodel.log_prob_so3(sd, ocfg, refl['flow_context_for_loglik'], v)
    with torch.no_grad():
        o = model(None, input_feats=feats.cuda(), compute_for_loglik=True, shape_for_loglik=shape_gt.cuda(), pose_R_for_loglik=R_gt.cuda(),
                  glob_R_for_loglik=glob_gt.cuda(), compute_point_est=False)
        lp = model.log_prob_all_parts(o['flow_context_for_loglik'], R_gt.cuda()).cpu()
        lpv = torch.stack([o['conditioned_pose_so3flow_dists_for_loglik'][j].log_prob(v[:, j].cuda()) for j in range(23)], 1).cpu()
    assert (lp - lp_ref).abs().max() < 1e-3 and (lpv - lpv_ref).abs().max() < 1e-3


def test_default_flow_through_generic_kernels_equals_mfma_kernels():
    model, sd, cfg = _variant('permute', 'spline_coupling')
    assert model.default_flow
    B, N = 4, 33
    g = torch.Generator().manual_seed(5)
    feats = torch.randn(B, 512, generator=g) * 0.5
    noise = {'shape': torch.randn(N, B, 10, generator=g).cuda(), 'pose': torch.randn(23, B, N, 3, generator=g).cuda()}
    R_gt = random_rotmats(B * 23, 0.7, seed=3).view(B, 23, 3, 3).cuda()
    with torch.no_grad():
        a = model(None, input_feats=feats.cuda(), num_samples=N, noise=noise, compute_for_loglik=True, shape_for_loglik=torch.zeros(B, 10).cuda(),
                  pose_R_for_loglik=R_gt, glob_R_for_loglik=R_gt[:, 0])
        lpa = model.log_prob_all_parts(a['flow_context_for_loglik'], R_gt)
        model.force_generic_flow = True
        model.invalidate_flow_weights()
        b = model(None, input_feats=feats.cuda(), num_samples=N, noise=noise, compute_for_loglik=True, shape_for_loglik=torch.zeros(B, 10).cuda(),
                  pose_R_for_loglik=R_gt, glob_R_for_loglik=R_gt[:, 0])
        lpb = model.log_prob_all_parts(b['flow_context_for_loglik'], R_gt)
    assert (a['pose_rotmats_samples'] - b['pose_rotmats_samples']).abs().max().item() < 2e-5
    assert (a['pose_rotmats_point_est'] - b['pose_rotmats_point_est']).abs().max().item() < 2e-5
    assert torch.equal(a['shape_samples'], b['shape_samples'])
    assert (lpa - lpb).abs().max().item() < 1e-4


def _sd64(sd):
    return {k: (v.double().clone().requires_grad_(True) if v.is_floating_point() else v.clone()) for k, v in sd.items()
            if not k.startswith('image_encoder')}


def _check(name, got, ref, rel=GREL):
    ref, got = ref.detach().double().cpu(), got.detach().double().cpu()
    scale = max(ref.abs().max().item(), 1e-12)
    err = (got - ref).abs().max().item()
    assert err <= rel * scale, '%s: max |diff| %.3e vs scale %.3e (rel %.2e)' % (name, err, scale, err / scale)


def _check_flow_params(model, sd64, rel=GREL):
    n = 0
    for name, p in model.named_parameters():
        if not name.startswith(('pose_so3flow_transform_modules', 'fc_flow_context', 'fc_input_shape_glob_cam_feats', 'fc_cam', 'fc_glob', 'fc_shape', 'fc1')):
            continue
        ref = sd64[name].grad
        assert ref is not None and p.grad is not None, name
        _check(name, p.grad, ref, rel)
        n += 1
    return n


@pytest.mark.parametrize('permute,transform,T,hidden,plu_hidden', CASES)
def test_variant_gradients_match_oracle_autograd(permute, transform, T, hidden, plu_hidden):
    """The training graph of the reference through a variant flow (train/train_humaniflow.py:248-330 with a non-default NORM_FLOW): reparameterised
    samples (ancestors feeding descendants) + shape samples + the teacher-forced SO(3) log-densities, every gradient vs float64 autograd
    through the oracle: transform hyper-networks, LU parameters / PLU networks, context layers, the isgc layer, the heads, the features."""
    model, sd, cfg = _variant(permute, transform, T, hidden, plu_hidden, seed=5)
    model.train()
    ocfg = oracle_cfg_from(cfg.MODEL)
    B, N = 3, 5
    g = torch.Generator().manual_seed(T * 7 + len(hidden))
    feats = torch.randn(B, 512, generator=g) * 0.5
    noise = {'shape': torch.randn(N, B, 10, generator=g), 'pose': torch.randn(23, B, N, 3, generator=g)}
    R_gt = random_rotmats(B * 23, 0.7, seed=11).view(B, 23, 3, 3)
    from oracle.so3 import so3_exp
    R_gt[2, 2] = so3_exp(torch.tensor([[0.6, 0.0, 0.8]], dtype=torch.float64) * 2.0)[0].float()      # second pre-image inside the support
    shape_gt, glob_gt = torch.randn(B, 10, generator=g), random_rotmats(B, 1.0, seed=12)
    w = torch.randn(B, N, 23, 3, 3, generator=g)
    ws = torch.randn(B, N, 10, generator=g)
    wl = torch.randn(B, 23, generator=g)
    sd64 = _sd64(sd)
    f64 = feats.double().requires_grad_(True)
    with ot.float64_defaults():
        out = ot.forward_grad(sd64, ocfg, input_feats=f64, num_samples=N, noise={k: v.double() for k, v in noise.items()}, compute_for_loglik=True,
                              shape_for_loglik=shape_gt.double(), pose_R_for_loglik=R_gt.double(), glob_R_for_loglik=glob_gt.double())
        lp_ref = ot.log_prob_SO3_grad(sd64, ocfg, out['flow_context_for_loglik'], R_gt.double())
    ((out['pose_rotmats_samples'] * w.double()).sum() + (out['shape_samples'] * ws.double()).sum() + (lp_ref * wl.double()).sum()).backward()
    fg = feats.cuda().requires_grad_(True)
    model.zero_grad()
    o = model(None, input_feats=fg, num_samples=N, noise={k: v.cuda() for k, v in noise.items()}, compute_for_loglik=True, shape_for_loglik=shape_gt.cuda(),
              pose_R_for_loglik=R_gt.cuda(), glob_R_for_loglik=glob_gt.cuda())
    assert (o['pose_rotmats_samples'].detach().cpu() - out['pose_rotmats_samples'].detach().float()).abs().max() < RTOL
    lp = model.log_prob_all_parts(o['flow_context_for_loglik'], R_gt.cuda())
    assert (lp.detach().cpu() - lp_ref.detach().float()).abs().max() < 1e-3
    ((o['pose_rotmats_samples'] * w.cuda()).sum() + (o['shape_samples'] * ws.cuda()).sum() + (lp * wl.cuda()).sum()).backward()
    _check('d input_feats', fg.grad, f64.grad)
    n = _check_flow_params(model, sd64)
    per_t = 2 * (len(hidden) + 1) + (1 if permute == 'linear_plu' else 2 * (len(plu_hidden) + 1) if permute == 'conditional_linear_plu' else 0)
    assert n == 23 * T * per_t + 46 + 2 + 8


def test_default_flow_gradients_through_generic_kernels_equal_the_default_reverse_sweeps():
    """Two independent implementations of the same gradients: the default configuration through flow_grad.hip and, forced, through the variants'
    reverse sweeps of flow_generic.hip."""
    model, sd, cfg = _variant('permute', 'spline_coupling', seed=8)
    model.train()
    B, N = 4, 6
    g = torch.Generator().manual_seed(15)
    feats = torch.randn(B, 512, generator=g) * 0.5
    noise = {'shape': torch.randn(N, B, 10, generator=g).cuda(), 'pose': torch.randn(23, B, N, 3, generator=g).cuda()}
    R_gt = random_rotmats(B * 23, 0.7, seed=3).view(B, 23, 3, 3).cuda()
    w = torch.randn(B, N, 23, 3, 3, generator=g).cuda()
    wl = torch.randn(B, 23, generator=g).cuda()
    grads = []
    for forced in (False, True):
        model.force_generic_flow = forced
        model.invalidate_flow_weights()
        model.zero_grad()
        fg = feats.cuda().requires_grad_(True)
        o = model(None, input_feats=fg, num_samples=N, noise=noise, compute_for_loglik=True, shape_for_loglik=torch.zeros(B, 10).cuda(),
                  pose_R_for_loglik=R_gt, glob_R_for_loglik=R_gt[:, 0])
        lp = model.log_prob_all_parts(o['flow_context_for_loglik'], R_gt)
        ((o['pose_rotmats_samples'] * w).sum() + (lp * wl).sum()).backward()
        grads.append(({n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}, fg.grad.clone()))
    (ga, fa), (gb, fb) = grads
    assert ga.keys() == gb.keys() and len(ga) >= 23 * 16 + 46
    _check('d input_feats', fb, fa)
    for n in ga:
        _check(n, gb[n], ga[n])
