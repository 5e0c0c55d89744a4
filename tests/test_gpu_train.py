"""GPU parity of the training step's backward passes (BASELINE configs[4]): every gradient the HIP reverse sweeps produce
vs torch autograd through the float64 oracle (oracle/train.py) on the same weights, inputs, targets and base noise.
Tolerance: 2e-4 of each tensor's largest reference gradient entry (fp32 kernels and fp32 reductions over up to 576 rows vs float64)."""
import math

import numpy as np
import pytest
import torch

import humaniflow_amd as H
from helpers import init_model_for_tests, oracle_cfg_from, random_rotmats, smpl_oracle_model
from oracle import model as omodel
from oracle import smpl as osmpl
from oracle import train as ot

pytestmark = pytest.mark.gpu
REL = 2e-4


def _model(layers=18, seed=3):
    cfg = H.get_humaniflow_cfg_defaults()
    cfg.MODEL.NUM_RESNET_LAYERS = layers
    m = init_model_for_tests(H.HumaniflowModel(torch.device('cpu'), cfg.MODEL, H.SMPL_PARENTS), seed=seed).eval()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    return m.to('cuda:0'), sd, cfg


def _sd64(sd):
    return {k: (v.double().clone().requires_grad_(True) if v.is_floating_point() else v.clone()) for k, v in sd.items()
            if not k.startswith('image_encoder')}


def _check(name, got, ref, rel=REL):
    ref = ref.detach().double().cpu()
    got = got.detach().double().cpu()
    scale = max(ref.abs().max().item(), 1e-12)
    err = (got - ref).abs().max().item()
    assert err <= rel * scale, '%s: max |diff| %.3e vs scale %.3e (rel %.2e)' % (name, err, scale, err / scale)


def _check_params(model, sd64, prefixes, rel=REL):
    worst = 0.0
    n = 0
    for name, p in model.named_parameters():
        if not name.startswith(tuple(prefixes)):
            continue
        ref = sd64[name].grad
        assert ref is not None and p.grad is not None, name
        _check(name, p.grad, ref, rel)
        n += 1
    assert n > 0
    return n


def test_log_prob_backward_matches_oracle_autograd():
    model, sd, cfg = _model(seed=4)
    ocfg = oracle_cfg_from(cfg.MODEL)
    B = 9
    g = torch.Generator().manual_seed(3)
    ctx = torch.randn(B, 23, 64, generator=g) * 0.7
    R_gt = random_rotmats(B * 23, 0.7, seed=1).view(B, 23, 3, 3)
    R_gt[0, 0] = torch.eye(3)                                                  # identity target (quirk 1/2)
    from oracle.so3 import so3_exp
    ax = torch.tensor([[0.6, 0.0, 0.8]], dtype=torch.float64)
    R_gt[1, 1] = so3_exp(ax * (math.pi - 1e-4))[0].float()                      # near-pi branch
    R_gt[2, 2] = so3_exp(ax * 2.0)[0].float()                                  # second pre-image inside the support
    w = torch.randn(B, 23, generator=g)
    sd64 = _sd64(sd)
    c64 = ctx.double().requires_grad_(True)
    with ot.float64_defaults():
        lp_ref = ot.log_prob_SO3_grad(sd64, ocfg, c64, R_gt.double())
    (lp_ref * w.double()).sum().backward()
    cg = ctx.cuda().requires_grad_(True)
    model.zero_grad()
    lp = model.log_prob_all_parts(cg, R_gt.cuda())
    assert (lp.detach().cpu() - lp_ref.detach().float()).abs().max() < 1e-3
    (lp * w.cuda()).sum().backward()
    _check('d ctx', cg.grad, c64.grad)
    assert _check_params(model, sd64, ['pose_so3flow_transform_modules']) == 23 * 16
    # per-part call pattern of the reference loss (one Function per part) gives the same gradients
    ref_grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    model.zero_grad()
    cg2 = ctx.cuda().requires_grad_(True)
    tot = 0
    for j in range(23):
        tot = tot + (model.log_prob_parts(cg2[:, j:j + 1], R_gt[:, j:j + 1].cuda(), j, 1)[:, 0] * w[:, j].cuda()).sum()
    tot.backward()
    _check('d ctx (per part)', cg2.grad, cg.grad, 1e-5)
    for n, p in model.named_parameters():
        if n in ref_grads:
            _check(n + ' (per part)', p.grad, ref_grads[n], 1e-5)


def test_teacher_forced_path_backward_through_context_and_isgc_layers():
    model, sd, cfg = _model(seed=5)
    ocfg = oracle_cfg_from(cfg.MODEL)
    B = 6
    g = torch.Generator().manual_seed(6)
    feats = torch.randn(B, 512, generator=g) * 0.5
    R_gt = random_rotmats(B * 23, 0.6, seed=7).view(B, 23, 3, 3)
    shape_gt, glob_gt = torch.randn(B, 10, generator=g), random_rotmats(B, 1.0, seed=8)
    w = torch.randn(B, 23, generator=g)
    sd64 = _sd64(sd)
    f64 = feats.double().requires_grad_(True)
    with ot.float64_defaults():
        out = ot.forward_grad(sd64, ocfg, input_feats=f64, compute_for_loglik=True, shape_for_loglik=shape_gt.double(),
                              pose_R_for_loglik=R_gt.double(), glob_R_for_loglik=glob_gt.double(), compute_point_est=False)
        lp_ref = ot.log_prob_SO3_grad(sd64, ocfg, out['flow_context_for_loglik'], R_gt.double())
    (lp_ref * w.double()).sum().backward()
    fg = feats.cuda().requires_grad_(True)
    model.zero_grad()
    o = model(None, input_feats=fg, compute_for_loglik=True, shape_for_loglik=shape_gt.cuda(), pose_R_for_loglik=R_gt.cuda(),
              glob_R_for_loglik=glob_gt.cuda(), compute_point_est=False)
    dists = o['conditioned_pose_SO3flow_dists_for_loglik']
    lp = torch.stack([dists[j].log_prob(R_gt[:, j].double().cuda()) for j in range(23)], dim=1)          # the reference's loop
    assert (lp.detach().cpu() - lp_ref.detach().float()).abs().max() < 1e-3
    (lp * w.cuda()).sum().backward()
    _check('d input_feats', fg.grad, f64.grad)
    _check_params(model, sd64, ['pose_so3flow_transform_modules', 'fc_flow_context', 'fc_input_shape_glob_cam_feats', 'fc_cam', 'fc1'])


def test_gradient_wrt_the_teacher_forced_inputs_matches_oracle_autograd():
    """Round 6: the context side of the flow-as-prior call pattern (optimise/optimise_humaniflow.py:96-109): with the rotations the densities are evaluated AT
    held fixed, d sum_j w_j log p(R_j | ctx_j) / d (pose_R_for_loglik, shape_for_loglik, glob_R_for_loglik) -- the ancestors' rotations through the 23 context
    layers, shape and global rotation through the isgc layer -- vs float64 autograd through the oracle; with the weights frozen (the optimise case) and not."""
    model, sd, cfg = _model(seed=9)
    ocfg = oracle_cfg_from(cfg.MODEL)
    B = 5
    g = torch.Generator().manual_seed(10)
    feats = torch.randn(B, 512, generator=g) * 0.5
    R_eval = random_rotmats(B * 23, 0.6, seed=11).view(B, 23, 3, 3)                       # where the densities are evaluated (no gradient)
    R_anc = random_rotmats(B * 23, 0.6, seed=12).view(B, 23, 3, 3)                        # pose_R_for_loglik (gradient)
    shape_gt, glob_gt = torch.randn(B, 10, generator=g), random_rotmats(B, 1.0, seed=13)
    w = torch.randn(B, 23, generator=g)
    sd64 = _sd64(sd)
    Ra64, sh64, gl64 = (t.double().requires_grad_(True) for t in (R_anc, shape_gt, glob_gt))
    with ot.float64_defaults():
        out = ot.forward_grad(sd64, ocfg, input_feats=feats.double(), compute_for_loglik=True, shape_for_loglik=sh64, pose_R_for_loglik=Ra64,
                              glob_R_for_loglik=gl64, compute_point_est=False)
        lp_ref = ot.log_prob_SO3_grad(sd64, ocfg, out['flow_context_for_loglik'], R_eval.double())
    (lp_ref * w.double()).sum().backward()
    assert Ra64.grad[:, 22].abs().max() == 0 and Ra64.grad[:, 0].abs().max() > 0           # a leaf part is nobody's ancestor; the first spine joint is
    for frozen in (True, False):
        for p_ in model.parameters():
            p_.requires_grad_(not frozen)
        model.zero_grad()
        Ra, sh, gl = (t.cuda().requires_grad_(True) for t in (R_anc, shape_gt, glob_gt))
        o = model(None, input_feats=feats.cuda(), compute_for_loglik=True, shape_for_loglik=sh, pose_R_for_loglik=Ra, glob_R_for_loglik=gl,
                  compute_point_est=False)
        lp = model.log_prob_all_parts(o['flow_context_for_loglik'], R_eval.cuda())
        assert (lp.detach().cpu() - lp_ref.detach().float()).abs().max() < 1e-3
        (lp * w.cuda()).sum().backward()
        _check('d pose_R_for_loglik (frozen=%s)' % frozen, Ra.grad, Ra64.grad)
        _check('d shape_for_loglik', sh.grad, sh64.grad)
        _check('d glob_R_for_loglik', gl.grad, gl64.grad)
        if not frozen:
            _check_params(model, sd64, ['pose_so3flow_transform_modules', 'fc_flow_context', 'fc_input_shape_glob_cam_feats'])
    for p_ in model.parameters():
        p_.requires_grad_(True)


def test_gradient_wrt_the_rotation_a_density_is_evaluated_at_matches_oracle_autograd():
    """Round 6: `dist.log_prob(R)` with R.requires_grad -- the prior term of optimise/optimise_humaniflow.py:96-109 -- is differentiable w.r.t. R and the context.
    Forward value = the fused kernel's; d/dR and d/d(pose_R_for_loglik, shape) vs float64 autograd through the oracle (which differentiates the reference's own op
    chain: so3_log incl. its near-pi branch, the 2 pi k pre-images, the exp map's log-determinant, the inverse flow).  Rows with one and with two pre-images in the
    support, and one row within 1e-3 of pi."""
    from oracle.so3 import so3_exp
    model, sd, cfg = _model(seed=14)
    for p_ in model.parameters():
        p_.requires_grad_(False)                                         # the optimise driver's flow is frozen
    ocfg = oracle_cfg_from(cfg.MODEL)
    B = 6
    g = torch.Generator().manual_seed(15)
    feats = torch.randn(B, 512, generator=g) * 0.5
    ax = torch.randn(B * 23, 3, generator=g, dtype=torch.float64)
    ax = ax / ax.norm(dim=-1, keepdim=True)
    ang = torch.rand(B * 23, 1, generator=g, dtype=torch.float64) * 2.6 + 0.2          # 0.2 .. 2.8 rad: beyond pi / 2 the -2 pi pre-image is inside the 1.5 pi support
    ang[5] = math.pi - 1e-3                                                             # the near-pi branch of the logarithm
    R = so3_exp(ax * ang).view(B, 23, 3, 3).float().double()             # the driver's rotations are fp32 (batch_rodrigues): both sides get the same fp32-representable matrices
    shape_gt, glob_gt = torch.randn(B, 10, generator=g), random_rotmats(B, 1.0, seed=16)
    w = torch.randn(B, 23, generator=g)
    sd64 = _sd64(sd)
    R64 = R.clone().requires_grad_(True)
    sh64 = shape_gt.double().requires_grad_(True)
    with ot.float64_defaults():
        out = ot.forward_grad(sd64, ocfg, input_feats=feats.double(), compute_for_loglik=True, shape_for_loglik=sh64, pose_R_for_loglik=R64,
                              glob_R_for_loglik=glob_gt.double(), compute_point_est=False)
        lp_ref = ot.log_prob_SO3_grad(sd64, ocfg, out['flow_context_for_loglik'], R64)
    (lp_ref * w.double()).sum().backward()
    Rg = R.float().cuda().requires_grad_(True)                            # the driver's rotations are fp32 (batch_rodrigues), cast to double at the call
    sh = shape_gt.cuda().requires_grad_(True)
    o = model(None, input_feats=feats.cuda(), compute_for_loglik=True, shape_for_loglik=sh, pose_R_for_loglik=Rg, glob_R_for_loglik=glob_gt.cuda(),
              compute_point_est=False)
    dists = o['conditioned_pose_SO3flow_dists_for_loglik']
    lp = torch.stack([dists[j].log_prob(Rg[:, j].double()) for j in range(23)], dim=1)              # the reference's loop (optimise_humaniflow.py:104-109)
    with torch.no_grad():
        fused = model.log_prob_all_parts(o['flow_context_for_loglik'].detach(), Rg.detach())
    assert (lp.detach() - fused).abs().max().item() < 2e-4                                          # same value as the inference kernel
    assert (lp.detach().cpu() - lp_ref.detach().float()).abs().max() < 1e-3
    (lp * w.cuda()).sum().backward()
    near = torch.zeros(B * 23, dtype=torch.bool)
    near[5] = True
    near = near.view(B, 23)
    _check('d R (evaluated-at + ancestors), regular rows', Rg.grad.cpu()[~near], R64.grad[~near], rel=1e-3)
    # within 1e-3 of pi the logarithm takes its square-root branch (d/dR ~ 1 / component): the gradient is ~1e3 larger and ill-conditioned in its input
    _check('d R, near-pi row', Rg.grad.cpu()[near], R64.grad[near], rel=2e-2)
    _check('d shape_for_loglik', sh.grad, sh64.grad, rel=1e-3)
    for p_ in model.parameters():
        p_.requires_grad_(True)


def test_point_estimate_gradients_and_algebra_density_gradients_match_oracle_autograd():
    """Round 6: (i) grad_for_pose_point_est=True (models/humaniflow_model.py:198,291): the z = 0 point-estimate rotations are differentiable w.r.t. the features and
    the flow / heads weights through the sample chain's reverse sweep at zero noise; (ii) the algebra-density objects (conditioned_pose_so3flow_dists_for_loglik)
    are differentiable w.r.t. their argument and the context.  Both vs float64 autograd through the oracle."""
    model, sd, cfg = _model(seed=17)
    ocfg = oracle_cfg_from(cfg.MODEL)
    B = 4
    g = torch.Generator().manual_seed(18)
    feats = torch.randn(B, 512, generator=g) * 0.5
    w = torch.randn(B, 23, 3, 3, generator=g)
    sd64 = _sd64(sd)
    f64 = feats.double().requires_grad_(True)
    with ot.float64_defaults():
        out = ot.forward_grad(sd64, ocfg, input_feats=f64, num_samples=0)
    (out['pose_rotmats_point_est'] * w.double()).sum().backward()
    fg = feats.cuda().requires_grad_(True)
    model.zero_grad()
    o = model(None, input_feats=fg, grad_for_pose_point_est=True)
    assert o['pose_rotmats_point_est'].requires_grad and (o['pose_rotmats_point_est'].detach().cpu() - out['pose_rotmats_point_est'].detach().float()).abs().max() < 5e-5
    (o['pose_rotmats_point_est'] * w.cuda()).sum().backward()
    _check('d input_feats (point estimate)', fg.grad, f64.grad, rel=1e-3)
    _check_params(model, sd64, ['pose_so3flow_transform_modules', 'fc_flow_context', 'fc_input_shape_glob_cam_feats', 'fc_shape', 'fc1'], rel=1e-3)
    with torch.no_grad():                                                  # default: no graph on the point estimate, exactly as the reference
        assert not model(None, input_feats=feats.cuda())['pose_rotmats_point_est'].requires_grad
    assert not model(None, input_feats=feats.cuda().requires_grad_(True))['pose_rotmats_point_est'].requires_grad
    # (ii) algebra density objects
    for p_ in model.parameters():
        p_.requires_grad_(False)
    R_gt = random_rotmats(B * 23, 0.6, seed=19).view(B, 23, 3, 3)
    shape_gt, glob_gt = torch.randn(B, 10, generator=g), random_rotmats(B, 1.0, seed=20)
    v = torch.randn(B, 23, 3, generator=g) * 0.8
    wv = torch.randn(B, 23, generator=g)
    sd64b = _sd64(sd)
    v64 = v.double().requires_grad_(True)
    sh64 = shape_gt.double().requires_grad_(True)
    with ot.float64_defaults():
        ol = ot.forward_grad(sd64b, ocfg, input_feats=feats.double(), compute_for_loglik=True, shape_for_loglik=sh64, pose_R_for_loglik=R_gt.double(),
                             glob_R_for_loglik=glob_gt.double(), compute_point_est=False)
        lp_ref = omodel.log_prob_so3.__wrapped__(sd64b, ocfg, ol['flow_context_for_loglik'], v64)
    (lp_ref * wv.double()).sum().backward()
    vg, sh = v.cuda().requires_grad_(True), shape_gt.cuda().requires_grad_(True)
    ol2 = model(None, input_feats=feats.cuda(), compute_for_loglik=True, shape_for_loglik=sh, pose_R_for_loglik=R_gt.cuda(), glob_R_for_loglik=glob_gt.cuda(),
                compute_point_est=False)
    lp = torch.stack([ol2['conditioned_pose_so3flow_dists_for_loglik'][j].log_prob(vg[:, j]) for j in range(23)], dim=1)
    assert (lp.detach().cpu() - lp_ref.detach().float()).abs().max() < 1e-3
    (lp * wv.cuda()).sum().backward()
    _check('d v (algebra density)', vg.grad, v64.grad, rel=1e-3)
    _check('d shape_for_loglik (through the context)', sh.grad, sh64.grad, rel=1e-3)
    for p_ in model.parameters():
        p_.requires_grad_(True)


def test_sample_chain_backward_matches_oracle_autograd():
    """Reparameterised samples: d (sum w . R_samples) / d every flow parameter, the features and the heads, ancestors feeding descendants."""
    model, sd, cfg = _model(seed=6)
    ocfg = oracle_cfg_from(cfg.MODEL)
    B, N = 3, 5
    g = torch.Generator().manual_seed(9)
    feats = torch.randn(B, 512, generator=g) * 0.5
    noise = {'shape': torch.randn(N, B, 10, generator=g), 'pose': torch.randn(23, B, N, 3, generator=g)}
    w = torch.randn(B, N, 23, 3, 3, generator=g)
    ws = torch.randn(B, N, 10, generator=g)
    sd64 = _sd64(sd)
    f64 = feats.double().requires_grad_(True)
    with ot.float64_defaults():
        out = ot.forward_grad(sd64, ocfg, input_feats=f64, num_samples=N, noise={k: v.double() for k, v in noise.items()})
    ((out['pose_rotmats_samples'] * w.double()).sum() + (out['shape_samples'] * ws.double()).sum()).backward()
    fg = feats.cuda().requires_grad_(True)
    model.zero_grad()
    o = model(None, input_feats=fg, num_samples=N, noise={k: v.cuda() for k, v in noise.items()})
    assert (o['pose_rotmats_samples'].detach().cpu() - out['pose_rotmats_samples'].detach().float()).abs().max() < 5e-5
    assert not o['pose_rotmats_point_est'].requires_grad                        # as the reference: no graph through the z = 0 chain
    ((o['pose_rotmats_samples'] * w.cuda()).sum() + (o['shape_samples'] * ws.cuda()).sum()).backward()
    _check('d input_feats', fg.grad, f64.grad)
    n = _check_params(model, sd64, ['pose_so3flow_transform_modules', 'fc_flow_context', 'fc_input_shape_glob_cam_feats', 'fc_cam', 'fc_glob',
                                    'fc_shape', 'fc1'])
    assert n >= 23 * 16 + 46 + 2


def test_smpl_joints_backward_matches_oracle_autograd():
    smpl = H.SMPL(model_data=H.synthetic_smpl_model(seed=0), create_transl=False).to('cuda:0')
    om = {k: (v.double() if v.is_floating_point() else v) for k, v in smpl_oracle_model(smpl.model_dict()).items()}
    M = 7
    g = torch.Generator().manual_seed(12)
    R = random_rotmats(M * 24, 0.6, seed=13).view(M, 24, 3, 3)
    betas = torch.randn(M, 10, generator=g)
    w = torch.randn(M, 90, 3, generator=g)
    R64, b64 = R.double().requires_grad_(True), betas.double().requires_grad_(True)
    with ot.float64_defaults():
        _, j_ref = osmpl.smpl_forward(om, b64, R64[:, 1:], R64[:, :1], pose2rot=False)
    (j_ref * w.double()).sum().backward()
    body = R[:, 1:].cuda().requires_grad_(True)
    glob = R[:, :1].cuda().requires_grad_(True)
    be = betas.cuda().requires_grad_(True)
    out = smpl(body_pose=body, global_orient=glob, betas=be, pose2rot=False)
    assert (out.joints.detach().cpu() - j_ref.detach().float()).abs().max() < 1e-4
    (out.joints * w.cuda()).sum().backward()
    _check('d body_pose', body.grad, R64.grad[:, 1:])
    _check('d global_orient', glob.grad, R64.grad[:, :1])
    _check('d betas', be.grad, b64.grad)
    out2 = smpl(body_pose=body, global_orient=glob, betas=be, pose2rot=False)
    assert out2.joints.requires_grad and out2.vertices.requires_grad         # round 6: the vertices carry the graph too (tests/test_gpu_smpl.py::test_vertex_gradients_*)


def test_full_train_step_loss_and_gradients_at_batch_72():
    """TRAIN.BATCH_SIZE = 72, LOSS.NUM_J2D_SAMPLES = 8 (configs/humaniflow_config.py:40,99): the reference's scalar loss from the encoder
    features onwards and the gradient of every parameter it reaches, vs the float64 oracle."""
    from humaniflow_amd.train import train_forward_loss
    model, sd, cfg = _model(seed=7)
    ocfg = oracle_cfg_from(cfg.MODEL)
    smpl = H.SMPL(model_data=H.synthetic_smpl_model(seed=0), create_transl=False).to('cuda:0')
    om = {k: (v.double() if v.is_floating_point() else v) for k, v in smpl_oracle_model(smpl.model_dict()).items()}
    B, N = 72, 8
    g = torch.Generator().manual_seed(21)
    feats = torch.randn(B, 512, generator=g) * 0.5
    targets = {'pose_rotmats': random_rotmats(B * 23, 0.5, seed=22).view(B, 23, 3, 3), 'glob_rotmats': random_rotmats(B, 1.0, seed=23),
               'shape': torch.randn(B, 10, generator=g) * 1.25, 'joints2D': torch.rand(B, 17, 2, generator=g) * 256,
               'joints2D_vis': torch.rand(B, 17, generator=g) > 0.2}
    noise = {'shape': torch.randn(N, B, 10, generator=g), 'pose': torch.randn(23, B, N, 3, generator=g)}
    def oracle(dt):
        sdd = {k: (v.to(dt).clone().requires_grad_(True) if v.is_floating_point() else v.clone()) for k, v in sd.items() if not k.startswith('image_encoder')}
        f = feats.to(dt).clone().requires_grad_(True)
        t = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in targets.items()}
        omd = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in om.items()}
        old = torch.get_default_dtype()
        torch.set_default_dtype(dt)
        try:
            loss_o, _ = ot.train_step_loss(sdd, ocfg, omd, cfg.LOSS, 256, f, t, {k: v.to(dt) for k, v in noise.items()}, N)
        finally:
            torch.set_default_dtype(old)
        loss_o.backward()
        return loss_o.item(), f.grad.double(), {k: v.grad.double() for k, v in sdd.items() if v.is_floating_point()}
    loss_ref, gf64, g64 = oracle(torch.float64)
    # The same oracle evaluated in float32: the densities' log-derivatives are only C^0 at spline knots / break points, so a target that
    # lands within rounding of one takes the other branch in float32 and moves a few parameter gradients by ~1e-3 of their scale
    # (measured: modules.40.nn.layers.0 2.4e-3, fc_flow_context.20 5.3e-4 with these seeds -- in the float32 ORACLE as much as in the kernels).
    # The bar is therefore: within 5e-4 of the float64 gradient, plus twice the float32 oracle's own distance from it.
    _, gf32, g32 = oracle(torch.float32)
    def run(heads_kernel):
        model.heads_kernel = heads_kernel
        fg = feats.cuda().requires_grad_(True)
        model.zero_grad()
        loss, _ = train_forward_loss(model, smpl, H.HumaniflowLoss(cfg.LOSS, 256), None, {k: v.cuda() for k, v in targets.items()}, N,
                                     noise={k: v.cuda() for k, v in noise.items()}, input_feats=fg)
        assert abs(loss.item() - loss_ref) <= 2e-4 * abs(loss_ref), (loss.item(), loss_ref)
        loss.backward()
        miss = []

        def close(name, got, ref, ref32):
            got = got.detach().double().cpu()
            scale = max(ref.abs().max().item(), 1e-12)
            err, slack = (got - ref).abs().max().item(), (ref32 - ref).abs().max().item()
            if err > 5e-4 * scale + 2.0 * slack:
                miss.append((name, '%s: |hip - f64| %.3e, |f32 oracle - f64| %.3e, scale %.3e' % (name, err, slack, scale)))
                assert err <= 1e-2 * scale, miss[-1]
        close('d input_feats', fg.grad, gf64, gf32)
        n = 0
        for name, p in model.named_parameters():
            if name.startswith('image_encoder'):
                continue
            assert p.grad is not None, name
            close(name, p.grad, g64[name], g32[name])
            n += 1
        assert n == 424                                                          # every non-encoder parameter of the checkpoint layout
        return miss
    # Heads on torch operators (round 4's graph): every tensor at the bar.  Heads on the library (HeadsTrain, round 5): the forward values of cam /
    # glob_R / shape move in their last bit, which with these seeds puts ONE sample of ONE body part (15) on the other side of a spline break point
    # -- the same C^0 effect as above, now on the kernels' side only.  Tensors that miss the bar must then all belong to that one part (its two
    # coupling hyper-networks pose_so3flow_transform_modules.{2p, 2p+1} and its context layer fc_flow_context.p), by no more than 1e-2 of their
    # scale (measured: modules.30.nn.layers.2.weight 3.2e-3, fc_flow_context.15.weight 1.3e-3); everything else stays at the bar.
    assert run(False) == []
    miss = run(True)
    model.heads_kernel = True
    def part_of(name):
        bits = name.split('.')
        if bits[0] == 'fc_flow_context':
            return int(bits[1])
        if bits[0] == 'pose_so3flow_transform_modules':
            return int(bits[1]) // 2
        return None
    parts = {part_of(nm) for nm, _ in miss}
    assert len(parts) <= 1 and None not in parts, miss


def test_train_step_from_the_proxy_input_at_configs4_size():
    """BASELINE configs[4] in one piece at its size: TRAIN.BATCH_SIZE = 72 proxy inputs of 256x256, ResNet-18, LOSS.NUM_J2D_SAMPLES = 8
    (configs/humaniflow_config.py:40,99; the step is train/train_humaniflow.py:248-352).  Product: HIP training graph of the encoder
    (batch-statistics BatchNorm) -> heads -> teacher-forced log-density + reparameterised samples -> SMPL joints -> HumaniflowLoss ->
    backward through all of it.  Oracle: the plain module encoder in float64 on the CPU (pinned to the reference's resnet18 in train mode by
    tests/golden/encoder_train_golden.npz) feeding oracle/train.py in float64, torch autograd.  Loss within 2e-4; every parameter gets a
    finite gradient; features 2e-5 x scale; head / flow gradients at the bar of the feature-entry test against the oracle evaluated at the HIP features
    (see below); encoder gradients in relative L2 (deep-ReLU mask flips)."""
    import copy
    from humaniflow_amd import train as T
    from humaniflow_amd.encoder import ResNetEncoder
    model, sd, cfg = _model(seed=9)
    model.train()
    ocfg = oracle_cfg_from(cfg.MODEL)
    dev = torch.device('cuda:0')
    smpl = H.SMPL(model_data=H.synthetic_smpl_model(seed=0), create_transl=False).to(dev)
    for p in smpl.parameters():
        p.requires_grad_(False)
    om = {k: (v.double() if v.is_floating_point() else v) for k, v in smpl_oracle_model(smpl.model_dict()).items()}
    B, N, S = 72, 8, cfg.DATA.PROXY_REP_SIZE
    assert S == 256 and cfg.TRAIN.BATCH_SIZE == 72 and cfg.LOSS.NUM_J2D_SAMPLES == 8
    x, targets = T.synthetic_train_batch(smpl, B, S, dev, seed=31, data_cfg=cfg.DATA)
    assert tuple(x.shape) == (B, 18, S, S)
    g = torch.Generator().manual_seed(32)
    noise = {'shape': torch.randn(N, B, 10, generator=g), 'pose': torch.randn(23, B, N, 3, generator=g)}
    enc_sd = {k[len('image_encoder.'):]: v for k, v in sd.items() if k.startswith('image_encoder.')}

    def oracle(dt, feats_in=None):
        if feats_in is not None:                      # heads + flow + loss only, from given features
            sdd = {k: (v.to(dt).clone().requires_grad_(True) if v.is_floating_point() else v.clone()) for k, v in sd.items() if not k.startswith('image_encoder')}
            t = {k: (v.cpu().to(dt) if v.is_floating_point() else v.cpu()) for k, v in targets.items()}
            omd = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in om.items()}
            old = torch.get_default_dtype()
            torch.set_default_dtype(dt)
            try:
                loss_o, _ = ot.train_step_loss(sdd, ocfg, omd, cfg.LOSS, S, feats_in.to(dt), t, {k: v.to(dt) for k, v in noise.items()}, N)
            finally:
                torch.set_default_dtype(old)
            loss_o.backward()
            return loss_o.item(), {k: v.grad.double() for k, v in sdd.items() if v.is_floating_point()}
        enc = ResNetEncoder(18, 18)
        enc.load_state_dict(enc_sd, strict=True)
        enc = enc.to(dt).train()
        sdd = {k: (v.to(dt).clone().requires_grad_(True) if v.is_floating_point() else v.clone()) for k, v in sd.items() if not k.startswith('image_encoder')}
        t = {k: (v.cpu().to(dt) if v.is_floating_point() else v.cpu()) for k, v in targets.items()}
        omd = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in om.items()}
        old = torch.get_default_dtype()
        torch.set_default_dtype(dt)
        try:
            feats = enc.forward_module(x.cpu().to(dt))
            loss_o, _ = ot.train_step_loss(sdd, ocfg, omd, cfg.LOSS, S, feats, t, {k: v.to(dt) for k, v in noise.items()}, N)
        finally:
            torch.set_default_dtype(old)
        loss_o.backward()
        grads = {k: v.grad.double() for k, v in sdd.items() if v.is_floating_point()}
        grads.update({'image_encoder.' + k: p.grad.double() for k, p in enc.named_parameters()})
        stats = {'image_encoder.' + k: b.detach().double() for k, b in enc.named_buffers() if not k.endswith('num_batches_tracked')}
        return loss_o.item(), feats.detach().double(), grads, stats
    loss_ref, f64, g64, st64 = oracle(torch.float64)
    _, f32, g32, _ = oracle(torch.float32)
    model.zero_grad()
    seen = {}
    hook = model.image_encoder.register_forward_hook(lambda m, i, o: seen.__setitem__('feats', o.detach().double().cpu()))
    loss, pred = T.train_forward_loss(model, smpl, H.HumaniflowLoss(cfg.LOSS, S), x, targets, N, noise={k: v.to(dev) for k, v in noise.items()})
    hook.remove()
    assert abs(loss.item() - loss_ref) <= 2e-4 * abs(loss_ref), (loss.item(), loss_ref)
    loss.backward()
    # The encoder's features against the float64 module: 2e-5 x scale (the inference bar).  The head / flow gradients are then held to the float64
    # oracle evaluated AT the HIP features: the spline's bin search, the SO(3) pre-image choice and the ReLUs of the flow make the loss piecewise
    # smooth in the features, and one of the 72 x 23 teacher-forced rotations sitting within 1e-6 of a bin edge moves every flow gradient by ~1/72
    # of its scale when a feature moves by one fp32 rounding (seen: the direct stem lands on one side with the float64 reference, the polyphase
    # stem -- whose stem output is CLOSER to float64 -- on the other).  Parity of the two halves is what the step needs; the knife edge is the
    # reference's own.
    f_err = (seen['feats'] - f64).abs().max().item()
    assert f_err <= 2e-5 * max(1.0, f64.abs().max().item()), f_err
    loss_h, g64_h = oracle(torch.float64, feats_in=seen['feats'])
    _, g32_h = oracle(torch.float32, feats_in=seen['feats'])
    assert abs(loss.item() - loss_h) <= 2e-4 * abs(loss_h), (loss.item(), loss_h)
    l2 = lambda a, b: ((a.double().cpu() - b).norm() / b.norm().clamp(min=1e-300)).item()
    n_enc = n_rest = 0
    worst_enc = (0.0, '')
    worst_rest, offenders = (0.0, ''), []
    for name, p in model.named_parameters():
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), name             # every parameter of the checkpoint layout is reached
        enc_p = name.startswith('image_encoder')
        ref, ref32 = (g64[name], g32[name]) if enc_p else (g64_h[name], g32_h[name])
        if enc_p:
            r, r32 = l2(p.grad, ref), ((ref32 - ref).norm() / ref.norm().clamp(min=1e-300)).item()
            worst_enc = max(worst_enc, (r, name))
            assert r < max(5e-3, 2 * r32), (name, r, r32)
            n_enc += 1
        else:
            got = p.grad.detach().double().cpu()
            scale = max(ref.abs().max().item(), 1e-12)
            err, slack = (got - ref).abs().max().item(), (ref32 - ref).abs().max().item()
            worst_rest = max(worst_rest, (err / (5e-4 * scale + 2.0 * slack), name))
            if err > 5e-4 * scale + 2.0 * slack:
                offenders.append((name, err, slack, scale))
            n_rest += 1
    print('feature error vs float64: %.3e of max |f|; worst head / flow gradient %s at %.2f of its bar' % (f_err / f64.abs().max().item(), worst_rest[1], worst_rest[0]))
    # ReLU knife edges.  The flow's hyper-networks hold 2 x 128 ReLU units per part and row: ~3.4 M pre-activations per step, so a few of them lie
    # within fp32 rounding of zero in every run.  Such a unit contributes nothing to the forward value, but its mask -- recomputed in fp32 by the
    # backward kernels -- can differ from float64's, and ONE row's contribution to the gradients of that coupling's layers and of the part's
    # context layer then differs (seen: part 3, one of the 576 sample rows, when the stem's rounding changed).  A parameter may therefore miss the
    # 5e-4 bar by at most a few rows' share of its gradient (4 x scale / (B N)), and only a handful of the 424 tensors may do so.
    for name, err, slack, scale in offenders:
        assert err <= 4.0 * scale / (B * N) + 2.0 * slack, '%s: |hip - f64| %.3e, |f32 oracle - f64| %.3e, scale %.3e' % (name, err, slack, scale)
    assert len(offenders) <= 16, [o[0] for o in offenders]
    if offenders:
        print('ReLU-mask knife edges: %d tensors outside the 5e-4 bar, worst %s at %.2e of its scale' % (
            len(offenders), max(offenders, key=lambda o: o[1] / o[3])[0], max(o[1] / o[3] for o in offenders)))
    assert n_enc == 60 and n_rest == 424
    for name, b in model.named_buffers():                                                  # running statistics of the one forward
        if name in st64:
            assert l2(b, st64[name]) < 5e-4, name
    print('loss %.6f vs %.6f; worst encoder gradient %s rel L2 %.2e' % (loss.item(), loss_ref, worst_enc[1], worst_enc[0]))


def test_rows_atb_batched_matches_float64():
    """hmf_rows_atb_batched (the flow's weight-gradient reductions dW = delta^T . a, db = column sums) vs float64 matmul: pitched operands with unused
    columns, ragged 32-tiles (m = 62, n = 65), odd row counts, an operand shared across the batch (batch stride 0)."""
    from humaniflow_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    for batch, rows, m, n, lda, ldb, shared in [(46, 576, 64, 65, 64, 65, False), (46, 216, 62, 32, 64, 32, False), (23, 577, 64, 463, 64, 463, True),
                                                (3, 1, 5, 7, 8, 7, False), (2, 31, 33, 31, 40, 31, True)]:
        A = torch.randn(batch, rows, lda, generator=g)
        Bm = torch.randn(1 if shared else batch, rows, ldb, generator=g)
        ref = torch.matmul(A[..., :m].double().transpose(1, 2), Bm[..., :n].double().expand(batch, -1, -1))
        rs = A[..., :m].double().sum(1)
        Ad, Bd = A.cuda(), Bm.cuda()
        Cd = torch.full((batch, m, n), float('nan'), device='cuda')
        sd = torch.full((batch, m), float('nan'), device='cuda')
        _lib.check(lib.hmf_rows_atb_batched(Ad.data_ptr(), Bd.data_ptr(), Cd.data_ptr(), sd.data_ptr(), batch, rows, m, n, lda, ldb, rows * lda,
                                            0 if shared else rows * ldb, _lib.current_stream()), 'atb')
        torch.cuda.synchronize()
        assert (Cd.cpu().double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item()), (batch, rows, m, n)
        assert (sd.cpu().double() - rs).abs().max().item() < 2e-5 * max(1.0, rs.abs().max().item())
    assert lib.hmf_rows_atb_batched(Ad.data_ptr(), Bd.data_ptr(), Cd.data_ptr(), None, 1, 4, 8, 8, 4, 8, 0, 0, None) == -1      # pitch < columns


def test_training_graph_guards_and_packed_weight_handle_persistence():
    """Guards of the hand-written training graph (it reads the LIVE weights in backward, frees its activations after one backward) and the
    packed flow-weight handle's life cycle (kept across train()/eval()/load_state_dict and re-packed in place: a captured HIP graph of a step
    holds its device pointers; mark_params_changed() after replays)."""
    model, sd, cfg = _model(seed=5)
    enc = model.image_encoder.train()
    x = torch.randn(2, 18, 64, 64, device='cuda')
    f = enc(x)
    with torch.no_grad():
        enc.layer1[0].conv1.weight.mul_(1.0)                              # in-place update between forward and backward
    with pytest.raises(RuntimeError, match='modified in place'):
        f.sum().backward()
    f = enc(x)
    f.sum().backward(retain_graph=True)
    with pytest.raises(RuntimeError, match='already back-propagated'):
        f.sum().backward()
    enc.layer2[0].bn1.momentum = None
    with pytest.raises(NotImplementedError, match='momentum'):
        enc(x)
    enc.layer2[0].bn1.momentum = 0.1
    enc.layer3[0].bn2.eval()
    with pytest.raises(NotImplementedError, match='eval'):
        enc(x)
    enc.layer3[0].bn2.train()
    model.load_state_dict({k: v.cuda() for k, v in sd.items()}, strict=True)      # the train-mode forwards above moved the BatchNorm statistics
    model.eval()
    with torch.no_grad():
        a = model(x, num_samples=3, noise={'shape': torch.zeros(3, 2, 10, device='cuda'), 'pose': torch.zeros(23, 2, 3, 3, device='cuda')})
    h0 = model._handles[('cuda', 0)].value
    model.train()
    model.eval()
    model.load_state_dict({k: v.cuda() for k, v in sd.items()}, strict=True)
    with torch.no_grad():
        b = model(x, num_samples=3, noise={'shape': torch.zeros(3, 2, 10, device='cuda'), 'pose': torch.zeros(23, 2, 3, 3, device='cuda')})
    assert model._handles[('cuda', 0)].value == h0                        # same arena: re-packed in place
    assert torch.allclose(a['pose_rotmats_samples'], b['pose_rotmats_samples'], atol=1e-6)
    # re-binding .data does not bump the version counter but moves the storage: seen through the (version, address) key (round 6)
    w = next(model.fc_flow_context[0].parameters())
    w.data = w.data * 1.5
    with torch.no_grad():
        c = model(x, num_samples=3, noise={'shape': torch.zeros(3, 2, 10, device='cuda'), 'pose': torch.zeros(23, 2, 3, 3, device='cuda')})
    assert not torch.equal(c['pose_rotmats_samples'], b['pose_rotmats_samples'])
    assert model._handles[('cuda', 0)].value == h0
    # a parameter changed behind BOTH (what a graph replay of an optimiser step does: same storage, no counter): stale until mark_params_changed()
    raw = torch.empty(0, device='cuda').set_(w.data.untyped_storage(), 0, w.shape, w.stride())      # an alias autograd does not know of
    raw.mul_(1.0 / 1.5)
    with torch.no_grad():
        c2 = model(x, num_samples=3, noise={'shape': torch.zeros(3, 2, 10, device='cuda'), 'pose': torch.zeros(23, 2, 3, 3, device='cuda')})
    assert torch.equal(c2['pose_rotmats_samples'], c['pose_rotmats_samples'])     # stale pack (documented)
    b = c
    model.mark_params_changed()
    with torch.no_grad():
        d = model(x, num_samples=3, noise={'shape': torch.zeros(3, 2, 10, device='cuda'), 'pose': torch.zeros(23, 2, 3, 3, device='cuda')})
    assert not torch.equal(d['pose_rotmats_samples'], b['pose_rotmats_samples'])
    assert model._handles[('cuda', 0)].value == h0


def test_train_step_skips_non_finite_updates_on_the_device():
    """train/train_humaniflow.py:264-278 (NaN roll-back): with the fused Adam the skip is decided on the device -- parameters, moments and step
    counts stay untouched when a gradient is non-finite, and a clean step afterwards updates as usual."""
    import humaniflow_amd as H
    from humaniflow_amd import train as T
    from humaniflow_amd.losses import HumaniflowLoss
    from humaniflow_amd.synthetic import init_benchmark_weights
    dev = torch.device('cuda')
    cfg = H.get_humaniflow_cfg_defaults()
    cfg.MODEL.NUM_RESNET_LAYERS = 18
    torch.manual_seed(0)
    model = init_benchmark_weights(H.HumaniflowModel(torch.device('cpu'), cfg.MODEL, H.SMPL_PARENTS), seed=0).to(dev).train()
    smpl = H.SMPL(model_data=H.synthetic_smpl_model(seed=0), batch_size=4, create_transl=False).to(dev)
    crit = HumaniflowLoss(cfg.LOSS, cfg.DATA.PROXY_REP_SIZE)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-4, fused=True)
    red = T.GradBucketReducer(params, None)
    batch = T.synthetic_train_batch(smpl, 4, cfg.DATA.PROXY_REP_SIZE, dev, seed=5, data_cfg=cfg.DATA)
    loss, sk = T.train_step(model, smpl, crit, opt, red, *batch, 2)
    assert float(sk) == 0.0 and torch.isfinite(loss)
    before = [p.detach().clone() for p in params]
    steps_before = [float(opt.state[p]['step']) for p in params if p in opt.state]
    poison = params[-1].register_hook(lambda g: g * float('nan'))
    loss, sk = T.train_step(model, smpl, crit, opt, red, *batch, 2)
    poison.remove()
    assert float(sk) == 1.0
    assert all(torch.equal(a, b.detach()) for a, b in zip(before, params))
    assert steps_before == [float(opt.state[p]['step']) for p in params if p in opt.state]
    loss, sk = T.train_step(model, smpl, crit, opt, red, *batch, 2)
    assert float(sk) == 0.0 and not all(torch.equal(a, b.detach()) for a, b in zip(before, params))


def test_train_step_rollback_policy_undoes_the_last_update_and_redoes_the_batch():
    """train/train_humaniflow.py:261-278 + :348-349 as nan_policy='rollback': an update that leaves NaN weights behind is found by the next
    forward's outputs, undone (parameters, BatchNorm buffers, Adam moments and step counts as before that update) and the batch is run again:
    after it exactly ONE update from the initial weights has been applied."""
    import humaniflow_amd as H
    from humaniflow_amd import train as T
    from humaniflow_amd.losses import HumaniflowLoss
    from humaniflow_amd.synthetic import init_benchmark_weights
    dev = torch.device('cuda')
    cfg = H.get_humaniflow_cfg_defaults()
    cfg.MODEL.NUM_RESNET_LAYERS = 18
    model = init_benchmark_weights(H.HumaniflowModel(torch.device('cpu'), cfg.MODEL, H.SMPL_PARENTS), seed=0).to(dev).train()
    smpl = H.SMPL(model_data=H.synthetic_smpl_model(seed=0), batch_size=4, create_transl=False).to(dev)
    crit = HumaniflowLoss(cfg.LOSS, cfg.DATA.PROXY_REP_SIZE)
    params = [p for p in model.parameters() if p.requires_grad]
    lr = 1e-4
    opt = torch.optim.Adam(params, lr=lr, fused=True)
    red = T.GradBucketReducer(params, None)
    keep = T.LastBatchState(model, opt)
    b0 = T.synthetic_train_batch(smpl, 4, cfg.DATA.PROXY_REP_SIZE, dev, seed=5, data_cfg=cfg.DATA)
    b1 = T.synthetic_train_batch(smpl, 4, cfg.DATA.PROXY_REP_SIZE, dev, seed=6, data_cfg=cfg.DATA)
    init = [p.detach().clone() for p in params]
    ptrs = [p.data_ptr() for p in params]
    loss, sk = T.train_step(model, smpl, crit, opt, red, *b0, 2, nan_policy='rollback', last_state=keep)
    assert float(sk) == 0.0 and torch.isfinite(loss) and keep.valid()
    with torch.no_grad():
        model.fc1.weight[0, 0] = float('nan')                                  # "some numerical instability" of the update, :262
    loss, sk = T.train_step(model, smpl, crit, opt, red, *b1, 2, nan_policy='rollback', last_state=keep)
    assert float(sk) == 0.0 and torch.isfinite(loss)
    assert ptrs == [p.data_ptr() for p in params]
    assert all(float(opt.state[p]['step']) == 1.0 for p in params if p in opt.state)       # update 0 undone, update on b1 applied
    moved = 0
    for a, p in zip(init, params):
        d = (p.detach() - a).abs()
        assert torch.isfinite(p).all() and float(d.max()) <= lr * 1.001            # Adam's first step moves a weight by <= lr
        moved += int((d > 0).sum())
    assert moved > 0
    # the kept state is now "before the update on b1": the initial weights again
    live, _ = keep._tensors()
    by_ptr = {t.data_ptr(): k for t, k in zip(live, keep._kept)}
    assert all(torch.equal(by_ptr[p.data_ptr()], a) for a, p in zip(init, params))
    # without a NaN the policy is the plain step
    loss, sk = T.train_step(model, smpl, crit, opt, red, *b0, 2, nan_policy='rollback', last_state=keep)
    assert float(sk) == 0.0 and all(float(opt.state[p]['step']) == 2.0 for p in params if p in opt.state)


@pytest.mark.parametrize('reduction', ['mean', 'sum'])
def test_fused_training_loss_equals_the_torch_operator_form(reduction):
    """hmf_train_loss (csrc/train_loss.py: joint selection, flip, projection, the four loss terms and their gradients in two launches) against the same
    step with HumaniflowLoss on torch operators (pinned to the reference's class by tests/golden/loss_golden.npz): loss and every parameter gradient."""
    import copy
    from humaniflow_amd import train as T
    model, sd, cfg = _model(seed=4)
    model.train()
    dev = torch.device('cuda:0')
    smpl = H.SMPL(model_data=H.synthetic_smpl_model(seed=0), create_transl=False).to(dev)
    for p in smpl.parameters():
        p.requires_grad_(False)
    B, N, S = 6, 3, cfg.DATA.PROXY_REP_SIZE
    lcfg = copy.deepcopy(cfg.LOSS)
    lcfg.REDUCTION = reduction
    x, targets = T.synthetic_train_batch(smpl, B, S, dev, seed=5, data_cfg=cfg.DATA)
    targets['joints2D_vis'][0, :5] = False                                   # some invisible joints
    g = torch.Generator().manual_seed(6)
    noise = {'shape': torch.randn(N, B, 10, generator=g).to(dev), 'pose': torch.randn(23, B, N, 3, generator=g).to(dev)}
    feats = torch.randn(B, model.input_feats_dim, generator=g).to(dev)
    res = {}
    for fused in (True, False):
        T.FUSED_LOSS = fused
        try:
            model.zero_grad()
            f = feats.clone().requires_grad_(True)
            loss, _ = T.train_forward_loss(model, smpl, H.HumaniflowLoss(lcfg, S), x, targets, N, noise=noise, input_feats=f)
            loss.backward()
        finally:
            T.FUSED_LOSS = True
        res[fused] = (loss.item(), f.grad.clone(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
    assert abs(res[True][0] - res[False][0]) <= 2e-6 * abs(res[False][0]), (res[True][0], res[False][0])
    assert (res[True][1] - res[False][1]).abs().max().item() <= 2e-5 * res[False][1].abs().max().item()
    assert set(res[True][2]) == set(res[False][2])
    for n, gr in res[False][2].items():
        assert (res[True][2][n] - gr).abs().max().item() <= 2e-5 * max(gr.abs().max().item(), 1e-8), n


@pytest.mark.parametrize('layers,B', [(18, 72), (50, 5)])
def test_heads_under_autograd_run_on_the_library_and_match_float64_autograd(layers, B):
    """models/humaniflow_model.py:232-253 under autograd = HeadsTrain (hmf_heads_train forward, hmf_heads_backward + two hmf_rows_atb_batched
    calls backward): outputs and every gradient (image features, fc1, fc_cam, fc_glob, fc_shape) vs torch autograd in float64 through the
    same arithmetic (Linear, ELU, init adds, the reference's 6-D -> rotation map from oracle/so3.py)."""
    from humaniflow_amd.autograd import HeadsTrain
    from oracle import so3
    model, sd, cfg = _model(layers=layers, seed=6)
    g = torch.Generator().manual_seed(B)
    F = model.fc1.in_features
    feats = (torch.randn(B, F, generator=g) * 0.5).cuda().requires_grad_(True)
    ps = [model.fc1.weight, model.fc1.bias, model.fc_cam.weight, model.fc_cam.bias, model.fc_glob.weight, model.fc_glob.bias, model.fc_shape.weight,
          model.fc_shape.bias]
    for p in ps:
        p.grad = None
    cam, R, mode, lstd = HeadsTrain.apply(feats, *ps, model.init_cam, model.init_glob)
    wts = [torch.randn(t.shape, generator=g).cuda() for t in (cam, R, mode, lstd)]
    (cam * wts[0]).sum().add((R * wts[1]).sum()).add((mode * wts[2]).sum()).add((lstd * wts[3]).sum()).backward()
    # float64 reference
    f64 = feats.detach().double().cpu().requires_grad_(True)
    q = [p.detach().double().cpu().requires_grad_(True) for p in ps]
    x = torch.nn.functional.elu(f64 @ q[0].t() + q[1])
    cam_r = x @ q[2].t() + q[3] + model.init_cam.double().cpu()
    glob = x @ q[4].t() + q[5] + model.init_glob.double().cpu()
    R_r = so3.rot6d_to_rotmat(glob)
    sp = x @ q[6].t() + q[7]
    nb = sp.shape[1] // 2
    outs_r = (cam_r, R_r, sp[:, :nb], sp[:, nb:])
    sum((o * w.double().cpu()).sum() for o, w in zip(outs_r, wts)).backward()
    for name, got, ref in zip(('cam', 'glob_R', 'shape_mode', 'shape_log_std'), (cam, R, mode, lstd), outs_r):
        assert (got.detach().double().cpu() - ref.detach().reshape(got.shape)).abs().max() < 2e-5, name
    _check('d feats', feats.grad, f64.grad)
    for name, p, r in zip(('fc1.weight', 'fc1.bias', 'fc_cam.weight', 'fc_cam.bias', 'fc_glob.weight', 'fc_glob.bias', 'fc_shape.weight', 'fc_shape.bias'), ps, q):
        _check(name, p.grad, r.grad)
    # the model's own differentiable forward takes this path (no nn.Linear for the heads): same values as the inference kernels
    model.train()
    out = model(None, input_feats=feats.detach().requires_grad_(True), num_samples=0, compute_point_est=False)
    assert out['cam_wp'].grad_fn is not None and 'HeadsTrain' in type(out['cam_wp'].grad_fn).__name__
    with torch.no_grad():
        ref = model.eval()(None, input_feats=feats.detach(), num_samples=0, compute_point_est=False)
    assert torch.equal(out['cam_wp'].detach(), ref['cam_wp']) and torch.equal(out['glob_rotmat'].detach(), ref['glob_rotmat'])


def test_isgc_layer_under_autograd_runs_on_the_library_and_matches_float64_autograd():
    """compute_input_shape_glob_cam_feats (models/humaniflow_model.py:116-150) for both uses of a training pass = autograd.IsgcTrain: outputs and every
    gradient (image features, the layer's weight and bias, predicted global rotation, camera, per-sample shapes) vs torch autograd in float64."""
    from humaniflow_amd.autograd import IsgcTrain
    model, sd, cfg = _model(seed=8)
    B, N, nb, F = 9, 5, 10, 512
    g = torch.Generator().manual_seed(4)
    mk = lambda *sh: torch.randn(*sh, generator=g)
    feats, cam, ss, sl = mk(B, F) * 0.5, mk(B, 3) * 0.3, mk(B, N, nb), mk(B, nb)
    R, Rl = random_rotmats(B, 0.8, seed=5), random_rotmats(B, 0.8, seed=6)
    lin = model.fc_input_shape_glob_cam_feats
    lin.weight.grad = lin.bias.grad = None
    dv = lambda t: t.cuda().requires_grad_(True)
    f_, R_, cam_, ss_ = dv(feats), dv(R), dv(cam), dv(ss)
    Hs, Hl = IsgcTrain.apply(model, f_, lin.weight, lin.bias, R_, cam_, ss_, Rl.cuda(), sl.cuda())
    w1, w2 = mk(B, N, 256), mk(B, 256)
    ((Hs * w1.cuda()).sum() + (Hl * w2.cuda()).sum()).backward()
    # float64 reference: the reference's cat + Linear + ELU
    d64 = lambda t: t.double().requires_grad_(True)
    f6, R6, c6, s6, W6, b6 = d64(feats), d64(R), d64(cam), d64(ss), d64(lin.weight.detach().cpu()), d64(lin.bias.detach().cpu())
    def layer(shape, rot):
        lead = shape.shape[:-1]
        ex = lambda t: t if len(lead) == 1 else t[:, None].expand(-1, N, -1)
        x = torch.cat([ex(f6), shape, ex(rot.reshape(B, 9)), ex(c6)], dim=-1)
        return torch.nn.functional.elu(x @ W6.t() + b6)
    Hs6, Hl6 = layer(s6, R6), layer(sl.double(), Rl.double())
    ((Hs6 * w1.double()).sum() + (Hl6 * w2.double()).sum()).backward()
    assert (Hs.detach().double().cpu() - Hs6.detach()).abs().max() < 2e-5 and (Hl.detach().double().cpu() - Hl6.detach()).abs().max() < 2e-5
    for name, got, ref in (('d feats', f_.grad, f6.grad), ('d glob_R', R_.grad, R6.grad), ('d cam', cam_.grad, c6.grad), ('d shape_samples', ss_.grad, s6.grad),
                           ('d weight', lin.weight.grad, W6.grad), ('d bias', lin.bias.grad, b6.grad)):
        _check(name, got, ref)


def test_the_optimise_objective_is_differentiable_end_to_end_and_sgd_lowers_it():
    """Round 6 capstone: the objective of optimise/optimise_humaniflow.py:70-115 -- 2-D joints through SMPL joints of axis-angle pose / shape (pose2rot=True), minus
    the flow's pose prior evaluated AT the optimised rotations (which are also the ancestors the prior is conditioned on), minus the shape prior -- runs through the
    boundary objects with the reference's call pattern, every optimisation variable receives a finite gradient, and a few SGD steps lower the objective."""
    from humaniflow_amd import evaluate as ev
    from humaniflow_amd.rigid_transform_utils import batch_rodrigues
    from humaniflow_amd.smpl import _rodrigues_torch
    model, sd, cfg = _model(seed=22)
    for p_ in model.parameters():
        p_.requires_grad_(False)
    dev = torch.device('cuda:0')
    smpl = H.SMPL(model_data=H.synthetic_smpl_model(seed=0), create_transl=False).to(dev)
    B, S = 3, 256
    g = torch.Generator().manual_seed(23)
    feats = (torch.randn(B, 512, generator=g) * 0.5).to(dev)
    with torch.no_grad():
        init = model(None, input_feats=feats)                                # the driver starts from the network's point estimate
    pose = init['pose_axisangle_point_est'].reshape(B, 69).clone().requires_grad_(True)
    glob = (torch.randn(B, 3, generator=g) * 0.2).to(dev).requires_grad_(True)
    shape = init['shape_mode'].clone().requires_grad_(True)
    cam = init['cam_wp'].clone().requires_grad_(True)
    target2d = (torch.rand(B, 17, 2, generator=g) * S).to(dev)
    opt = torch.optim.SGD([pose, glob, shape, cam], lr=1e-3)
    hist = []
    for it in range(6):
        o = smpl(body_pose=pose, global_orient=glob, betas=shape, pose2rot=True)
        j2d = ev.undo_keypoint_normalisation(ev.orthographic_project_torch(ev.flip_about_x(ev.select_joints(o.joints, 'coco')), cam), S)
        loss_j2d = ((j2d - target2d) ** 2).mean()
        R = _rodrigues_torch(pose.reshape(-1, 3)).view(B, 23, 3, 3)           # batch_rodrigues of the driver (:96), differentiable
        Rg = _rodrigues_torch(glob)
        d = model(None, input_feats=feats, compute_point_est=False, num_samples=0, compute_for_loglik=True, shape_for_loglik=shape,
                  pose_R_for_loglik=R, glob_R_for_loglik=Rg)
        lp = sum(d['conditioned_pose_SO3flow_dists_for_loglik'][j].log_prob(R[:, j].double()).sum() for j in range(23)) / B
        ls = d['shape_dist_for_loglik'].log_prob(shape).sum() / B
        loss = 1e-4 * loss_j2d - 0.02 * lp - 0.02 * ls
        opt.zero_grad()
        loss.backward()
        for name, t in (('pose', pose), ('glob', glob), ('shape', shape), ('cam', cam)):
            assert t.grad is not None and torch.isfinite(t.grad).all() and t.grad.abs().max() > 0, name
        opt.step()
        hist.append(float(loss.detach()))
    assert hist[-1] < hist[0], hist
    with torch.no_grad():                                                    # the kernel Rodrigues and its torch-operator form agree
        assert (batch_rodrigues(pose.detach().reshape(-1, 3)) - _rodrigues_torch(pose.detach().reshape(-1, 3))).abs().max().item() < 2e-6
    for p_ in model.parameters():
        p_.requires_grad_(True)
