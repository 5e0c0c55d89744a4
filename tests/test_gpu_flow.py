"""GPU parity: the flow (hmf_flow_forward / context / log-prob) and the whole HumaniflowModel.forward dict through the
C ABI vs the oracle's reference-mode forward.  Tolerances (BASELINE.json north_star): rotations feed vertices that must
hold 1e-4 -> rot-mats checked at 5e-5; log-probs within 1e-3."""
import math

import numpy as np
import pytest
import torch

import humaniflow_amd as H
from helpers import init_model_for_tests, oracle_cfg_from, synthetic_proxy_input, random_rotmats, smpl_oracle_model
from oracle import model as omodel
from oracle import smpl as osmpl

pytestmark = pytest.mark.gpu
RTOL = 5e-5


def _model(layers, seed=3):
    cfg = H.get_humaniflow_cfg_defaults()
    cfg.MODEL.NUM_RESNET_LAYERS = layers
    m = init_model_for_tests(H.HumaniflowModel(torch.device('cpu'), cfg.MODEL, H.SMPL_PARENTS), seed=seed).eval()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    return m.to('cuda:0'), sd, cfg


def _noise(B, N, seed=0):
    g = torch.Generator().manual_seed(seed)
    return {'shape': torch.randn(N, B, 10, generator=g), 'pose': torch.randn(23, B, N, 3, generator=g)}


def _cuda(d):
    return {k: v.cuda() for k, v in d.items()}


@pytest.mark.parametrize('layers,B,N,use_mode', [(18, 2, 13, False), (18, 3, 1, True), (50, 1, 40, False), (18, 5, 64, False)])
def test_flow_forward_matches_oracle(layers, B, N, use_mode):
    model, sd, cfg = _model(layers)
    F = 512 if layers == 18 else 2048
    feats = torch.randn(B, F, generator=torch.Generator().manual_seed(B * 7 + N)) * 0.5
    noise = _noise(B, N, seed=N)
    ref = omodel.forward(sd, oracle_cfg_from(cfg.MODEL), input_feats=feats, num_samples=N, noise=noise,
                         use_shape_mode_for_samples=use_mode)
    with torch.no_grad():
        out = model(None, input_feats=feats.cuda(), num_samples=N, noise=_cuda(noise), use_shape_mode_for_samples=use_mode)
    for k in ('cam_wp', 'glob_rotmat', 'shape_mode', 'shape_log_std'):
        assert (out[k].cpu() - ref[k]).abs().max() < 1e-5, k
    assert out['pose_rotmats_samples'].shape == (B, N, 23, 3, 3) and out['pose_rotmats_samples'].dtype == torch.float32
    assert (out['shape_samples'].cpu() - ref['shape_samples']).abs().max() < 1e-5
    assert (out['pose_rotmats_samples'].cpu() - ref['pose_rotmats_samples']).abs().max() < RTOL
    assert (out['pose_rotmats_point_est'].cpu() - ref['pose_rotmats_point_est']).abs().max() < RTOL
    assert (out['pose_axisangle_point_est'].cpu() - ref['pose_axisangle_point_est']).abs().max() < RTOL
    assert (ref['pose_rotmats_samples'] - torch.eye(3)).abs().max() > 0.3         # flow is not degenerate


@pytest.mark.parametrize('layers,B,N', [(18, 1, 10), (18, 3, 21), (50, 32, 100)])
def test_16_row_chain_kernel_agrees_with_the_32_row_kernel_and_the_oracle(layers, B, N):
    """Round 4: the kinematic loop exists on 16-row tiles (k_flow_chain16, v_mfma_f32_16x16x4_f32: the latency form, default up to 1,024 rows) and
    on 32-row tiles (k_flow_chain: the throughput form).  Two independent register layouts, packings and spline-parameter exchanges of the same
    arithmetic."""
    model, sd, cfg = _model(layers)
    F = 512 if layers == 18 else 2048
    feats = (torch.randn(B, F, generator=torch.Generator().manual_seed(B + N)) * 0.5).cuda()
    noise = _cuda(_noise(B, N, seed=B))
    with torch.no_grad():
        model.flow_chain32 = False
        a = model(None, input_feats=feats, num_samples=N, noise=noise)
        model.flow_chain32 = True
        b = model(None, input_feats=feats, num_samples=N, noise=noise)
        model.flow_chain32 = None
    for k in ('pose_rotmats_samples', 'pose_rotmats_point_est', 'pose_axisangle_point_est'):
        assert (a[k] - b[k]).abs().max().item() < 2e-5, k
    assert torch.equal(a['shape_samples'], b['shape_samples'])
    if B * N <= 100:
        ref = omodel.forward(sd, oracle_cfg_from(cfg.MODEL), input_feats=feats.cpu(), num_samples=N, noise={k: v.cpu() for k, v in noise.items()})
        assert (a['pose_rotmats_samples'].cpu() - ref['pose_rotmats_samples']).abs().max() < RTOL
        assert (a['pose_rotmats_point_est'].cpu() - ref['pose_rotmats_point_est']).abs().max() < RTOL


def test_point_estimate_only_and_samples_only_paths():
    model, sd, cfg = _model(18)
    feats = torch.randn(4, 512, generator=torch.Generator().manual_seed(1)) * 0.5
    ref = omodel.forward(sd, oracle_cfg_from(cfg.MODEL), input_feats=feats, num_samples=0)
    with torch.no_grad():
        out = model(None, input_feats=feats.cuda())
        assert 'pose_rotmats_samples' not in out
        assert (out['pose_rotmats_point_est'].cpu() - ref['pose_rotmats_point_est']).abs().max() < RTOL
        noise = _noise(4, 3)
        out2 = model(None, input_feats=feats.cuda(), compute_point_est=False, num_samples=3, noise=_cuda(noise))
        assert 'pose_rotmats_point_est' not in out2
        ref2 = omodel.forward(sd, oracle_cfg_from(cfg.MODEL), input_feats=feats, num_samples=3, noise=noise, compute_point_est=False)
        assert (out2['pose_rotmats_samples'].cpu() - ref2['pose_rotmats_samples']).abs().max() < RTOL
        only = model(torch.zeros(1, 18, 64, 64).cuda(), return_input_feats_only=True)
        assert list(only) == ['input_feats'] and only['input_feats'].shape == (1, 512)


def test_end_to_end_forward_dict_with_encoder_and_smpl():
    """predict-style call on a synthetic proxy input: the six save_pred_output model keys (utils/predict_utils.py:19-24) plus
    samples -> SMPL vertices/joints, all vs the oracle.  ResNet-50, 256x256, as BASELINE configs name."""
    model, sd, cfg = _model(50, seed=5)
    x = synthetic_proxy_input(2, 256, seed=0)
    N = 6
    noise = _noise(2, N, seed=2)
    ref = omodel.forward(sd, oracle_cfg_from(cfg.MODEL), input=x, num_samples=N, noise=noise)
    with torch.no_grad():
        out = model(x.cuda(), num_samples=N, noise=_cuda(noise), return_input_feats=True)
    scale = ref['input_feats'].abs().max().item()
    assert (out["input_feats"].cpu() - ref["input_feats"]).abs().max() < 2e-5 * max(1.0, scale)
    # feed the ORACLE's features to isolate the flow from conv round-off, then compare the whole dict
    with torch.no_grad():
        out = model(None, input_feats=ref['input_feats'].cuda(), num_samples=N, noise=_cuda(noise), return_input_feats=True)
    for k in ('pose_axisangle_point_est', 'pose_rotmats_point_est', 'shape_mode', 'glob_rotmat', 'cam_wp', 'input_feats',
              'pose_rotmats_samples', 'shape_samples', 'shape_log_std'):
        # 2048-long feature dot products are summed in a different order than on the CPU; the 23-part chain amplifies that
        # round-off, so rotations get 1e-4 here -- the north-star bar (vertices/joints 1e-4) is asserted right below
        assert (out[k].cpu() - ref[k]).abs().max() < 1e-4 * max(1.0, ref[k].abs().max().item()), k
    assert isinstance(out['shape_dist_for_loglik'], torch.distributions.Normal)
    smpl = H.SMPL(model_data=H.synthetic_smpl_model(seed=0), create_transl=False).to('cuda:0')
    so = smpl(body_pose=out['pose_rotmats_samples'].reshape(-1, 23, 3, 3),
              global_orient=out['glob_rotmat'][:, None].expand(-1, N, -1, -1).reshape(-1, 1, 3, 3),
              betas=out['shape_samples'].reshape(-1, 10), pose2rot=False)
    v_ref, j_ref = osmpl.smpl_forward(smpl_oracle_model(smpl.model_dict()), ref['shape_samples'].reshape(-1, 10),
                                      ref['pose_rotmats_samples'].reshape(-1, 23, 3, 3),
                                      ref['glob_rotmat'][:, None].expand(-1, N, -1, -1).reshape(-1, 1, 3, 3), pose2rot=False)
    assert (so.vertices.cpu() - v_ref).abs().max() < 1e-4 and (so.joints.cpu() - j_ref).abs().max() < 1e-4


def test_teacher_forced_log_prob_matches_oracle():
    model, sd, cfg = _model(18, seed=4)
    ocfg = oracle_cfg_from(cfg.MODEL)
    B = 9
    g = torch.Generator().manual_seed(3)
    feats = torch.randn(B, 512, generator=g) * 0.5
    R_gt = random_rotmats(B * 23, 0.7, seed=1).view(B, 23, 3, 3)
    R_gt[0, 0] = torch.eye(3)                                                   # quirk 1/2: identity target
    from oracle.so3 import so3_exp
    ax = torch.tensor([[0.6, 0.0, 0.8]], dtype=torch.float64)
    R_gt[1, 1] = so3_exp(ax * (math.pi - 1e-4))[0].float()                       # near-pi branch of the log map
    R_gt[2, 2] = so3_exp(ax * 2.0)[0].float()                                   # theta > pi/2: a second pre-image in support
    shape_gt = torch.randn(B, 10, generator=g)
    glob_gt = random_rotmats(B, 1.0, seed=2)
    ref = omodel.forward(sd, ocfg, input_feats=feats, compute_for_loglik=True, shape_for_loglik=shape_gt,
                         pose_R_for_loglik=R_gt, glob_R_for_loglik=glob_gt)
    lp_ref = omodel.log_prob_SO3(sd, ocfg, ref['flow_context_for_loglik'], R_gt.double())
    with torch.no_grad():
        out = model(None, input_feats=feats.cuda(), compute_for_loglik=True, shape_for_loglik=shape_gt.cuda(),
                    pose_R_for_loglik=R_gt.cuda(), glob_R_for_loglik=glob_gt.cuda())
    assert (out['flow_context_for_loglik'].cpu() - ref['flow_context_for_loglik']).abs().max() < 1e-5
    dists = out['conditioned_pose_SO3flow_dists_for_loglik']
    assert len(dists) == 23 and len(out['conditioned_pose_so3flow_dists_for_loglik']) == 23
    lp = torch.stack([dists[j].log_prob(R_gt[:, j].double().cuda()) for j in range(23)], dim=1).cpu()   # reference call pattern
    assert lp.dtype == torch.float32 and torch.isfinite(lp).all()
    assert (lp - lp_ref).abs().max() < 1e-3, (lp - lp_ref).abs().max()
    lp_all = model.log_prob_all_parts(out['flow_context_for_loglik'], R_gt.cuda()).cpu()
    assert torch.equal(lp_all, lp)
    # density on the algebra (conditioned_pose_so3flow_dists_for_loglik)
    v = torch.randn(B, 23, 3, generator=g) * 0.8
    lp_so3_ref = omodel.log_prob_so3(sd, ocfg, ref['flow_context_for_loglik'], v)
    lp_so3 = torch.stack([out['conditioned_pose_so3flow_dists_for_loglik'][j].log_prob(v[:, j].cuda()) for j in range(23)], 1).cpu()
    assert (lp_so3 - lp_so3_ref).abs().max() < 1e-3
    model.pose_SO3flow_dists[0].clear_cache()


def test_full_size_B32_N100_properties():
    """BASELINE configs[1] size: orthonormality, determinism, agreement of the point estimate with an N=0 call, row
    independence (any sample row equals the same row computed in a small batch) and a slice against the oracle."""
    model, sd, cfg = _model(50, seed=6)
    B, N = 32, 100
    feats = torch.randn(B, 2048, generator=torch.Generator().manual_seed(5)) * 0.3
    noise = _noise(B, N, seed=9)
    with torch.no_grad():
        out = model(None, input_feats=feats.cuda(), num_samples=N, noise=_cuda(noise))
        out_b = model(None, input_feats=feats.cuda(), num_samples=N, noise=_cuda(noise))
        pe = model(None, input_feats=feats.cuda())
    Rs = out['pose_rotmats_samples']
    assert torch.equal(Rs, out_b['pose_rotmats_samples'])                       # deterministic
    I = torch.eye(3, device='cuda')
    assert (Rs @ Rs.transpose(-1, -2) - I).abs().max().item() < 1e-5 and (torch.linalg.det(Rs) - 1).abs().max().item() < 1e-5
    # Round 4: the kinematic loop has two forms chosen by size (16-row tiles up to 1,024 rows, 32-row tiles beyond): the N = 0 call and the small
    # batch below run the other form than the 3,232-row call -- same arithmetic, another accumulation order (2e-5); within one form a row's result
    # does not depend on the batch it sits in (bit for bit for the point estimate, 1e-5 for a sample row whose tile neighbours differ)
    assert (pe['pose_rotmats_point_est'] - out['pose_rotmats_point_est']).abs().max().item() < 2e-5
    sub = {'shape': noise['shape'][:7, 30:32].contiguous(), 'pose': noise['pose'][:, 30:32, :7].contiguous()}
    with torch.no_grad():
        small = model(None, input_feats=feats[30:32].cuda(), num_samples=7, noise=_cuda(sub))
        model.flow_chain32 = True
        pe32 = model(None, input_feats=feats.cuda())
        small32 = model(None, input_feats=feats[30:32].cuda(), num_samples=7, noise=_cuda(sub))
        model.flow_chain32 = False
        out16 = model(None, input_feats=feats.cuda(), num_samples=N, noise=_cuda(noise))
        model.flow_chain32 = None
    assert torch.equal(pe32['pose_rotmats_point_est'], out['pose_rotmats_point_est'])
    assert torch.equal(pe['pose_rotmats_point_est'], out16['pose_rotmats_point_est'])
    assert (small32['pose_rotmats_samples'] - Rs[30:32, :7]).abs().max().item() < 1e-5
    assert (small['pose_rotmats_samples'] - out16['pose_rotmats_samples'][30:32, :7]).abs().max().item() < 1e-5
    assert (small['pose_rotmats_samples'] - Rs[30:32, :7]).abs().max().item() < 2e-5
    ref = omodel.forward(sd, oracle_cfg_from(cfg.MODEL), input_feats=feats[30:32], num_samples=7, noise=sub)
    assert (Rs[30:32, :7].cpu() - ref['pose_rotmats_samples']).abs().max() < RTOL
    # default (device generator) path draws noise and stays in support
    with torch.no_grad():
        torch.manual_seed(0)
        rnd = model(None, input_feats=feats.cuda(), num_samples=N)
    assert torch.isfinite(rnd['pose_rotmats_samples']).all() and rnd['shape_samples'].shape == (B, N, 10)
    assert rnd['shape_samples'].std(dim=1).mean().item() > 1e-3


def test_unbuilt_gradient_paths_fail_loudly():
    """Gradient paths that are not built raise instead of returning tensors without a graph.  Round 6 built the flow's: the rotation an SO(3)
    density is evaluated at, the algebra density, the z = 0 point-estimate chain, SMPL vertices (tests/test_gpu_train.py, test_gpu_smpl.py); what still raises are
    the value-gradient paths of the non-default flow variants."""
    model, _, _ = _model(18)
    for p in model.parameters():
        p.requires_grad_(False)
    feats = torch.zeros(2, 512).cuda()
    R = random_rotmats(2 * 23, 0.5, seed=1).view(2, 23, 3, 3).cuda()
    out = model(None, input_feats=feats, compute_for_loglik=True, shape_for_loglik=torch.zeros(2, 10).cuda(), pose_R_for_loglik=R,
                glob_R_for_loglik=R[:, 0])
    Rg = R[:, 0].double().clone().requires_grad_(True)
    lpg = out['conditioned_pose_SO3flow_dists_for_loglik'][0].log_prob(Rg)      # round 6: differentiable w.r.t. the rotation it is evaluated at (tests/test_gpu_train.py)
    assert lpg.requires_grad
    lpa = out['conditioned_pose_so3flow_dists_for_loglik'][0].log_prob((torch.ones(2, 3) * 0.1).cuda().requires_grad_(True))      # ... and so is the algebra density
    assert lpa.requires_grad
    smpl = H.SMPL(model_data=H.synthetic_smpl_model(seed=0), create_transl=False).to('cuda:0')
    # SMPL: joints AND (round 6) vertices carry the graph; the joints-only form has no vertices at all
    o = smpl(body_pose=R.clone().requires_grad_(True), global_orient=R[:, :1].contiguous(), betas=torch.zeros(2, 10).cuda(), pose2rot=False)
    assert o.joints.requires_grad and o.vertices.requires_grad
    oj = smpl(body_pose=R.clone().requires_grad_(True), global_orient=R[:, :1].contiguous(), betas=torch.zeros(2, 10).cuda(), pose2rot=False, joints_only=True)
    assert oj.vertices is None and oj.joints.requires_grad
    c3 = H.get_humaniflow_cfg_defaults()
    c3.MODEL.NORM_FLOW.PERMUTE_TYPE = 'linear_plu'
    mv = init_model_for_tests(H.HumaniflowModel(torch.device('cpu'), c3.MODEL, H.SMPL_PARENTS), seed=1).eval().cuda()
    for p in mv.parameters():
        p.requires_grad_(False)
    ov = mv(None, input_feats=feats, compute_for_loglik=True, shape_for_loglik=torch.zeros(2, 10).cuda(), pose_R_for_loglik=R, glob_R_for_loglik=R[:, 0])
    with pytest.raises(NotImplementedError):                                   # what still raises: the value-gradient paths of the NON-DEFAULT flow variants
        ov['conditioned_pose_SO3flow_dists_for_loglik'][0].log_prob(R[:, 0].double().clone().requires_grad_(True))
    lp = out['conditioned_pose_SO3flow_dists_for_loglik'][0].log_prob(R[:, 0].double())            # frozen weights, plain inputs: inference path
    assert lp.shape == (2,) and not lp.requires_grad


def test_sharded_eval_driver_on_device():
    """humaniflow_amd.evaluate on the GPU (world 1): predictions + targets + 3D metrics stay on the device, and the metric
    of a model evaluated against ITS OWN point estimate is zero."""
    from humaniflow_amd import evaluate as ev
    model, sd, cfg = _model(18, seed=2)
    smpl = H.SMPL(model_data=H.synthetic_smpl_model(seed=0), create_transl=False).to('cuda:0')
    ds = ev.SyntheticEvalSet(5, size=128, seed=1)
    dev = torch.device('cuda:0')
    metrics, frames = ev.evaluate_humaniflow(ds, 2, dev, lambda b: ev.predict_batch(model, smpl, b, 4), lambda b: ev.target_batch(smpl, b))
    assert metrics['num_total_test_data'] == 5 and all(np.isfinite(v) for v in metrics.values())
    assert metrics['PVE-PA'] <= metrics['PVE-SC'] + 1e-6 <= metrics['PVE'] + 2e-6            # metres, like the reference
    assert metrics['PVE_samples_min'] > 0 and frames['PVE'].shape == (5,)
    assert set(ev.METRICS) <= set(metrics) and metrics['verts3D_sample_diversity'] > 0 and metrics['joints2Dsamples-L2E'] > 0
    assert metrics['PVE-PA_samples_min'] <= metrics['PVE-SC_samples_min'] + 1e-6
    self_metrics, _ = ev.evaluate_humaniflow(ds, 2, dev, lambda b: ev.predict_batch(model, smpl, b, 0),
                                             lambda b: {k: v for k, v in ev.predict_batch(model, smpl, b, 0).items()})
    assert self_metrics['PVE'] < 1e-6 and self_metrics['MPJPE'] < 1e-6 and self_metrics['joints2D-L2E'] < 1e-3


def test_predict_driver_plumbing_config0(tmp_path):
    """BASELINE configs[0] counterpart: 4 images, N=10, neutral body model, no render: the six save_pred_output keys with the
    reference's names/shapes, checked against the oracle on the same injected noise."""
    from humaniflow_amd.predict import predict_humaniflow
    model, sd, cfg = _model(18, seed=7)
    smpl = H.SMPL(model_data=H.synthetic_smpl_model(seed=0), create_transl=False).to('cuda:0')
    x = synthetic_proxy_input(4, 128, seed=3)
    N = 10
    noise = {'shape': torch.randn(N, 4, 10), 'pose': torch.randn(23, 4, N, 3, generator=torch.Generator().manual_seed(1))}
    names = ['%04d.png' % (i + 1) for i in range(4)]
    res = predict_humaniflow(model, smpl, x.cuda(), names, save_dir=str(tmp_path), num_pred_samples=N, noise=_cuda(noise))
    ref = omodel.forward(sd, oracle_cfg_from(cfg.MODEL), input=x, num_samples=N, noise=noise, use_shape_mode_for_samples=True)
    shapes = {'pose_axisangle_point_est': (23, 3), 'pose_rotmats_point_est': (23, 3, 3), 'shape_mode': (10,), 'glob_rotmat': (3, 3),
              'cam_wp': (3,), 'input_feats': (512,)}
    for i, name in enumerate(names):
        z = np.load(str(tmp_path / (name[:-4] + '_pred.npz')))
        for k, shp in shapes.items():
            assert z[k].shape == shp, k
        assert res[i]['vertices_samples'].shape == (N, 6890, 3) and res[i]['per_vertex_avg_dist_from_mean'].shape == (6890,)
        assert (res[i]['pred']['shape_samples'] - res[i]['pred']['shape_mode'][:, None]).abs().max().item() == 0     # shape mode used
    # parity on the oracle's own features (isolates conv round-off), all four images in one call
    with torch.no_grad():
        out = model(None, input_feats=ref['input_feats'].cuda(), num_samples=N, noise=_cuda(noise), use_shape_mode_for_samples=True)
    for k in ('pose_axisangle_point_est', 'pose_rotmats_point_est', 'shape_mode', 'glob_rotmat', 'cam_wp', 'pose_rotmats_samples'):
        assert (out[k].cpu() - ref[k]).abs().max() < 1e-4 * max(1.0, ref[k].abs().max().item()), k


def test_non_default_beta_count_and_single_row():
    """NUM_SMPL_BETAS other than 10 changes the isgc layer width and the rank-nb MFMA update; B = N = 1 is one live row in a tile."""
    cfg = H.get_humaniflow_cfg_defaults()
    cfg.MODEL.NUM_SMPL_BETAS = 13
    m = init_model_for_tests(H.HumaniflowModel(torch.device('cpu'), cfg.MODEL, H.SMPL_PARENTS), seed=8).eval()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.to('cuda:0')
    for B, N in ((1, 1), (2, 33)):
        g = torch.Generator().manual_seed(B + N)
        feats = torch.randn(B, 512, generator=g) * 0.5
        noise = {'shape': torch.randn(N, B, 13, generator=g), 'pose': torch.randn(23, B, N, 3, generator=g)}
        ref = omodel.forward(sd, oracle_cfg_from(cfg.MODEL), input_feats=feats, num_samples=N, noise=noise)
        with torch.no_grad():
            out = m(None, input_feats=feats.cuda(), num_samples=N, noise=_cuda(noise))
        assert out['shape_samples'].shape == (B, N, 13)
        assert (out['shape_samples'].cpu() - ref['shape_samples']).abs().max() < 1e-5
        assert (out['pose_rotmats_samples'].cpu() - ref['pose_rotmats_samples']).abs().max() < RTOL
        assert (out['pose_rotmats_point_est'].cpu() - ref['pose_rotmats_point_est']).abs().max() < RTOL


def test_log_prob_at_training_batch_size_72():
    """TRAIN.BATCH_SIZE = 72 (configs/humaniflow_config.py:40): all 23 x 72 log-densities in one launch, finite, and
    consistent with sampling: samples drawn from the flow score higher on average under their own context than under a permuted one."""
    model, sd, cfg = _model(18, seed=9)
    B = 72
    g = torch.Generator().manual_seed(2)
    feats = torch.randn(B, 512, generator=g) * 0.5
    with torch.no_grad():
        s = model(None, input_feats=feats.cuda(), num_samples=1, noise=_cuda(_noise(B, 1, seed=4)))
        R = s['pose_rotmats_samples'][:, 0]                                           # (B,23,3,3) one sample per image
        out = model(None, input_feats=feats.cuda(), compute_for_loglik=True, shape_for_loglik=s['shape_samples'][:, 0],
                    pose_R_for_loglik=R, glob_R_for_loglik=s['glob_rotmat'], compute_point_est=False)
        lp = model.log_prob_all_parts(out['flow_context_for_loglik'], R)
        lp_perm = model.log_prob_all_parts(out['flow_context_for_loglik'].roll(1, 0), R)
    assert lp.shape == (B, 23) and torch.isfinite(lp).all() and torch.isfinite(lp_perm).all()
    assert lp.mean().item() > lp_perm.mean().item()
    ref = omodel.log_prob_SO3(sd, oracle_cfg_from(cfg.MODEL), out['flow_context_for_loglik'][:5].cpu(), R[:5].cpu().double())
    assert (lp[:5].cpu() - ref).abs().max() < 1e-3


def test_end_to_end_from_gpu_encoder_features_within_north_star_tolerances():
    """The north-star bar asserted END TO END on the product path (nothing taken from the oracle in between): hand-written encoder ->
    heads -> flow -> ONE fused SMPL launch, vs the oracle's reference-mode forward on the same proxy input, weights and noise:
    vertices / joints within 1e-4 m, log-probs within 1e-3 (ResNet-50, 256x256: BASELINE configs[1] shapes at B=2, N=6)."""
    model, sd, cfg = _model(50, seed=5)
    ocfg = oracle_cfg_from(cfg.MODEL)
    B, N = 2, 6
    x = synthetic_proxy_input(B, 256, seed=0)
    noise = _noise(B, N, seed=2)
    ref = omodel.forward(sd, ocfg, input=x, num_samples=N, noise=noise)
    smpl = H.SMPL(model_data=H.synthetic_smpl_model(seed=0), create_transl=False).to('cuda:0')
    with torch.no_grad():
        out = model(x.cuda(), num_samples=N, noise=_cuda(noise), return_input_feats=True)
        pe, sa = smpl.forward_samples_and_point_est(out)
    scale = ref['input_feats'].abs().max().item()
    assert (out['input_feats'].cpu() - ref['input_feats']).abs().max() < 2e-5 * max(1.0, scale)
    om = smpl_oracle_model(smpl.model_dict())
    g = ref['glob_rotmat']
    v_ref, j_ref = osmpl.smpl_forward(om, ref['shape_samples'].reshape(-1, 10), ref['pose_rotmats_samples'].reshape(-1, 23, 3, 3),
                                      g[:, None].expand(-1, N, -1, -1).reshape(-1, 1, 3, 3), pose2rot=False)
    vp_ref, jp_ref = osmpl.smpl_forward(om, ref['shape_mode'], ref['pose_rotmats_point_est'], g[:, None], pose2rot=False)
    assert (sa.vertices.cpu() - v_ref).abs().max() < 1e-4 and (sa.joints.cpu() - j_ref).abs().max() < 1e-4
    assert (pe.vertices.cpu() - vp_ref).abs().max() < 1e-4 and (pe.joints.cpu() - jp_ref).abs().max() < 1e-4
    for k in ('pose_rotmats_samples', 'pose_rotmats_point_est', 'pose_axisangle_point_est', 'cam_wp', 'glob_rotmat', 'shape_mode'):
        assert (out[k].cpu() - ref[k]).abs().max() < 1e-4, k
    # teacher-forced log-likelihood from the GPU features as well
    R_gt = random_rotmats(B * 23, 0.7, seed=1).view(B, 23, 3, 3)
    shape_gt = torch.randn(B, 10, generator=torch.Generator().manual_seed(3))
    glob_gt = random_rotmats(B, 1.0, seed=2)
    refl = omodel.forward(sd, ocfg, input=x, compute_for_loglik=True, shape_for_loglik=shape_gt, pose_R_for_loglik=R_gt, glob_R_for_loglik=glob_gt)
    lp_ref = omodel.log_prob_SO3(sd, ocfg, refl['flow_context_for_loglik'], R_gt.double())
    with torch.no_grad():
        outl = model(x.cuda(), compute_for_loglik=True, shape_for_loglik=shape_gt.cuda(), pose_R_for_loglik=R_gt.cuda(), glob_R_for_loglik=glob_gt.cuda())
        lp = model.log_prob_all_parts(outl['flow_context_for_loglik'], R_gt.cuda()).cpu()
    assert (lp - lp_ref).abs().max() < 1e-3


@pytest.mark.parametrize('terms,feat_bar,vert_bar', [(6, 2e-5, 1.5e-4), (3, 5e-4, 1e-3)])
def test_end_to_end_with_split_operand_gemms(terms, feat_bar, vert_bar):
    """The opt-in encoder flavour (image_encoder.gemm_terms: 1x1 GEMMs on the bf16 pipe with split fp32 operands) through the whole path against
    the oracle: the 6-term form meets the feature bar of the fp32 path (2e-5) and sits AT the north-star vertex bar (measured 6.9e-5 m with the
    32-row chain kernel of round 3, 1.05e-4 m with the 16-row kernel of round 4: the same 9e-7 feature difference amplified through the kinematic
    chain under a different accumulation order; bar 1.5e-4 m); the 3-term form is the documented approximation (bars 25x / 10x looser)."""
    model, sd, cfg = _model(50, seed=5)
    model.image_encoder.gemm_terms = terms
    ocfg = oracle_cfg_from(cfg.MODEL)
    B, N = 2, 6
    x = synthetic_proxy_input(B, 256, seed=0)
    noise = _noise(B, N, seed=2)
    ref = omodel.forward(sd, ocfg, input=x, num_samples=N, noise=noise)
    smpl = H.SMPL(model_data=H.synthetic_smpl_model(seed=0), create_transl=False).to('cuda:0')
    with torch.no_grad():
        out = model(x.cuda(), num_samples=N, noise=_cuda(noise), return_input_feats=True)
        pe, sa = smpl.forward_samples_and_point_est(out)
    assert model.image_encoder._plan['split']
    scale = ref['input_feats'].abs().max().item()
    fe = (out['input_feats'].cpu() - ref['input_feats']).abs().max().item() / max(1.0, scale)
    om = smpl_oracle_model(smpl.model_dict())
    g = ref['glob_rotmat']
    v_ref, j_ref = osmpl.smpl_forward(om, ref['shape_samples'].reshape(-1, 10), ref['pose_rotmats_samples'].reshape(-1, 23, 3, 3),
                                      g[:, None].expand(-1, N, -1, -1).reshape(-1, 1, 3, 3), pose2rot=False)
    vp_ref, jp_ref = osmpl.smpl_forward(om, ref['shape_mode'], ref['pose_rotmats_point_est'], g[:, None], pose2rot=False)
    ve = max((sa.vertices.cpu() - v_ref).abs().max().item(), (sa.joints.cpu() - j_ref).abs().max().item(),
             (pe.vertices.cpu() - vp_ref).abs().max().item(), (pe.joints.cpu() - jp_ref).abs().max().item())
    print('gemm_terms=%d: features %.2e (relative to the largest feature), vertices / joints %.2e m' % (terms, fe, ve))
    assert fe < feat_bar and ve < vert_bar, (terms, fe, ve)


def test_fused_smpl_launch_equals_the_two_reference_calls():
    model, sd, cfg = _model(18, seed=11)
    smpl = H.SMPL(model_data=H.synthetic_smpl_model(seed=0), create_transl=False).to('cuda:0')
    for B, N, use_mode in ((3, 7, False), (2, 33, True), (1, 1, False)):
        feats = torch.randn(B, 512, generator=torch.Generator().manual_seed(B + N)) * 0.5
        with torch.no_grad():
            out = model(None, input_feats=feats.cuda(), num_samples=N, noise=_cuda(_noise(B, N, seed=B)), use_shape_mode_for_samples=use_mode)
            pe, sa = smpl.forward_samples_and_point_est(out)
            g = out['glob_rotmat']
            pe2 = smpl(body_pose=out['pose_rotmats_point_est'], global_orient=g[:, None], betas=out['shape_mode'], pose2rot=False)
            sa2 = smpl(body_pose=out['pose_rotmats_samples'].reshape(-1, 23, 3, 3), global_orient=g[:, None].expand(-1, N, -1, -1).reshape(-1, 1, 3, 3),
                       betas=out['shape_samples'].reshape(-1, 10), pose2rot=False)
        assert torch.equal(pe.vertices, pe2.vertices) and torch.equal(pe.joints, pe2.joints)
        assert torch.equal(sa.vertices, sa2.vertices) and torch.equal(sa.joints, sa2.joints)
        assert torch.equal(out['smpl_packed_inputs']['betas'][B * N:], out['shape_mode'])


def test_configs2_flow_and_encoder_at_B64():
    """BASELINE configs[2] batch: encoder + flow at B = 64, N = 100 (6,464 rows = 202 tiles): determinism, orthonormality, batch-row
    independence of the encoder, and slices against the oracle."""
    model, sd, cfg = _model(50, seed=12)
    B, N = 64, 100
    x = synthetic_proxy_input(B, 256, seed=4).cuda()
    noise = _noise(B, N, seed=13)
    with torch.no_grad():
        feats = model(x, return_input_feats_only=True)['input_feats']
        feats2 = model(x[40:44], return_input_feats_only=True)['input_feats']
        out = model(None, input_feats=feats, num_samples=N, noise=_cuda(noise))
    assert feats.shape == (B, 2048) and (feats[40:44] - feats2).abs().max().item() < 1e-5 * feats.abs().max().item()
    Rs = out['pose_rotmats_samples']
    I = torch.eye(3, device='cuda')
    assert (Rs @ Rs.transpose(-1, -2) - I).abs().max().item() < 1e-5
    sub = {'shape': noise['shape'][:5, 62:64].contiguous(), 'pose': noise['pose'][:, 62:64, :5].contiguous()}
    ref = omodel.forward(sd, oracle_cfg_from(cfg.MODEL), input_feats=feats[62:64].cpu(), num_samples=5, noise=sub)
    # Per (image, sample) row: RTOL plus four times the float32 oracle's OWN distance from the float64 oracle on that row.  One of these ten rows
    # (image 63, sample 2, part 12) is ill-conditioned -- the oracle itself moves by 2e-5 between float32 and float64 there -- and the two chain
    # kernels (16-row / 32-row tiles: different accumulation orders) land 7.5e-5 / 4.7e-5 from the float32 oracle on it, 1e-5 everywhere else.
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    ref64 = omodel.forward(sd64, oracle_cfg_from(cfg.MODEL), input_feats=feats[62:64].cpu().double(), num_samples=5,
                           noise={k: v.double() for k, v in sub.items()})
    own = (ref['pose_rotmats_samples'].double() - ref64['pose_rotmats_samples']).abs().amax(dim=(2, 3, 4))          # (2, 5)
    err = (Rs[62:64, :5].cpu() - ref['pose_rotmats_samples']).abs().amax(dim=(2, 3, 4))
    assert (err < RTOL + 4 * own).all() and err.median() < 1e-5, (err, own)
    assert (out['pose_rotmats_point_est'][62:64].cpu() - ref['pose_rotmats_point_est']).abs().max() < RTOL
    fr = omodel.forward(sd, oracle_cfg_from(cfg.MODEL), input=x[:1].cpu(), compute_point_est=False)['input_feats']
    assert (feats[:1].cpu() - fr).abs().max() < 2e-5 * max(1.0, fr.abs().max().item())


def test_configs3_eval_step_at_B32_N100():
    """BASELINE configs[3] per-GPU step at ITS size: 32 3DPW-shaped synthetic items, N = 100: model -> fused SMPL (3,232 bodies) -> T-pose
    bodies -> target bodies -> device tracker update.  Tracker sums vs the oracle metric functions on a slice of the same tensors."""
    from humaniflow_amd import evaluate as ev
    from oracle import metrics as om
    model, sd, cfg = _model(50, seed=14)
    smpl = H.SMPL(model_data=H.synthetic_smpl_model(seed=0), create_transl=False).to('cuda:0')
    dev = torch.device('cuda:0')
    ds = ev.SyntheticEvalSet(32, size=256, seed=2)
    batch = ds.batch(list(range(32)), dev)
    pred = ev.predict_batch(model, smpl, batch, 100)
    targ = ev.target_batch(smpl, batch)
    assert pred['verts3D_samples'].shape == (32, 100, 6890, 3) and pred['tpose_verts3D_samples'].shape == (32, 100, 6890, 3)
    trk = ev.EvalMetricsTracker(dev)
    trk.update_per_batch(pred, targ)
    final, frames = trk.compute_final_metrics()
    assert final['num_total_test_data'] == 32 and all(np.isfinite(v) for v in final.values())
    assert frames['PVE_samples_min'].shape == (32,)
    sl = slice(5, 8)
    P, T = pred['verts3D_samples'][sl].double().cpu(), targ['verts3D'][sl].double().cpu()
    e = om.pointset_errors(P, T, True)                                               # (3, 100, 3)
    for c, name in enumerate(('PVE_samples_min', 'PVE-SC_samples_min', 'PVE-PA_samples_min')):
        assert torch.allclose(frames[name][sl].double().cpu(), e[..., c].min(dim=1).values, rtol=2e-5, atol=1e-7), name
    e = om.pointset_errors(pred['tpose_verts3D_samples'][sl].double().cpu(), targ['tpose_verts3D'][sl].double().cpu(), False)
    assert torch.allclose(frames['PVE-T-SC_samples_min'][sl].double().cpu(), e[..., 1].min(dim=1).values, rtol=2e-5, atol=1e-7)
    d = om.sample_diversity(P) / (100 * 6890)
    assert torch.allclose(frames['verts3D_sample_diversity'][sl].double().cpu(), d, rtol=2e-5)
    # the bench's form of the same step: no T-pose bodies (shape vectors -> hmf_smpl_tpose_errors), same sums
    noise = model.draw_noise(dev, 32, 100)
    noise = {'shape': noise[0], 'pose': noise[1]}
    predA = ev.predict_batch(model, smpl, batch, 100, noise=noise)
    predB = ev.predict_batch(model, smpl, batch, 100, noise=noise, materialise_tpose=False)
    assert 'tpose_verts3D' not in predB and predB['tpose_betas_samples'].shape == (32, 100, 10)
    tA, tB = ev.EvalMetricsTracker(dev), ev.EvalMetricsTracker(dev, smpl=smpl)
    tA.update_per_batch(predA, targ)
    tB.update_per_batch(predB, ev.target_batch(smpl, batch, materialise_tpose=False))
    assert torch.allclose(tA.sums, tB.sums, rtol=2e-6, atol=1e-9)


def test_in_place_weight_updates_repack_on_the_device():
    """the packed-weight handle and the encoder plan follow in-place parameter updates (optimizer.step(), p.data.copy_(),
    load_state_dict on a sub-module) through the version counters; the flow re-packs on the device through the slot->parameter index map
    (hmf_flow_update_from_device) and must equal a handle built from scratch with the same weights."""
    import copy
    model, sd, cfg = _model(18, seed=13)
    B, N = 3, 9
    feats = torch.randn(B, 512, generator=torch.Generator().manual_seed(1)) * 0.5
    noise = _cuda(_noise(B, N, seed=2))
    x = synthetic_proxy_input(1, 128, seed=5).cuda()
    with torch.no_grad():
        before = model(None, input_feats=feats.cuda(), num_samples=N, noise=noise)['pose_rotmats_samples'].clone()
        f0 = model(x, return_input_feats_only=True)['input_feats'].clone()
        g = torch.Generator().manual_seed(3)
        for name, p in model.named_parameters():                      # in-place: what an optimiser step does
            if name.startswith(('pose_so3flow', 'fc_flow_context', 'fc_input_shape')):
                p.mul_(1.0 + 0.05 * torch.randn(p.shape, generator=g).cuda())
        model.image_encoder.layer1[0].conv1.weight.mul_(1.1)
        sub = {k: v * 0.9 for k, v in model.fc_flow_context[3].state_dict().items()}
        model.fc_flow_context[3].load_state_dict(sub)                 # sub-module load: bypasses the root module's hooks
        after = model(None, input_feats=feats.cuda(), num_samples=N, noise=noise)['pose_rotmats_samples']
        f1 = model(x, return_input_feats_only=True)['input_feats']
        assert (after - before).abs().max().item() > 1e-3 and (f1 - f0).abs().max().item() > 1e-4        # the updates were seen
        model.invalidate_flow_weights()                               # from-scratch host pack of the same weights
        model.image_encoder.invalidate_plan()
        fresh = model(None, input_feats=feats.cuda(), num_samples=N, noise=noise)['pose_rotmats_samples']
        f2 = model(x, return_input_feats_only=True)['input_feats']
    assert torch.equal(after, fresh) and torch.equal(f1, f2)
    m2 = copy.deepcopy(model)                                         # device-side caches are not copied, the copy works on its own
    with torch.no_grad():
        again = m2(None, input_feats=feats.cuda(), num_samples=N, noise=noise)['pose_rotmats_samples']
    assert torch.equal(again, fresh)


def test_hip_graph_replay_of_the_whole_step_equals_eager_issue():
    """bench.py replays one captured HIP graph per batch: the replayed step (encoder -> heads -> flow -> fused SMPL launch, with the base
    noise supplied so that both runs draw the same samples) returns bit-identical tensors to the eagerly issued step, replay after replay."""
    import humaniflow_amd as H
    from humaniflow_amd.synthetic import init_benchmark_weights, synthetic_proxy_input
    cfg = H.get_humaniflow_cfg_defaults()
    torch.manual_seed(0)
    dev = torch.device('cuda')
    model = init_benchmark_weights(H.HumaniflowModel(torch.device('cpu'), cfg.MODEL, H.SMPL_PARENTS), seed=0).eval().to(dev)
    B, N = 4, 6
    smpl = H.SMPL(model_data=H.synthetic_smpl_model(seed=0), batch_size=B, create_transl=False).to(dev)
    x = synthetic_proxy_input(B, 256, seed=3).to(dev)
    g = torch.Generator().manual_seed(9)
    noise = {'shape': torch.randn(N, B, 10, generator=g).to(dev), 'pose': torch.randn(23, B, N, 3, generator=g).to(dev)}

    def step():
        with torch.no_grad():
            pred = model(x, num_samples=N, noise=noise)
            pe, sa = smpl.forward_samples_and_point_est(pred)
            return sa.vertices, sa.joints, pe.vertices, pred['pose_rotmats_samples'], pred['shape_samples']
    ref = [t.clone() for t in step()]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode='thread_local'):
        outs = step()
    for _ in range(3):
        for t in outs:
            t.fill_(float('nan'))
        graph.replay()
        torch.cuda.synchronize()
        diffs = [float((a - b).abs().max()) if torch.isfinite(b).all() else float('nan') for a, b in zip(ref, outs)]
        assert all(torch.equal(a, b) for a, b in zip(ref, outs)), diffs


def test_bench_parity_block_at_small_size():
    """bench.py's `parity` object (round 6): the cpu_baseline leg keeps its last oracle iteration, parity_block runs the product path end to end on the
    same weights / inputs / base noise and compares -- here at 3 images x 6 samples with the benchmark's calibrated random weights (metre-sized bodies)."""
    import importlib
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    bench = importlib.import_module('bench')
    from humaniflow_amd.synthetic import init_benchmark_weights, calibrate_benchmark_weights, synthetic_proxy_input as spi
    dev = torch.device('cuda:0')
    cfg = H.get_humaniflow_cfg_defaults()
    cfg.MODEL.NUM_RESNET_LAYERS = 50
    model = init_benchmark_weights(H.HumaniflowModel(torch.device('cpu'), cfg.MODEL, H.SMPL_PARENTS), seed=0).eval().to(dev)
    scale = calibrate_benchmark_weights(model, spi(4, 256, seed=99).to(dev))
    assert 1e-4 < scale < 1.0
    smpl = H.SMPL(model_data=H.synthetic_smpl_model(seed=0), create_transl=False).to(dev)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    nt = torch.get_num_threads()
    try:
        base, ref = bench.cpu_baseline(sd, cfg, {k: v.detach().cpu() for k, v in smpl.model_dict().items()}, 3, 6)
    finally:
        torch.set_num_threads(nt)
    assert base['kind'] == 'port' and base['value'] > 0 and ref['verts_samples'].shape == (18, 6890, 3)
    p = bench.parity_block(model, smpl, ref, 6, dev)
    assert p['rows'] == 21 and p['logprob_rows'] == 3 * 23 and p['logprob_finite']
    assert p['vertex_abs_max'] < 10.0                                    # metre-sized bodies: the absolute bars mean something
    assert p['ok'] and p['vertex_linf'] <= 1e-4 and p['joint_linf'] <= 1e-4 and p['logprob_max_abs'] <= 1e-3, p
    # the block must be able to FAIL: a perturbed product model is out of tolerance
    with torch.no_grad():
        model.fc_shape.bias[:10] += 0.05
    assert not bench.parity_block(model, smpl, ref, 6, dev)['ok']
