"""Pin the oracle against the golden vectors generated from the reference's own Python
(tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, deterministic_state_dict
from oracle import so3, flow
from oracle.resnet import resnet_forward


def _g(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def test_so3_exp_matches_reference():
    z = _g('so3_golden.npz')
    R = so3.so3_exp(torch.from_numpy(z['v']))
    assert np.abs(R.numpy() - z['R']).max() < 1e-14


def test_so3_log_matches_reference_incl_near_pi():
    z = _g('so3_golden.npz')
    v = so3.so3_vee(so3.so3_log(torch.from_numpy(z['R_log_in'])))
    assert np.abs(v.numpy() - z['log_v']).max() < 1e-12


def test_so3_xset_and_logdet_match_reference():
    z = _g('so3_golden.npz')
    v = torch.from_numpy(z['v'])
    xs = so3.so3_xset(v, 1).numpy()
    assert np.array_equal(np.isnan(xs), np.isnan(z['xset']))            # quirk 2: NaN at v == 0
    assert np.nanmax(np.abs(xs - z['xset'])) < 1e-13
    ld = so3.so3_log_abs_det_jacobian(v).numpy()
    assert np.abs(ld - z['logdet']).max() < 1e-13
    assert abs(ld[64] - np.log(11.0 / 12.0)) < 1e-15                     # quirk 1 at v == 0


def test_rot6d_matches_reference():
    z = _g('so3_golden.npz')
    R = so3.rot6d_to_rotmat(torch.from_numpy(z['x6']))
    assert np.abs(R.numpy() - z['R6']).max() < 1e-6
    r6 = so3.rotmat_to_rot6d(torch.from_numpy(z['R'][:8]).float())
    assert np.abs(r6.numpy() - z['rot6d_of_R']).max() == 0


def test_radial_tanh_matches_reference():
    z = _g('flow_golden.npz')
    r = float(z['radius'])
    assert np.abs(flow.radial_tanh_forward(torch.from_numpy(z['xs']), r).numpy() - z['ys']).max() < 1e-6
    xinv = flow.radial_tanh_inverse(torch.from_numpy(z['ys_in']), r)
    assert np.abs(xinv.numpy() - z['xinv']).max() < 1e-5
    ld = flow.radial_tanh_log_abs_det(xinv, torch.from_numpy(z['ys_in']), r)
    assert np.abs(ld.numpy() - z['ld']).max() < 1e-5


def test_pushforward_log_prob_plumbing_matches_reference():
    """LocalDiffeoTransformedDistribution.log_prob with a radial-tanh-only base: pins log-map, pre-image set, masks,
    exp log-det, dtype hops and the logsumexp over pre-images."""
    z = _g('flow_golden.npz')
    r = float(z['radius'])
    Rq = torch.from_numpy(z['Rq'])
    x, xset, mask = so3.so3_inverse_set(Rq, r)
    assert np.abs(x.numpy() - z['inv_x']).max() < 1e-12
    assert np.array_equal(mask.numpy(), z['inv_mask'])
    assert np.abs(np.nan_to_num(xset.numpy()) - z['inv_xset']).max() < 1e-12
    pf = flow.PartFlow(couplings=[], radius=r, base_std=0.6)            # no couplings: base -> radial tanh only
    lp = pf.log_prob_SO3(Rq, torch.zeros(Rq.shape[0], 64))
    assert np.abs(lp.numpy() - z['lp']).max() < 2e-5


def test_resnet_matches_reference_and_key_names():
    z = _g('resnet_golden.npz')
    from humaniflow_amd.encoder import ResNetEncoder
    for layers in (50, 18):
        enc = ResNetEncoder(layers, 18)
        keys = sorted(enc.state_dict().keys())
        assert keys == [str(k) for k in z['keys%d' % layers]]          # checkpoint layout == reference's
        sd = deterministic_state_dict({k: tuple(t.shape) for k, t in enc.state_dict().items()}, seed=7)
        x = torch.randn(2, 18, 96, 96, generator=torch.Generator().manual_seed(5))
        y = resnet_forward(sd, x, layers)
        ref = z['feats%d' % layers]
        assert np.abs(y.numpy() - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())
        enc.load_state_dict(sd, strict=True)
        enc.eval()
        with torch.no_grad():
            y2 = enc.forward_module(x)                                   # the module graph the checkpoint layout belongs to (what training runs)
        assert np.abs(y2.numpy() - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())
        with pytest.raises(RuntimeError, match='no CPU fallback'):
            enc(x)                                                       # the product forward has no host path


def test_training_mode_encoder_module_matches_reference_resnet_golden():
    """The module graph the training step differentiates (`ResNetEncoder.forward_module`, train mode: batch-statistics BatchNorm, running
    statistics update) vs the reference's models/resnet.py:202-217 resnet18(in_channels=18).train() run in float64 by
    tests/golden/make_golden_train.py: features, all 60 parameter gradients, all 40 running statistics.  This module in float64 is the
    encoder half of the oracle of the full-size train-step test (tests/test_gpu_train.py); the HIP training graph is held to the same
    golden in tests/test_gpu_encoder_train.py."""
    from helpers import check_encoder_against_train_golden, encoder_train_golden_inputs
    from humaniflow_amd.encoder import ResNetEncoder
    z = _g('encoder_train_golden.npz')
    x, w = encoder_train_golden_inputs()
    for dt, bar in ((torch.float64, 1e-9), (torch.float32, 2e-3)):
        enc = ResNetEncoder(18, 18)
        enc.load_state_dict(deterministic_state_dict({k: tuple(t.shape) for k, t in enc.state_dict().items()}, seed=11), strict=True)
        enc = enc.to(dt).train()
        feats = enc.forward_module(x.to(dt))
        (feats * w.to(dt)).sum().backward()
        # stored in float32: 6e-8 relative; float64 module == float64 reference up to that
        df, worst = check_encoder_against_train_golden(enc, feats, z, grad_bar=max(bar, 2e-7), feat_bar=max(bar, 2e-7), stat_bar=max(bar, 2e-7))


def test_linear_plu_oracle_matches_reference_transform():
    """oracle.flow.plu_forward / plu_inverse / conditional diagonal constraint vs the reference's ConditionedLinearPLUTransform
    (models/norm_flows/transforms/linearPLU_transform.py:12-82, generated by tests/golden/make_golden_train.py)."""
    from oracle import flow as oflow
    z = _g('linear_plu_golden.npz')
    for pi in range(3):
        perm = [int(v) for v in z['p%d_perm' % pi]]
        LU, x, y = torch.from_numpy(z['p%d_LU' % pi]), torch.from_numpy(z['p%d_x' % pi]), torch.from_numpy(z['p%d_y' % pi])
        assert np.abs(oflow.plu_forward(perm, LU, x).numpy() - z['p%d_y' % pi]).max() < 1e-6
        xi, ld = oflow.plu_inverse(perm, LU, y)
        assert np.abs(xi.numpy() - z['p%d_xinv' % pi]).max() < 2e-5 and np.abs(ld.numpy() - z['p%d_logdet' % pi]).max() < 1e-6
    LU = torch.from_numpy(z['u_LU'])
    assert np.abs(oflow.plu_forward([1, 2, 0], LU, torch.from_numpy(z['u_x'])).numpy() - z['u_y']).max() < 1e-6
    xi, ld = oflow.plu_inverse([1, 2, 0], LU, torch.from_numpy(z['u_y']))
    assert np.abs(xi.numpy() - z['u_xinv']).max() < 2e-5 and np.abs(ld.expand(16).numpy() - z['u_logdet']).max() < 1e-6      # negative u_ii: log|u_ii|
    # softplus(beta = 0.75) of the conditioned diagonal: a 0-layer "network" whose bias is the raw value
    raw = torch.from_numpy(z['sp_raw'])
    for r, ref in zip(raw, z['sp_out']):
        LUc = oflow.conditional_lu([(torch.zeros(9, 4), torch.full((9,), float(r)))], torch.zeros(1, 4))
        assert abs(LUc[0, 0, 0].item() - float(ref)) < 1e-5 * max(1.0, abs(float(ref))) and abs(LUc[0, 0, 1].item() - float(r)) < 1e-6
