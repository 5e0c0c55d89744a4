"""Pin the oracle's pyro-ppl 1.7.0 / smplx 0.1.26 restatements (oracle/spline.py, oracle/flow.py, oracle/smpl.py: "parity unpinned") against
golden vectors generated FROM those packages by tests/golden/make_golden_thirdparty.py (or its reference-free twin make_golden_thirdparty_pip.py).  Neither package is installable in the build
container, so tests/golden/thirdparty_golden.npz may be absent: every test here then skips, and turns into a real check the day the file is
generated on a box that has the packages (one command, see the generator's docstring)."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN

PATH = os.path.join(GOLDEN, 'thirdparty_golden.npz')
pytestmark = pytest.mark.skipif(not os.path.exists(PATH), reason='tests/golden/thirdparty_golden.npz not generated (needs pyro-ppl==1.7.0 and '
                                'smplx==0.1.26: python tests/golden/make_golden_thirdparty.py)')
BOUND, BASE_STD = 1.5 * np.pi, 0.6


def _z():
    return np.load(PATH, allow_pickle=False)


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _layers(z, prefix, n=4):
    return [(_t(z['%sw%d' % (prefix, i)]), _t(z['%sb%d' % (prefix, i)])) for i in range(n)]


def test_dense_nn_and_param_split_match_pyro():
    from oracle import spline
    z = _z()
    layers = _layers(z, 'cpl_')
    x, ctx = _t(z['cpl_x']), _t(z['cpl_ctx'])
    raw = spline.dense_nn(layers, x[:, :1], ctx)                                   # ConditionalDenseNN: cat([context, x1]), ReLU between layers
    parts = spline.split_params(raw, 8, 2)
    for i, p in enumerate(parts):
        assert np.abs(p.numpy() - z['cpl_raw%d' % i]).max() < 1e-5, i


def test_spline_coupling_forward_inverse_logdet_match_pyro():
    from oracle import spline
    z = _z()
    layers = _layers(z, 'cpl_')
    x, ctx = _t(z['cpl_x']), _t(z['cpl_ctx'])
    y, ld = spline.spline_coupling(layers, ctx, x, BOUND)
    assert np.abs(y.numpy() - z['cpl_y']).max() < 2e-5
    assert np.abs(ld.numpy() - z['cpl_logdet']).max() < 1e-4
    xi, ldi = spline.spline_coupling(layers, ctx, _t(z['cpl_y']), BOUND, inverse=True)
    assert np.abs(xi.numpy() - z['cpl_x_from_y']).max() < 2e-5
    assert np.abs(ldi.numpy() - z['cpl_logdet']).max() < 2e-4                      # forward log-det recovered from the inverse direction


def test_permute_matches_pyro():
    z = _z()
    x = _t(z['cpl_x'])
    assert np.array_equal(x.index_select(-1, torch.tensor([1, 2, 0])).numpy(), z['permute_120_of_x'])
    inv = [[1, 2, 0].index(i) for i in range(3)]
    assert np.array_equal(x.index_select(-1, torch.tensor(inv)).numpy(), z['permute_120_inv_of_x'])


def test_whole_part_flow_forward_and_log_prob_match_the_reference_factory():
    from oracle.flow import PartFlow
    z = _z()
    k = int(z['flow_num_couplings'])
    pf = PartFlow([_layers(z, 'flow_t%d_' % t) for t in range(k)], BOUND, BASE_STD, 8)
    ctx, zz = _t(z['flow_ctx']), _t(z['flow_z'])
    if 'flow_v' not in z.files:            # file written by the reference-free generator: the pyro stack alone, before the reference's radial tanh
        u = pf.forward_couplings(zz, ctx)
        assert np.abs(u.numpy() - z['flow_u']).max() < 5e-5
        lp = pf.log_prob_couplings(_t(z['flow_u']), ctx)
        assert np.abs(lp.numpy() - z['flow_log_prob_of_u']).max() < 1e-3
        return
    v = pf.forward(zz, ctx)
    assert np.abs(v.numpy() - z['flow_v']).max() < 5e-5
    lp = pf.log_prob_so3(_t(z['flow_v']), ctx)
    assert np.abs(lp.numpy() - z['flow_log_prob_of_v']).max() < 1e-3                # the north-star log-prob tolerance


def test_batch_rodrigues_matches_smplx():
    from oracle.so3 import batch_rodrigues
    z = _z()
    R = batch_rodrigues(_t(z['smpl_aa']).reshape(-1, 3)).reshape(z['smpl_rodrigues'].shape)
    assert np.abs(R.numpy() - z['smpl_rodrigues']).max() < 1e-6


def test_lbs_and_joint_selector_match_smplx():
    from humaniflow_amd.smpl import synthetic_smpl_model, load_extra_regressors
    from helpers import smpl_oracle_model
    from oracle import smpl as osmpl
    z = _z()
    md = dict(synthetic_smpl_model(seed=0))
    md.update(load_extra_regressors())
    model = smpl_oracle_model(md)
    aa, betas, R = _t(z['smpl_aa']), _t(z['smpl_betas']), _t(z['smpl_rodrigues'])
    v, j = osmpl.smpl_forward(model, betas, aa[:, 3:], aa[:, :3], pose2rot=True)
    assert np.abs(v.numpy() - z['smpl_verts_aa']).max() < 2e-5 and np.abs(j[:, :45].numpy() - z['smpl_joints45_aa']).max() < 2e-5
    v, j = osmpl.smpl_forward(model, betas, R[:, 1:], R[:, :1], pose2rot=False)
    assert np.abs(v.numpy() - z['smpl_verts_R']).max() < 2e-5 and np.abs(j[:, :45].numpy() - z['smpl_joints45_R']).max() < 2e-5
    assert np.abs(torch.cat([R[:, :1], R[:, 1:]], dim=1).numpy().reshape(z['smpl_full_pose_R'].shape) - z['smpl_full_pose_R']).max() == 0
    v, j = osmpl.smpl_forward(model, betas[:1], R[:, 1:], R[:, :1], pose2rot=False)
    assert np.abs(v.numpy() - z['smpl_verts_R_beta0']).max() < 2e-5 and np.abs(j[:, :45].numpy() - z['smpl_joints45_R_beta0']).max() < 2e-5
