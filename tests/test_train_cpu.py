"""Training-step pieces that need no GPU: the loss mirror and the oracle's loss restatement against the reference's own HumaniflowLoss
(tests/golden/loss_golden.npz: values and input gradients for both reductions), and the oracle's differentiable forward."""
import os
import types

import numpy as np
import torch

from helpers import GOLDEN, init_model_for_tests, oracle_cfg_from
from oracle import train as ot


def _cases():
    z = np.load(os.path.join(GOLDEN, 'loss_golden.npz'))
    for ci in range(2):
        pre = 'c%d_' % ci
        yield {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}


def _cfg(red):
    return types.SimpleNamespace(REDUCTION=red, APPLY_POINT_EST_LOSS=False,
                                 WEIGHTS=types.SimpleNamespace(POSE=150.0, SHAPE=50.0, JOINTS2D=25000.0, GLOB_ROTMATS=5000.0, VERTS3D=0.0, JOINTS3D=0.0))


def test_loss_mirror_and_oracle_loss_match_the_reference_class():
    from humaniflow_amd.losses import HumaniflowLoss
    for c in _cases():
        red = str(c['reduction'])
        T = lambda k, g=False: torch.tensor(c[k]).requires_grad_(g)
        for which in ('mirror', 'oracle'):
            lp, mode, ls, j2, gp = T('lp', True), T('mode', True), T('log_std', True), T('j2d_pred', True), T('glob_p', True)
            vis = torch.tensor(c['vis'])
            if which == 'oracle':
                loss = ot.humaniflow_loss(_cfg(red), 256, lp, mode, ls, T('shape_t'), j2, T('j2d_t'), vis, gp, T('glob_t'))
            else:
                class Stub:                                           # the reference's call pattern: a list of per-part distributions
                    def __init__(self, j):
                        self.j = j

                    def log_prob(self, value):
                        assert value.dtype == torch.float64
                        return lp[:, self.j]
                crit = HumaniflowLoss(_cfg(red), img_wh=256)
                pred = {'pose_dist': [Stub(j) for j in range(23)], 'shape_dist': torch.distributions.Normal(mode, torch.exp(ls), validate_args=False),
                        'joints2D': j2, 'glob_rotmats': gp}
                targ = {'pose_params_rotmats': torch.zeros(lp.shape[0], 23, 3, 3), 'shape_params': T('shape_t'), 'joints2D': T('j2d_t'),
                        'joints2D_vis': vis, 'glob_rotmats': T('glob_t')}
                loss = crit(targ, pred)
            loss.backward()
            assert abs(loss.item() - float(c['loss'])) <= 1e-6 * abs(float(c['loss'])), (which, red)
            for t, k in ((lp, 'g_lp'), (mode, 'g_mode'), (ls, 'g_log_std'), (j2, 'g_j2d_pred'), (gp, 'g_glob_p')):
                ref = c[k]
                assert np.abs(t.grad.numpy() - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), (which, red, k)


def test_oracle_forward_is_differentiable_in_float64():
    """The ground truth of the GPU gradient tests: the pinned oracle forward under autograd with float64 defaults; a finite-difference
    probe of one hyper-network weight and one context weight confirms its Jacobians."""
    import humaniflow_amd as H
    cfg = H.get_humaniflow_cfg_defaults()
    m = init_model_for_tests(H.HumaniflowModel(torch.device('cpu'), cfg.MODEL, H.SMPL_PARENTS), seed=3)
    sd = {k: (v.detach().double().clone().requires_grad_(True) if v.is_floating_point() else v.clone()) for k, v in m.state_dict().items()
          if not k.startswith('image_encoder')}
    ocfg = oracle_cfg_from(cfg.MODEL)
    g = torch.Generator().manual_seed(0)
    B, N = 2, 2
    feats = (torch.randn(B, 512, generator=g) * 0.5).double()
    noise = {'shape': torch.randn(N, B, 10, generator=g).double(), 'pose': torch.randn(23, B, N, 3, generator=g).double()}
    wts = torch.randn(B, N, 23, 3, 3, generator=g).double()

    def loss_of(sd_):
        with ot.float64_defaults():
            out = ot.forward_grad(sd_, ocfg, input_feats=feats, num_samples=N, noise=noise)
        return (out['pose_rotmats_samples'] * wts).sum()
    loss = loss_of(sd)
    loss.backward()
    for key, idx in (('pose_so3flow_transform_modules.7.nn.layers.3.weight', (5, 3)), ('fc_flow_context.9.weight', (11, 260)),
                     ('fc_input_shape_glob_cam_feats.weight', (17, 515))):
        assert sd[key].grad is not None and torch.isfinite(sd[key].grad).all()
        h = 1e-6
        with torch.no_grad():
            sp = {k: v.detach().clone() for k, v in sd.items()}
            sp[key][idx] += h
            sm = {k: v.detach().clone() for k, v in sd.items()}
            sm[key][idx] -= h
            fd = (loss_of(sp) - loss_of(sm)).item() / (2 * h)
        assert abs(fd - sd[key].grad[idx].item()) <= 1e-4 * max(1.0, abs(fd)), (key, fd, sd[key].grad[idx].item())


def test_bench_conv_flop_accounting_is_invariant_under_seam_fusion():
    """bench.py books algorithmic FLOPs per launch of the encoder's plan: fusing bottleneck seams into single launches changes the launch
    list (6 launches fewer for ResNet-50 at 256 x 256), never the total (391.0 GFLOP at B = 32)."""
    import importlib.util
    import os
    from helpers import ROOT
    from humaniflow_amd.encoder import ResNetEncoder
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    enc = ResNetEncoder(50, 18)
    f0, k0 = bench.encoder_conv_flops(enc, 32, 256)                      # no plan yet: one launch per convolution (downsample branches fused)
    assert len(f0) == 49 and abs(sum(f0) / 1e9 - 391.0) < 0.1
    enc._plan = {'pair': {0: None, 1: None, 2: None, 4: None, 5: None, 6: None}}     # the seams the device plan fuses (block indices)
    f1, k1 = bench.encoder_conv_flops(enc, 32, 256)
    assert len(f1) == 43 and abs(sum(f1) - sum(f0)) < 1.0 and sum(1 for k in k1 if 'seam' in k) == 6
    f2, _ = bench.encoder_conv_flops(enc, 1, 224)                         # layer1 56 x 56 = 49 * 64 pixels: fused; layer2 28 x 28 = 784: not a multiple of 64
    assert len(f2) == 46
