"""Golden vectors for the training-step pieces, generated from the REFERENCE's own Python (build container only; only inputs and
the outputs the reference's functions produce are stored):

    python tests/golden/make_golden_train.py [/root/reference]

* loss_golden.npz      -- losses/humaniflow_loss.py HumaniflowLoss.forward (both REDUCTION modes, J2D visibility masking) on seeded
                          tensors, with stub per-part distributions returning fixed log-probabilities; values and input gradients.
* linear_plu_golden.npz -- models/norm_flows/transforms/linearPLU_transform.py ConditionedLinearPLUTransform (_call, _inverse,
                          log_abs_det_jacobian) with given permutation / LU matrices (pure torch once the pyro base classes are stubbed).
* encoder_train_golden.npz -- models/resnet.py resnet18(in_channels=18) in TRAIN mode (batch-statistics BatchNorm, the encoder of the
                          training step, train/train_humaniflow.py:248-352) at B=4, 64x64: features, every BatchNorm's running-statistics
                          update and the gradient of every parameter (small tensors in full, large ones as a fixed strided sample + their L2
                          norm) for the loss sum(feats * w).  Weights = helpers.deterministic_state_dict(seed=11), input / w from numpy's
                          RandomState (stable by contract): nothing but outputs is stored.  The reference ran in float64 (ground truth) and
                          in float32 (its own rounding distance from float64 = the slack of the GPU test).
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


class NS(types.SimpleNamespace):
    pass


def loss_golden():
    hl = _load('ref_humaniflow_loss', os.path.join(REF, 'losses', 'humaniflow_loss.py'))
    out = {}
    g = torch.Generator().manual_seed(77)
    B, P, S, nb = 5, 23, 4, 10
    for ci, red in enumerate(('mean', 'sum')):
        cfg = NS(REDUCTION=red, APPLY_POINT_EST_LOSS=False,
                 WEIGHTS=NS(POSE=150.0, SHAPE=50.0, JOINTS2D=25000.0, GLOB_ROTMATS=5000.0, VERTS3D=0.0, JOINTS3D=0.0))
        crit = hl.HumaniflowLoss(cfg, img_wh=256)
        lp = (torch.randn(B, P, generator=g) * 2).requires_grad_(True)
        mode = torch.randn(B, nb, generator=g).requires_grad_(True)
        log_std = (torch.randn(B, nb, generator=g) * 0.3).requires_grad_(True)
        shape_t = torch.randn(B, nb, generator=g)
        j2d_pred = (torch.randn(B, S, 17, 2, generator=g) * 0.5).requires_grad_(True)
        j2d_t = torch.rand(B, 17, 2, generator=g) * 256
        vis = torch.rand(B, 17, generator=g) > 0.3
        glob_p = torch.randn(B, 3, 3, generator=g).requires_grad_(True)
        glob_t = torch.randn(B, 3, 3, generator=g)
        R_t = torch.randn(B, P, 3, 3, generator=g)

        class Stub:
            def __init__(self, j):
                self.j = j

            def log_prob(self, value):
                assert value.dtype == torch.float64 and tuple(value.shape) == (B, 3, 3)
                return lp[:, self.j]
        pred = {'pose_dist': [Stub(j) for j in range(P)], 'shape_dist': torch.distributions.Normal(mode, torch.exp(log_std), validate_args=False),
                'joints2D': j2d_pred, 'glob_rotmats': glob_p}
        targ = {'pose_params_rotmats': R_t, 'shape_params': shape_t, 'joints2D': j2d_t, 'joints2D_vis': vis, 'glob_rotmats': glob_t}
        loss = crit(targ, pred)
        loss.backward()
        pre = 'c%d_' % ci
        out.update({pre + 'reduction': np.array(red), pre + 'lp': lp.detach().numpy(), pre + 'mode': mode.detach().numpy(),
                    pre + 'log_std': log_std.detach().numpy(), pre + 'shape_t': shape_t.numpy(), pre + 'j2d_pred': j2d_pred.detach().numpy(),
                    pre + 'j2d_t': j2d_t.numpy(), pre + 'vis': vis.numpy(), pre + 'glob_p': glob_p.detach().numpy(), pre + 'glob_t': glob_t.numpy(),
                    pre + 'loss': loss.detach().numpy(), pre + 'g_lp': lp.grad.numpy(), pre + 'g_mode': mode.grad.numpy(),
                    pre + 'g_log_std': log_std.grad.numpy(), pre + 'g_j2d_pred': j2d_pred.grad.numpy(), pre + 'g_glob_p': glob_p.grad.numpy()})
    np.savez_compressed(os.path.join(OUT, 'loss_golden.npz'), **out)


def linear_plu_golden():
    # pyro base classes are only inherited from / used as type tags: stub them, the arithmetic is plain torch
    pyro = types.ModuleType('pyro')
    pnn = types.ModuleType('pyro.nn')
    pnn.DenseNN = object
    pd = types.ModuleType('pyro.distributions')
    pd.constraints = torch.distributions.constraints
    pc = types.ModuleType('pyro.distributions.conditional')
    pc.ConditionalTransformModule = type('ConditionalTransformModule', (torch.nn.Module,), {})
    pt = types.ModuleType('pyro.distributions.torch_transform')
    pt.TransformModule = type('TransformModule', (torch.distributions.Transform, torch.nn.Module), {})
    sys.modules.update({'pyro': pyro, 'pyro.nn': pnn, 'pyro.distributions': pd, 'pyro.distributions.conditional': pc,
                        'pyro.distributions.torch_transform': pt})
    plu = _load('ref_linear_plu', os.path.join(REF, 'models', 'norm_flows', 'transforms', 'linearPLU_transform.py'))
    g = torch.Generator().manual_seed(5)
    out = {}
    n = 16
    for pi, perm in enumerate(([0, 1, 2], [1, 2, 0], [2, 0, 1])):
        P = torch.eye(3)[torch.tensor(perm)]
        LU = torch.randn(n, 3, 3, generator=g)
        LU[:, range(3), range(3)] = torch.nn.functional.softplus(LU[:, range(3), range(3)], beta=0.75)      # as ConditionalLinearPLUTransform.condition
        t = plu.ConditionedLinearPLUTransform(P, LU)
        x = torch.randn(n, 3, generator=g)
        y = t._call(x)
        out.update({'p%d_perm' % pi: np.array(perm), 'p%d_LU' % pi: LU.numpy(), 'p%d_x' % pi: x.numpy(), 'p%d_y' % pi: y.numpy(),
                    'p%d_xinv' % pi: t._inverse(y).numpy(), 'p%d_logdet' % pi: t.log_abs_det_jacobian(x, y).numpy()})
    # unconditioned LU with a negative diagonal entry (LinearPLUTransform keeps it unconstrained: log|u_ii|), one matrix for the whole batch
    LU1 = torch.tensor([[0.7, -1.2, 0.3], [0.5, -0.9, 1.1], [-0.4, 0.8, 1.6]])
    t = plu.ConditionedLinearPLUTransform(torch.eye(3)[torch.tensor([1, 2, 0])], LU1)
    x = torch.randn(n, 3, generator=g)
    y = t._call(x)
    out.update({'u_LU': LU1.numpy(), 'u_x': x.numpy(), 'u_y': y.numpy(), 'u_xinv': t._inverse(y).numpy(), 'u_logdet': t.log_abs_det_jacobian(x, y).numpy()})
    # softplus(beta=0.75) of the conditional variant's diagonal, as raw -> constrained pairs
    raw = torch.linspace(-6, 30, 25)
    out.update({'sp_raw': raw.numpy(), 'sp_out': torch.nn.functional.softplus(raw, beta=0.75).numpy()})
    np.savez_compressed(os.path.join(OUT, 'linear_plu_golden.npz'), **out)


def encoder_train_inputs(B=4, size=64, out_dim=512):
    """Input batch and loss weights of the encoder-training golden: numpy's legacy RandomState stream is stable across versions."""
    rs = np.random.RandomState(20240611)
    x = rs.standard_normal((B, 18, size, size)).astype(np.float32)
    w = rs.standard_normal((B, out_dim)).astype(np.float32)
    return torch.from_numpy(x), torch.from_numpy(w)


def grad_sample(t, full_below=8192, samples=4096):
    """Small tensors in full; large ones as every stride-th entry of the flattened tensor (stride fixed by the element count)."""
    f = t.reshape(-1)
    if f.numel() <= full_below:
        return f
    return f[::f.numel() // samples]


def encoder_train_golden():
    sys.path.insert(0, os.path.join(OUT, '..'))
    from helpers import deterministic_state_dict
    sys.path.insert(0, REF)
    import models.resnet as resnet
    x, w = encoder_train_inputs()
    runs = {}
    for dt in (torch.float64, torch.float32):
        net = resnet.resnet18(in_channels=18)
        sd = deterministic_state_dict({k: tuple(t.shape) for k, t in net.state_dict().items()}, seed=11)
        net.load_state_dict(sd, strict=True)
        net = net.to(dt).train()
        feats = net(x.to(dt))
        (feats * w.to(dt)).sum().backward()
        runs[dt] = (net, feats.detach())
    n64, f64 = runs[torch.float64]
    n32, f32 = runs[torch.float32]
    l2 = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp(min=1e-300)).item()
    out = {'feats': f64.numpy(), 'feats_f32_dist': np.array(l2(f32, f64))}
    names, dists = [], []
    for (name, p), (_, q) in zip(n64.named_parameters(), n32.named_parameters()):
        names.append(name)
        dists.append(l2(q.grad, p.grad))
        out['g_' + name] = grad_sample(p.grad).float().numpy()
        out['n_' + name] = np.array(p.grad.norm().item())
    out['param_names'] = np.array(names)
    out['grad_f32_dist'] = np.array(dists)
    for name, b in n64.named_buffers():
        if not name.endswith('num_batches_tracked'):
            out['b_' + name] = b.float().numpy()
        else:
            assert int(b) == 1
    np.savez_compressed(os.path.join(OUT, 'encoder_train_golden.npz'), **out)


if __name__ == '__main__':
    loss_golden()
    linear_plu_golden()
    encoder_train_golden()
    print('written loss_golden.npz, linear_plu_golden.npz, encoder_train_golden.npz to', OUT)
