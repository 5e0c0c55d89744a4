"""Close the "parity unpinned" gap of the oracle in ONE command on any box that has the reference's two third-party packages.

    pip install pyro-ppl==1.7.0 smplx==0.1.26          # requirements.txt:10-11 of the reference (not installable in the build container)
    python tests/golden/make_golden_thirdparty.py [/path/to/HuManiFlow]      -> tests/golden/thirdparty_golden.npz

What it stores (inputs and the packages' outputs only -- no source text):
  * pyro: the reference's own `conditional_spline_coupling(3, 64, [64, 32, 32], count_bins=8, bound=1.5*pi)`
    (models/norm_flows/transforms/conditional_spline_coupling_transform.py:51-78, as models/humaniflow_model.py:92-104 configures it) with
    name-seeded weights: forward, inverse and log|det| on inputs inside and outside the bound; `Permute`; and a whole per-part flow built by
    the reference's `create_conditional_norm_flow` (pyro_conditional_norm_flow.py:21-117): `forward_trans_conditional_norm_flow` on base
    samples and `.condition(ctx).log_prob(v)` of the ConditionalTransformedDistribution;
  * smplx: `smplx.lbs.batch_rodrigues`, and `smplx.SMPL(...)` on the repo's synthetic SMPL-shaped body model (written to a temporary
    smplx-layout pkl) for rotation-matrix and axis-angle inputs: vertices, the 45 joints of VertexJointSelector, full_pose -- what
    models/smpl.py:27-34 builds its 90 joints from.
tests/test_oracle_thirdparty_golden.py consumes the file when present (skips otherwise) and holds oracle/spline.py, oracle/flow.py and
oracle/smpl.py to it.  Without the two packages this script exits with status 3 and writes nothing.
"""
import math
import os
import pickle
import sys
import tempfile

import numpy as np
import torch

REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(OUT))
sys.path.insert(0, os.path.join(OUT, '..'))
sys.path.insert(0, ROOT)

HIDDEN = [64, 32, 32]
CTX, BINS, BOUND, BASE_STD = 64, 8, 1.5 * math.pi, 0.6


def seeded(shape, name, scale):
    """Name-seeded normal values: the consumer test regenerates nothing -- every tensor is stored -- but the file is reproducible."""
    import hashlib
    g = torch.Generator().manual_seed(int(hashlib.md5(name.encode()).hexdigest()[:8], 16))
    return torch.randn(shape, generator=g) * scale


def fill(module, prefix):
    with torch.no_grad():
        for n, p in module.named_parameters():
            p.copy_(seeded(p.shape, prefix + n, 1.0 / math.sqrt(max(p.shape[-1], 1)) if p.dim() > 1 else 0.1))


def pyro_part(out):
    import pyro  # noqa: F401  (raises ImportError when absent)
    from pyro.distributions.transforms import Permute
    sys.path.insert(0, REF)
    from models.norm_flows.transforms.conditional_spline_coupling_transform import conditional_spline_coupling
    out['pyro_version'] = np.array(pyro.__version__)

    # ---- one coupling: forward / inverse / log-det ------------------------------------------------------------------------------------
    t = conditional_spline_coupling(3, CTX, hidden_dims=HIDDEN, count_bins=BINS, bound=BOUND)
    fill(t, 'coupling.')
    for i, lin in enumerate(t.nn.layers):
        out['cpl_w%d' % i], out['cpl_b%d' % i] = lin.weight.detach().numpy(), lin.bias.detach().numpy()
    n = 96
    ctx = seeded((n, CTX), 'coupling.ctx', 1.0)
    x = seeded((n, 3), 'coupling.x', 2.5)
    x[:6] = torch.tensor([[0.3, BOUND + 0.5, -1.0], [0.1, -BOUND - 2.0, BOUND + 1.0], [0.0, 0.0, 0.0], [1.0, BOUND, -BOUND],
                          [-2.0, BOUND - 1e-4, 1e-4], [5.0, 4.0, -4.5]])
    with torch.no_grad():
        ct = t.condition(ctx)
        y = ct(x)
        ld = ct.log_abs_det_jacobian(x, y)
        xi = ct.inv(y)
        # the raw hyper-network output and pyro's parameter transform, to localise a mismatch
        raw = t.nn(x[:, :1], context=ctx)
    out.update(cpl_ctx=ctx.numpy(), cpl_x=x.numpy(), cpl_y=y.numpy(), cpl_logdet=ld.numpy(), cpl_x_from_y=xi.numpy())
    for i, r in enumerate(raw):
        out['cpl_raw%d' % i] = r.numpy()
    p = Permute(torch.tensor([1, 2, 0]))
    out['permute_120_of_x'] = p(x).numpy()
    out['permute_120_inv_of_x'] = p.inv(x).numpy()

    return out


def pyro_flow_part(out):
    """The per-part flow through the reference's own factory, called as models/humaniflow_model.py:92-104 calls it: forward_trans on base samples
    and log_prob on the algebra."""
    sys.path.insert(0, REF)
    from models.norm_flows import pyro_conditional_norm_flow as pf
    kw = dict(device=torch.device('cpu'), event_dim=3, context_dim=CTX, num_transforms=2, transform_type='spline_coupling',
              transform_hidden_dims=HIDDEN, permute_type='permute', permute_hidden_dims=None, bound=BOUND, count_bins=BINS,
              radial_tanh_radius=BOUND, base_dist_std=BASE_STD)
    res = pf.create_conditional_norm_flow(**kw)
    flow_dist, modules, transforms = res[0], res[1], res[2]
    fill(modules, 'flow.')
    k = 0
    for m in modules:
        if hasattr(m, 'nn'):
            for i, lin in enumerate(m.nn.layers):
                out['flow_t%d_w%d' % (k, i)], out['flow_t%d_b%d' % (k, i)] = lin.weight.detach().numpy(), lin.bias.detach().numpy()
            k += 1
    out['flow_num_couplings'] = np.array(k)
    out['flow_kwargs_used'] = np.array(sorted(kw))
    n = 64
    ctx = seeded((n, CTX), 'flow.ctx', 1.0)
    z = seeded((n, 3), 'flow.z', BASE_STD)
    z[0] = 0.0
    with torch.no_grad():
        v = pf.forward_trans_conditional_norm_flow(transforms, z, ctx)
        lp = flow_dist.condition(ctx).log_prob(v)
    out.update(flow_ctx=ctx.numpy(), flow_z=z.numpy(), flow_v=v.numpy(), flow_log_prob_of_v=lp.numpy())
    return out


def smplx_part(out):
    import smplx
    from smplx.lbs import batch_rodrigues
    from humaniflow_amd.smpl import synthetic_smpl_model
    out['smplx_version'] = np.array(getattr(smplx, '__version__', 'unknown'))
    md = synthetic_smpl_model(seed=0)
    kin = np.stack([np.array([2 ** 32 - 1] + list(md['parents'][1:]), np.int64), np.arange(24, dtype=np.int64)])
    pkl = {'v_template': md['v_template'].astype(np.float64), 'shapedirs': md['shapedirs'].astype(np.float64),
           'posedirs': md['posedirs'].T.reshape(6890, 3, 207).astype(np.float64), 'J_regressor': md['J_regressor'].astype(np.float64),
           'weights': md['lbs_weights'].astype(np.float64), 'kintree_table': kin, 'f': np.zeros((13776, 3), np.int64)}
    with tempfile.TemporaryDirectory() as tmp:
        with open(os.path.join(tmp, 'SMPL_NEUTRAL.pkl'), 'wb') as f:
            pickle.dump(pkl, f)
        body = smplx.SMPL(model_path=tmp, gender='neutral', num_betas=10, create_transl=False, batch_size=1)
    M = 12
    aa = seeded((M, 72), 'smpl.aa', 0.4)
    aa[0] = 0.0                                                     # identity pose: exercises batch_rodrigues' angle = |v + 1e-8|
    betas = seeded((M, 10), 'smpl.betas', 1.25)
    with torch.no_grad():
        R = batch_rodrigues(aa.reshape(-1, 3)).reshape(M, 24, 3, 3)
        o1 = body(betas=betas, global_orient=aa[:, :3], body_pose=aa[:, 3:], pose2rot=True)
        o2 = body(betas=betas, global_orient=R[:, :1], body_pose=R[:, 1:], pose2rot=False)
        o3 = body(betas=betas[:1], global_orient=R[:, :1], body_pose=R[:, 1:], pose2rot=False)            # betas broadcast (1 -> M)
    out.update(smpl_aa=aa.numpy(), smpl_betas=betas.numpy(), smpl_rodrigues=R.numpy(),
               smpl_verts_aa=o1.vertices.numpy(), smpl_joints45_aa=o1.joints.numpy(),
               smpl_verts_R=o2.vertices.numpy(), smpl_joints45_R=o2.joints.numpy(), smpl_full_pose_R=np.asarray(o2.full_pose),
               smpl_verts_R_beta0=o3.vertices.numpy(), smpl_joints45_R_beta0=o3.joints.numpy())
    return out


def main():
    try:
        import pyro  # noqa: F401
        import smplx  # noqa: F401
    except ImportError as e:
        print('make_golden_thirdparty: %s -- install pyro-ppl==1.7.0 and smplx==0.1.26 (requirements.txt:10-11 of the reference); nothing written' % e)
        return 3
    out = {}
    pyro_part(out)
    pyro_flow_part(out)
    smplx_part(out)
    path = os.path.join(OUT, 'thirdparty_golden.npz')
    np.savez_compressed(path, **out)
    print('wrote %s (%d arrays, %.1f MB)' % (path, len(out), os.path.getsize(path) / 1e6))
    return 0


if __name__ == '__main__':
    sys.exit(main())
