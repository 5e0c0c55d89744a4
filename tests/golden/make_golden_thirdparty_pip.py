"""Third-party-ONLY variant of make_golden_thirdparty.py: imports nothing from the reference checkout, so it can travel to a box that has
(or can `pip install`) the reference's two pinned packages but not the reference itself.

    pip install pyro-ppl==1.7.0 smplx==0.1.26          # requirements.txt:10-11 of the reference
    python tests/golden/make_golden_thirdparty_pip.py   -> tests/golden/thirdparty_golden.npz

The reference's wrapper class around pyro's coupling (models/norm_flows/transforms/conditional_spline_coupling_transform.py:11-48) only
forwards its arguments to `pyro.distributions.transforms.SplineCoupling` with `hypernet = partial(ConditionalDenseNN, context=ctx)`; this
script makes that call itself from pyro's public API with the arguments models/humaniflow_model.py:92-104 passes:
    SplineCoupling(input_dim=3, split_dim=1, hypernet=partial(ConditionalDenseNN(1, 64, [64, 32, 32], param_dims=[16, 16, 14, 16]), context=ctx),
                   count_bins=8, bound=1.5 pi, order='linear', identity=True)
and the per-part stack of models/norm_flows/pyro_conditional_norm_flow.py:44-52,55-101 WITHOUT the reference's own ScaledRadialTanh
(that transform is reference code, pinned by tests/golden/flow_golden.npz): Independent(Normal(0, 0.6), 1) pushed through
[Permute([0,1,2]), coupling 0, Permute([1,2,0]), coupling 1] -- stored as `flow_u` (pre-tanh algebra vector) and `flow_log_prob_of_u`.
The smplx part is the same as in make_golden_thirdparty.py (it never needed the reference).  Keys shared with that script mean the same
thing; tests/test_oracle_thirdparty_golden.py consumes either file.  Without the two packages: exit status 3, nothing written.
"""
import math
import os
import sys
from functools import partial

import numpy as np
import torch

OUT = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(OUT))
sys.path.insert(0, OUT)
sys.path.insert(0, ROOT)

HIDDEN = [64, 32, 32]
CTX, BINS, BOUND, BASE_STD = 64, 8, 1.5 * math.pi, 0.6
PARAM_DIMS = [2 * BINS, 2 * BINS, 2 * (BINS - 1), 2 * BINS]


def main():
    try:
        import pyro
        import smplx  # noqa: F401
    except ImportError as e:
        print('make_golden_thirdparty_pip: %s -- install pyro-ppl==1.7.0 and smplx==0.1.26; nothing written' % e)
        return 3
    from pyro.distributions import ConditionalTransformedDistribution, Independent, Normal  # noqa: F401
    from pyro.distributions.transforms import Permute, SplineCoupling
    from pyro.nn import ConditionalDenseNN
    import make_golden_thirdparty as shared                    # seeded(), fill(), smplx_part(): no reference import at module level

    out = {'pyro_version': np.array(pyro.__version__), 'generator': np.array('make_golden_thirdparty_pip.py (reference-free)')}

    def coupling(net, ctx):
        return SplineCoupling(input_dim=3, split_dim=1, hypernet=partial(net, context=ctx), count_bins=BINS, bound=BOUND, order='linear',
                              identity=True)

    # ---- one coupling: forward / inverse / log-det (same keys and inputs as make_golden_thirdparty.pyro_part) ---------------------------
    net = ConditionalDenseNN(1, CTX, HIDDEN, param_dims=PARAM_DIMS)
    shared.fill(net, 'coupling.nn.')
    for i, lin in enumerate(net.layers):
        out['cpl_w%d' % i], out['cpl_b%d' % i] = lin.weight.detach().numpy(), lin.bias.detach().numpy()
    n = 96
    ctx = shared.seeded((n, CTX), 'coupling.ctx', 1.0)
    x = shared.seeded((n, 3), 'coupling.x', 2.5)
    x[:6] = torch.tensor([[0.3, BOUND + 0.5, -1.0], [0.1, -BOUND - 2.0, BOUND + 1.0], [0.0, 0.0, 0.0], [1.0, BOUND, -BOUND],
                          [-2.0, BOUND - 1e-4, 1e-4], [5.0, 4.0, -4.5]])
    with torch.no_grad():
        ct = coupling(net, ctx)
        y = ct(x)
        ld = ct.log_abs_det_jacobian(x, y)
        xi = ct.inv(y)
        raw = net(x[:, :1], context=ctx)
    out.update(cpl_ctx=ctx.numpy(), cpl_x=x.numpy(), cpl_y=y.numpy(), cpl_logdet=ld.numpy(), cpl_x_from_y=xi.numpy())
    for i, r in enumerate(raw):
        out['cpl_raw%d' % i] = r.numpy()
    p = Permute(torch.tensor([1, 2, 0]))
    out['permute_120_of_x'] = p(x).numpy()
    out['permute_120_inv_of_x'] = p.inv(x).numpy()

    # ---- the per-part stack without the radial tanh, through torch's TransformedDistribution ---------------------------------------------
    nets = [ConditionalDenseNN(1, CTX, HIDDEN, param_dims=PARAM_DIMS) for _ in range(2)]
    for k, m in enumerate(nets):
        shared.fill(m, 'flow.%d.nn.' % k)
        for i, lin in enumerate(m.layers):
            out['flow_t%d_w%d' % (k, i)], out['flow_t%d_b%d' % (k, i)] = lin.weight.detach().numpy(), lin.bias.detach().numpy()
    out['flow_num_couplings'] = np.array(2)
    n = 64
    ctx = shared.seeded((n, CTX), 'flow.ctx', 1.0)
    z = shared.seeded((n, 3), 'flow.z', BASE_STD)
    z[0] = 0.0
    perms = [torch.tensor([0, 1, 2]), torch.tensor([1, 2, 0])]
    with torch.no_grad():
        ts = []
        for k in range(2):
            ts += [Permute(perms[k]), coupling(nets[k], ctx)]
        base = Independent(Normal(loc=torch.zeros(3), scale=torch.ones(3) * BASE_STD, validate_args=False), 1)
        u = z
        for t in ts:
            u = t(u)
        dist = torch.distributions.TransformedDistribution(base, ts, validate_args=False)
        lp = dist.log_prob(u)
    out.update(flow_ctx=ctx.numpy(), flow_z=z.numpy(), flow_u=u.numpy(), flow_log_prob_of_u=lp.numpy())

    shared.smplx_part(out)
    path = os.path.join(OUT, 'thirdparty_golden.npz')
    np.savez_compressed(path, **out)
    print('wrote %s (%d arrays, %.1f MB)' % (path, len(out), os.path.getsize(path) / 1e6))
    return 0


if __name__ == '__main__':
    sys.exit(main())
