"""Property tests anchoring the restated pyro spline / coupling (parity unpinned: SURVEY App. A)."""
import math

import torch

from oracle.spline import rational_linear_spline, spline_params, spline_coupling, split_params
from oracle.flow import PartFlow

B = 1.5 * math.pi


def _params(n, seed=0, scale=2.0):
    g = torch.Generator().manual_seed(seed)
    raw = torch.randn(n, 62, generator=g, dtype=torch.float64) * scale
    return spline_params(split_params(raw), 2, 8)


def test_forward_inverse_roundtrip_and_logdet_sign():
    w, h, d, l = _params(512)
    x = (torch.rand(512, 2, dtype=torch.float64, generator=torch.Generator().manual_seed(1)) * 2 - 1) * B * 1.2
    y, ld = rational_linear_spline(x, w, h, d, l, B)
    xr, ldi = rational_linear_spline(y, w, h, d, l, B, inverse=True)
    assert (xr - x).abs().max() < 1e-9
    assert (ld + ldi).abs().max() < 1e-9
    outside = (x.abs() > B)
    assert torch.equal(y[outside], x[outside]) and (ld[outside] == 0).all()


def test_monotone_and_logdet_is_log_derivative():
    w, h, d, l = _params(1, seed=3)
    xs = torch.linspace(-B * 0.999, B * 0.999, 4001, dtype=torch.float64)[:, None].expand(-1, 2).contiguous().requires_grad_(True)
    W, H, D, L = (t.expand(4001, -1, -1) for t in (w, h, d, l))
    y, ld = rational_linear_spline(xs, W, H, D, L, B)
    assert (y[1:] - y[:-1] > 0).all()                                   # strictly monotone
    (g,) = torch.autograd.grad(y.sum(), xs)
    assert (torch.log(g) - ld).abs().max() < 1e-8                         # logdet == log dy/dx
    # continuity of value and derivative where the formula switches: interior knots (+ the 1e-6 search offset)
    # and the in-bin break points lambda
    from oracle import spline as S
    widths = S.MIN_BIN_WIDTH + (1.0 - S.MIN_BIN_WIDTH * 8) * w
    ww, cw = S._knots(widths.clone(), -B, B)
    lam = (1 - 2 * S.MIN_LAMBDA) * l + S.MIN_LAMBDA
    pts = torch.cat([cw[0, :, 1:-1] + S.SEARCH_EPS, cw[0, :, :-1] + lam[0] * ww[0]], dim=1).T.contiguous()   # (15, 2)
    n = pts.shape[0]
    Wn, Hn, Dn, Ln = (t.expand(n, -1, -1) for t in (w, h, d, l))
    ya, lda = rational_linear_spline(pts - 1e-9, Wn, Hn, Dn, Ln, B)
    yb, ldb = rational_linear_spline(pts + 1e-9, Wn, Hn, Dn, Ln, B)
    slope = torch.exp(torch.maximum(lda, ldb))
    assert ((ya - yb).abs() <= 2e-9 * slope * 1.5 + 1e-12).all() and (lda - ldb).abs().max() < 1e-3


def test_boundary_behaviour():
    w, h, d, l = _params(4, seed=5)
    x = torch.tensor([[-B, B]] * 4, dtype=torch.float64)
    y, ld = rational_linear_spline(x, w, h, d, l, B)
    assert (y - x).abs().max() < 1e-9                                     # end knots are fixed points
    eps = 1e-7
    y2, ld2 = rational_linear_spline(x * (1 - eps), w, h, d, l, B)
    assert (ld2 - math.log(1 - 1e-3)).abs().max() < 1e-3                  # boundary derivative 1 - 1e-3 (quirk 8)


def _layers(seed, dt=torch.float32):
    g = torch.Generator().manual_seed(seed)
    dims = [65, 64, 32, 32, 62]
    return [(torch.randn(dims[i + 1], dims[i], generator=g, dtype=dt) * 0.3, torch.randn(dims[i + 1], generator=g, dtype=dt) * 0.1)
            for i in range(4)]


def test_coupling_roundtrip_and_context_first():
    layers = _layers(0, torch.float64)
    g = torch.Generator().manual_seed(2)
    ctx = torch.randn(16, 64, generator=g, dtype=torch.float64)
    x = torch.randn(16, 3, generator=g, dtype=torch.float64)
    y, ld = spline_coupling(layers, ctx, x, B)
    assert torch.equal(y[:, 0], x[:, 0])                                  # identity=True: first coordinate passes
    xr, ld2 = spline_coupling(layers, ctx, y, B, inverse=True)
    assert (xr - x).abs().max() < 1e-9 and (ld - ld2).abs().max() < 1e-9
    # ConditionalDenseNN concatenates [context, x]: the LAST input column multiplies x1
    layers2 = [(layers[0][0].clone(), layers[0][1])] + layers[1:]
    layers2[0][0][:, 64] = 0
    y0, _ = spline_coupling(layers2, ctx, x, B)
    x2 = x.clone(); x2[:, 0] += 1.0
    y1, _ = spline_coupling(layers2, ctx, x2, B)
    assert (y0[:, 1:] - y1[:, 1:]).abs().max() < 1e-12


def test_part_flow_density_integrates_and_inverts():
    # fp64 weights: with these (deliberately wild) hyper-networks the spline derivatives reach e^-15 and the
    # fp32 inverse is ill-conditioned; the algebra is what is being checked here
    pf = PartFlow([_layers(1, torch.float64), _layers(2, torch.float64)], B, 0.6)
    g = torch.Generator().manual_seed(4)
    ctx = torch.randn(8, 64, generator=g, dtype=torch.float64)
    eps = torch.randn(8, 3, generator=g, dtype=torch.float64)
    v = pf.forward(eps * 0.6, ctx)
    assert (v.norm(dim=-1) < B).all()
    lp = pf.log_prob_so3(v, ctx)
    assert torch.isfinite(lp).all()
    # change of variables: log p(v) = log N(z) - log|det dv/dz| with the Jacobian from autograd
    for i in range(3):
        z = (eps[i] * 0.6).clone().requires_grad_(True)
        J = torch.autograd.functional.jacobian(lambda t: pf.forward(t[None], ctx[i:i + 1])[0], z)
        ref = torch.distributions.Normal(0, 0.6).log_prob(z.detach()).sum() - torch.linalg.slogdet(J)[1]
        assert abs(ref.item() - lp[i].item()) < 1e-2
