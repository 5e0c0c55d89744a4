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


# ---------------------------------------------------------------------------------------------------------------------
# Parameter SEMANTICS, stated independently of the restated pyro code path ("which parameter means what").
# The hyper-network emits the 62-vector in the order of `param_dims` at the reference call site
# (models/norm_flows/transforms/conditional_spline_coupling_transform.py:64-70): [widths 2x8 | heights 2x8 | derivatives 2x7 |
# lambdas 2x8].  Below, the knots are rebuilt with plain numpy from the slices of that vector and the SPLINE FUNCTION is probed
# as a black box: it must interpolate them, have the stated slopes at them, and react to a perturbation of ONE raw entry only
# where that entry's meaning says it should.  A swapped w/h or d/l slice, a transposed (2,8) view or a shifted derivative index
# fails these.
# ---------------------------------------------------------------------------------------------------------------------
import numpy as np


def _np_softmax(a):
    e = np.exp(a - a.max(-1, keepdims=True))
    return e / e.sum(-1, keepdims=True)


def _np_knots(raw_slice):                       # (2, 8) logits -> (2, 9) knot positions on [-B, B] with >= 1e-3 relative bin size
    frac = 1e-3 + (1 - 8e-3) * _np_softmax(raw_slice)
    k = np.concatenate([np.zeros((2, 1)), np.cumsum(frac, -1)], -1) * 2 * B - B
    k[:, 0], k[:, -1] = -B, B
    return k


def _spline_from_raw(raw, x, inverse=False):
    """raw (62,) float64, x (n, 2) -> (y, logdet) through the oracle's split + activations + spline."""
    r = torch.as_tensor(raw)[None].expand(x.shape[0], -1)
    w, h, d, l = spline_params(split_params(r), 2, 8)
    return rational_linear_spline(torch.as_tensor(x), w, h, d, l, B, inverse=inverse)


def test_semantics_knots_are_interpolated_and_knot_slopes_are_the_derivative_slice():
    rng = np.random.RandomState(11)
    raw = rng.randn(62) * 1.5
    kx = _np_knots(raw[0:16].reshape(2, 8))             # widths slice: x-knots
    ky = _np_knots(raw[16:32].reshape(2, 8))            # heights slice: y-knots
    dk = 1e-3 + np.log1p(np.exp(raw[32:46].reshape(2, 7)))          # derivative slice: slopes at the 7 INTERIOR knots
    # f(x_k) = y_k at every knot (evaluated a hair inside so the bin search picks either neighbour; both must agree)
    for off in (-1e-9, 1e-9):
        xs = np.clip(kx.T + off, -B, B)                 # (9, 2): row k = knot k of both dimensions
        y, _ = _spline_from_raw(raw, xs)
        assert np.abs(y.numpy() - ky.T).max() < 1e-7, 'knots (x_k, y_k) are not interpolated'
    # slope at the interior knots from BOTH sides = 1e-3 + softplus(raw_d); slope at the two end knots = 1 - 1e-3
    xs_in = kx.T[1:-1]
    for off in (-1e-8, 1e-8):
        _, ld = _spline_from_raw(raw, xs_in + off)
        assert np.abs(np.exp(ld.numpy()) - dk.T).max() < 1e-5, 'interior knot slopes are not 1e-3 + softplus(d_k)'
    _, ld = _spline_from_raw(raw, np.stack([kx[:, 0] + 1e-8, kx[:, -1] - 1e-8]))
    assert np.abs(np.exp(ld.numpy()) - (1 - 1e-3)).max() < 1e-5
    # the inverse direction searches the HEIGHT knots: g(y_k) = x_k
    xb, _ = _spline_from_raw(raw, np.clip(ky.T + 1e-9, -B, B), inverse=True)
    assert np.abs(xb.numpy() - kx.T).max() < 1e-6


def test_semantics_one_raw_entry_moves_only_what_it_means():
    rng = np.random.RandomState(12)
    raw = rng.randn(62)
    kx = _np_knots(raw[0:16].reshape(2, 8))
    grid = np.stack([np.linspace(-B * 0.98, B * 0.98, 1501)] * 2, 1)
    y0, ld0 = (t.numpy() for t in _spline_from_raw(raw, grid))

    def bins_changed(idx, delta=0.7):
        r = raw.copy()
        r[idx] += delta
        y1, ld1 = (t.numpy() for t in _spline_from_raw(r, grid))
        out = []
        for q in range(2):
            b = np.clip(np.searchsorted(kx[q], grid[:, q]) - 1, 0, 7)               # bins of the UNPERTURBED spline
            ch = (np.abs(y1[:, q] - y0[:, q]) > 1e-10) | (np.abs(ld1[:, q] - ld0[:, q]) > 1e-10)
            out.append(sorted(set(b[ch].tolist())))
        return out
    # lambda of (dim q, bin k) lives at 46 + 8 q + k and touches ONLY that bin (knots and knot slopes are unchanged)
    for q, k in ((0, 0), (0, 5), (1, 3), (1, 7)):
        ch = bins_changed(46 + 8 * q + k)
        assert ch[q] == [k] and ch[1 - q] == [], ('lambda', q, k, ch)
    # derivative j of dim q lives at 32 + 7 q + j = slope at interior knot j + 1: touches exactly the two bins that meet there
    for q, j in ((0, 0), (0, 6), (1, 2)):
        ch = bins_changed(32 + 7 * q + j)
        assert ch[q] == [j, j + 1] and ch[1 - q] == [], ('derivative', q, j, ch)
    # a height logit moves y-knots but no x-knot: the bin membership of every grid point (decided on x) is unchanged,
    # a width logit moves the x-knots
    r = raw.copy(); r[16 + 3] += 0.7
    assert np.array_equal(_np_knots(r[0:16].reshape(2, 8)), kx)
    y1, _ = _spline_from_raw(r, kx.T[1:-1] + 1e-9)
    assert np.abs(y1.numpy()[:, 0] - _np_knots(r[16:32].reshape(2, 8))[0, 1:-1]).max() < 1e-7
    assert np.abs(y1.numpy()[:, 0] - _np_knots(raw[16:32].reshape(2, 8))[0, 1:-1]).max() > 1e-3      # dim 0 y-knots moved ...
    y1b, _ = _spline_from_raw(raw, kx.T[1:-1] + 1e-9)
    assert np.abs(y1.numpy()[:, 1] - y1b.numpy()[:, 1]).max() < 1e-12                                # ... dim 1 untouched


def test_semantics_lambda_places_the_break_point_inside_its_bin():
    """Within a bin the map is two linear-rational pieces glued at x_k + lambda_k w_k, lambda_k = 0.025 + 0.95 sigmoid(raw_l):
    the second derivative of a linear-rational piece never changes sign, so the only point of a bin where the (numerical) third-difference
    signature jumps is that break point."""
    rng = np.random.RandomState(13)
    raw = rng.randn(62)
    kx = _np_knots(raw[0:16].reshape(2, 8))
    lam = 0.025 + 0.95 / (1 + np.exp(-raw[46:62].reshape(2, 8)))
    for q, k in ((0, 2), (1, 5)):
        a, b = kx[q, k], kx[q, k + 1]
        xs = np.linspace(a + 1e-6, b - 1e-6, 20001)
        g = np.zeros((xs.size, 2)); g[:, q] = xs
        _, ld = _spline_from_raw(raw, g)
        s = ld.numpy()[:, q]                                 # log f'(x): smooth on each piece, kinked (C^0 only) at the break point
        kink = np.abs(np.diff(s, 2))
        x_break = xs[1:-1][np.argmax(kink)]
        assert abs(x_break - (a + lam[q, k] * (b - a))) < 2 * (xs[1] - xs[0]), (q, k, x_break, a + lam[q, k] * (b - a))
