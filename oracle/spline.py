"""Oracle: monotonic rational-LINEAR spline + conditional hyper-network. Test infra only.

PARITY UNPINNED: this arithmetic lives in pyro-ppl==1.7.0 (requirements.txt:11 of the
reference), which is neither vendored under /root/reference nor installable here.
It restates the published algorithm of
  pyro/distributions/transforms/spline.py  (_monotonic_rational_spline, _calculate_knots,
                                            _searchsorted, _select_bins, ConditionalSpline._params)
  pyro/distributions/transforms/spline_coupling.py (SplineCoupling._call/_inverse, identity=True)
  pyro/nn/dense_nn.py                      (ConditionalDenseNN.forward: cat([context, x]))
(Dolatabadi et al. 2020, "Invertible generative modeling using linear rational splines"),
anchored on the reference call sites
  models/norm_flows/transforms/conditional_spline_coupling_transform.py:35-48,58-78
  models/norm_flows/pyro_conditional_norm_flow.py:97-101
and on the property tests in tests/test_oracle_spline.py.
"""
import torch
import torch.nn.functional as F

MIN_BIN_WIDTH = 1e-3
MIN_BIN_HEIGHT = 1e-3
MIN_DERIVATIVE = 1e-3
MIN_LAMBDA = 0.025
SEARCH_EPS = 1e-6


def _knots(lengths, lower, upper):
    knots = torch.cumsum(lengths, dim=-1)
    knots = F.pad(knots, pad=(1, 0), mode='constant', value=0.0)
    knots = (upper - lower) * knots + lower
    knots[..., 0] = lower
    knots[..., -1] = upper
    return knots[..., 1:] - knots[..., :-1], knots


def _bin(x, idx):
    idx = idx.clamp(min=0, max=x.size(-1) - 1)
    return x.gather(-1, idx).squeeze(-1)


def spline_params(raw, input_dim=2, count_bins=8):
    """ConditionalSpline._params for order='linear' with a DenseNN hyper-network:
    raw is the tuple (w,h,d,l) of widths (..., D*K), (..., D*K), (..., D*(K-1)), (..., D*K)."""
    w, h, d, l = raw
    w = w.reshape(w.shape[:-1] + (input_dim, count_bins))
    h = h.reshape(h.shape[:-1] + (input_dim, count_bins))
    d = d.reshape(d.shape[:-1] + (input_dim, count_bins - 1))
    l = l.reshape(l.shape[:-1] + (input_dim, count_bins))
    return F.softmax(w, -1), F.softmax(h, -1), F.softplus(d), torch.sigmoid(l)


def rational_linear_spline(inputs, widths, heights, derivatives, lambdas, bound, inverse=False):
    """_monotonic_rational_spline(order='linear').  Returns (outputs, logabsdet) where
    logabsdet is log|d out / d in| of the direction evaluated (forward or inverse)."""
    left = bottom = -bound
    right = top = bound
    num_bins = widths.shape[-1]
    inside = (inputs >= left) & (inputs <= right)

    widths = MIN_BIN_WIDTH + (1.0 - MIN_BIN_WIDTH * num_bins) * widths
    heights = MIN_BIN_HEIGHT + (1.0 - MIN_BIN_HEIGHT * num_bins) * heights
    derivatives = MIN_DERIVATIVE + derivatives

    widths, cumwidths = _knots(widths, left, right)
    heights, cumheights = _knots(heights, bottom, top)
    derivatives = F.pad(derivatives, pad=(1, 1), mode='constant', value=1.0 - MIN_DERIVATIVE)

    ref = (cumheights if inverse else cumwidths) + SEARCH_EPS
    bin_idx = ((inputs[..., None] >= ref).sum(-1) - 1)[..., None]

    in_w = _bin(widths, bin_idx)
    in_cw = _bin(cumwidths, bin_idx)
    in_ch = _bin(cumheights, bin_idx)
    in_delta = _bin(heights / widths, bin_idx)
    in_d = _bin(derivatives, bin_idx)
    in_d1 = _bin(derivatives[..., 1:], bin_idx)
    in_h = _bin(heights, bin_idx)

    lambdas = (1 - 2 * MIN_LAMBDA) * lambdas + MIN_LAMBDA
    lam = _bin(lambdas, bin_idx)

    wa = 1.0
    wb = torch.sqrt(in_d / in_d1) * wa
    wc = (lam * wa * in_d + (1 - lam) * wb * in_d1) / in_delta
    ya = in_ch
    yb = in_h + in_ch
    yc = ((1.0 - lam) * wa * ya + lam * wb * yb) / ((1.0 - lam) * wa + lam * wb)

    if inverse:
        lo = (inputs <= yc).to(inputs.dtype)
        hi = (inputs > yc).to(inputs.dtype)
        num = (lam * wa * (ya - inputs)) * lo + ((wc - lam * wb) * inputs + lam * wb * yb - wc * yc) * hi
        den = ((wc - wa) * inputs + wa * ya - wc * yc) * lo + ((wc - wb) * inputs + wb * yb - wc * yc) * hi
        theta = num / den
        outputs = theta * in_w + in_cw
        dnum = (wa * wc * lam * (yc - ya) * lo + wb * wc * (1 - lam) * (yb - yc) * hi) * in_w
        logabsdet = torch.log(dnum) - 2 * torch.log(torch.abs(den))
    else:
        theta = (inputs - in_cw) / in_w
        lo = (theta <= lam).to(inputs.dtype)
        hi = (theta > lam).to(inputs.dtype)
        num = (wa * ya * (lam - theta) + wc * yc * theta) * lo + (wc * yc * (1 - theta) + wb * yb * (theta - lam)) * hi
        den = (wa * (lam - theta) + wc * theta) * lo + (wc * (1 - theta) + wb * (theta - lam)) * hi
        outputs = num / den
        dnum = (wa * wc * lam * (yc - ya) * lo + wb * wc * (1 - lam) * (yb - yc) * hi) / in_w
        logabsdet = torch.log(dnum) - 2 * torch.log(torch.abs(den))

    outputs = torch.where(inside, outputs, inputs)
    logabsdet = torch.where(inside, logabsdet, torch.zeros_like(logabsdet))
    return outputs, logabsdet


def dense_nn(layers, x, context):
    """pyro.nn.ConditionalDenseNN.forward with ReLU: h = cat([context, x]); ReLU after every
    layer but the last.  `layers` is a list of (weight, bias).  Returns the raw (…,62) output."""
    context = context.expand(x.shape[:-1] + (context.shape[-1],))
    h = torch.cat([context, x], dim=-1)
    for (w, b) in layers[:-1]:
        h = F.relu(F.linear(h, w, b))
    w, b = layers[-1]
    return F.linear(h, w, b)


def split_params(raw, count_bins=8, dims=2):
    """ConditionalDenseNN param_dims = [D*K, D*K, D*(K-1), D*K] consecutive slices."""
    k = dims * count_bins
    k1 = dims * (count_bins - 1)
    return raw[..., :k], raw[..., k:2 * k], raw[..., 2 * k:2 * k + k1], raw[..., 2 * k + k1:3 * k + k1]


def spline_coupling(layers, context, x, bound, inverse=False, count_bins=8):
    """pyro SplineCoupling._call / ._inverse with split_dim=1, identity=True on a 3-vector:
    x1 = x[...,:1] passes through; x[...,1:] go through the conditional spline whose
    parameters are dense_nn([context, x1]).  Returns (y, log|det dy/dx| of the FORWARD map)."""
    x1, x2 = x[..., :1], x[..., 1:]
    raw = dense_nn(layers, x1, context)
    w, h, d, l = spline_params(split_params(raw, count_bins, 2), 2, count_bins)
    y2, ld = rational_linear_spline(x2, w, h, d, l, bound, inverse=inverse)
    y = torch.cat([x1, y2], dim=-1)
    ld = ld.sum(-1)
    return y, (-ld if inverse else ld)
