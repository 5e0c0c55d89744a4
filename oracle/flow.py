"""Oracle: the per-body-part conditional flow on so(3) and its push-forward to SO(3).
Test infrastructure only.

Follows models/norm_flows/pyro_conditional_norm_flow.py:21-129 (flow construction and
forward_trans), models/norm_flows/transforms/scaled_radial_tanh_transform.py:28-59,
transforms/to_transform.py:9-29, transforms/so3_exp_transform.py:30-50 and
models/norm_flows/local_diffeo_transformed_distribution.py:72-142.
radial-tanh + push-forward log-prob plumbing are PINNED (tests/golden/flow_golden.npz);
the spline couplings inherit "parity unpinned" from oracle/spline.py.
"""
import math

import torch

from . import so3
from .spline import spline_coupling

LOG_2PI = math.log(2 * math.pi)


def permutations(event_dim, num_transforms):
    """pyro_conditional_norm_flow.py:46-47,60-62: cyclic shifts idx[i:]+idx[:i], one per transform."""
    idx = list(range(event_dim))
    return [idx[i % event_dim:] + idx[:i % event_dim] for i in range(num_transforms)]


def radial_tanh_forward(x, radius):
    """scaled_radial_tanh_transform.py:28-35."""
    n = x.norm(dim=-1, keepdim=True)
    mask = n > 1e-7
    n = torch.where(mask, n, torch.ones_like(n))
    return torch.where(mask, torch.tanh(n / radius) * (x / n) * radius, x)


def radial_tanh_inverse(y, radius):
    """scaled_radial_tanh_transform.py:37-46 (atanh in fp64, cast back)."""
    dt = y.dtype
    y = y.double()
    n = y.norm(dim=-1, keepdim=True)
    mask = n > 1e-7
    n = torch.where(mask, n, torch.ones_like(n))
    return torch.where(mask, torch.atanh(n / radius) * (y / n) * radius, y).to(dt)


def radial_tanh_log_abs_det(x, y, radius):
    """scaled_radial_tanh_transform.py:48-59."""
    xn = x.norm(dim=-1)
    yn = y.norm(dim=-1)
    ld = 2 * (torch.log(yn) - torch.log(xn)) + torch.log1p(-((yn / radius) ** 2))
    return torch.where(yn > 1e-7, ld, torch.zeros_like(ld))


def plu_matrices(perm, LU):
    """models/norm_flows/transforms/linearPLU_transform.py:23-33: P = eye[perm], L = tril(LU, -1) + I, U = triu(LU).  LU (..., 3, 3)."""
    P = torch.eye(3, dtype=LU.dtype)[torch.tensor(perm)]
    return P, LU.tril(diagonal=-1) + torch.eye(3, dtype=LU.dtype), LU.triu()


def plu_forward(perm, LU, x):
    """ConditionedLinearPLUTransform._call (:35-50): y = P L U x."""
    P, L, U = plu_matrices(perm, LU)
    return torch.matmul(P @ L @ U, x[..., None])[..., 0]


def plu_inverse(perm, LU, y):
    """ConditionedLinearPLUTransform._inverse (:52-74) and log_abs_det_jacobian (:76-82): returns (x, sum log|u_ii|)."""
    P, L, U = plu_matrices(perm, LU)
    LUx = torch.matmul(P.T, y[..., None])
    Ux = torch.linalg.solve_triangular(L.expand(LUx.shape[:-2] + (3, 3)), LUx, upper=False)
    x = torch.linalg.solve_triangular(U.expand(LUx.shape[:-2] + (3, 3)), Ux, upper=True)
    return x[..., 0], LU.diagonal(dim1=-2, dim2=-1).abs().log().sum(dim=-1)


def conditional_lu(layers, context):
    """ConditionalLinearPLUTransform.condition (:138-150): DenseNN(context) -> (…,3,3), diagonal through softplus(beta=0.75)."""
    import torch.nn.functional as F
    h = context
    for (w, b) in layers[:-1]:
        h = F.relu(F.linear(h, w, b))
    LU = F.linear(h, layers[-1][0], layers[-1][1])
    LU = LU.view(LU.shape[:-1] + (3, 3)).clone()
    LU[..., range(3), range(3)] = F.softplus(LU[..., range(3), range(3)], beta=0.75)
    return LU


def additive_coupling(layers, context, x, inverse=False):
    """transforms/additive_coupling_transform.py:28-66 with split_dim = 1: y1 = x1, y2 = x2 +/- nn([context, x1]); log-det 0."""
    from .spline import dense_nn
    mean = dense_nn(layers, x[..., :1], context)
    return torch.cat([x[..., :1], x[..., 1:] - mean if inverse else x[..., 1:] + mean], dim=-1)


class PartFlow:
    """One body part's flow: [Permute | LinearPLU | ConditionalLinearPLU, SplineCoupling | AdditiveCoupling] x T  ->  ScaledRadialTanh
    (models/norm_flows/pyro_conditional_norm_flow.py:55-113).  `couplings` is a list of T hyper-networks, each a list of (weight, bias);
    `plu` (optional) a list of T entries: a (3,3) LU tensor (linear_plu) or a list of (weight, bias) DenseNN layers (conditional)."""

    def __init__(self, couplings, radius, base_std, count_bins=8, transform_type='spline_coupling', plu=None):
        self.couplings = couplings
        self.radius = radius
        self.base_std = base_std
        self.count_bins = count_bins
        self.perms = permutations(3, len(couplings))
        self.transform_type = transform_type
        self.plu = plu

    def _lu(self, t, context):
        p = self.plu[t]
        return p if torch.is_tensor(p) else conditional_lu(p, context)

    # -- forward (sampling) direction: pyro_conditional_norm_flow.py:120-129 ------------
    def forward(self, z, context):
        return radial_tanh_forward(self.forward_couplings(z, context), self.radius)

    def forward_couplings(self, z, context):
        """The [permutation, coupling] x T stack alone (the pyro part of the flow), before the reference's own radial tanh."""
        x = z
        for t, (perm, layers) in enumerate(zip(self.perms, self.couplings)):
            if self.plu is None:
                x = x.index_select(-1, torch.tensor(perm))             # pyro Permute._call
            else:
                x = plu_forward(perm, self._lu(t, context), x)
            if self.transform_type == 'spline_coupling':
                x, _ = spline_coupling(layers, context, x, self.radius, count_bins=self.count_bins)
            else:
                x = additive_coupling(layers, context, x)
        return x

    # -- density on the algebra: torch TransformedDistribution.log_prob ------------------
    def log_prob_so3(self, v, context):
        """log p(v | context), v (...,3) fp32 on the Lie algebra (|v| < radius)."""
        x = radial_tanh_inverse(v, self.radius)
        return self.log_prob_couplings(x, context) - radial_tanh_log_abs_det(x, v, self.radius)

    def log_prob_couplings(self, u, context):
        """log density of the base Normal pushed through the [permutation, coupling] x T stack alone, at the pre-tanh vector u."""
        lp = 0.0
        y = u
        for t in reversed(range(len(self.couplings))):
            perm, layers = self.perms[t], self.couplings[t]
            if self.transform_type == 'spline_coupling':
                x, ld = spline_coupling(layers, context, y, self.radius, inverse=True, count_bins=self.count_bins)
                lp = lp - ld
            else:
                x = additive_coupling(layers, context, y, inverse=True)
            if self.plu is None:
                inv = [perm.index(i) for i in range(3)]
                y = x.index_select(-1, torch.tensor(inv))              # Permute inverse, log-det 0
            else:
                y, ld = plu_inverse(perm, self._lu(t, context), x)
                lp = lp - ld
        # pyro_conditional_norm_flow.py:49-52: Independent(Normal(0, base_std * ones(3)), 1)
        base = torch.distributions.Normal(torch.zeros(3), torch.ones(3) * self.base_std,
                                          validate_args=False).log_prob(y).sum(-1)
        return lp + base

    # -- density on the group: local_diffeo_transformed_distribution.py:91-142 -----------
    def log_prob_SO3(self, R, context):
        """log p(R | context), R (B,3,3) fp64 -> (B,) fp32; logsumexp over the 3 pre-images."""
        x, xset, mask = so3.so3_inverse_set(R, self.radius)
        x_term = -so3.so3_log_abs_det_jacobian(x).float() + self.log_prob_so3(x.float(), context).float()
        xs_ld = -so3.so3_log_abs_det_jacobian(xset).float()
        xs_lp = self.log_prob_so3(xset.float(), context).float()
        xs_term = torch.where(mask, xs_ld + xs_lp, torch.tensor([float('-inf')]))
        terms = torch.cat([x_term[None], xs_term])
        return torch.logsumexp(terms, dim=0)

    # -- sampling on the group: local_diffeo_transformed_distribution.py:72-82 ------------
    def sample_SO3(self, eps, context):
        """eps standard normal (...,3) fp32 -> R (...,3,3) fp64; also returns the algebra vector."""
        v = self.forward(eps * self.base_std, context)
        return so3.so3_exp(v.double()), v
