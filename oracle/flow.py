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


class PartFlow:
    """One body part's flow: [Permute, SplineCoupling] x T  ->  ScaledRadialTanh.
    `couplings` is a list of T hyper-networks, each a list of (weight, bias)."""

    def __init__(self, couplings, radius, base_std, count_bins=8):
        self.couplings = couplings
        self.radius = radius
        self.base_std = base_std
        self.count_bins = count_bins
        self.perms = permutations(3, len(couplings))

    # -- forward (sampling) direction: pyro_conditional_norm_flow.py:120-129 ------------
    def forward(self, z, context):
        x = z
        for perm, layers in zip(self.perms, self.couplings):
            x = x.index_select(-1, torch.tensor(perm))                 # pyro Permute._call
            x, _ = spline_coupling(layers, context, x, self.radius, count_bins=self.count_bins)
        return radial_tanh_forward(x, self.radius)

    # -- density on the algebra: torch TransformedDistribution.log_prob ------------------
    def log_prob_so3(self, v, context):
        """log p(v | context), v (...,3) fp32 on the Lie algebra (|v| < radius)."""
        lp = 0.0
        y = v
        x = radial_tanh_inverse(y, self.radius)
        lp = lp - radial_tanh_log_abs_det(x, y, self.radius)
        y = x
        for perm, layers in zip(reversed(self.perms), reversed(self.couplings)):
            x, ld = spline_coupling(layers, context, y, self.radius, inverse=True, count_bins=self.count_bins)
            lp = lp - ld
            inv = [perm.index(i) for i in range(3)]
            y = x.index_select(-1, torch.tensor(inv))                  # Permute inverse, log-det 0
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
