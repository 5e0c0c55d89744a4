"""Oracle: SO(3) / so(3) closed-form maps (CPU, torch).  Test infrastructure only.

Restates utils/rigid_transform_utils.py and utils/lin_alg_utils.py of the
reference, dtype hop for dtype hop.  Every function cites the reference lines it
follows; quirks are reproduced on purpose (SURVEY.md section 8a "Quirks").
PINNED by tests/golden/so3_golden.npz (generated from the reference itself).
"""
import math
from itertools import product

import torch
import torch.nn.functional as F


def rot6d_to_rotmat(x):
    """utils/rigid_transform_utils.py:86-100.  6D is row-major pairs
    [R11,R12,R21,R22,R31,R32]; Gram-Schmidt; columns stacked."""
    x = x.reshape(-1, 3, 2)
    a1, a2 = x[:, :, 0], x[:, :, 1]
    b1 = F.normalize(a1)                                   # a1 / max(|a1|, 1e-12)
    dot = torch.einsum('bi,bi->b', b1, a2).unsqueeze(-1)
    b2 = F.normalize(a2 - dot * b1)
    b3 = torch.linalg.cross(b1, b2, dim=-1)
    return torch.stack((b1, b2, b3), dim=-1)


def rotmat_to_rot6d(R):
    """utils/rigid_transform_utils.py:103-116 with stack_columns=False."""
    return R[:, :, :2].contiguous().view(-1, 6)


def batch_trace(m):
    """utils/lin_alg_utils.py:4-7."""
    return m[..., 0, 0] + m[..., 1, 1] + m[..., 2, 2]


def so3_hat(v):
    """utils/rigid_transform_utils.py:142-165."""
    z = torch.zeros_like(v[..., 0])
    rows = [torch.stack([z, -v[..., 2], v[..., 1]], -1),
            torch.stack([v[..., 2], z, -v[..., 0]], -1),
            torch.stack([-v[..., 1], v[..., 0], z], -1)]
    return torch.stack(rows, -2)


def so3_vee(m):
    """utils/rigid_transform_utils.py:168-179."""
    return torch.stack((-m[..., 1, 2], m[..., 0, 2], -m[..., 0, 1]), -1)


def so3_exp(v):
    """utils/rigid_transform_utils.py:182-201 (Rodrigues, fp64 only).
    Quirk 1: the Taylor branch is evaluated on theta AFTER it was replaced by 1."""
    assert v.dtype == torch.float64
    theta = v.norm(p=2, dim=-1)
    mask = theta > 1e-10
    theta = torch.where(mask, theta, torch.ones_like(theta))
    alpha = torch.where(mask, torch.sin(theta) / theta, 1 - theta ** 2 / 6)
    beta = torch.where(mask, (1 - torch.cos(theta)) / theta ** 2, 0.5 - theta ** 2 / 24)
    eye = torch.eye(3, dtype=v.dtype)
    x = so3_hat(v)
    return eye + alpha[..., None, None] * x + beta[..., None, None] * x @ x


def so3_log_pi(r, theta):
    """utils/rigid_transform_utils.py:240-279: near-pi branch; magnitudes from the
    diagonal of the symmetric part, signs by an 8-way search that re-exponentiates."""
    sym = 0.5 * (r + r.transpose(-1, -2))
    eye = torch.eye(3, dtype=r.dtype).expand_as(sym)
    z = theta ** 2 / (1 - torch.cos(theta)) * (sym - eye)
    q1, q2, q3 = z[..., 0, 0], z[..., 1, 1], z[..., 2, 2]
    x1 = torch.sqrt(torch.clamp(q1 - q2 - q3, min=1e-8) / 2)
    x2 = torch.sqrt(torch.clamp(-q1 + q2 - q3, min=1e-8) / 2)
    x3 = torch.sqrt(torch.clamp(-q1 - q2 + q3, min=1e-8) / 2)
    x = torch.stack([x1, x2, x3], -1).view(-1, 3)
    r = r.view(-1, 3, 3)
    signs = torch.tensor(list(product([0, 1], repeat=3)), dtype=x.dtype) * 2 - 1
    x_stack = signs.view(8, 1, 3) * x[None]
    r_stack = so3_exp(x_stack)
    diff = (r[None] - r_stack).pow(2).sum(-1).sum(-1)
    sel = torch.argmin(diff, dim=0)
    x = x_stack[sel, torch.arange(len(sel))]
    return so3_hat(x)


def so3_log(r):
    """utils/rigid_transform_utils.py:204-237 for a 1-D batch (B,3,3) fp64 (the only
    shape the reference's callers use; its mask indexing is 1-D-batch only).
    Returns the skew matrix (B,3,3)."""
    assert r.dtype == torch.float64 and r.dim() == 3
    anti = 0.5 * (r - r.transpose(-1, -2))
    cos_t = (0.5 * (batch_trace(r)[..., None, None] - 1)).clamp(-1, 1)
    theta = torch.acos(cos_t)
    ratio = theta / torch.sin(theta)
    small = theta[..., 0, 0] < 1e-20
    ratio = torch.where(small[:, None, None], 1 + theta ** 2 / 6, ratio)
    log = ratio * anti
    near_pi = (math.pi - theta[..., 0, 0]).abs() < 1e-2
    if near_pi.any():
        idx = near_pi.nonzero()[:, 0]
        log = log.clone()
        log[idx] = so3_log_pi(r[idx], theta[idx])
    return log


def so3_xset(x, k_max=1):
    """utils/rigid_transform_utils.py:282-295.  Quirk 2: NaN at x == 0."""
    x = x[None]
    n = x.norm(dim=-1, keepdim=True)
    k = torch.arange(1, k_max + 1, dtype=x.dtype)
    k = torch.cat([-k, k]).view([-1] + [1] * (x.dim() - 1))
    return x / n * (n + 2 * math.pi * k)


def so3_log_abs_det_jacobian(x):
    """utils/rigid_transform_utils.py:298-314.  Quirk 1: returns log(11/12) at |x|<=1e-10."""
    n = x.double().norm(dim=-1)
    mask = n > 1e-10
    n = torch.where(mask, n, torch.ones_like(n))
    ratio = torch.where(mask, (2 - 2 * torch.cos(n)) / n ** 2, 1 - n ** 2 / 12)
    return torch.log(ratio).to(x.dtype)


def so3_inverse_set(R, support_radius):
    """models/norm_flows/transforms/so3_exp_transform.py:33-41:
    (x, xset(2,...), mask(2,...)) with out-of-support pre-images zeroed."""
    x = so3_vee(so3_log(R))
    xset = so3_xset(x, 1)
    mask = xset.norm(dim=-1) < support_radius
    xset = xset.masked_fill(torch.logical_not(mask[..., None]), 0)
    return x, xset, mask


def batch_rodrigues(rot_vecs):
    """smplx==0.1.26 lbs.batch_rodrigues (NOT vendored; restated).
    Quirk 3: angle = |v + 1e-8| (added per component), fp32 throughout."""
    angle = torch.norm(rot_vecs + 1e-8, dim=1, keepdim=True)
    d = rot_vecs / angle
    cos = torch.cos(angle)[:, None]
    sin = torch.sin(angle)[:, None]
    rx, ry, rz = torch.split(d, 1, dim=1)
    zeros = torch.zeros_like(rx)
    K = torch.cat([zeros, -rz, ry, rz, zeros, -rx, -ry, rx, zeros], dim=1).view(-1, 3, 3)
    ident = torch.eye(3, dtype=rot_vecs.dtype)[None]
    return ident + sin * K + (1 - cos) * torch.bmm(K, K)
