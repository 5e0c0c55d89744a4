"""SO(3) helpers with the reference's names (utils/rigid_transform_utils.py), backed by the HIP kernels.

rot6d_to_rotmat :86-100, rotmat_to_rot6d :103-116, so3_hat :142-165, so3_vee :168-179, so3_exp :182-201,
so3_log :204-279, so3_xset :282-295, so3_log_abs_det_jacobian :298-314, plus smplx's batch_rodrigues.
Pure index/view helpers (rotmat_to_rot6d, hat, vee) are plain torch; everything with arithmetic calls
libhumaniflow_amd and therefore needs device tensors (no CPU fallback; CPU checks live in oracle/).
The cv2 / pytorch3d based `aa_*` driver helpers of the reference are out of scope.
"""
import math

import torch

from . import _lib


def _run(fn_name, x, in_cols, out_cols, dtype):
    _lib.require_cuda(x, 'input')
    lib = _lib.load()
    lead = x.shape[:-1] if in_cols[1] is None else x.shape[:-2]
    xin = x.detach().to(dtype).contiguous().reshape(-1, in_cols[0])
    n = xin.shape[0]
    out = torch.empty(n, out_cols, dtype=dtype, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(getattr(lib, fn_name)(_lib.dptr(xin), n, _lib.dptr(out), _lib.current_stream()), fn_name)
    return out, lead


def rot6d_to_rotmat_kernel(x):
    out, _ = _run('hmf_rot6d_to_rotmat', x.reshape(-1, 6), (6, None), 9, torch.float32)
    return out.view(-1, 3, 3)


def rot6d_to_rotmat(x):
    """(B,6) or (B,24*6) -> (B',3,3).  Differentiable (training: the global-rotation loss and everything downstream of glob_R)."""
    if torch.is_grad_enabled() and x.requires_grad:
        from .autograd import Rot6dToRotmat
        return Rot6dToRotmat.apply(x)
    return rot6d_to_rotmat_kernel(x)


def rotmat_to_rot6d(R, stack_columns=False):
    if stack_columns:
        return torch.cat([R[:, :, 0], R[:, :, 1]], dim=1)
    return R[:, :, :2].contiguous().view(-1, 6)


def so3_hat(v):
    z = torch.zeros_like(v[..., 0])
    return torch.stack([torch.stack([z, -v[..., 2], v[..., 1]], -1), torch.stack([v[..., 2], z, -v[..., 0]], -1),
                        torch.stack([-v[..., 1], v[..., 0], z], -1)], -2)


def so3_vee(m):
    return torch.stack((-m[..., 1, 2], m[..., 0, 2], -m[..., 0, 1]), -1)


def so3_exp(v):
    assert v.dtype == torch.double
    out, lead = _run('hmf_so3_exp', v, (3, None), 9, torch.float64)
    return out.view(*lead, 3, 3)


def so3_log(r, return_axis_angle=False):
    """Skew matrix (..., 3, 3), or the (…,3) vector when return_axis_angle."""
    assert r.dtype == torch.double
    out, lead = _run('hmf_so3_log', r.reshape(*r.shape[:-2], 9), (9, None), 3, torch.float64)
    v = out.view(*r.shape[:-2], 3)
    return v if return_axis_angle else so3_hat(v)


def so3_xset(x, k_max):
    x = x[None]
    n = x.norm(dim=-1, keepdim=True)
    k = torch.arange(1, k_max + 1, dtype=x.dtype, device=x.device)
    k = torch.cat([-k, k]).view([-1] + [1] * (x.dim() - 1))
    return x / n * (n + 2 * math.pi * k)


def so3_log_abs_det_jacobian(x):
    out, lead = _run('hmf_so3_log_abs_det_jacobian', x.double(), (3, None), 1, torch.float64)
    return out.view(*lead).to(x.dtype)


def batch_rodrigues(rot_vecs):
    out, _ = _run('hmf_batch_rodrigues', rot_vecs.reshape(-1, 3), (3, None), 9, torch.float32)
    return out.view(-1, 3, 3)
