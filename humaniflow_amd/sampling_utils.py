"""Sample selection helpers of the predict driver (utils/sampling_utils.py:10-74), on the device.

`compute_vertex_variance_from_samples` is the fused HIP reduction in smpl.py (hmf_vertex_variance).  The two functions here touch
17 joints per sample -- a handful of small device ops; only the final gather moves the (N,6890,3) meshes."""
import torch

from .evaluate import ALL_JOINTS_TO_COCO_MAP, flip_about_x, orthographic_project_torch, undo_keypoint_normalisation
from .smpl import compute_vertex_variance_from_samples  # noqa: F401  (re-exported under the reference's module name)


def quat_to_rotmat(quat):
    """utils/rigid_transform_utils.py:119-139: (B,4) quaternions (w, x, y, z), normalised here -> (B,3,3).
    Written in the vector form  R = (w^2 - |v|^2) I + 2 v v^T + 2 w [v]_x  (same matrix as the reference's component-wise table)."""
    q = quat / quat.norm(p=2, dim=1, keepdim=True)
    w, v = q[:, :1], q[:, 1:]
    eye = torch.eye(3, dtype=q.dtype, device=q.device)
    cross = torch.zeros(q.shape[0], 3, 3, dtype=q.dtype, device=q.device)
    cross[:, 0, 1], cross[:, 0, 2], cross[:, 1, 2] = -v[:, 2], v[:, 1], -v[:, 0]
    cross = cross - cross.transpose(1, 2)
    return (w * w - (v * v).sum(1, keepdim=True))[:, :, None] * eye + 2.0 * v[:, :, None] * v[:, None, :] + 2.0 * w[:, :, None] * cross


def so3_uniform_sampling_torch(num_matrices, device=None, generator=None):
    """utils/sampling_utils.py:10-19: Haar-uniform rotations from normalised Gaussian quaternions."""
    return quat_to_rotmat(torch.randn(num_matrices, 4, device=device, generator=generator))


def convert_heatmaps_to_2Djoints_coordinates_torch(joints2D_heatmaps, eps=1e-6, gaussian_heatmaps=False):
    """utils/label_conversions.py:128-163: arg-max location (x, y) of every heat-map and its visibility (max > eps)."""
    n, k, _, width = joints2D_heatmaps.shape
    max_vals, idx = torch.max(joints2D_heatmaps.reshape(n, k, -1), dim=-1)
    joints2D = torch.stack([(idx % width).float(), torch.floor(idx / float(width))], dim=-1)
    vis = max_vals > eps
    if gaussian_heatmaps:
        vis = torch.logical_and(vis, max_vals > (max_vals.max() - 1e-2))
    joints2D = torch.where(vis[..., None], joints2D, torch.full_like(joints2D, -1000.0))      # invisible joints: -1000 (:161)
    return joints2D, vis


def joints2D_error_sorted_verts_sampling(pred_vertices_samples, pred_joints_samples, input_joints2D_heatmaps, pred_cam_wp):
    """utils/sampling_utils.py:36-74: sort the N mesh samples by the largest image-plane distance between their projected COCO joints
    and the visible input 2D joints (ascending)."""
    j = flip_about_x(pred_joints_samples[:, ALL_JOINTS_TO_COCO_MAP, :])
    j2d = undo_keypoint_normalisation(orthographic_project_torch(j, pred_cam_wp), input_joints2D_heatmaps.shape[-1])
    inp, vis = convert_heatmaps_to_2Djoints_coordinates_torch(input_joints2D_heatmaps, eps=1e-6)
    err = torch.norm(j2d[:, vis[0], :] - inp[:, vis[0], :], dim=-1)                     # (N, visible joints)
    order = torch.sort(err.max(dim=-1).values, descending=False).indices
    return pred_vertices_samples[order]
