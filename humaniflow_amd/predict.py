"""Per-image prediction driver for the hot path: the model/SMPL part of predict/predict_humaniflow.py:23-273.

The reference driver also runs HRNet / Mask-RCNN (2D joints), Canny (edge map) and a pytorch3d renderer; those are outside the
path.  This counterpart starts from the proxy representation (1, 18, 256, 256) the reference builds at
predict_humaniflow.py:101-109 and produces what the path produces:
  * the six model keys of utils/predict_utils.py:19-24 (`pose_axisangle_point_est`, `pose_rotmats_point_est`, `shape_mode`,
    `glob_rotmat`, `cam_wp`, `input_feats`) in `<name>_pred.npz`,
  * point-estimate / T-pose / sample vertices and the per-vertex sample uncertainty of :128-131,147,160-169.
"""
import os

import numpy as np
import torch

from .smpl import compute_vertex_variance_from_samples


@torch.no_grad()
def predict_humaniflow(humaniflow_model, smpl_model, proxy_inputs, names, save_dir=None, num_pred_samples=50, noise=None):
    """proxy_inputs: (B, 18, S, S) on the model's device; processed one image at a time like the reference loop (:65)."""
    humaniflow_model.eval()
    humaniflow_model.pose_so3flow_transform_modules.eval()                                    # :60-61
    results = []
    N = num_pred_samples
    for i, name in enumerate(names):
        x = proxy_inputs[i:i + 1]
        nz = None if noise is None else {k: v[:, i:i + 1].contiguous() for k, v in noise.items()}
        pred = humaniflow_model(x, num_samples=N, use_shape_mode_for_samples=True, return_input_feats=True, noise=nz)   # :112-115
        out_np = {k: pred[k].cpu().numpy()[0] for k in ('pose_axisangle_point_est', 'pose_rotmats_point_est', 'shape_mode',
                                                          'glob_rotmat', 'cam_wp', 'input_feats')}          # predict_utils.py:19-24
        if save_dir is not None:
            os.makedirs(save_dir, exist_ok=True)
            np.savez(os.path.join(save_dir, os.path.splitext(name)[0] + '_pred.npz'), proxy_rep=x.cpu().numpy()[0], **out_np)
        if N > 0:                                                                                             # :128-131 and :160-165 in one launch
            pe, sa = smpl_model.forward_samples_and_point_est(pred)
        else:
            pe = smpl_model(body_pose=pred['pose_rotmats_point_est'], global_orient=pred['glob_rotmat'][:, None, :, :],
                            betas=pred['shape_mode'], pose2rot=False)
        res = {'name': name, 'pred': pred, 'vertices_point_est': pe.vertices, 'joints_point_est': pe.joints,
               'tpose_vertices_point_est': smpl_model.tpose(pred['shape_mode'])}                              # :147
        if N > 0:
            avg, xyz = compute_vertex_variance_from_samples(sa.vertices)                                      # :169
            res.update(vertices_samples=sa.vertices, joints_samples=sa.joints, per_vertex_avg_dist_from_mean=avg,
                       per_vertex_xyz_variance=xyz)
        results.append(res)
    return results
