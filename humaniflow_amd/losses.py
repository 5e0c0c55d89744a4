"""HumaniflowLoss with the reference's constructor and call signature (losses/humaniflow_loss.py:5-79).

    criterion = HumaniflowLoss(cfg.LOSS, img_wh=cfg.DATA.PROXY_REP_SIZE)
    loss = criterion(target_dict, pred_dict)

pred_dict:   'pose_dist' (list of 23 conditioned SO(3) flow distributions), 'shape_dist' (torch Normal), 'joints2D' (B,S,17,2),
             'glob_rotmats' (B,3,3) [+ 'verts', 'joints3D' with APPLY_POINT_EST_LOSS]
target_dict: 'pose_params_rotmats' (B,23,3,3), 'shape_params' (B,nb), 'joints2D' (B,17,2) pixels, 'joints2D_vis' (B,17) bool,
             'glob_rotmats' (B,3,3) [+ 'verts', 'joints3D']
Arithmetic as the reference, term for term; the one difference is HOW the pose NLL is evaluated when the 23 distributions are this
package's conditioned flow objects: one batched launch (and one backward launch) instead of 23 -- same numbers.
Pinned by tests/golden/loss_golden.npz (values and input gradients produced by the reference's own class).
"""
import torch
from torch import nn


class HumaniflowLoss(nn.Module):
    def __init__(self, loss_cfg, img_wh):
        super().__init__()
        self.loss_cfg = loss_cfg
        self.img_wh = img_wh
        self.joints2D_loss = nn.MSELoss(reduction=loss_cfg.REDUCTION)                 # :17-18
        self.glob_rotmats_loss = nn.MSELoss(reduction=loss_cfg.REDUCTION)
        if loss_cfg.APPLY_POINT_EST_LOSS:                                             # :20-22
            self.verts3D_loss = nn.MSELoss(reduction=loss_cfg.REDUCTION)
            self.joints3D_loss = nn.MSELoss(reduction=loss_cfg.REDUCTION)

    @staticmethod
    def pose_log_probs(pred_pose_dists, target_pose_rotmats):
        """(B, 23) log-densities: one launch when the list is a model's own conditioned distributions, else the reference's loop."""
        first = pred_pose_dists[0]
        parent = getattr(first, 'parent', None)
        if parent is not None and getattr(first, 'group', False) and len(pred_pose_dists) == parent[0].shape[1] and \
                all(getattr(d, 'parent', None) is not None and d.parent[0] is parent[0] and d.parent[1] == j and d.group
                    for j, d in enumerate(pred_pose_dists)):
            return first.model.log_prob_all_parts(parent[0], target_pose_rotmats)
        return torch.stack([d.log_prob(target_pose_rotmats[:, j]) for j, d in enumerate(pred_pose_dists)], dim=1)

    def forward(self, target_dict, pred_dict):
        cfg = self.loss_cfg
        # ----------------- Pose NLL (:25-38) -----------------
        target_pose_rotmats = target_dict['pose_params_rotmats'].double()
        batch_size, num_bodyparts = target_pose_rotmats.shape[:2]
        pose_nll = -self.pose_log_probs(pred_dict['pose_dist'], target_pose_rotmats).sum()
        if cfg.REDUCTION == 'mean':
            pose_nll = pose_nll / (batch_size * num_bodyparts)
        # ----------------- Shape NLL (:40-46) -----------------
        shape_nll = -(pred_dict['shape_dist'].log_prob(target_dict['shape_params']).sum(dim=1))
        shape_nll = torch.mean(shape_nll) if cfg.REDUCTION == 'mean' else torch.sum(shape_nll)
        # ----------------- Visible 2D joints reprojection MSE (:48-60) -----------------
        pred_joints2D = pred_dict['joints2D']
        target_joints2D = target_dict['joints2D'][:, None, :, :].expand_as(pred_joints2D)
        vis = target_dict['joints2D_vis'][:, None, :].expand(-1, pred_joints2D.shape[1], -1)
        # the reference compacts the visible joints with a boolean index (a host synchronisation for the count) and applies nn.MSELoss;
        # the same value as a masked reduction stays on the device (and inside a captured HIP graph)
        d2 = (pred_joints2D - ((2.0 * target_joints2D) / self.img_wh - 1.0)) ** 2 * vis[..., None].to(pred_joints2D.dtype)
        joints2D_loss = d2.sum()
        if cfg.REDUCTION == 'mean':
            joints2D_loss = joints2D_loss / (vis.sum() * d2.shape[-1]).clamp(min=1).to(pred_joints2D.dtype)
        elif cfg.REDUCTION != 'sum':
            raise NotImplementedError("humaniflow_amd: LOSS.REDUCTION '%s' for the 2D-joint term (mean / sum are built)" % cfg.REDUCTION)
        glob_rotmats_loss = self.glob_rotmats_loss(pred_dict['glob_rotmats'], target_dict['glob_rotmats'])     # :63
        total = pose_nll * cfg.WEIGHTS.POSE + shape_nll * cfg.WEIGHTS.SHAPE + joints2D_loss * cfg.WEIGHTS.JOINTS2D \
            + glob_rotmats_loss * cfg.WEIGHTS.GLOB_ROTMATS
        if cfg.APPLY_POINT_EST_LOSS:                                                  # :72-77
            total = total + self.verts3D_loss(pred_dict['verts'], target_dict['verts']) * cfg.WEIGHTS.VERTS3D
            total = total + self.joints3D_loss(pred_dict['joints3D'], target_dict['joints3D']) * cfg.WEIGHTS.JOINTS3D
        return total
