"""Oracle: the differentiable part of the reference's training step on CPU, float64.  Test infrastructure only.

Restates train/train_humaniflow.py:248-330 (forward with compute_for_loglik under grad, SMPL joints of the point estimate and of the
samples, flip about x + orthographic projection, loss) and losses/humaniflow_loss.py:23-79 with torch autograd through the oracle's own
forward (oracle/model.py -- the same code that is pinned for inference, run with gradients enabled and float64 defaults so that
torch's autograd Jacobians are the ground truth the HIP reverse sweeps are checked against).
The loss restatement is PINNED by tests/golden/loss_golden.npz (the reference's own HumaniflowLoss, values and gradients).
"""
import contextlib

import torch

from . import model as omodel
from . import smpl as osmpl

ALL_JOINTS_TO_COCO_MAP = [24, 26, 25, 28, 27, 16, 17, 18, 19, 20, 21, 1, 2, 4, 5, 7, 8]        # utils/label_conversions.py:17


@contextlib.contextmanager
def float64_defaults():
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        yield
    finally:
        torch.set_default_dtype(old)


def forward_grad(sd, cfg, **kw):
    """oracle.model.forward with autograd enabled (the pinned function minus its no_grad decorator)."""
    return omodel.forward.__wrapped__(sd, cfg, **kw)


def log_prob_SO3_grad(sd, cfg, ctx, R):
    return omodel.log_prob_SO3.__wrapped__(sd, cfg, ctx, R)


def humaniflow_loss(loss_cfg, img_wh, pose_logprob, shape_mode, shape_log_std, shape_target, joints2D_pred, joints2D_target, joints2D_vis,
                    glob_pred, glob_target):
    """losses/humaniflow_loss.py:23-79 on tensors: pose_logprob (B,23) = log_prob of every conditioned distribution."""
    B, P = pose_logprob.shape
    red = loss_cfg.REDUCTION
    pose_nll = -pose_logprob.sum()
    if red == 'mean':
        pose_nll = pose_nll / (B * P)
    shape_nll = -(torch.distributions.Normal(shape_mode, torch.exp(shape_log_std), validate_args=False).log_prob(shape_target).sum(dim=1))
    shape_nll = shape_nll.mean() if red == 'mean' else shape_nll.sum()
    tgt = joints2D_target[:, None].expand_as(joints2D_pred)
    vis = joints2D_vis[:, None, :].expand(-1, joints2D_pred.shape[1], -1)
    mse = lambda a, b: ((a - b) ** 2).mean() if red == 'mean' else ((a - b) ** 2).sum()
    j2d = mse(joints2D_pred[vis, :], (2.0 * tgt[vis, :]) / img_wh - 1.0)
    glob = mse(glob_pred, glob_target)
    W = loss_cfg.WEIGHTS
    return pose_nll * W.POSE + shape_nll * W.SHAPE + j2d * W.JOINTS2D + glob * W.GLOB_ROTMATS


def project_coco(joints, cam):
    """train/train_humaniflow.py:291-297: COCO joints, 180 degrees about x (x, -y, -z), orthographic s * (xy + t)."""
    jc = joints[:, ALL_JOINTS_TO_COCO_MAP] * torch.tensor([1.0, -1.0, -1.0], dtype=joints.dtype)
    return cam[:, None, 0:1] * (jc[:, :, :2] + cam[:, None, 1:])


def train_step_loss(sd, cfg, smpl_model, loss_cfg, img_wh, input_feats, targets, noise, num_samples):
    """The scalar the reference back-propagates (J2D_LOSS_ON = 'point_est+samples'), from encoder features onwards.
    targets: pose_rotmats (B,23,3,3), glob_rotmats (B,3,3), shape (B,nb), joints2D (B,17,2) pixels, joints2D_vis (B,17)."""
    N = num_samples
    out = forward_grad(sd, cfg, input_feats=input_feats, num_samples=N, noise=noise, compute_for_loglik=True,
                       shape_for_loglik=targets['shape'], pose_R_for_loglik=targets['pose_rotmats'], glob_R_for_loglik=targets['glob_rotmats'])
    lp = log_prob_SO3_grad(sd, cfg, out['flow_context_for_loglik'], targets['pose_rotmats'].double())
    g = out['glob_rotmat']
    B = g.shape[0]
    # the point-estimate chain carries no gradient in the reference (torch.set_grad_enabled(grad_for_pose_point_est), :294)
    _, j_pe = osmpl.smpl_forward(smpl_model, out['shape_mode'], out['pose_rotmats_point_est'].detach(), g[:, None], pose2rot=False)
    j2d_pe = project_coco(j_pe, out['cam_wp'])
    _, j_s = osmpl.smpl_forward(smpl_model, out['shape_samples'].reshape(B * N, -1), out['pose_rotmats_samples'].reshape(B * N, 23, 3, 3),
                                g[:, None].expand(-1, N, -1, -1).reshape(B * N, 1, 3, 3), pose2rot=False)
    j2d_s = project_coco(j_s, out['cam_wp'][:, None].expand(-1, N, -1).reshape(B * N, 3)).reshape(B, N, 17, 2)
    j2d = torch.cat([j2d_pe[:, None], j2d_s], dim=1)                                       # :312-314
    return humaniflow_loss(loss_cfg, img_wh, lp, out['shape_mode'], out['shape_log_std'], targets['shape'], j2d, targets['joints2D'],
                           targets['joints2D_vis'], g, targets['glob_rotmats']), out
