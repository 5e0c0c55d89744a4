"""Oracle: SMPL blend-shapes + linear-blend skinning + the reference's 90-joint set.
Test infrastructure only.

PARITY UNPINNED for the LBS arithmetic: it lives in smplx==0.1.26 (requirements.txt:10),
not vendored under /root/reference and not installable here.  This restates the published
algorithm of smplx/lbs.py (lbs, blend_shapes, vertices2joints, batch_rigid_transform,
batch_rodrigues), smplx/body_models.py (SMPL.forward) and smplx/vertex_joint_selector.py
(SURVEY.md Appendix B), anchored on the reference's own wrapper models/smpl.py:13-41
(which IS restated line for line: get_skin, three extra regressors appended -> 90 joints)
and on the LBS identities in tests/test_oracle_smpl.py.

The op order is the "as written" one on purpose (dense einsum blend-shapes, dense
(M,207)x(207,20670) pose blend, 24-step chain, materialised (M,6890,4,4) skinning
transforms, dense extra-joint regressors): this is also the CPU baseline bench.py times.
"""
import torch
import torch.nn.functional as F

from .so3 import batch_rodrigues

# smplx/vertex_ids.py 'smplh' table, in VertexJointSelector order: face, feet, l-hand tips, r-hand tips
EXTRA_VERTEX_IDS = [332, 6260, 2800, 4071, 583,                 # nose reye leye rear lear
                    3216, 3226, 3387, 6617, 6624, 6787,          # LBigToe LSmallToe LHeel RBigToe RSmallToe RHeel
                    2746, 2319, 2445, 2556, 2673,                # l thumb index middle ring pinky
                    6191, 5782, 5905, 6016, 6133]                # r thumb index middle ring pinky


def blend_shapes(betas, shapedirs):
    return torch.einsum('bl,mkl->bmk', betas, shapedirs)


def vertices2joints(J_regressor, vertices):
    return torch.einsum('bik,ji->bjk', vertices, J_regressor)


def batch_rigid_transform(rot_mats, joints, parents):
    """smplx lbs.batch_rigid_transform: G_i = G_parent(i) @ [[R_i, J_i - J_parent(i)],[0,1]];
    A_i = G_i with the rest joint removed (so it acts on rest-pose vertices)."""
    joints = joints.unsqueeze(-1)
    rel = joints.clone()
    rel[:, 1:] -= joints[:, parents[1:]]
    B, J = rot_mats.shape[:2]
    M = torch.cat([F.pad(rot_mats.reshape(-1, 3, 3), [0, 0, 0, 1]),
                   F.pad(rel.reshape(-1, 3, 1), [0, 0, 0, 1], value=1.0)], dim=2).reshape(B, J, 4, 4)
    chain = [M[:, 0]]
    for i in range(1, J):
        chain.append(torch.matmul(chain[parents[i]], M[:, i]))
    G = torch.stack(chain, dim=1)
    posed = G[:, :, :3, 3]
    jh = F.pad(joints, [0, 0, 0, 1])
    A = G - F.pad(torch.matmul(G, jh), [3, 0, 0, 0, 0, 0, 0, 0])
    return posed, A


def lbs(betas, pose, model, pose2rot):
    """smplx lbs.lbs.  pose is (M,72) axis-angle if pose2rot else (M,24,3,3)."""
    M = max(betas.shape[0], pose.shape[0])
    v_shaped = model['v_template'] + blend_shapes(betas, model['shapedirs'])
    J = vertices2joints(model['J_regressor'], v_shaped)
    ident = torch.eye(3, dtype=betas.dtype)
    if pose2rot:
        rot = batch_rodrigues(pose.reshape(-1, 3)).view(M, -1, 3, 3)
    else:
        rot = pose.view(M, -1, 3, 3)
    pose_feature = (rot[:, 1:] - ident).reshape(M, -1)
    v_posed = v_shaped + torch.matmul(pose_feature, model['posedirs']).view(M, -1, 3)
    J_t, A = batch_rigid_transform(rot, J, model['parents'])
    W = model['lbs_weights'].unsqueeze(0).expand(M, -1, -1)
    T = torch.matmul(W, A.view(M, -1, 16)).view(M, -1, 4, 4)
    homo = torch.cat([v_posed, torch.ones(M, v_posed.shape[1], 1, dtype=betas.dtype)], dim=2)
    verts = torch.matmul(T, homo.unsqueeze(-1))[:, :, :3, 0]
    return verts, J_t


def smpl_forward(model, betas, body_pose, global_orient, pose2rot=True):
    """models/smpl.py:27-41 on top of smplx SMPL.forward: returns (vertices (M,6890,3), joints (M,90,3)).
    body_pose/global_orient: (M,69)/(M,3) axis-angle, or (M,23,3,3)/(M,1,3,3) with pose2rot=False."""
    full_pose = torch.cat([global_orient, body_pose], dim=1)
    M = max(betas.shape[0], global_orient.shape[0], body_pose.shape[0])
    if betas.shape[0] != M:
        betas = betas.expand(int(M / betas.shape[0]), -1)
    verts, joints = lbs(betas, full_pose, model, pose2rot)
    extra = verts[:, torch.tensor(EXTRA_VERTEX_IDS)]
    joints = torch.cat([joints, extra], dim=1)                                    # 24 + 21 = 45
    more = [vertices2joints(model[k], verts) for k in ('J_regressor_extra', 'J_regressor_cocoplus', 'J_regressor_h36m')]
    return verts, torch.cat([joints] + more, dim=1)                                # 45 + 9 + 19 + 17 = 90


def vertex_variance(vertices_samples):
    """utils/sampling_utils.py:22-33."""
    mean = vertices_samples.mean(dim=0)
    diff = vertices_samples - mean
    return torch.norm(diff, dim=-1).mean(dim=0), torch.sqrt((diff ** 2).mean(dim=0))
