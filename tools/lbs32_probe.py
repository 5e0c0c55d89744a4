import sys, torch, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import humaniflow_amd as H
from humaniflow_amd import _lib
from oracle.so3 import so3_exp
lib = _lib.load(); dev = torch.device('cuda:0')
smpl = H.SMPL(model_data=H.synthetic_smpl_model(seed=0), create_transl=False).to(dev)
for M in (1, 31, 33, 100, 3232):
    g = torch.Generator().manual_seed(M)
    R = so3_exp(torch.randn(M * 24, 3, generator=g, dtype=torch.float64) * 0.5).float().view(M, 24, 3, 3).to(dev)
    betas = (torch.randn(M, 10, generator=g) * 1.25).to(dev)
    a = smpl(body_pose=R[:, 1:], global_orient=R[:, :1], betas=betas, pose2rot=False, impl=0)
    b = smpl(body_pose=R[:, 1:], global_orient=R[:, :1], betas=betas, pose2rot=False, impl=4)
    print(M, 'verts equal', torch.equal(a.vertices, b.vertices), 'joints equal', torch.equal(a.joints, b.joints), (a.vertices - b.vertices).abs().max().item())
M = 3232
ts = {0: [], 4: []}
for rnd in range(6):
    for impl in (0, 4):
        lib.hmf_profile_enable(1)
        smpl(body_pose=R[:, 1:], global_orient=R[:, :1], betas=betas, pose2rot=False, impl=impl)
        ms = np.zeros(16); tag = np.zeros(16, np.int32); rows = np.zeros(16, np.int32)
        n = lib.hmf_profile_collect(ms.ctypes.data_as(_lib.c_double_p), _lib.iptr(tag), _lib.iptr(rows), 16)
        lib.hmf_profile_enable(0)
        if rnd: ts[impl].append([ms[i] for i in range(n) if tag[i] == 1][0] * 1e3)
print('k_lbs_mfma (64-row groups) %.1f us   k_lbs_mfma32 %.1f us' % (np.median(ts[0]), np.median(ts[4])))
