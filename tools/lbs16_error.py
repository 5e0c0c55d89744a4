"""Error of the configs[2] kernel (bf16 hi + lo operands, matrix-pipe skinning epilogue) against the fp32 kernel on the same inputs.   python tools/lbs16_error.py [M]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
import humaniflow_amd as H
from oracle.so3 import so3_exp
dev = 'cuda:0'
M = int(sys.argv[1]) if len(sys.argv) > 1 else 70
s32 = H.SMPL(model_data=H.synthetic_smpl_model(seed=0), create_transl=False).to(dev)
s16 = H.SMPL(model_data=H.synthetic_smpl_model(seed=0), create_transl=False, blend_dtype='bf16').to(dev)
g = torch.Generator().manual_seed(0)
R = so3_exp(torch.randn(M * 24, 3, generator=g, dtype=torch.float64) * 0.5).float().view(M, 24, 3, 3).to(dev)
b = (torch.randn(M, 10, generator=g) * 1.25).to(dev)
a = s32(body_pose=R[:, 1:], global_orient=R[:, :1], betas=b, pose2rot=False)
c = s16(body_pose=R[:, 1:], global_orient=R[:, :1], betas=b, pose2rot=False)
d = (a.vertices - c.vertices).abs()
print('M = %d: vertices max |diff| %.3e (mean %.3e, |v| max %.2f), joints %.3e' % (M, d.max().item(), d.mean().item(), a.vertices.abs().max().item(), (a.joints - c.joints).abs().max().item()))
print('worst rows', torch.topk(d.amax(dim=(1, 2)), min(5, M)).values.tolist())
print('worst vertices', torch.topk(d.amax(dim=(0, 2)), 5))
