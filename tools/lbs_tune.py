"""A/B the k_lbs_mfma tuning variants (impl = 16*OPT) in one process, interleaved rounds, HIP-event timing from the library."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import humaniflow_amd as H
from humaniflow_amd import _lib
lib = _lib.use_experiments_library()      # tools/libhumaniflow_amd_experiments.so: product entry points + timing variants
dev = torch.device('cuda:0')
smpl = H.SMPL(model_data=H.synthetic_smpl_model(seed=0), create_transl=False).to(dev)
M = int(os.environ.get('M', 3200))
g = torch.Generator().manual_seed(0)
from math import pi
v = torch.randn(M * 24, 3, generator=g, dtype=torch.float64) * 0.5
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from oracle.so3 import so3_exp
R = so3_exp(v).float().view(M, 24, 3, 3).to(dev)
betas = (torch.randn(M, 10, generator=g) * 1.25).to(dev)
variants = [int(a) for a in sys.argv[1:]] or [0, 1, 3, 7]
ref = smpl(body_pose=R[:, 1:], global_orient=R[:, :1], betas=betas, pose2rot=False, impl=0).vertices.clone()
res = {o: [] for o in variants}
cyc = torch.zeros(512, dtype=torch.int64, device=dev)
lib.hmf_debug_set_lbs_cycles(cyc.data_ptr())
clk = {o: [] for o in variants}
for rnd in range(6):
    for o in variants:
        lib.hmf_profile_enable(1)
        out = smpl(body_pose=R[:, 1:], global_orient=R[:, :1], betas=betas, pose2rot=False, impl=16 * o)
        ms = np.zeros(16); tag = np.zeros(16, np.int32); rows = np.zeros(16, np.int32)
        n = lib.hmf_profile_collect(ms.ctypes.data_as(_lib.c_double_p), _lib.iptr(tag), _lib.iptr(rows), 16)
        lib.hmf_profile_enable(0)
        t = [ms[i] for i in range(n) if tag[i] == 1][0]
        if rnd > 0:
            res[o].append(t * 1e3)
            clk[o].append(cyc[:256].max().item() / (t * 1e-3) / 1e9)
        if rnd == 0 and o == 0:
            print('OPT %d max |diff| vs default: %.2e' % (o, (out.vertices - ref).abs().max().item()))
for o in variants:
    a = np.array(res[o])
    print('OPT %2d: median %.1f us  min %.1f us   effective shader clock %.2f GHz (max block span / event time)' % (o, np.median(a), a.min(), np.median(clk[o])))
