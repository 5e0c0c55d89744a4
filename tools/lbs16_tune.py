"""Where k_lbs_mfma16 (configs[2]: bf16 hi + lo operands) spends its time: timing ablations of the experiments build (impl = 3 + 16 * ABL), interleaved
rounds, HIP-event timing from the library.  ABL & 1 no skinning epilogue, & 2 no B-operand stream, & 4 no MFMAs, & 8 skinning without vertex stores."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import humaniflow_amd as H
from humaniflow_amd import _lib
lib = _lib.use_experiments_library()
dev = torch.device('cuda:0')
smpl = H.SMPL(model_data=H.synthetic_smpl_model(seed=0), create_transl=False).to(dev)
M = int(os.environ.get('M', 6464))
g = torch.Generator().manual_seed(0)
from oracle.so3 import so3_exp
R = so3_exp(torch.randn(M * 24, 3, generator=g, dtype=torch.float64) * 0.5).float().view(M, 24, 3, 3).to(dev)
betas = (torch.randn(M, 10, generator=g) * 1.25).to(dev)
variants = [int(a) for a in sys.argv[1:]] or [0, 8, 1, 2, 3, 4, 5, 6, 7]
names = {0: 'product', 1: 'no epilogue', 2: 'no B stream', 3: 'no epilogue, no B stream', 4: 'no MFMAs', 5: 'no MFMAs, no epilogue', 6: 'no MFMAs, no B stream',
         7: 'no MFMAs, no B stream, no epilogue (A reads + staging only)', 16: 'round 6: B ring 2 k-groups ahead (pinned) + late epilogue operands', 17: 'round 5 form: requests unpinned, epilogue operands held through the loop',
         18: 'round 6: 2 ahead (pinned)', 19: 'round 6: 3 ahead (pinned, spills) + late', 21: 'round 6: product without the B stream', 22: 'round 6: product + half-tile rounds (3.5 rounds, waves de-phased)', 23: 'round 6: product + next tile first k-group requested before the epilogue', 24: 'round 6: product with the transforms staged through registers (no LDS-DMA overlap)', 25: 'round 6: MATRIX-PIPE skinning epilogue (dense weights, bf16 hi + lo, element pairs fenced)',
         26: 'round 6: matrix-pipe epilogue, pairs not fenced', 27: 'round 6: matrix-pipe epilogue without the vertex stores', 64: 'product + phase stamps', 65: 'half tiles + phase stamps', 66: 'round-5 form + phase stamps', 8: 'skinning without vertex stores', 10: 'no B stream, no vertex stores', 12: 'no MFMAs, no vertex stores'}
res = {o: [] for o in variants}
phases = {}
import ctypes
for rnd in range(6):
    for o in variants:
        lib.hmf_profile_enable(1)
        smpl(body_pose=R[:, 1:], global_orient=R[:, :1], betas=betas, pose2rot=False, impl=3 + 16 * o)
        ms = np.zeros(16); tag = np.zeros(16, np.int32); rows = np.zeros(16, np.int32)
        n = lib.hmf_profile_collect(ms.ctypes.data_as(_lib.c_double_p), _lib.iptr(tag), _lib.iptr(rows), 16)
        lib.hmf_profile_enable(0)
        t = [ms[i] for i in range(n) if tag[i] == 12][0]
        buf = (ctypes.c_ulonglong * 8)()
        lib.hmf_debug_lbs16_phases(ctypes.cast(buf, ctypes.c_void_p))
        phases[o] = list(buf)
        if rnd > 0:
            res[o].append(t * 1e3)
print('k_lbs_mfma16<4, split> at M = %d (vertex tensor %.0f MB)' % (M, M * 82680 / 1e6))
for o in variants:
    a = np.array(res[o])
    print('ABL %2d %-60s median %.1f us  min %.1f us' % (o, names.get(o, ''), np.median(a), a.min()))
    p = phases[o]
    if p[4]:
        print('        per wave tile (k cycles, mean over %d/2 wave tiles): start-up %.1f | main loop %.1f | epilogue %.1f ; staging + barrier waits per wave: %.0f k cycles in all (%d waves)'
              % (p[4], 2 * p[1] / p[4] / 1e3, 2 * p[2] / p[4] / 1e3, 2 * p[3] / p[4] / 1e3, p[0] / (8 * 256) / 1e3, 8 * 256))
