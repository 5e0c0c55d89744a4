import sys, numpy as np, torch
sys.path.insert(0, '.')
import humaniflow_amd as H
from humaniflow_amd import _lib
from humaniflow_amd.synthetic import init_benchmark_weights
lib = _lib.load()
cfg = H.get_benchmark_cfg()
model = init_benchmark_weights(H.HumaniflowModel(torch.device('cpu'), cfg.MODEL, H.SMPL_PARENTS), seed=0).eval().cuda()
feats = torch.randn(32, 2048, device='cuda') * 0.1
with torch.no_grad():
    for _ in range(3): model(None, input_feats=feats, num_samples=100)
    torch.cuda.synchronize()
    lib.hmf_profile_enable(1)
    for _ in range(10): model(None, input_feats=feats, num_samples=100)
ms = np.zeros(512); tag = np.zeros(512, np.int32); rows = np.zeros(512, np.int32)
n = lib.hmf_profile_collect(ms.ctypes.data_as(_lib.c_double_p), _lib.iptr(tag), _lib.iptr(rows), 512)
lib.hmf_profile_enable(0)
for t, nm in ((5, 'ctx0'), (6, 'chain'), (7, 'isgc')):
    print(nm, round(float(ms[:n][tag[:n] == t].mean()) * 1e3, 1), 'us')
