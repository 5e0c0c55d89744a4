"""Probe: capture one step (encoder -> flow -> SMPL) in a HIP graph per stream and replay.   python tools/graph_probe.py [streams]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import humaniflow_amd as H
from humaniflow_amd.synthetic import synthetic_proxy_input, init_benchmark_weights

S = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device('cuda:0')
cfg = H.get_benchmark_cfg()
model = init_benchmark_weights(H.HumaniflowModel(torch.device('cpu'), cfg.MODEL, H.SMPL_PARENTS), seed=0).eval().to(dev)
smpl = H.SMPL(model_data=H.synthetic_smpl_model(seed=0), batch_size=32, create_transl=False).to(dev)
x = synthetic_proxy_input(32, 256, seed=100).to(dev)
N = 100


def step():
    with torch.no_grad():
        pred = model(x, num_samples=N)
        g = pred['glob_rotmat']
        pe = smpl(body_pose=pred['pose_rotmats_point_est'], global_orient=g[:, None], betas=pred['shape_mode'], pose2rot=False)
        sa = smpl(body_pose=pred['pose_rotmats_samples'].reshape(-1, 23, 3, 3), global_orient=g[:, None].expand(-1, N, -1, -1).reshape(-1, 1, 3, 3),
                  betas=pred['shape_samples'].reshape(-1, 10), pose2rot=False)
    return sa.vertices


for _ in range(3):
    step()
torch.cuda.synchronize()
from humaniflow_amd import _lib
import numpy as np
lib = _lib.load()
PROF = os.environ.get('PROF') == '1'
if PROF:
    lib.hmf_profile_enable(1)
graphs, outs, streams = [], [], [torch.cuda.Stream() for _ in range(S)]
for s in streams:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        for _ in range(2):
            step()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s, capture_error_mode=os.environ.get("CAPMODE", "global")):
        o = step()
    graphs.append(g); outs.append(o)
torch.cuda.synchronize()
ref = step()
graphs[0].replay(); torch.cuda.synchronize()
print('replay finite:', bool(torch.isfinite(outs[0]).all()), 'shape', tuple(outs[0].shape))
K = 21
for _ in range(3):
    for i in range(S):
        with torch.cuda.stream(streams[i]):
            graphs[i].replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(K):
    with torch.cuda.stream(streams[i % S]):
        graphs[i % S].replay()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print('%d streams, graph replay: %.3f ms/step, %.0f images*samples/s' % (S, dt * 1e3, 3200 / dt))

if PROF:
    ms = np.zeros(16384); tag = np.zeros(16384, np.int32); rows = np.zeros(16384, np.int32)
    n = lib.hmf_profile_collect(ms.ctypes.data_as(_lib.c_double_p), _lib.iptr(tag), _lib.iptr(rows), 16384)
    print('profile records', n, 'conv igemm mean us', float(ms[:n][tag[:n] == 8].mean()) * 1e3 if n > 0 else None, 'lbs', ms[:n][tag[:n] == 1][-3:] * 1e3 if n > 0 else None)
