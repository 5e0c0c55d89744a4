"""Latency of the `predict` shape (predict/predict_humaniflow.py:112-131: one image, N = 10 hierarchical samples, point estimate + samples through
SMPL): nothing hides a kernel behind another batch here, so this is where the latency-class flow kernels (k_flow_ctx0 / k_flow_chain: eight
dependent rounds of the kinematic tree) show.  Reports the step eagerly on one stream, as a replayed HIP graph, and per library kernel.
    python tools/predict_latency.py [B] [N] [resnet layers: 18 (the reference's default) | 50]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
import humaniflow_amd as H
from humaniflow_amd import _lib
from humaniflow_amd.synthetic import synthetic_proxy_input, init_benchmark_weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
lib = _lib.load()
dev = torch.device('cuda:0')
cfg = H.get_humaniflow_cfg_defaults()
if len(sys.argv) > 3:
    cfg.MODEL.NUM_RESNET_LAYERS = int(sys.argv[3])            # default: the reference's configuration default (configs/humaniflow_config.py: 18)
torch.manual_seed(0)
model = init_benchmark_weights(H.HumaniflowModel(torch.device('cpu'), cfg.MODEL, H.SMPL_PARENTS), seed=0).eval().to(dev)
smpl = H.SMPL(model_data=H.synthetic_smpl_model(seed=0), batch_size=B, create_transl=False).to(dev)
x = synthetic_proxy_input(B, 256, seed=1).to(dev)


def step():
    with torch.no_grad():
        pred = model(x, num_samples=N)
        pe, sa = smpl.forward_samples_and_point_est(pred)
        return sa.vertices, pe.vertices, pred['pose_rotmats_samples']


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


print('predict shape: B = %d image(s), N = %d samples, ResNet-%d + heads + flow + SMPL (M = %d bodies), fp32' % (B, N, cfg.MODEL.NUM_RESNET_LAYERS, B * N + B))
print('eager, one stream            : %8.1f us per step (host-issue bound: ~60 launches)' % timed(step))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    step()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s, capture_error_mode='thread_local'):
    out = step()
with torch.cuda.stream(s):
    t = timed(g.replay)
print('HIP-graph replay             : %8.1f us per step' % t)
lib.hmf_profile_enable(1)
reps = 20
for _ in range(reps):
    step()
torch.cuda.synchronize()
ms, tag, rows = np.zeros(65536), np.zeros(65536, np.int32), np.zeros(65536, np.int32)
n = lib.hmf_profile_collect(ms.ctypes.data_as(_lib.c_double_p), _lib.iptr(tag), _lib.iptr(rows), 65536)
lib.hmf_profile_enable(0)
names = {1: 'k_lbs_mfma', 2: 'k_rigid_chain', 3: 'k_joints_compact', 5: 'k_flow_ctx0', 6: 'k_flow_chain', 7: 'k_isgc', 8: '1x1 convolutions (k_conv_gemm / k_conv_igemm / k_conv_pair)',
         9: 'stem (k_stem_wino + k_stem_pool_rim)', 10: 'k_conv_patch (+ split-K finish)', 11: 'k_conv_winograd (+ split-K finish)', 15: 'k_heads_fc1 + k_heads_out'}
# A record is ONE ProfScope = one event pair around the launches of one library call (a split-K 3x3 layer: main kernel + finish pass); the pair's
# time is start-of-first to end-of-last kernel of the call, i.e. it does not include the gap to the NEXT call.  (Round 4 printed "includes ~2 us
# of launch gap each" on every row and a row label with a launch count the tag does not have: the 1x1 row is 30 records of ~0.9-3 us each.)
tot = 0.0
print('  per library call class: records per step, mean per record, total per step (HIP event pair around each call; gaps BETWEEN calls not included)')
for tg in sorted(set(tag[:n].tolist())):
    sel = tag[:n] == tg
    us = ms[:n][sel].sum() / reps * 1e3
    tot += us
    print('  %-62s %4.0f x %6.2f us = %8.1f us per step' % (names.get(tg, 'tag %d' % tg), sel.sum() / reps, us / (sel.sum() / reps), us))
print('  %-62s %24.1f us (kernel spans; the graph replay above adds the gaps between the %d calls)' % ('sum', tot, int((tag[:n] >= 0).sum() / reps)))
lay = ms[:n][tag[:n] == 11].reshape(reps, -1).mean(0) * 1e3
print('  k_conv_winograd calls in plan order (us): ' + ' '.join('%.1f' % v for v in lay))
lay = ms[:n][tag[:n] == 8].reshape(reps, -1).mean(0) * 1e3
print('  1x1 calls in plan order (us): ' + ' '.join('%.1f' % v for v in lay))
