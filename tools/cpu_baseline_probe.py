"""Probe how the CPU baseline (oracle, reference mode) scales with host threads on the bench host."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import humaniflow_amd as H
from humaniflow_amd.synthetic import init_benchmark_weights, synthetic_proxy_input
from oracle import model as omodel, smpl as osmpl

print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)), flush=True)
cfg = H.get_benchmark_cfg()
m = init_benchmark_weights(H.HumaniflowModel(torch.device('cpu'), cfg.MODEL, H.SMPL_PARENTS)).eval()
sd = {k: v.detach() for k, v in m.state_dict().items()}
smpl = H.SMPL(model_data=H.synthetic_smpl_model(seed=0), create_transl=False)
sm = {k: (v.long() if k == 'parents' else v.float()) for k, v in smpl.model_dict().items()}
ocfg = omodel.OracleCfg(num_resnet_layers=50)
images, N = int(sys.argv[1]) if len(sys.argv) > 1 else 4, 100
x = synthetic_proxy_input(images, 256, seed=0)
for nt in [int(a) for a in sys.argv[2:]] or [8, 16, 32, 64]:
    torch.set_num_threads(nt)
    res = []
    for it in range(2):
        t0 = time.perf_counter()
        feats = omodel.resnet_forward(sd, x, 50, 'image_encoder.')
        t1 = time.perf_counter()
        out = omodel.forward(sd, ocfg, input_feats=feats, num_samples=N)
        t2 = time.perf_counter()
        g = out['glob_rotmat']
        osmpl.smpl_forward(sm, out['shape_samples'].reshape(-1, 10), out['pose_rotmats_samples'].reshape(-1, 23, 3, 3),
                           g[:, None].expand(-1, N, -1, -1).reshape(-1, 1, 3, 3), pose2rot=False)
        t3 = time.perf_counter()
        res = (t1 - t0, t2 - t1, t3 - t2)
    print('threads %3d: encoder %.3f s  flow %.3f s  smpl %.3f s  -> %.0f img*samples/s' % (nt, *res, images * N / sum(res)), flush=True)
