"""Time the evaluation flavour of the path on one GPU: predict_batch (model + SMPL on point estimate, samples, T-poses, 2D projection),
target_batch and the metric tracker update, B=32, N=100.   python tools/eval_bench.py [B] [N]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import humaniflow_amd as H
from humaniflow_amd import evaluate as ev
from humaniflow_amd.synthetic import init_benchmark_weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device('cuda:0')
cfg = H.get_benchmark_cfg()                       # ResNet-50 encoder (the reference default config is ResNet-18)
model = init_benchmark_weights(H.HumaniflowModel(torch.device('cpu'), cfg.MODEL, H.SMPL_PARENTS), seed=0).eval().to(dev)
smpl = H.SMPL(model_data=H.synthetic_smpl_model(seed=0), create_transl=False).to(dev)
ds = ev.SyntheticEvalSet(B, size=256, seed=0)
batch = ds.batch(list(range(B)), dev)


def timed(fn, reps=5):
    for _ in range(2):
        r = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, r


t_pred, pred = timed(lambda: ev.predict_batch(model, smpl, batch, N))
t_targ, targ = timed(lambda: ev.target_batch(smpl, batch))
trk = ev.EvalMetricsTracker(dev)
t_upd, _ = timed(lambda: trk.update_per_batch(pred, targ))
print('B=%d N=%d: predict_batch %.2f ms, target_batch %.2f ms, tracker.update_per_batch %.2f ms' % (B, N, t_pred, t_targ, t_upd))
for name in ev.METRICS:
    t = ev.EvalMetricsTracker(dev, [name])
    ms, _ = timed(lambda: t.update_per_batch(pred, targ), 3)
    print('   %-28s %.2f ms' % (name, ms))
