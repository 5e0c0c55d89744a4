"""Does a real MIOpen find (benchmarking) pick faster fp32 conv solvers than immediate mode?  Prints per-mode encoder time."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import humaniflow_amd as H
from humaniflow_amd.synthetic import init_benchmark_weights, synthetic_proxy_input
print('MIOPEN env:', {k: v for k, v in os.environ.items() if k.startswith('MIOPEN')}, flush=True)
dev = torch.device('cuda:0')
model = init_benchmark_weights(H.HumaniflowModel(torch.device('cpu'), H.get_benchmark_cfg().MODEL, H.SMPL_PARENTS)).eval().to(dev)
enc = model.image_encoder
x = synthetic_proxy_input(32, 256, seed=0).to(dev)
torch.backends.cudnn.benchmark = True
def timeit(fn, reps=10):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
with torch.no_grad():
    for lay in sys.argv[1:] or ['nchw', 'nhwc']:
        enc.layout = lay; enc.invalidate_plan()
        t0 = time.time(); enc.forward_folded(x); torch.cuda.synchronize()
        print('%s first call %.1f s' % (lay, time.time() - t0), flush=True)
        enc.forward_folded(x)
        print('%s: %.3f ms' % (lay, timeit(lambda: enc.forward_folded(x))), flush=True)
