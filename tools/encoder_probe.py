"""Time ResNet-50 encoder variants on the GPU (MIOpen): layout, BN folding, find mode."""
import sys, os, time
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import humaniflow_amd as H
from humaniflow_amd.synthetic import init_benchmark_weights, synthetic_proxy_input

dev = torch.device('cuda:0')
cfg = H.get_benchmark_cfg()
model = init_benchmark_weights(H.HumaniflowModel(torch.device('cpu'), cfg.MODEL, H.SMPL_PARENTS)).eval().to(dev)
enc = model.image_encoder
x = synthetic_proxy_input(32, 256, seed=0).to(dev)


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


with torch.no_grad():
    for bench in (False, True):
        torch.backends.cudnn.benchmark = bench
        t0 = time.time()
        print('benchmark=%s' % bench, flush=True)
        print('  module path NCHW (BN unfused): %.3f ms' % timeit(lambda: enc.forward_module(x)), flush=True)
        xcl = x.contiguous(memory_format=torch.channels_last)
        print('  module path NHWC input       : %.3f ms' % timeit(lambda: enc.forward_module(xcl)), flush=True)
        for lay in ('nchw', 'nhwc'):
            enc.layout = lay
            enc.invalidate_plan()
            y = enc.forward_folded(x)
            ref = enc.forward_module(x)
            print('  folded %s + fused epilogue   : %.3f ms   (max rel err vs module path %.2e)'
                  % (lay, timeit(lambda: enc.forward_folded(x)), ((y - ref).abs().max() / ref.abs().max()).item()), flush=True)
        print('  (setup+timing took %.1f s)' % (time.time() - t0), flush=True)
