import sys, os
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import humaniflow_amd as H
from humaniflow_amd.synthetic import init_benchmark_weights, synthetic_proxy_input
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0')
model = init_benchmark_weights(H.HumaniflowModel(torch.device('cpu'), H.get_benchmark_cfg().MODEL, H.SMPL_PARENTS)).eval().to(dev)
enc = model.image_encoder
x = synthetic_proxy_input(32, 256, seed=0).to(dev)
print('env', {k: v for k, v in os.environ.items() if 'MIOPEN' in k or 'NHWC' in k})
with torch.no_grad():
    for mode in ('module', 'nchw', 'nhwc'):
        fn = (lambda: enc.forward_module(x)) if mode == 'module' else (lambda: enc.forward_folded(x))
        if mode != 'module':
            enc.layout = mode
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            fn()
            torch.cuda.synchronize()
        print('=== ', mode)
        print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=12, max_name_column_width=70))
    w = enc._plan['blocks'][0][0][0][0]
    y = F.conv2d(torch.randn(32, 64, 64, 64, device=dev), w.contiguous(), None)
    print('conv out strides', y.stride(), y.shape, 'w strides', w.stride())
