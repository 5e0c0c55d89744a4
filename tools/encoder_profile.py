"""Run N encoder forwards in one mode (for rocprofv3): python tools/encoder_profile.py {module|nchw|nhwc} [iters]"""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import humaniflow_amd as H
from humaniflow_amd.synthetic import init_benchmark_weights, synthetic_proxy_input
mode = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device('cuda:0')
model = init_benchmark_weights(H.HumaniflowModel(torch.device('cpu'), H.get_benchmark_cfg().MODEL, H.SMPL_PARENTS)).eval().to(dev)
enc = model.image_encoder
x = synthetic_proxy_input(32, 256, seed=0).to(dev)
with torch.no_grad():
    for _ in range(iters):
        if mode == 'module':
            enc.forward_module(x)
        else:
            enc.layout = mode
            enc.forward_folded(x)
    torch.cuda.synchronize()
