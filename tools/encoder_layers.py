"""Per-launch kernel times of the encoder's inference plan (HIP events on the launch stream via hmf_profile_*), in launch order,
with the FLOP rate of every convolution.   python tools/encoder_layers.py [B] [layers]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
from humaniflow_amd import _lib
from humaniflow_amd.encoder import ResNetEncoder
from humaniflow_amd.synthetic import synthetic_proxy_input

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 50
lib = _lib.load()
enc = ResNetEncoder(layers, 18).eval().cuda()
x = synthetic_proxy_input(B, 256, seed=0).cuda()
with torch.no_grad():
    for _ in range(3):
        enc(x)
    torch.cuda.synchronize()
    reps = 10
    lib.hmf_profile_enable(1)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        enc(x)
    b.record()
    torch.cuda.synchronize()
    total = a.elapsed_time(b) / reps
ms = np.zeros(4096); tag = np.zeros(4096, np.int32); rows = np.zeros(4096, np.int32)
n = lib.hmf_profile_collect(ms.ctypes.data_as(_lib.c_double_p), _lib.iptr(tag), _lib.iptr(rows), 4096)
lib.hmf_profile_enable(0)
per = n // reps
us = ms[:n].reshape(reps, per).mean(0) * 1e3
# launch order and FLOPs: bench.encoder_conv_flops (stem, then per block [downsample,] conv1.., last [+ fused downsample])
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bench import encoder_conv_flops
flops, kinds = encoder_conv_flops(enc, B, 256)
names = ['%02d %s' % (i, k) for i, k in enumerate(kinds)]
assert len(names) == per, (len(names), per)
for nm, t, f in zip(names, us, flops):
    print('%-34s %8.1f us  %6.1f TFLOP/s' % (nm, t, f / t / 1e6))
print('conv kernels: %.1f us (%.1f TFLOP/s over %.1f GFLOP); whole encoder forward (events around it, incl. max-pool, mean, launch gaps): %.1f us'
      % (us.sum(), sum(flops) / us.sum() / 1e6, sum(flops) / 1e9, total * 1e3))
