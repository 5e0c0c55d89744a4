"""Feasibility probe: fp32 GEMM on the bf16 MFMA by a three-way operand split (tools/split_probe/split_gemm.hip) vs the product's fp32-MFMA
1x1 convolution kernel on ResNet-50 1x1 shapes: time and error against float64.   python tools/split_probe/probe.py"""
import ctypes as C
import os
import subprocess
import sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
import torch
from humaniflow_amd import _lib
so = os.path.join(HERE, 'libsplit_gemm.so')
if not os.path.exists(so):
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', os.path.join(HERE, 'split_gemm.hip'), '-o', so])
sp = C.CDLL(so)
sp.split_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]
sp.split_gemm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
lib = _lib.load()
dev = 'cuda:0'


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, B, hw, K, N in (('layer3 conv1', 32, 16, 1024, 256), ('layer3 conv3', 32, 16, 256, 1024), ('layer4 conv1', 32, 8, 2048, 512), ('layer2 conv1', 32, 32, 512, 128)):
    M = B * hw * hw
    g = torch.Generator().manual_seed(K)
    a = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    ref = (a.double() @ w.double().t())
    st = _lib.current_stream()
    ws = torch.empty(3, N, K, dtype=torch.int16, device=dev)
    assert sp.split_rows(w.data_ptr(), ws.data_ptr(), N * K, st) == 0
    c = torch.empty(M, N, device=dev)
    f_split = lambda: sp.split_gemm(a.data_ptr(), ws.data_ptr(), c.data_ptr(), M, N, K, st)
    assert f_split() == 0
    torch.cuda.synchronize()
    e_split = (c.double() - ref).abs().max().item()
    y = torch.empty(M, N, device=dev)
    f_f32 = lambda: _lib.check(lib.hmf_conv2d_nhwc(a.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), B, hw, hw, K, N, 1, 1, 0, 0, st), 'conv')
    f_f32()
    torch.cuda.synchronize()
    e_f32 = (y.double() - ref).abs().max().item()
    e_t = ((a @ w.t()).double() - ref).abs().max().item()
    t_split, t_f32 = timeit(f_split), timeit(f_f32)
    fl = 2.0 * M * N * K
    print('%-13s M=%6d N=%5d K=%5d | fp32 MFMA %6.1f us %6.1f TF  err %.2e | bf16 3-way split %6.1f us %6.1f TF(fp32-equivalent)  err %.2e | torch fp32 err %.2e (scale %.1f)'
          % (name, M, N, K, t_f32, fl / t_f32 / 1e6, e_f32, t_split, fl / t_split / 1e6, e_split, e_t, ref.abs().max().item()), flush=True)
