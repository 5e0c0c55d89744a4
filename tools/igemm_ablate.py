"""Ablations of k_conv_igemm on the ResNet-50 1x1 shapes (timing only): what does the loop cost without global fetches / barriers / epilogue?
    python tools/igemm_ablate.py"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from humaniflow_amd import _lib
lib = _lib.use_experiments_library()      # tools/libhumaniflow_amd_experiments.so: product entry points + timing variants
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


B = 32
for name, hw, cin, cout, res in (('l1.c1', 64, 256, 64, False), ('l1.c3', 64, 64, 256, True), ('l2.c1', 32, 512, 128, False), ('l2.c3', 32, 128, 512, True),
                                 ('l3.c1', 16, 1024, 256, False), ('l3.c3', 16, 256, 1024, True), ('l4.c1', 8, 2048, 512, False), ('l4.c3', 8, 512, 2048, True)):
    x = torch.randn(B, cin, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 1, 1, device=dev) / cin ** 0.5).contiguous(memory_format=torch.channels_last)
    bias = torch.randn(cout, device=dev)
    r = torch.randn(B, cout, hw, hw, device=dev).contiguous(memory_format=torch.channels_last) if res else None
    y = torch.empty(B, cout, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    st = _lib.current_stream()
    fl = 2.0 * B * hw * hw * cin * cout
    line = '%-6s %5.2f GF' % (name, fl / 1e9)
    for tile, lab in ((64064, 'full'), (8064064, 'one-ahead'), (1064064, 'no-fetch'), (2064064, 'no-barrier'), (3064064, 'no-fetch,no-barrier'), (4064064, 'no-epilogue'), (7064064, 'bare')):
        fn = lambda: _lib.check(lib.hmf_conv2d_nhwc(x.data_ptr(), w.data_ptr(), bias.data_ptr(), None if r is None else r.data_ptr(), y.data_ptr(),
                                                    B, hw, hw, cin, cout, 1, 1, 1, tile, st), 'conv')
        t = timeit(fn)
        line += ' | %s %6.1f us %5.1f TF' % (lab, t, fl / t / 1e6)
    print(line, flush=True)
