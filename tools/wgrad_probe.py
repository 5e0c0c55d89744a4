"""hmf_conv_wgrad on the ResNet-18 / ResNet-50 3x3 training shapes (B = 72, 256 x 256 input) + the stem.   python tools/wgrad_probe.py [B]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from humaniflow_amd import _lib
lib = _lib.load()
dev = 'cuda:0'
CL = torch.channels_last


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


B = int(sys.argv[1]) if len(sys.argv) > 1 else 72
for name, hw, cin, cout, ks, stride, stem in (('stem 7x7/2', 256, 18, 64, 7, 2, 1), ('layer1 3x3', 64, 64, 64, 3, 1, 0), ('layer2.0 3x3/2', 64, 64, 128, 3, 2, 0),
                                              ('layer2 3x3', 32, 128, 128, 3, 1, 0), ('layer3.0 3x3/2', 32, 128, 256, 3, 2, 0), ('layer3 3x3', 16, 256, 256, 3, 1, 0),
                                              ('layer4.0 3x3/2', 16, 256, 512, 3, 2, 0), ('layer4 3x3', 8, 512, 512, 3, 1, 0), ('downsample 1x1/2', 64, 64, 128, 1, 2, 0)):
    oh = (hw + 2 * (ks // 2) - ks) // stride + 1
    x = torch.randn(B, cin, hw, hw, device=dev)
    if not stem:
        x = x.contiguous(memory_format=CL)
    gy = torch.randn(B, cout, oh, oh, device=dev).contiguous(memory_format=CL)
    dw = torch.empty(cout * cin * ks * ks, device=dev)
    st = _lib.current_stream()
    fn = lambda: _lib.check(lib.hmf_conv_wgrad(gy.data_ptr(), x.data_ptr(), dw.data_ptr(), B, hw, hw, cin, cout, ks, stride, stem, st), 'wgrad')
    t = timeit(fn)
    fl = 2.0 * B * oh * oh * cout * cin * ks * ks
    print('%-18s %6.2f GF  %8.1f us  %6.1f TFLOP/s' % (name, fl / 1e9, t, fl / t / 1e6), flush=True)
