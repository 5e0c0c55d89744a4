"""Seam fusion (hmf_conv_pair) vs the two separate 1x1 launches on the ResNet-50 layer1 / layer2 shapes.
    python tools/pair_probe.py [B]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from humaniflow_amd import _lib
lib = _lib.use_experiments_library()      # tools/libhumaniflow_amd_experiments.so: product entry points + timing variants
dev = 'cuda:0'
CL = torch.channels_last


def timeit(fn, iters=50):
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


B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for name, hw, k1, k2, C, n2, res in (('layer1.0->1 (+downsample)', 64, 64, 64, 256, 64, False), ('layer1.1->2', 64, 64, 0, 256, 64, True),
                                     ('layer1.2->layer2.0', 64, 64, 0, 256, 128, True), ('layer2.1->2', 32, 128, 0, 512, 128, True),
                                     ('layer2.3->layer3.0', 32, 128, 0, 512, 256, True)):
    h = torch.randn(B, k1, hw, hw, device=dev).contiguous(memory_format=CL)
    x2 = torch.randn(B, k2, hw, hw, device=dev).contiguous(memory_format=CL) if k2 else None
    w3 = torch.randn(C, k1 + k2) / (k1 + k2) ** 0.5
    w1 = torch.randn(n2, C) / C ** 0.5
    p3, p1 = torch.empty_like(w3), torch.empty_like(w1)
    _lib.check(lib.hmf_conv_pair_pack_host(w3.data_ptr(), w1.data_ptr(), p3.data_ptr(), p1.data_ptr(), k1 + k2, C, n2), 'pack')
    w3d, w1d, p3d, p1d = w3.to(dev), w1.to(dev), p3.to(dev), p1.to(dev)
    b3, b1 = torch.randn(C, device=dev), torch.randn(n2, device=dev)
    r = torch.randn(B, C, hw, hw, device=dev).contiguous(memory_format=CL) if res else None
    y = torch.empty(B, C, hw, hw, device=dev).contiguous(memory_format=CL)
    z = torch.empty(B, n2, hw, hw, device=dev).contiguous(memory_format=CL)
    st = _lib.current_stream()

    def separate():
        if k2:
            _lib.check(lib.hmf_conv1x1_dual_nhwc(h.data_ptr(), x2.data_ptr(), w3d.data_ptr(), b3.data_ptr(), y.data_ptr(), B, hw, hw, k1, hw, hw, k2, 1, C, 1, st), 'dual')
        else:
            _lib.check(lib.hmf_conv2d_nhwc(h.data_ptr(), w3d.data_ptr(), b3.data_ptr(), None if r is None else r.data_ptr(), y.data_ptr(), B, hw, hw, k1, C, 1, 1, 1, 0, st), 'c3')
        _lib.check(lib.hmf_conv2d_nhwc(y.data_ptr(), w1d.data_ptr(), b1.data_ptr(), None, z.data_ptr(), B, hw, hw, C, n2, 1, 1, 1, 0, st), 'c1')

    def fused(variant=0):
        _lib.check(lib.hmf_conv_pair(h.data_ptr(), None if x2 is None else x2.data_ptr(), p3d.data_ptr(), b3.data_ptr(), None if r is None else r.data_ptr(),
                                     y.data_ptr(), p1d.data_ptr(), b1.data_ptr(), z.data_ptr(), B, hw, hw, k1, hw, hw, k2, 1, C, n2, variant, st), 'pair')

    fl = 2.0 * B * hw * hw * ((k1 + k2) * C + C * n2)
    mb = B * hw * hw * 4 * (k1 + k2 + (C if res else 0) + C + n2) / 1e6
    ts, tf = timeit(separate), timeit(fused)
    print('%-26s %5.2f GF | separate %6.1f us %5.1f TF | fused %6.1f us %5.1f TF  (%.0f MB min -> %.2f TB/s)' % (name, fl / 1e9, ts, fl / ts / 1e6, tf, fl / tf / 1e6, mb, mb / tf))
    if k1 + k2 == 64 and n2 == 64:
        print('    variants:', ' | '.join('%s %.1f us' % (lab, timeit(lambda v=v: fused(v))) for v, lab in
                                          ((1, 'no y store'), (2, 'no residual'), (3, 'neither'), (4, 'no barrier'), (7, 'bare'), (8, 'residual nt'), (16, 'y store cached'), (20, '2 waves/SIMD'), (27, 'bare 2 waves/SIMD'))))
