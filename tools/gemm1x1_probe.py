"""Time hmf_conv1x1_gemm (csrc/conv_gemm.hip) on every 1x1 shape of ResNet-50 at B = 32 (256x256 input), every tile, against k_conv_igemm:
    python tools/gemm1x1_probe.py [B]
Interleaved rounds, HIP events around batches of launches (launch overhead amortised), median."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
from humaniflow_amd import _lib

lib = _lib.use_experiments_library()
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 32
st = _lib.current_stream()
CL = torch.channels_last
# (name, HW, K1, N, K2, H2, stride2, residual)
SHAPES = [
    ('layer1.0.conv1   64->64   @64', 64, 64, 64, 0, 0, 1, False),
    ('layer1.0 dual  64+64->256 @64', 64, 64, 256, 64, 64, 1, False),
    ('layer1.x.conv1  256->64   @64', 64, 256, 64, 0, 0, 1, False),
    ('layer1.x.conv3   64->256  @64+r', 64, 64, 256, 0, 0, 1, True),
    ('layer2.0.conv1  256->128  @64', 64, 256, 128, 0, 0, 1, False),
    ('layer3.0.conv1  512->256  @32', 32, 512, 256, 0, 0, 1, False),
    ('layer2.1.conv1  512->128  @32', 32, 512, 128, 0, 0, 1, False),
    ('layer2.x.conv3  128->512  @32+r', 32, 128, 512, 0, 0, 1, True),
    ('layer2.0 dual 128+256->512 @32', 32, 128, 512, 256, 64, 2, False),
    ('layer3.x.conv1 1024->256  @16', 16, 1024, 256, 0, 0, 1, False),
    ('layer3.x.conv3  256->1024 @16+r', 16, 256, 1024, 0, 0, 1, True),
    ('layer3.0 dual 256+512->1024 @16', 16, 256, 1024, 512, 32, 2, False),
    ('layer4.0.conv1 1024->512  @16', 16, 1024, 512, 0, 0, 1, False),
    ('layer4.x.conv1 2048->512  @8', 8, 2048, 512, 0, 0, 1, False),
    ('layer4.x.conv3  512->2048 @8+r', 8, 512, 2048, 0, 0, 1, True),
    ('layer4.0 dual 512+1024->2048 @8', 8, 512, 2048, 1024, 16, 2, False),
]
TILES = [3064128, 2064128, 64064, 3064064, 2064064]


def timeit(fn, reps=20, rounds=5):
    out = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) / reps * 1e3)
    return float(np.median(out))


TIMELINE = '--timeline' in sys.argv
if TIMELINE:
    TILES = [64064, 3064064, 3064128]


def timeline(fn, blocks):
    """Per-phase shader-clock breakdown of one launch (experiments build): means over work-groups, in cycles."""
    import ctypes
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    fn()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3
    n = min(blocks, 4096)
    buf = np.zeros((n, 6), np.int64)
    _lib.check(lib.hmf_debug_gemm_stamps(buf.ctypes.data_as(ctypes.c_void_p), n), 'stamps')
    # shader-clock counters are per CU group and not synchronised: only per-work-group differences mean something.  The 100 MHz counter IS global.
    cyc = (buf[:, 3] - buf[:, 0]).astype(np.float64)
    wall = (buf[:, 5] - buf[:, 4]).astype(np.float64) / 100.0                     # us
    ok = wall > 1.0
    mhz = float(np.median(cyc[ok] / wall[ok])) if ok.any() else float('nan')
    first, last_start, last_end = buf[:, 4].min(), buf[:, 4].max(), buf[:, 5].max()
    return ('%5.1f us by one event pair | first start -> last end %5.1f us, last start +%4.1f us | per work-group: prologue %5d  main %6d  epilogue %5d cyc = %5.1f us at %4.0f MHz'
            % (us, (last_end - first) / 100.0, (last_start - first) / 100.0, (buf[:, 1] - buf[:, 0]).mean(), (buf[:, 2] - buf[:, 1]).mean(),
               (buf[:, 3] - buf[:, 2]).mean(), float(np.median(wall)), mhz))


print('B = %d; us per launch (back-to-back launches on one stream, so ~1.5 us of launch boundary included)' % B)
for name, hw, k1, n, k2, h2, s2, res in SHAPES:
    x = torch.randn(B, k1, hw, hw, device='cuda').contiguous(memory_format=CL)
    x2 = torch.randn(B, k2, h2, h2, device='cuda').contiguous(memory_format=CL) if k2 else None
    w = torch.randn(n, k1 + k2, device='cuda') / (k1 + k2) ** 0.5
    bias = torch.randn(n, device='cuda')
    r = torch.randn(B, n, hw, hw, device='cuda').contiguous(memory_format=CL) if res else None
    y = torch.empty(B, n, hw, hw, device='cuda').contiguous(memory_format=CL)
    flops = 2.0 * B * hw * hw * n * (k1 + k2)
    if k2:
        old = lambda: _lib.check(lib.hmf_conv1x1_dual_nhwc(x.data_ptr(), x2.data_ptr(), w.data_ptr(), bias.data_ptr(), y.data_ptr(), B, hw, hw, k1, h2, h2, k2, s2, n, 1, st), 'dual')
    else:
        old = lambda: _lib.check(lib.hmf_conv2d_nhwc(x.data_ptr(), w.data_ptr(), bias.data_ptr(), None if r is None else r.data_ptr(), y.data_ptr(), B, hw, hw, k1, n, 1, 1, 1, 0, st), 'igemm')
    t_old = timeit(old)
    cols = ['igemm %6.1f (%5.1f TF)' % (t_old, flops / t_old / 1e6)]
    for tile in TILES:
        if (tile % 1000) == 128 and n % 128:
            cols.append('%d      -      ' % tile)
            continue
        new = lambda: _lib.check(lib.hmf_conv1x1_gemm(x.data_ptr(), None if x2 is None else x2.data_ptr(), w.data_ptr(), bias.data_ptr(), None if r is None else r.data_ptr(),
                                                      y.data_ptr(), B, hw, hw, k1, h2, h2, k2, s2, n, 1, tile, st), 'gemm')
        t = timeit(new)
        cols.append('%d %6.1f (%5.1f)' % (tile, t, flops / t / 1e6))
        if TIMELINE:
            bm, bn = (tile % 1000000) // 1000, tile % 1000
            print('    %-9d %s' % (tile, timeline(new, ((B * hw * hw + bm - 1) // bm) * (n // bn))))
    print('%-34s %5.2f GF | %s' % (name, flops / 1e9, ' | '.join(cols)))
