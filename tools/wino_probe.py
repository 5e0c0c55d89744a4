"""Per-phase timeline of k_conv_winograd on the ResNet-50 stride-1 3x3 shapes (experiments build: shader-clock stamps per work-group).
    python tools/wino_probe.py [B]"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
from humaniflow_amd import _lib
lib = _lib.use_experiments_library()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
st = _lib.current_stream()
for name, hw, c in (('layer1', 64, 64), ('layer2', 32, 128), ('layer3', 16, 256), ('layer4', 8, 512)):
    x = torch.randn(B, c, hw, hw, device='cuda').contiguous(memory_format=torch.channels_last)
    w = (torch.randn(c, 3, 3, c) / (3 * c ** 0.5)).contiguous()
    packed = torch.empty(int(lib.hmf_conv_winograd_packed_floats(c, c)), dtype=torch.float32)
    _lib.check(lib.hmf_conv_winograd_pack_host(w.data_ptr(), packed.data_ptr(), c, c), 'pack')
    wk = packed.cuda()
    bias = torch.randn(c, device='cuda')
    y = torch.empty(B, c, hw, hw, device='cuda').contiguous(memory_format=torch.channels_last)
    fn = lambda: _lib.check(lib.hmf_conv_winograd(x.data_ptr(), wk.data_ptr(), bias.data_ptr(), None, y.data_ptr(), B, hw, hw, c, c, 1, st), 'wino')
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 20 * 1e3)
    pair = hw <= 8
    tiles = (1 if pair else (hw + 15) // 16) * ((hw + 7) // 8)
    blocks = ((B + 1) // 2 if pair else B) * tiles * (c // 32)
    n = min(blocks, 4096)
    buf = np.zeros((n, 8), np.int64)
    _lib.check(lib.hmf_debug_wino_stamps(buf.ctypes.data_as(ctypes.c_void_p), n), 'stamps')
    cyc = (buf[:, 3] - buf[:, 0]).astype(np.float64)
    wall = (buf[:, 6] - buf[:, 5]).astype(np.float64) / 100.0
    nslab = c // 64
    mfma = nslab * 8 * 16 * 64
    print('%s %3d ch @%2d: %5.1f us/launch (median of back-to-back), %d work-groups; kernel first start -> last end %.1f us, last start +%.1f us | per work-group %.1f us at %.0f MHz: '
          'first patch %d cyc, main loop %d (of which fetch + barriers %d; MFMA issue time of a wave alone %d), output %d'
          % (name, c, hw, float(np.median(ts)), blocks, (buf[:, 6].max() - buf[:, 5].min()) / 100.0, (buf[:, 5].max() - buf[:, 5].min()) / 100.0, float(np.median(wall)),
             float(np.median(cyc / wall)), (buf[:, 1] - buf[:, 0]).mean(), (buf[:, 2] - buf[:, 1]).mean(), (buf[:, 4]).mean() - (buf[:, 1] - buf[:, 0]).mean(), mfma,
             (buf[:, 3] - buf[:, 2]).mean()))
