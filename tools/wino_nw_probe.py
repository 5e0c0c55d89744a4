"""k_conv_winograd (32 output channels per work-group, positions split over the waves) vs k_conv_wino_nw (128 output channels per work-group, a wave
owns all sixteen positions of one n-tile) on the encoder's stride-1 3x3 layer shapes: bitwise equality, and time per launch (round-robin rounds,
20 back-to-back launches per sample, torch events).   python tools/wino_nw_probe.py [B]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from humaniflow_amd import _lib
MODES = [int(m) for m in os.environ.get('NW_MODES', '').split(',') if m]
lib = _lib.use_experiments_library()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device('cuda:0')
st = torch.cuda.current_stream().cuda_stream
shapes = [('layer1 64x64 64->64 (old only)', 64, 64, 64), ('layer2 32x32 128->128', 32, 128, 128), ('layer3 16x16 256->256', 16, 256, 256), ('64x64 128->128', 64, 128, 128),
          ('32x32 256->256', 32, 256, 256), ('64x64 64->128', 64, 64, 128)]
g = torch.Generator().manual_seed(0)
for name, hw, cin, cout in shapes:
    x = torch.randn(B, hw, hw, cin, generator=g).to(dev)
    w = (torch.randn(cout, 3, 3, cin, generator=g) / (3 * cin ** 0.5))
    packed = torch.empty(int(lib.hmf_conv_winograd_packed_floats(cin, cout)), dtype=torch.float32)
    _lib.check(lib.hmf_conv_winograd_pack_host(w.data_ptr(), packed.data_ptr(), cin, cout))
    packed = packed.to(dev)
    bias = torch.randn(cout, generator=g).to(dev)
    res = torch.randn(B, hw, hw, cout, generator=g).to(dev)
    y0, y1 = torch.empty(B, hw, hw, cout, device=dev), torch.empty(B, hw, hw, cout, device=dev)
    old = lambda: _lib.check(lib.hmf_conv_winograd(x.data_ptr(), packed.data_ptr(), bias.data_ptr(), res.data_ptr(), y0.data_ptr(), B, hw, hw, cin, cout, 1, st))
    new = lambda: lib.hmf_conv_winograd_nw(x.data_ptr(), packed.data_ptr(), bias.data_ptr(), res.data_ptr(), y1.data_ptr(), B, hw, hw, cin, cout, 1, st)
    old()
    rc = new()
    torch.cuda.synchronize()
    if rc != 0:
        print('%-34s new form: unsupported (rc %d)' % (name, rc))
        forms = [('old', old)]
    else:
        print('%-34s max |new - old| = %.3g   bitwise equal: %s' % (name, (y1 - y0).abs().max().item(), torch.equal(y0, y1)))
        forms = [('old', old), ('new', new)]
        for m in MODES:
            forms.append(('new/mode%d' % m, (lambda m_: (lambda: (lib.hmf_debug_set_wino_nw_mode(m_), new(), lib.hmf_debug_set_wino_nw_mode(0))))(m)))
    ts = {k: [] for k, _ in forms}
    for rnd in range(7):
        for k, fn in forms:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                fn()
            b.record()
            torch.cuda.synchronize()
            if rnd:
                ts[k].append(a.elapsed_time(b) / 20 * 1e3)
    fl = 2.0 * B * hw * hw * cin * cout * 9
    print('    ' + '   '.join('%s %.1f us (%.0f TFLOP/s direct-form, %.0f executed)' % (k, np.median(v), fl / np.median(v) / 1e6, fl / np.median(v) / 1e6 * 4 / 9) for k, v in ts.items()))
