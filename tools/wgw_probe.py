"""Winograd weight-gradient kernel on the ResNet-18 stride-1 3x3 shapes at B = 72: time vs the number of tile slices (atomic passes) and vs the
direct kernel.   python tools/wgw_probe.py [B]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import statistics
import torch
from humaniflow_amd import _lib
lib = _lib.use_experiments_library()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 72
st = _lib.current_stream()
CL = torch.channels_last


def timeit(fn, iters=20, rounds=5):
    out = []
    for _ in range(rounds):
        fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) / iters * 1e3)
    return statistics.median(out)


for name, hw, c in (('layer1', 64, 64), ('layer2', 32, 128), ('layer3', 16, 256), ('layer4', 8, 512)):
    x = torch.randn(B, c, hw, hw, device='cuda').contiguous(memory_format=CL)
    gy = torch.randn(B, c, hw, hw, device='cuda').contiguous(memory_format=CL)
    dw = torch.empty(c, c, 3, 3, device='cuda')
    dwd = torch.empty(c, 3, 3, c, device='cuda')
    ws = torch.empty(int(lib.hmf_conv_wgrad_winograd_workspace_floats(c, c)), device='cuda')
    fl = 2.0 * B * hw * hw * c * c * 9
    line = '%s %3d ch @%2d, B=%d: %.1f GF direct' % (name, c, hw, B, fl / 1e9)
    t = timeit(lambda: _lib.check(lib.hmf_conv_wgrad(gy.data_ptr(), x.data_ptr(), dwd.data_ptr(), B, hw, hw, c, c, 3, 1, 0, st), 'direct'))
    line += ' | direct %.1f us (%.0f TF)' % (t, fl / t / 1e6)
    for single in (1, 0, 2, 4):
        lib.hmf_debug_set_wgw_single_stage(single)
        for target in ((512,) if single >= 2 else (256, 512, 768, 1024)):
            lib.hmf_debug_set_wgw_target_wgs(target)
            t = timeit(lambda: _lib.check(lib.hmf_conv_wgrad_winograd(gy.data_ptr(), x.data_ptr(), dw.data_ptr(), ws.data_ptr(), B, hw, hw, c, c, st), 'wino'))
            line += ' | %s, %d WGs: %.1f us (%.0f TF)' % ({1: 'single-staged 32-tile chunks', 0: 'two stages of 16 tiles (product)', 2: 'two stages, NO accumulation pass (timing only)', 4: 'two stages, float ATOMICS instead of partial stores + reduce (timing only)'}[single], target, t, fl / t / 1e6)
    lib.hmf_debug_set_wgw_target_wgs(512)
    lib.hmf_debug_set_wgw_single_stage(0)
    # the zero fill of M and the inverse transform alone: one image's worth of tiles
    x1, g1 = x[:1].contiguous(memory_format=CL), gy[:1].contiguous(memory_format=CL)
    t1 = timeit(lambda: _lib.check(lib.hmf_conv_wgrad_winograd(g1.data_ptr(), x1.data_ptr(), dw.data_ptr(), ws.data_ptr(), 1, hw, hw, c, c, st), 'wino'))
    line += ' | B=1 (partial stores + reduce + inverse transform floor): %.1f us' % t1
    print(line, flush=True)
