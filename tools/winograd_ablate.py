"""Timing experiments on k_conv_winograd over the ResNet-50 stride-1 3x3 shapes (variants other than 0 may compute wrong results):
where does a launch spend its time -- patch fetch, output transform, or the LDS + transform + MFMA loop?
    python tools/winograd_ablate.py [B]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from humaniflow_amd import _lib
lib = _lib.use_experiments_library()      # tools/libhumaniflow_amd_experiments.so: product entry points + timing variants
dev = 'cuda:0'


def timeit(fn, iters=100):
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
VARIANTS = ((100, 'fenced transform, 2 waves/SIMD'), (30, 'pipelined + 3 waves/SIMD'), (40, '64 channels per work-group'), (41, '64 channels, pipelined'), (10, '3 waves/SIMD'), (20, 'pipelined transform'), (1, 'no-fetch'), (2, 'no-epilogue'), (3, 'bare loop'))
if '--pf' in sys.argv:            # round 4: k_conv_winograd (300 = the round-3 defaults) vs the stage-pipelined k_conv_winograd_pf
    VARIANTS = ((300, 'product (round-3 kernel)'), (301, 'pf: 3 waves/SIMD + 2 tiles | 2 waves/SIMD'), (311, 'pf pipe occ3 1 tile'), (312, 'pf pipe occ3 2 tiles'), (314, 'pf pipe occ3 4 tiles'),
                (321, 'pf fenced occ2 1 tile'), (322, 'pf fenced occ2 2 tiles'), (332, 'pf pipe occ2 2 tiles'))
elif '--dma' in sys.argv:         # round 4: LDS-DMA patch fetch (500) and, with it, 64 output channels per work-group (540 fenced / 541 pipelined transform)
    VARIANTS = ((300, 'product'), (500, 'DMA fetch'), (540, 'DMA + 64 ch, fenced'), (541, 'DMA + 64 ch, pipelined'))
elif '--full' not in sys.argv:
    VARIANTS = VARIANTS[:4]
ROUNDS = 7           # variants are timed round-robin and the median over rounds is reported: a single pass is dominated by clock ramp / order effects (+-15 %)
for name, hw, c in (('layer1', 64, 64), ('layer2', 32, 128), ('layer3', 16, 256), ('layer4', 8, 512)):
    x = torch.randn(B, c, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(c, 3, 3, c) / (3 * c ** 0.5)).contiguous()
    packed = torch.empty(int(lib.hmf_conv_winograd_packed_floats(c, c)), dtype=torch.float32)
    _lib.check(lib.hmf_conv_winograd_pack_host(w.data_ptr(), packed.data_ptr(), c, c), 'pack')
    wk = packed.to(dev)
    bias = torch.randn(c, device=dev)
    y = torch.empty(B, c, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    st = _lib.current_stream()
    if '--pf' in sys.argv or '--dma' in sys.argv:        # the variants compute the same convolution: check them bit for bit against the round-3 kernel first
        res = torch.randn(B, c, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
        ref = torch.empty_like(y)
        _lib.check(lib.hmf_conv_winograd_variant(x.data_ptr(), wk.data_ptr(), bias.data_ptr(), res.data_ptr(), ref.data_ptr(), B, hw, hw, c, c, 1, 300, st), 'wino')
        for var, lab in VARIANTS[1:]:
            if hw <= 8 and var in (540, 541):
                continue
            y.zero_()
            _lib.check(lib.hmf_conv_winograd_variant(x.data_ptr(), wk.data_ptr(), bias.data_ptr(), res.data_ptr(), y.data_ptr(), B, hw, hw, c, c, 1, var, st), 'wino')
            assert torch.equal(y, ref), (name, var, (y - ref).abs().max().item())
    fl = 2.0 * B * hw * hw * c * c * 9
    line = '%-6s %5.2f GF direct (%.2f executed)' % (name, fl / 1e9, fl / 1e9 * 4 / 9)
    import statistics
    ts = {var: [] for var, _ in VARIANTS}
    for _ in range(ROUNDS):
        for var, lab in VARIANTS:
            fn = lambda: _lib.check(lib.hmf_conv_winograd_variant(x.data_ptr(), wk.data_ptr(), bias.data_ptr(), None, y.data_ptr(), B, hw, hw, c, c, 1, var, st), 'wino')
            try:
                ts[var].append(timeit(fn, iters=30))
            except RuntimeError:                      # variant not built for this shape
                ts[var].append(float('nan'))
    for var, lab in VARIANTS:
        line += ' | %s %5.1f us (%.1f-%.1f)' % (lab, statistics.median(ts[var]), min(ts[var]), max(ts[var]))
    print(line, flush=True)
