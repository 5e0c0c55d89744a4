"""Per-layer probe of the fused implicit-GEMM convolution (hmf_conv2d_nhwc) against eager torch (vendor convolution + bias/residual/ReLU passes) on the ResNet-50
layer shapes at B=32, 256x256 input: max abs difference and time per tile configuration.   python tools/conv_probe.py [B]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import torch.nn.functional as F
from humaniflow_amd import _lib

lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = 'cuda:0'


def shapes():
    out, cin, hw = [], 64, 64
    for li, (width, n) in enumerate(zip((64, 128, 256, 512), (3, 4, 6, 3))):
        for bi in range(n):
            s = 2 if (li > 0 and bi == 0) else 1
            out.append(('l%d.%d.c1' % (li + 1, bi), hw, cin, width, 1, 1, False))
            out.append(('l%d.%d.c2' % (li + 1, bi), hw, width, width, 3, s, False))
            hw2 = hw // s
            out.append(('l%d.%d.c3' % (li + 1, bi), hw2, width, width * 4, 1, 1, True))
            if bi == 0:
                out.append(('l%d.%d.ds' % (li + 1, bi), hw, cin, width * 4, 1, s, False))
            cin, hw = width * 4, hw2
    return out


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


seen, tot_mi, tot_best, rows = {}, 0.0, 0.0, []
for name, hw, cin, cout, ks, s, res in shapes():
    key = (hw, cin, cout, ks, s, res)
    if key in seen:
        tot_mi += seen[key][0]
        tot_best += seen[key][1]
        continue
    g = torch.Generator(device='cpu').manual_seed(hash(key) & 0xffff)
    x = torch.randn(B, cin, hw, hw, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, ks, ks, generator=g) / (cin * ks * ks) ** 0.5).to(dev).contiguous(memory_format=torch.channels_last)
    bias = torch.randn(cout, generator=g).to(dev)
    oh = hw // s
    r = torch.randn(B, cout, oh, oh, generator=g).to(dev).contiguous(memory_format=torch.channels_last) if res else None
    st = _lib.current_stream()

    def miopen():
        y = F.conv2d(x, w, bias, s, ks // 2)
        return torch.relu_(y if r is None else y.add_(r))
    ref = miopen()
    t_mi = timeit(miopen)
    y = torch.empty_like(ref)
    best, line = None, ''
    for tile in (64128, 64064):
        if tile % 1000 == 128 and cout % 128:
            continue
        fn = lambda: _lib.check(lib.hmf_conv2d_nhwc(x.data_ptr(), w.data_ptr(), bias.data_ptr(), None if r is None else r.data_ptr(), y.data_ptr(),
                                                    B, hw, hw, cin, cout, ks, s, 1, tile, st), 'conv')
        fn()
        err = (y - ref).abs().max().item()
        t = timeit(fn)
        line += ' %6d:%7.1f' % (tile, t)
        if err > 1e-3:
            line += '(ERR %.2e)' % err
        if best is None or t < best[0]:
            best = (t, tile)
    if True:
        wh = w.permute(0, 2, 3, 1).contiguous().cpu()                     # (Cout, ks, ks, Cin) row-major
        for cfg in ((16420, 16410, 8210) if ks == 3 else ()):
            sc = 128 if cfg % 10 else 64
            bn = (4 // ((cfg // 100) % 10)) * ((cfg // 10) % 10) * 32
            if (ks == 3 and s == 2 and cfg != 8210) or cin % sc or cout % bn or (ks == 3 and cfg % 10) or (cfg // 1000 == 8 and oh % 8) or (cfg // 1000 == 16 and oh % 16 and oh > 8):
                continue
            packed = torch.empty(cout * ks * ks * cin)
            _lib.check(lib.hmf_conv_patch_pack_host(wh.data_ptr(), packed.data_ptr(), cin, cout, ks, cfg), 'pack')
            pd = packed.to(dev)
            fn = lambda: _lib.check(lib.hmf_conv_patch(x.data_ptr(), pd.data_ptr(), bias.data_ptr(), None if r is None else r.data_ptr(), y.data_ptr(),
                                                       B, hw, hw, cin, cout, ks, s, 1, cfg, st), 'conv_patch')
            y.fill_(float('nan'))
            fn()
            err = (y - ref).abs().max().item()
            t = timeit(fn)
            line += ' p%5d:%7.1f' % (cfg, t)
            if not err < 1e-3:
                line += '(ERR %.2e)' % err
            elif best is None or t < best[0]:
                best = (t, cfg)
    if ks == 3 and s == 1 and cin % 64 == 0 and (hw >= 16 or hw == 8):
        wh = w.permute(0, 2, 3, 1).contiguous().cpu()
        packed = torch.empty(int(lib.hmf_conv_winograd_packed_floats(cin, cout)))
        _lib.check(lib.hmf_conv_winograd_pack_host(wh.data_ptr(), packed.data_ptr(), cin, cout), 'wpack')
        pd = packed.to(dev)
        for tc in (32,):
            fn = lambda: _lib.check(lib.hmf_conv_winograd(x.data_ptr(), pd.data_ptr(), bias.data_ptr(), None if r is None else r.data_ptr(), y.data_ptr(),
                                                          B, hw, hw, cin, cout, 1, st), 'wino')
            y.fill_(float('nan'))
            fn()
            err = (y - ref).abs().max().item()
            t = timeit(fn)
            line += ' wino%d:%7.1f' % (tc, t)
            if not err < 1e-3:
                line += '(ERR %.2e)' % err
            elif t < best[0]:
                best = (t, 'wino%d' % tc)
    fl = 2.0 * B * oh * oh * cout * cin * ks * ks
    print('%-9s hw%3d %4d->%4d k%d s%d  miopen %7.1f us (%5.1f TF)  ours best %7.1f us (%5.1f TF) tile %6s |%s' %
          (name, hw, cin, cout, ks, s, t_mi, fl / t_mi / 1e6, best[0], fl / best[0] / 1e6, best[1], line), flush=True)
    seen[key] = (t_mi, best[0])
    tot_mi += t_mi
    tot_best += best[0]
print('sum over the 52 non-stem convolutions: miopen+epilogue %.1f us, ours(best tile) %.1f us' % (tot_mi, tot_best))
