"""Randomised shape fuzz of the encoder kernels against float64 torch convolutions (GPU).   python tools/conv_fuzz.py [cases] [seed]"""
import os
import random
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import torch.nn.functional as F
from humaniflow_amd import _lib

lib = _lib.load()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
st = None
worst = {}


def check(name, y, ref, tol=3e-5):
    torch.cuda.synchronize()
    err = (y.cpu().double() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    assert torch.isfinite(y).all(), name
    worst[name.split()[0]] = max(worst.get(name.split()[0], 0.0), err)
    assert err < tol, (name, err)


for it in range(cases):
    g = torch.Generator().manual_seed(rnd.randrange(1 << 30))
    kind = rnd.choice(['igemm', 'patch', 'wino', 'dual', 'stem'])
    B, H, W = rnd.randint(1, 5), rnd.randint(1, 40), rnd.randint(1, 40)
    res, relu = rnd.random() < 0.5, rnd.random() < 0.5
    cl = lambda t: t.cuda().contiguous(memory_format=torch.channels_last)
    if kind == 'stem':
        H, W = rnd.randint(2, 70), rnd.randint(2, 70)
        x, w, b = torch.randn(B, 18, H, W, generator=g), torch.randn(64, 18, 7, 7, generator=g) / 30, torch.randn(64, generator=g)
        ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), 2, 3))
        pk = torch.empty(int(lib.hmf_stem_packed_floats()))
        _lib.check(lib.hmf_stem_pack_host(w.contiguous().data_ptr(), pk.data_ptr()), 'pack')
        y = torch.full(ref.shape, float('nan'), device='cuda').contiguous(memory_format=torch.channels_last)
        xd, pd, bd = x.cuda(), pk.cuda(), b.cuda()
        _lib.check(lib.hmf_stem_conv(xd.data_ptr(), pd.data_ptr(), bd.data_ptr(), y.data_ptr(), B, 18, H, W, 1, st), 'stem')
        check('stem B%d %dx%d' % (B, H, W), y, ref)
        z = torch.full(F.max_pool2d(ref, 3, 2, 1).shape, float('nan'), device='cuda').contiguous(memory_format=torch.channels_last)
        _lib.check(lib.hmf_maxpool3x3s2_nhwc(y.data_ptr(), z.data_ptr(), B, ref.shape[2], ref.shape[3], 64, st), 'pool')
        torch.cuda.synchronize()
        assert torch.equal(z, F.max_pool2d(y, 3, 2, 1))
        continue
    if kind == 'dual':
        cin, cin2, cout, s2 = 32 * rnd.randint(1, 6), 32 * rnd.randint(1, 6), 64 * rnd.randint(1, 4), rnd.choice([1, 2])
        H2, W2 = (H - 1) * s2 + 1 + rnd.randint(0, s2 - 1), (W - 1) * s2 + 1 + rnd.randint(0, s2 - 1)
        h, x = torch.randn(B, cin, H, W, generator=g), torch.randn(B, cin2, H2, W2, generator=g)
        w3, wd, b = torch.randn(cout, cin, generator=g) / cin ** 0.5, torch.randn(cout, cin2, generator=g) / cin2 ** 0.5, torch.randn(cout, generator=g)
        ref = F.conv2d(h.double(), w3.double()[:, :, None, None]) + F.conv2d(x.double(), wd.double()[:, :, None, None], None, s2) + b.double()[None, :, None, None]
        ref = ref.clamp_min(0) if relu else ref
        y = torch.full(ref.shape, float('nan'), device='cuda').contiguous(memory_format=torch.channels_last)
        hd, xd, wc, bd = cl(h), cl(x), torch.cat([w3, wd], 1).contiguous().cuda(), b.cuda()
        _lib.check(lib.hmf_conv1x1_dual_nhwc(hd.data_ptr(), xd.data_ptr(), wc.data_ptr(), bd.data_ptr(), y.data_ptr(), B, H, W, cin, H2, W2, cin2, s2, cout,
                                             int(relu), st), 'dual')
        check('dual B%d %dx%d %d+%d->%d s%d' % (B, H, W, cin, cin2, cout, s2), y, ref)
        continue
    ks = 3 if kind in ('patch', 'wino') else rnd.choice([1, 3])
    stride = 1 if kind == 'wino' else rnd.choice([1, 2])
    mult = 64 if kind in ('patch', 'wino') else 32
    cin, cout = mult * rnd.randint(1, 4), (32 if kind == 'wino' else 64) * rnd.randint(1, 4)
    x, w, b = torch.randn(B, cin, H, W, generator=g), torch.randn(cout, cin, ks, ks, generator=g) / (cin * ks * ks) ** 0.5, torch.randn(cout, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride, ks // 2)
    r = torch.randn(ref.shape, generator=g) if res else None
    ref = ref + r.double() if res else ref
    ref = ref.clamp_min(0) if relu else ref
    y = torch.full(ref.shape, float('nan'), device='cuda').contiguous(memory_format=torch.channels_last)
    xd, bd, rd = cl(x), b.cuda(), (cl(r) if res else None)
    rp = None if rd is None else rd.data_ptr()
    wh = w.permute(0, 2, 3, 1).contiguous()
    name = '%s B%d %dx%d %d->%d k%d s%d' % (kind, B, H, W, cin, cout, ks, stride)
    if kind == 'igemm':
        tile = rnd.choice([0, 64064] + ([64128] if cout % 128 == 0 else []))
        wd = cl(w)
        _lib.check(lib.hmf_conv2d_nhwc(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), rp, y.data_ptr(), B, H, W, cin, cout, ks, stride, int(relu), tile, st), name)
    elif kind == 'patch':
        cfg = 8210 if stride == 2 else rnd.choice([16410, 8210] + ([16420] if cout % 64 == 0 else []))
        pk = torch.empty(int(lib.hmf_conv_patch_packed_floats(cin, cout, 3)))
        _lib.check(lib.hmf_conv_patch_pack_host(wh.data_ptr(), pk.data_ptr(), cin, cout, 3, cfg), 'pack')
        pd = pk.cuda()
        _lib.check(lib.hmf_conv_patch(xd.data_ptr(), pd.data_ptr(), bd.data_ptr(), rp, y.data_ptr(), B, H, W, cin, cout, 3, stride, int(relu), cfg, st), name)
    else:
        pk = torch.empty(int(lib.hmf_conv_winograd_packed_floats(cin, cout)))
        _lib.check(lib.hmf_conv_winograd_pack_host(wh.data_ptr(), pk.data_ptr(), cin, cout), 'pack')
        pd = pk.cuda()
        _lib.check(lib.hmf_conv_winograd(xd.data_ptr(), pd.data_ptr(), bd.data_ptr(), rp, y.data_ptr(), B, H, W, cin, cout, int(relu), st), name)
    check(name, y, ref, 5e-5 if kind == 'wino' else 3e-5)
print('%d cases passed; worst relative error per kernel:' % cases, {k: '%.2e' % v for k, v in worst.items()})
