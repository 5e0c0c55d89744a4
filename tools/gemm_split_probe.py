"""hmf_conv1x1_gemm_split (csrc/conv_gemm_split.hip, bf16 pieces of fp32 operands) against hmf_conv1x1_gemm (fp32 MFMA) on every 1x1 shape of ResNet-50:
error against a float64 product and time per launch, every instantiated tile.
    python tools/gemm_split_probe.py [B]"""
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
ABL = '--ablate' in sys.argv
TILES = {3: [3064128, 9000001, 9000002, 9000003, 9000004, 9000007], 6: []} if ABL else {3: [2128128, 2064128, 3064064, 4064064, 3064128, 4064128, 3128128, 2128064, 3128064], 6: [2128128, 2064128, 3064064, 4064064, 3064128, 3128128, 2128064]}


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


print('B = %d; us per launch; max |err| against a float64 product over a 4096-row sample, in units of the output rms' % B)
for name, hw, k1, n, k2, h2, s2, res in SHAPES:
    x = torch.randn(B, k1, hw, hw, device='cuda').contiguous(memory_format=CL)
    x2 = torch.randn(B, k2, h2, h2, device='cuda').contiguous(memory_format=CL) if k2 else None
    w = torch.randn(n, k1 + k2, device='cuda') / (k1 + k2) ** 0.5
    bias = torch.randn(n, device='cuda')
    r = torch.randn(B, n, hw, hw, device='cuda').contiguous(memory_format=CL) if res else None
    y = torch.empty(B, n, hw, hw, device='cuda').contiguous(memory_format=CL)
    flops = 2.0 * B * hw * hw * n * (k1 + k2)
    # float64 reference on a sample of rows
    xm = x.permute(0, 2, 3, 1).reshape(-1, k1)
    rows = torch.randperm(xm.shape[0], device='cuda')[:4096]
    a64 = xm[rows].double()
    if k2:
        x2m = x2[:, :, ::s2, ::s2].permute(0, 2, 3, 1).reshape(-1, k2)
        a64 = torch.cat([a64, x2m[rows].double()], 1)
    ref = a64 @ w.double().t() + bias.double()
    if res:
        ref = ref + r.permute(0, 2, 3, 1).reshape(-1, n)[rows].double()
    ref = torch.relu(ref)
    rms = float(ref.pow(2).mean().sqrt())

    def err():
        torch.cuda.synchronize()
        return float((y.permute(0, 2, 3, 1).reshape(-1, n)[rows].double() - ref).abs().max()) / rms

    args = (x.data_ptr(), None if x2 is None else x2.data_ptr())
    tail = (bias.data_ptr(), None if r is None else r.data_ptr(), y.data_ptr(), B, hw, hw, k1, h2, h2, k2, s2, n, 1)
    f32 = lambda: _lib.check(lib.hmf_conv1x1_gemm(*args, w.data_ptr(), *tail, 0, st), 'gemm')
    f32()
    e32 = err()
    t32 = timeit(f32)
    cols = ['fp32 %6.1f (%5.1f TF) err %.1e' % (t32, flops / t32 / 1e6, e32)]
    for terms in ((3,) if ABL else (3, 6)):
        wp = torch.empty(int(lib.hmf_conv1x1_split_packed_bytes(n, k1 + k2, terms)), dtype=torch.uint8, device='cuda')
        _lib.check(lib.hmf_conv1x1_split_pack(w.data_ptr(), wp.data_ptr(), n, k1 + k2, terms, st), 'pack')
        best = None
        for tile in [0] + TILES[terms]:
            if ((tile % 1000) == 128 or tile >= 9000000) and n % 128:
                continue
            fn = lambda: _lib.check(lib.hmf_conv1x1_gemm_split(*args, wp.data_ptr(), *tail, terms, tile, st), 'split')
            y.zero_()
            fn()
            e = err()
            t = timeit(fn)
            cols.append('x%d/%d %5.1f%s' % (terms, tile, t, '' if (e < (1e-4 if terms == 3 else 2e-5) or tile >= 9000000) else ' ERR %.0e' % e))
    print('%-34s %5.2f GF | %s' % (name, flops / 1e9, ' | '.join(cols)))
