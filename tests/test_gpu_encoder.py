"""Fused implicit-GEMM convolution of the encoder (hmf_conv2d_nhwc) through the C ABI vs a float64 torch convolution on the CPU
(the plain-torch reference for a floating-point kernel; tolerance 2e-5 relative to the output scale -- fp32 accumulation over
K <= 4608 terms), and the whole folded encoder vs the oracle's op-for-op ResNet (models/resnet.py restatement)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _case(B, H, W, cin, cout, ks, stride, res, relu, tile, seed=0):
    from humaniflow_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, ks, ks, generator=g) / (cin * ks * ks) ** 0.5
    bias = torch.randn(cout, generator=g)
    ref = F.conv2d(x.double(), w.double(), bias.double(), stride, ks // 2)
    r = torch.randn(ref.shape, generator=g) if res else None
    if res:
        ref = ref + r.double()
    if relu:
        ref = ref.clamp_min(0)
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    wd = w.cuda().contiguous(memory_format=torch.channels_last)
    bd = bias.cuda()
    rd = r.cuda().contiguous(memory_format=torch.channels_last) if res else None
    y = torch.full(ref.shape, float('nan'), device='cuda').contiguous(memory_format=torch.channels_last)
    _lib.check(lib.hmf_conv2d_nhwc(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), None if rd is None else rd.data_ptr(), y.data_ptr(),
                                   B, H, W, cin, cout, ks, stride, int(relu), tile, _lib.current_stream()), 'hmf_conv2d_nhwc')
    torch.cuda.synchronize()
    err = (y.cpu().double() - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), (B, H, W, cin, cout, ks, stride, res, relu, tile, err)


@pytest.mark.parametrize('tile', [0, 64128, 64064])
@pytest.mark.parametrize('ks,stride', [(1, 1), (1, 2), (3, 1), (3, 2)])
def test_conv_matches_float64_reference(tile, ks, stride):
    _case(2, 12, 12, 64, 128, ks, stride, res=True, relu=True, tile=tile)


def test_conv_ragged_rows_and_flags():
    # M = 49 / 147 rows: partial tiles, rows beyond M neither read nor written; bias-only / residual-only / no ReLU paths
    _case(1, 7, 7, 32, 64, 3, 1, res=False, relu=False, tile=64064, seed=1)
    _case(3, 7, 7, 96, 192, 3, 1, res=True, relu=False, tile=64064, seed=2)
    _case(3, 7, 7, 96, 192, 1, 1, res=False, relu=True, tile=64064, seed=3)
    _case(1, 9, 5, 64, 128, 3, 2, res=True, relu=True, tile=64128, seed=4)          # non-square, odd sizes, stride 2
    _case(2, 8, 8, 2048, 512, 1, 1, res=False, relu=True, tile=0, seed=5)           # deepest K of the 1x1 layers


def _gemm_case(B, OH, OW, k1, n, tile, k2=0, stride2=1, res=False, relu=True, bias=True, seed=0, terms=0):
    """hmf_conv1x1_gemm (LDS-DMA pipelined GEMM, csrc/conv_gemm.hip) vs float64 torch: y = relu?(conv1x1(x) (+ conv1x1(x2, stride2)) + bias (+ res))."""
    from humaniflow_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, k1, OH, OW, generator=g)
    w = torch.randn(n, k1 + k2, generator=g) / (k1 + k2) ** 0.5
    bv = torch.randn(n, generator=g) if bias else None
    ref = F.conv2d(x.double(), w[:, :k1].double()[:, :, None, None], None if bv is None else bv.double())
    x2 = None
    H2 = W2 = 0
    if k2:
        H2, W2 = (OH - 1) * stride2 + 1 + (stride2 - 1) * (seed % 2), (OW - 1) * stride2 + 1           # odd and even source sizes
        x2 = torch.randn(B, k2, H2, W2, generator=g)
        ref = ref + F.conv2d(x2.double(), w[:, k1:].double()[:, :, None, None], None, stride2)
    r = torch.randn(ref.shape, generator=g) if res else None
    if res:
        ref = ref + r.double()
    if relu:
        ref = ref.clamp_min(0)
    CL = torch.channels_last
    xd, wd = x.cuda().contiguous(memory_format=CL), w.cuda().contiguous()
    x2d = x2.cuda().contiguous(memory_format=CL) if k2 else None
    bd = bv.cuda() if bias else None
    rd = r.cuda().contiguous(memory_format=CL) if res else None
    y = torch.full(ref.shape, float('nan'), device='cuda').contiguous(memory_format=CL)
    guard = torch.full((4096,), 7.0, device='cuda')                      # allocated right after y: a store beyond M would likely land here
    if terms:
        wp = torch.empty(int(lib.hmf_conv1x1_split_packed_bytes(n, k1 + k2, terms)), dtype=torch.uint8, device='cuda')
        _lib.check(lib.hmf_conv1x1_split_pack(wd.data_ptr(), wp.data_ptr(), n, k1 + k2, terms, _lib.current_stream()), 'hmf_conv1x1_split_pack')
        _lib.check(lib.hmf_conv1x1_gemm_split(xd.data_ptr(), None if x2d is None else x2d.data_ptr(), wp.data_ptr(), None if bd is None else bd.data_ptr(),
                                              None if rd is None else rd.data_ptr(), y.data_ptr(), B, OH, OW, k1, H2, W2, k2, stride2, n, int(relu), terms,
                                              tile, _lib.current_stream()), 'hmf_conv1x1_gemm_split')
    else:
        _lib.check(lib.hmf_conv1x1_gemm(xd.data_ptr(), None if x2d is None else x2d.data_ptr(), wd.data_ptr(), None if bd is None else bd.data_ptr(),
                                        None if rd is None else rd.data_ptr(), y.data_ptr(), B, OH, OW, k1, H2, W2, k2, stride2, n, int(relu), tile,
                                        _lib.current_stream()), 'hmf_conv1x1_gemm')
    torch.cuda.synchronize()
    assert bool((guard == 7.0).all())
    err = (y.cpu().double() - ref).abs().max().item()
    # fp32 MFMA and the 6-term split (fp32-level products) share one bound; the 3-term split keeps ~16 mantissa bits per product
    tol = 2e-4 if terms == 3 else 2e-5
    assert err < tol * max(1.0, ref.abs().max().item()), (B, OH, OW, k1, k2, n, tile, terms, res, relu, err)
    return y


@pytest.mark.parametrize('tile', [0, 64064, 3064064, 3064128, 2064064, 2064128])
def test_conv1x1_gemm_matches_float64_reference(tile):
    _gemm_case(2, 16, 16, 256, 128, tile, res=True, relu=True)                                   # M = 512: whole tiles
    _gemm_case(3, 7, 7, 96, 256, tile, res=True, relu=False, seed=1)                             # M = 147: ragged last tile, 3 k-chunks (< pipeline depth)
    _gemm_case(1, 5, 9, 32, 128, tile, res=False, relu=True, bias=False, seed=2)                 # ONE k-chunk, M = 45 < every tile height
    _gemm_case(2, 8, 8, 2048, 512, tile, res=False, relu=True, seed=3)                           # deepest K of the 1x1 layers (64 chunks)
    _gemm_case(2, 6, 6, 128, 256, tile, k2=64, stride2=2, relu=True, seed=4)                     # downsample branch as a second K range, stride 2
    _gemm_case(2, 6, 6, 64, 128, tile, k2=160, stride2=2, res=True, relu=True, seed=5)           # ... odd source size, 2 + 5 chunks
    _gemm_case(1, 12, 12, 64, 256, tile, k2=64, stride2=1, relu=False, seed=6)                   # layer1.0: stride-1 branch


@pytest.mark.parametrize('terms', [3, 6])
@pytest.mark.parametrize('tile', [0, 2128128, 2064128, 3064064, 4064064])
def test_conv1x1_gemm_split_matches_float64_reference(tile, terms):
    """The opt-in split-operand GEMM (bf16 pieces of fp32 operands on the bf16 matrix pipe) on the cases of the fp32 GEMM: the 6-term form meets
    the fp32 kernel's bound, the 3-term form 10x of it."""
    _gemm_case(2, 16, 16, 256, 128, tile, res=True, relu=True, terms=terms)
    _gemm_case(3, 7, 7, 96, 256, tile, res=True, relu=False, seed=1, terms=terms)                # ragged last tile (M = 147 < 2 x 128)
    _gemm_case(1, 5, 9, 32, 128, tile, res=False, relu=True, bias=False, seed=2, terms=terms)    # ONE k-chunk, M = 45
    _gemm_case(2, 8, 8, 2048, 512, tile, res=False, relu=True, seed=3, terms=terms)              # 64 chunks
    _gemm_case(2, 6, 6, 128, 256, tile, k2=64, stride2=2, relu=True, seed=4, terms=terms)        # downsample branch as a second K range
    _gemm_case(2, 6, 6, 64, 128, tile, k2=160, stride2=2, res=True, relu=True, seed=5, terms=terms)
    _gemm_case(1, 12, 12, 64, 256, tile, k2=64, stride2=1, relu=False, seed=6, terms=terms)


def test_conv1x1_gemm_split_rejects_bad_arguments():
    from humaniflow_amd import _lib
    lib = _lib.load()
    p = torch.zeros(1 << 16, device='cuda').data_ptr()
    assert lib.hmf_conv1x1_split_packed_bytes(128, 64, 3) == 128 * 64 * 4 and lib.hmf_conv1x1_split_packed_bytes(128, 64, 6) == 128 * 64 * 6
    assert lib.hmf_conv1x1_split_packed_bytes(128, 48, 3) == -1 and lib.hmf_conv1x1_split_packed_bytes(128, 64, 4) == -1
    assert lib.hmf_conv1x1_split_pack(p, p, 128, 64, 5, None) == -1
    assert lib.hmf_conv1x1_gemm_split(p, None, p, None, None, p, 1, 4, 4, 64, 0, 0, 0, 1, 64, 1, 4, 0, None) == -1          # terms
    assert lib.hmf_conv1x1_gemm_split(p, None, p, None, None, p, 1, 4, 4, 48, 0, 0, 0, 1, 64, 1, 3, 0, None) == -2          # K % 32
    assert lib.hmf_conv1x1_gemm_split(p, None, p, None, None, p, 1, 4, 4, 64, 0, 0, 0, 1, 64, 1, 3, 2064128, None) == -1    # tile needs N % 128
    assert lib.hmf_conv1x1_gemm_split(p, None, p, None, None, p, 1, 4, 4, 64, 0, 0, 0, 1, 128, 1, 3, 3064128, None) == -1   # not a product tile
    assert lib.hmf_conv1x1_gemm_split(p, None, p + 4, None, None, p, 1, 4, 4, 64, 0, 0, 0, 1, 64, 1, 3, 0, None) == -1      # misaligned


def test_conv1x1_gemm_agrees_with_igemm_and_rejects_bad_shapes():
    from humaniflow_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(9)
    B, H, W, cin, cout = 4, 16, 16, 1024, 256                              # layer3 conv1 shape at a small batch
    x = torch.randn(B, cin, H, W, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, generator=g) / cin ** 0.5).cuda()
    b = torch.randn(cout, generator=g).cuda()
    ya = torch.empty(B, cout, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
    yb = torch.empty_like(ya)
    _lib.check(lib.hmf_conv2d_nhwc(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, ya.data_ptr(), B, H, W, cin, cout, 1, 1, 1, 0, _lib.current_stream()), 'igemm')
    _lib.check(lib.hmf_conv1x1_gemm(x.data_ptr(), None, w.data_ptr(), b.data_ptr(), None, yb.data_ptr(), B, H, W, cin, 0, 0, 0, 1, cout, 1, 0,
                                    _lib.current_stream()), 'gemm')
    assert (ya - yb).abs().max().item() < 1e-5 * max(1.0, ya.abs().max().item())
    p = x.data_ptr()
    assert lib.hmf_conv1x1_gemm(p, None, p, None, None, p, 1, 4, 4, 48, 0, 0, 0, 1, 64, 1, 0, None) == -2          # K % 32
    assert lib.hmf_conv1x1_gemm(p, None, p, None, None, p, 1, 4, 4, 64, 0, 0, 0, 1, 96, 1, 0, None) == -2          # N % 64
    assert lib.hmf_conv1x1_gemm(p, None, p, None, None, p, 1, 4, 4, 64, 0, 0, 0, 1, 64, 1, 3064128, None) == -1    # tile needs N % 128
    assert lib.hmf_conv1x1_gemm(p, None, p, None, None, p, 1, 4, 4, 64, 0, 0, 0, 1, 128, 1, 128128, None) == -1    # not a product tile
    assert lib.hmf_conv1x1_gemm(p, p, p, None, None, p, 1, 4, 4, 64, 9, 9, 64, 2, 64, 1, 0, None) == -1            # x2 size / stride mismatch
    assert lib.hmf_conv1x1_gemm(p + 4, None, p, None, None, p, 1, 4, 4, 64, 0, 0, 0, 1, 64, 1, 0, None) == -1      # misaligned


def _patch_case(B, H, W, cin, cout, ks, stride, res, relu, cfg, seed=0):
    from humaniflow_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, ks, ks, generator=g) / (cin * ks * ks) ** 0.5
    bias = torch.randn(cout, generator=g)
    ref = F.conv2d(x.double(), w.double(), bias.double(), stride, ks // 2)
    r = torch.randn(ref.shape, generator=g) if res else None
    if res:
        ref = ref + r.double()
    if relu:
        ref = ref.clamp_min(0)
    wh = w.permute(0, 2, 3, 1).contiguous()
    packed = torch.empty(int(lib.hmf_conv_patch_packed_floats(cin, cout, ks)))
    _lib.check(lib.hmf_conv_patch_pack_host(wh.data_ptr(), packed.data_ptr(), cin, cout, ks, cfg), 'pack')
    xd, pd, bd = x.cuda().contiguous(memory_format=torch.channels_last), packed.cuda(), bias.cuda()
    rd = r.cuda().contiguous(memory_format=torch.channels_last) if res else None
    y = torch.full(ref.shape, float('nan'), device='cuda').contiguous(memory_format=torch.channels_last)
    _lib.check(lib.hmf_conv_patch(xd.data_ptr(), pd.data_ptr(), bd.data_ptr(), None if rd is None else rd.data_ptr(), y.data_ptr(),
                                  B, H, W, cin, cout, ks, stride, int(relu), cfg, _lib.current_stream()), 'hmf_conv_patch')
    torch.cuda.synchronize()
    err = (y.cpu().double() - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), (B, H, W, cin, cout, ks, stride, res, relu, cfg, err)
    # the small-launch form (slab loop split over work-groups, partial sums added in a fixed order)
    nbytes = int(lib.hmf_conv_patch_workspace_bytes(B, H, W, cin, cout, ks, stride, cfg))
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device='cuda')
    y2, y3 = torch.full_like(y, float('nan')), torch.full_like(y, float('nan'))
    for out in (y2, y3):
        ws.fill_(255)
        _lib.check(lib.hmf_conv_patch_ws(xd.data_ptr(), pd.data_ptr(), bd.data_ptr(), None if rd is None else rd.data_ptr(), out.data_ptr(), B, H, W, cin, cout, ks,
                                         stride, int(relu), cfg, ws.data_ptr() if nbytes else None, nbytes, _lib.current_stream()), 'hmf_conv_patch_ws')
    torch.cuda.synchronize()
    assert torch.equal(y2, y3)
    if nbytes:
        assert (y2.cpu().double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item()), (B, H, W, cin, cout, 'split')
    else:
        assert torch.equal(y2, y)
    return nbytes


@pytest.mark.parametrize('cfg', [16420, 16410, 8210])
def test_patch_conv3x3_matches_float64_reference(cfg):
    _patch_case(2, 16, 16, 128, 128, 3, 1, True, True, cfg)                 # two channel slabs, whole tiles
    _patch_case(1, 11, 19, 64, 64, 3, 1, False, False, cfg, seed=1)         # ragged tiles: masked stores, zero-filled patch borders


def test_patch_conv3x3_stride2():
    _patch_case(2, 16, 16, 128, 64, 3, 2, False, True, 8210)
    _patch_case(1, 13, 9, 64, 128, 3, 2, True, False, 8210, seed=3)
    # the predict shape (one image): layer3.0 / layer4.0 sized reductions split over 4 / 8 slab ranges
    assert _patch_case(1, 32, 32, 256, 256, 3, 2, False, True, 8210, seed=4) > 0
    assert _patch_case(1, 16, 16, 512, 512, 3, 2, True, True, 8210, seed=5) > 0
    # 32-channel slabs (cfg 8211: half the LDS per work-group)
    _patch_case(2, 16, 16, 128, 64, 3, 2, False, True, 8211, seed=6)
    _patch_case(1, 13, 9, 64, 128, 3, 2, True, False, 8211, seed=7)
    assert _patch_case(1, 16, 16, 512, 512, 3, 2, True, True, 8211, seed=8) > 0


def _wino_case(B, H, W, cin, cout, res, relu, seed=0):
    from humaniflow_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    bias = torch.randn(cout, generator=g)
    ref = F.conv2d(x.double(), w.double(), bias.double(), 1, 1)
    r = torch.randn(ref.shape, generator=g) if res else None
    if res:
        ref = ref + r.double()
    if relu:
        ref = ref.clamp_min(0)
    wh = w.permute(0, 2, 3, 1).contiguous()
    packed = torch.empty(int(lib.hmf_conv_winograd_packed_floats(cin, cout)))
    _lib.check(lib.hmf_conv_winograd_pack_host(wh.data_ptr(), packed.data_ptr(), cin, cout), 'pack')
    xd, pd, bd = x.cuda().contiguous(memory_format=torch.channels_last), packed.cuda(), bias.cuda()
    rd = r.cuda().contiguous(memory_format=torch.channels_last) if res else None
    y = torch.full(ref.shape, float('nan'), device='cuda').contiguous(memory_format=torch.channels_last)
    _lib.check(lib.hmf_conv_winograd(xd.data_ptr(), pd.data_ptr(), bd.data_ptr(), None if rd is None else rd.data_ptr(), y.data_ptr(),
                                     B, H, W, cin, cout, int(relu), _lib.current_stream()), 'hmf_conv_winograd')
    torch.cuda.synchronize()
    err = (y.cpu().double() - ref).abs().max().item()
    assert err < 3e-5 * max(1.0, ref.abs().max().item()), (B, H, W, cin, cout, res, relu, err)      # Winograd: a few more roundings than direct
    # the small-batch form (slab loop split over work-groups, partial sums added in a fixed order): same bar, deterministic, and the unsplit entry's
    # result when the shape is not split
    nbytes = int(lib.hmf_conv_winograd_workspace_bytes(B, H, W, cin, cout))
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device='cuda')
    y2 = torch.full(ref.shape, float('nan'), device='cuda').contiguous(memory_format=torch.channels_last)
    y3 = torch.full(ref.shape, float('nan'), device='cuda').contiguous(memory_format=torch.channels_last)
    for out in (y2, y3):
        ws.fill_(255)
        _lib.check(lib.hmf_conv_winograd_ws(xd.data_ptr(), pd.data_ptr(), bd.data_ptr(), None if rd is None else rd.data_ptr(), out.data_ptr(),
                                            B, H, W, cin, cout, int(relu), ws.data_ptr() if nbytes else None, nbytes, _lib.current_stream()), 'hmf_conv_winograd_ws')
    torch.cuda.synchronize()
    assert torch.equal(y2, y3)
    if nbytes:
        assert (y2.cpu().double() - ref).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item()), (B, H, W, cin, cout, 'split')
        assert lib.hmf_conv_winograd_ws(xd.data_ptr(), pd.data_ptr(), bd.data_ptr(), None, y3.data_ptr(), B, H, W, cin, cout, int(relu), ws.data_ptr(), nbytes - 16,
                                        _lib.current_stream()) < 0
    else:
        assert torch.equal(y2, y)
    return nbytes


def test_winograd_conv3x3_matches_float64_reference():
    _wino_case(2, 16, 16, 128, 64, True, True)                  # two slabs, two channel tiles, whole pixel tiles
    _wino_case(1, 8, 32, 64, 32, False, False, seed=1)          # one tile row, two tile columns
    _wino_case(2, 11, 19, 64, 96, True, True, seed=2)           # ragged: odd sizes, masked stores, zero padding inside tiles
    _wino_case(1, 32, 32, 256, 256, False, True, seed=3)        # l3-like depth
    _wino_case(4, 8, 8, 128, 64, True, True, seed=6)            # 8-pixel maps: two images per tile
    _wino_case(3, 8, 8, 64, 32, False, False, seed=7)           # ... odd batch: the last tile holds one image
    _wino_case(2, 5, 7, 64, 32, True, True, seed=8)             # ... ragged narrow map
    _wino_case(2, 11, 19, 64, 128, True, False, seed=5)
    # the predict shape's deep layers (one image): split over 2 / 4 / 8 slab ranges
    assert _wino_case(1, 32, 32, 128, 128, True, True, seed=9) > 0
    assert _wino_case(1, 16, 16, 256, 256, True, True, seed=10) > 0
    assert _wino_case(1, 8, 8, 512, 512, True, True, seed=11) > 0
    assert _wino_case(2, 8, 8, 512, 512, False, False, seed=12) > 0
    assert _wino_case(1, 64, 64, 64, 64, True, True, seed=13) == 0           # one slab: nothing to split


def test_dual_1x1_matches_float64_reference():
    """conv3 + downsample branch as one K-concatenated GEMM (hmf_conv1x1_dual_nhwc), strides 1 and 2, ragged M."""
    from humaniflow_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    for B, OH, OW, cin, cin2, cout, s2 in [(2, 8, 8, 64, 128, 256, 2), (1, 7, 5, 32, 64, 64, 1), (2, 6, 6, 128, 256, 128, 2)]:
        H2, W2 = (OH - 1) * s2 + 1 + (s2 - 1), (OW - 1) * s2 + 1 + (s2 - 1)
        h, x = torch.randn(B, cin, OH, OW, generator=g), torch.randn(B, cin2, H2, W2, generator=g)
        w3, wd = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5, torch.randn(cout, cin2, 1, 1, generator=g) / cin2 ** 0.5
        bias = torch.randn(cout, generator=g)
        ref = (F.conv2d(h.double(), w3.double()) + F.conv2d(x.double(), wd.double(), None, s2) + bias.double()[None, :, None, None]).clamp_min(0)
        wcat = torch.cat([w3.reshape(cout, cin), wd.reshape(cout, cin2)], 1).contiguous().cuda()
        hd, xd = h.cuda().contiguous(memory_format=torch.channels_last), x.cuda().contiguous(memory_format=torch.channels_last)
        y = torch.full(ref.shape, float('nan'), device='cuda').contiguous(memory_format=torch.channels_last)
        _lib.check(lib.hmf_conv1x1_dual_nhwc(hd.data_ptr(), xd.data_ptr(), wcat.data_ptr(), bias.cuda().data_ptr(), y.data_ptr(), B, OH, OW, cin,
                                             H2, W2, cin2, s2, cout, 1, _lib.current_stream()), 'dual')
        torch.cuda.synchronize()
        assert (y.cpu().double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


def test_bottleneck_seam_pair_matches_float64_reference():
    """hmf_conv_pair: relu([h | x2] W3 + b3 (+ R)) and relu(y W1 + b1) in one launch vs float64 -- plain, with the downsample branch as a
    second K range (stride 1 and 2), both output widths, odd image count."""
    from humaniflow_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(11)
    for B, OH, OW, k1, k2, s2, C, n2, res in [(2, 8, 8, 64, 0, 1, 256, 64, True), (1, 16, 8, 64, 64, 1, 256, 64, False), (3, 8, 8, 128, 0, 1, 512, 128, True),
                                              (2, 8, 16, 64, 0, 1, 256, 128, True), (2, 8, 8, 64, 64, 2, 128, 128, False), (1, 8, 8, 128, 0, 1, 64, 64, False),
                                              (2, 8, 8, 128, 0, 1, 512, 256, True), (1, 8, 8, 64, 0, 1, 128, 256, False)]:
        H2, W2 = (OH - 1) * s2 + 1 + (s2 - 1), (OW - 1) * s2 + 1 + (s2 - 1)
        h = torch.randn(B, k1, OH, OW, generator=g)
        x2 = torch.randn(B, k2, H2, W2, generator=g) if k2 else None
        w3 = torch.randn(C, k1 + k2, generator=g) / (k1 + k2) ** 0.5
        w1 = torch.randn(n2, C, generator=g) / C ** 0.5
        b3, b1 = torch.randn(C, generator=g), torch.randn(n2, generator=g)
        r = torch.randn(B, C, OH, OW, generator=g) if res else None
        yr = F.conv2d(h.double(), w3[:, :k1].double()[:, :, None, None]) + b3.double()[None, :, None, None]
        if k2:
            yr = yr + F.conv2d(x2.double(), w3[:, k1:].double()[:, :, None, None], None, s2)
        if res:
            yr = yr + r.double()
        yr = yr.clamp_min(0)
        zr = (F.conv2d(yr, w1.double()[:, :, None, None]) + b1.double()[None, :, None, None]).clamp_min(0)
        p3, p1 = torch.empty_like(w3), torch.empty_like(w1)
        _lib.check(lib.hmf_conv_pair_pack_host(w3.data_ptr(), w1.data_ptr(), p3.data_ptr(), p1.data_ptr(), k1 + k2, C, n2), 'pack')
        assert sorted(p3.flatten().tolist()) == sorted(w3.flatten().tolist())          # the packing is a permutation
        cl = torch.channels_last
        hd = h.cuda().contiguous(memory_format=cl)
        xd = x2.cuda().contiguous(memory_format=cl) if k2 else None
        rd = r.cuda().contiguous(memory_format=cl) if res else None
        y = torch.full(yr.shape, float('nan'), device='cuda').contiguous(memory_format=cl)
        z = torch.full(zr.shape, float('nan'), device='cuda').contiguous(memory_format=cl)
        p3d, p1d, b3d, b1d = p3.cuda(), p1.cuda(), b3.cuda(), b1.cuda()
        _lib.check(lib.hmf_conv_pair(hd.data_ptr(), None if xd is None else xd.data_ptr(), p3d.data_ptr(), b3d.data_ptr(), None if rd is None else rd.data_ptr(),
                                     y.data_ptr(), p1d.data_ptr(), b1d.data_ptr(), z.data_ptr(), B, OH, OW, k1, H2, W2, k2, s2, C, n2, 0, _lib.current_stream()), 'pair')
        torch.cuda.synchronize()
        ey, ez = (y.cpu().double() - yr).abs().max().item(), (z.cpu().double() - zr).abs().max().item()
        assert ey < 2e-5 * max(1.0, yr.abs().max().item()) and ez < 2e-5 * max(1.0, zr.abs().max().item()), (B, OH, OW, k1, k2, s2, C, n2, res, ey, ez)
    x = torch.zeros(1, 5, 5, 64, device='cuda')                                          # 25 pixels: not a multiple of 64
    assert lib.hmf_conv_pair(x.data_ptr(), None, p3d.data_ptr(), b3d.data_ptr(), None, x.data_ptr(), p1d.data_ptr(), b1d.data_ptr(), x.data_ptr(),
                             1, 5, 5, 64, 0, 0, 0, 1, 64, 64, 0, None) < 0


def test_encoder_seam_fusion_equals_separate_launches():
    """ResNet-50 inference plan with and without the fused seams (layer1 / layer2): same features to fp32 summation order."""
    from humaniflow_amd.encoder import ResNetEncoder
    from humaniflow_amd.synthetic import synthetic_proxy_input
    torch.manual_seed(3)
    enc = ResNetEncoder(50, 18).eval().cuda()
    x = synthetic_proxy_input(4, 128, seed=2).cuda()
    calls = []
    inner = enc._conv_pair
    enc._conv_pair = lambda *a, **k: (calls.append(1), inner(*a, **k))[1]
    with torch.no_grad():
        a = enc(x)
        # with the 1x1 layers on hmf_conv1x1_gemm only the two 64-channel-input seams of layer1 stay fused (measured: two GEMM launches beat the
        # seam kernel elsewhere)
        assert sorted(enc._plan['pair']) == [1, 2] and len(calls) == 2
        saved, enc._plan['pair'] = enc._plan['pair'], {}
        b = enc(x)
        enc._plan['pair'] = saved
        assert len(calls) == 2
        enc.gemm_1x1 = False                                            # the round-2 plan: k_conv_igemm + all six seams the pair kernel is built for
        enc.invalidate_plan()
        c = enc(x)
        assert len(enc._plan['pair']) == 6 and len(calls) == 8          # layer1: 3 seams (the first with the downsample branch), layer2: 3
    assert (a - b).abs().max().item() <= 2e-5 * a.abs().max().item()
    assert (c - b).abs().max().item() <= 2e-5 * a.abs().max().item()


def test_conv_rejects_unsupported_shapes():
    from humaniflow_amd import _lib
    lib = _lib.load()
    x = torch.zeros(1, 8, 8, 48, device='cuda')
    for args in [(48, 64, 3, 1), (64, 96, 3, 1), (64, 64, 5, 1), (64, 64, 3, 3)]:
        cin, cout, ks, s = args
        rc = lib.hmf_conv2d_nhwc(x.data_ptr(), x.data_ptr(), None, None, x.data_ptr(), 1, 8, 8, cin, cout, ks, s, 1, 0, None)
        assert rc < 0 and lib.hmf_last_error()


STEM_FORMS = {'direct': ('hmf_stem_packed_floats', 'hmf_stem_pack_host', 'hmf_stem_conv', 'hmf_stem_conv_pool'),
              'polyphase': ('hmf_stem_wino_packed_floats', 'hmf_stem_wino_pack_host', 'hmf_stem_wino', 'hmf_stem_wino_pool')}


@pytest.mark.parametrize('form', ['direct', 'polyphase'])
@pytest.mark.parametrize('B,H,W', [(2, 64, 64), (1, 48, 80), (3, 30, 22)])
def test_stem_conv_and_maxpool_match_float64_reference(B, H, W, form):
    """hmf_stem_conv / hmf_stem_wino (NCHW in, NHWC out, bias + ReLU fused) and hmf_maxpool3x3s2_nhwc vs float64 torch; odd sizes exercise partial
    tiles and the zero-padded patch borders."""
    from humaniflow_amd import _lib
    lib = _lib.load()
    f_floats, f_pack, f_conv, _ = (getattr(lib, nm) for nm in STEM_FORMS[form])
    g = torch.Generator().manual_seed(B * 1000 + H)
    x = torch.randn(B, 18, H, W, generator=g)
    w = torch.randn(64, 18, 7, 7, generator=g) / 30.0
    bias = torch.randn(64, generator=g) * 0.1
    ref = F.relu(F.conv2d(x.double(), w.double(), bias.double(), 2, 3))
    refp = F.max_pool2d(ref, 3, 2, 1)
    packed = torch.empty(int(f_floats()))
    _lib.check(f_pack(w.contiguous().data_ptr(), packed.data_ptr()), 'pack')
    xd, pd, bd = x.cuda(), packed.cuda(), bias.cuda()
    OH, OW = ref.shape[2:]
    y = torch.full((B, 64, OH, OW), float('nan'), device='cuda').contiguous(memory_format=torch.channels_last)
    st = _lib.current_stream()
    _lib.check(f_conv(xd.data_ptr(), pd.data_ptr(), bd.data_ptr(), y.data_ptr(), B, 18, H, W, 1, st), 'stem')
    z = torch.full(tuple(refp.shape), float('nan'), device='cuda').contiguous(memory_format=torch.channels_last)
    _lib.check(lib.hmf_maxpool3x3s2_nhwc(y.data_ptr(), z.data_ptr(), B, OH, OW, 64, st), 'hmf_maxpool3x3s2_nhwc')
    torch.cuda.synchronize()
    assert (y.cpu().double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    assert (z.cpu().double() - refp).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    assert torch.equal(z, F.max_pool2d(y, 3, 2, 1))              # the pool itself is exact


@pytest.mark.parametrize('form', ['direct', 'polyphase'])
@pytest.mark.parametrize('B,H,W', [(2, 64, 64), (1, 100, 76), (3, 34, 130), (1, 256, 256), (2, 18, 22), (1, 130, 258)])
def test_stem_with_fused_maxpool_equals_the_two_launches(B, H, W, form):
    """hmf_stem_conv_pool / hmf_stem_wino_pool (pool applied per tile, shared windows by atomic max) == the stem + hmf_maxpool3x3s2_nhwc, bit for bit
    (ragged tiles, odd pooled sizes, several images)."""
    from humaniflow_amd import _lib
    lib = _lib.load()
    f_floats, f_pack, f_conv, f_pool = (getattr(lib, nm) for nm in STEM_FORMS[form])
    g = torch.Generator().manual_seed(H + W)
    x = torch.randn(B, 18, H, W, generator=g).cuda()
    w = (torch.randn(64, 18, 7, 7, generator=g) / 30).contiguous()
    bias = torch.randn(64, generator=g).cuda()
    packed = torch.empty(int(f_floats()))
    _lib.check(f_pack(w.data_ptr(), packed.data_ptr()), 'pack')
    pk = packed.cuda()
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    PH, PW = (OH - 1) // 2 + 1, (OW - 1) // 2 + 1
    st = _lib.current_stream()
    y = torch.empty(B, OH, OW, 64, device='cuda')
    z = torch.empty(B, PH, PW, 64, device='cuda')
    zf = torch.full((B, PH, PW, 64), float('nan'), device='cuda')
    _lib.check(f_conv(x.data_ptr(), pk.data_ptr(), bias.data_ptr(), y.data_ptr(), B, 18, H, W, 1, st), 'stem')
    _lib.check(lib.hmf_maxpool3x3s2_nhwc(y.data_ptr(), z.data_ptr(), B, OH, OW, 64, st), 'pool')
    if form == 'polyphase':
        ws = torch.empty(int(lib.hmf_stem_wino_pool_workspace_bytes(B, H, W)), dtype=torch.uint8, device='cuda')
        _lib.check(f_pool(x.data_ptr(), pk.data_ptr(), bias.data_ptr(), zf.data_ptr(), B, 18, H, W, ws.data_ptr(), ws.numel(), st), 'stem+pool')
        assert lib.hmf_stem_wino_pool(x.data_ptr(), pk.data_ptr(), bias.data_ptr(), zf.data_ptr(), B, 18, H, W, ws.data_ptr(), ws.numel() - 1, st) < 0      # too small
    else:
        _lib.check(f_pool(x.data_ptr(), pk.data_ptr(), bias.data_ptr(), zf.data_ptr(), B, 18, H, W, st), 'stem+pool')
    torch.cuda.synchronize()
    assert torch.equal(z, zf)


def test_polyphase_stem_filters_packed_on_the_device_equal_the_host_packer_and_no_relu_no_bias_form():
    """hmf_stem_wino_pack_device (training re-packs the live filter every step) == hmf_stem_wino_pack_host bit for bit; the relu = 0 / bias = NULL
    form (the training graph's stem) against float64."""
    from humaniflow_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(77)
    w = (torch.randn(64, 18, 7, 7, generator=g) / 30).contiguous()
    n = int(lib.hmf_stem_wino_packed_floats())
    host = torch.empty(n)
    _lib.check(lib.hmf_stem_wino_pack_host(w.data_ptr(), host.data_ptr()), 'pack')
    dev = torch.full((n,), float('nan'), device='cuda')
    wd = w.cuda()
    st = _lib.current_stream()
    _lib.check(lib.hmf_stem_wino_pack_device(wd.data_ptr(), dev.data_ptr(), st), 'pack device')
    torch.cuda.synchronize()
    assert torch.equal(dev.cpu(), host)
    B, H, W = 2, 96, 64
    x = torch.randn(B, 18, H, W, generator=g)
    ref = F.conv2d(x.double(), w.double(), None, 2, 3)
    y = torch.full((B, 64, H // 2, W // 2), float('nan'), device='cuda').contiguous(memory_format=torch.channels_last)
    xd = x.cuda()
    _lib.check(lib.hmf_stem_wino(xd.data_ptr(), dev.data_ptr(), None, y.data_ptr(), B, 18, H, W, 0, st), 'stem')
    torch.cuda.synchronize()
    assert (y.cpu().double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    assert (y < 0).any()                                          # no ReLU was applied


def test_folded_encoder_matches_module_path():
    """The whole inference plan (stem + fused residual stages, ResNet-50 and ResNet-18) vs the plain nn.Module path on the same weights."""
    import humaniflow_amd as H
    from humaniflow_amd.encoder import ResNetEncoder
    from humaniflow_amd.synthetic import synthetic_proxy_input
    for layers in (18, 50):
        torch.manual_seed(layers)
        enc = ResNetEncoder(layers, 18)
        for m in enc.modules():                       # non-trivial running statistics so the BN fold is exercised
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.data.uniform_(0.5, 1.5)
                m.bias.data.normal_(0, 0.1)
        enc = enc.eval().cuda()
        x = synthetic_proxy_input(2, 128, seed=layers).cuda()
        with torch.no_grad():
            got = enc(x)
            ref = enc.forward_module(x.double()) if False else enc.double().forward_module(x.double())
        assert got.shape == (2, enc.out_dim)
        assert (got.double() - ref).abs().max().item() < 5e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize('terms,tol', [(6, 5e-5), (3, 1e-3)])
def test_encoder_with_split_operand_gemms(terms, tol):
    """gemm_terms (opt-in): ResNet-50 with every 1x1 GEMM on the split-operand kernel against the float64 module path; the plan follows the flag."""
    from humaniflow_amd.encoder import ResNetEncoder
    from humaniflow_amd.synthetic import synthetic_proxy_input
    torch.manual_seed(50)
    enc = ResNetEncoder(50, 18)
    for m in enc.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.1)
    enc = enc.eval().cuda()
    x = synthetic_proxy_input(2, 128, seed=50).cuda()
    with torch.no_grad():
        base = enc(x)
        assert not enc._plan['split'] and enc._plan['pair']
        enc.gemm_terms = terms
        got = enc(x)
        assert len(enc._plan['split']) == 16 + 16 + 4 - 4 and not enc._plan['pair']      # conv1 + conv3 of 16 blocks; 4 of the conv3 carry the downsample branch
        enc.gemm_terms = 0
        again = enc(x)
        assert torch.equal(again, base)
        ref = enc.double().forward_module(x.double())
    scale = max(1.0, ref.abs().max().item())
    assert (got.double() - ref).abs().max().item() < tol * scale
    assert not torch.equal(got, base)
    enc.float()
    enc.gemm_terms = 5
    with pytest.raises(ValueError), torch.no_grad():
        enc(x)


def test_encoder_full_size_properties():
    """BASELINE size (B=32, 256x256, ResNet-50): size-independent properties of the hand-written encoder -- run-to-run determinism (bitwise),
    independence of the images of a batch (an image's features do not depend on its neighbours or on the batch size), agreement of the
    Winograd / direct 3x3 kernels through the whole network, and finite outputs."""
    import humaniflow_amd as H
    from humaniflow_amd.encoder import ResNetEncoder
    from humaniflow_amd.synthetic import synthetic_proxy_input
    torch.manual_seed(1)
    enc = ResNetEncoder(50, 18).eval().cuda()
    x = synthetic_proxy_input(32, 256, seed=5).cuda()
    with torch.no_grad():
        a, b = enc(x), enc(x)
        assert torch.equal(a, b) and torch.isfinite(a).all() and a.shape == (32, 2048)
        sub = enc(x[5:8])
        scale = a.abs().max().item()
        assert (sub - a[5:8]).abs().max().item() <= 1e-6 * scale
        one = enc(x[31:32])
        assert (one - a[31:32]).abs().max().item() <= 1e-6 * scale
        # same network with the Winograd layers forced onto the direct patch kernel
        plan = enc._plan
        saved, plan['wino'] = plan['wino'], {}
        c = enc(x)
        plan['wino'] = saved
        assert (c - a).abs().max().item() <= 2e-5 * scale and not torch.equal(c, a)
