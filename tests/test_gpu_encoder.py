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


@pytest.mark.parametrize('tile', [0, 128128, 128064, 64128, 64064])
@pytest.mark.parametrize('ks,stride', [(1, 1), (1, 2), (3, 1), (3, 2)])
def test_conv_matches_float64_reference(tile, ks, stride):
    _case(2, 12, 12, 64, 128, ks, stride, res=True, relu=True, tile=tile)


def test_conv_ragged_rows_and_flags():
    # M = 49 / 147 rows: partial tiles, rows beyond M neither read nor written; bias-only / residual-only / no ReLU paths
    _case(1, 7, 7, 32, 64, 3, 1, res=False, relu=False, tile=64064, seed=1)
    _case(3, 7, 7, 96, 192, 3, 1, res=True, relu=False, tile=128064, seed=2)
    _case(3, 7, 7, 96, 192, 1, 1, res=False, relu=True, tile=64064, seed=3)
    _case(1, 9, 5, 64, 128, 3, 2, res=True, relu=True, tile=128128, seed=4)          # non-square, odd sizes, stride 2
    _case(2, 8, 8, 2048, 512, 1, 1, res=False, relu=True, tile=0, seed=5)           # deepest K of the 1x1 layers


def test_conv_rejects_unsupported_shapes():
    from humaniflow_amd import _lib
    lib = _lib.load()
    x = torch.zeros(1, 8, 8, 48, device='cuda')
    for args in [(48, 64, 3, 1), (64, 96, 3, 1), (64, 64, 5, 1), (64, 64, 3, 3)]:
        cin, cout, ks, s = args
        rc = lib.hmf_conv2d_nhwc(x.data_ptr(), x.data_ptr(), None, None, x.data_ptr(), 1, 8, 8, cin, cout, ks, s, 1, 0, None)
        assert rc < 0 and lib.hmf_last_error()
