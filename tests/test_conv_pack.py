"""Host-side weight packers of the encoder kernels (no GPU): the packed streams must be exactly the permutation the kernels assume --
lane l = 32*q + j of k-group g supplies W[n = 32*t + j][k = 8*g + 4*q + u] as its u-th float (v_mfma_f32_32x32x2_f32 B operand)."""
import numpy as np
import torch

from humaniflow_amd import _lib


def test_patch_pack_is_the_documented_permutation():
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    for cin, cout, ks, cfg, sc in [(128, 64, 3, 8210, 64), (64, 96, 3, 16410, 64), (256, 64, 3, 16420, 64)]:
        w = torch.randn(cout, ks, ks, cin, generator=g)                  # channels_last weight: [Cout][kh][kw][Cin]
        packed = torch.empty(int(lib.hmf_conv_patch_packed_floats(cin, cout, ks)))
        assert packed.numel() == w.numel()
        _lib.check(lib.hmf_conv_patch_pack_host(w.data_ptr(), packed.data_ptr(), cin, cout, ks, cfg), 'pack')
        p = packed.reshape(cout // 32, cin // sc, ks * ks, sc // 8, 2, 32, 4)          # [t][slab][tap][g][q][j][u]
        back = p.permute(0, 5, 2, 1, 3, 4, 6).reshape(cout, ks * ks, cin)              # [t, j][tap][slab, g, q, u]
        assert torch.equal(back, w.reshape(cout, ks * ks, cin))
    assert lib.hmf_conv_patch_pack_host(w.data_ptr(), packed.data_ptr(), 96, 64, 3, 8210) < 0      # Cin not a multiple of the slab
    assert lib.hmf_conv_patch_pack_host(w.data_ptr(), packed.data_ptr(), 64, 64, 3, 1234) < 0       # unknown cfg


def test_stem_pack_is_the_documented_permutation():
    lib = _lib.load()
    w = torch.randn(64, 18, 7, 7, generator=torch.Generator().manual_seed(1))
    n = int(lib.hmf_stem_packed_floats())
    assert n == 111 * 2 * 64 * 4
    packed = torch.empty(n)
    _lib.check(lib.hmf_stem_pack_host(w.data_ptr(), packed.data_ptr()), 'pack')
    p = packed.reshape(111, 2, 2, 32, 4)                                              # [g][t][q][j][u]
    steps = p.permute(0, 4, 2, 1, 3).reshape(444, 2, 64)                              # [s = 4g+u][q][n = 32t+j]
    assert torch.all(steps[441:] == 0)                                                # padding k-steps carry zero weights
    ref = w.reshape(64, 9, 2, 49).permute(1, 3, 2, 0).reshape(441, 2, 64)             # s = (cp, kh, kw); channel = 2cp + q
    assert torch.equal(steps[:441], ref)
