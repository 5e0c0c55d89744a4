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


def test_seam_pair_pack_is_the_documented_permutation():
    """hmf_conv_pair_pack_host: both filters of a bottleneck seam in B-operand lane order per (64-channel chunk, n-tile, k-group)."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(2)
    for kt, C, n2 in [(64, 256, 64), (128, 512, 128), (128, 64, 64), (64, 128, 128), (128, 512, 256)]:
        w3, w1 = torch.randn(C, kt, generator=g), torch.randn(n2, C, generator=g)
        p3, p1 = torch.empty_like(w3), torch.empty_like(w1)
        _lib.check(lib.hmf_conv_pair_pack_host(w3.data_ptr(), w1.data_ptr(), p3.data_ptr(), p1.data_ptr(), kt, C, n2), 'pack')
        a = p3.reshape(C // 64, 2, kt // 8, 2, 32, 4)                     # [c][wn][g][q][j][u] = w3[64c + 32wn + j][8g + 4q + u]
        assert torch.equal(a.permute(0, 1, 4, 2, 3, 5).reshape(C, kt), w3)
        b = p1.reshape(C // 64, n2 // 32, 8, 2, 32, 4)                    # [c][t][g][q][j][u] = w1[32t + j][64c + 8g + 4q + u]
        assert torch.equal(b.permute(1, 4, 0, 2, 3, 5).reshape(n2, C), w1)
    assert lib.hmf_conv_pair_pack_host(w3.data_ptr(), w1.data_ptr(), p3.data_ptr(), p1.data_ptr(), 96, 64, 64) < 0       # K must be 64 or 128
    assert lib.hmf_conv_pair_pack_host(w3.data_ptr(), w1.data_ptr(), p3.data_ptr(), p1.data_ptr(), 64, 64, 96) < 0       # N2 must be 64, 128 or 256
