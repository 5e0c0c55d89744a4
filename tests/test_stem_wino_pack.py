"""CPU replay of k_stem_wino's arithmetic (csrc/stem_wino.hip) on the C++-packed filter operand: the polyphase F(2,3)-class products, their four
classes per dimension, the wave roles and the class combination, against a float64 convolution.  Pins the packer's layout and the algebra without
a GPU; the kernel itself is held to the same reference in tests/test_gpu_encoder.py."""
import numpy as np
import torch
import torch.nn.functional as F

# 1-D product table of the kernel: (ja, jb, mode, class); classes 0 = A0 (y0), 1 = A1 (y1), 2 = P (+ +), 3 = Q (+ -)
PROD = [(0, 4, -1, 0), (2, 4, 1, 2), (4, 2, -1, 3), (6, 2, -1, 1), (6, 6, 0, 0), (8, 8, 0, 1), (1, 5, -1, 0), (3, 5, 1, 2), (5, 3, -1, 3), (7, 3, -1, 1)]
VLIST = [[0, 4, 6, 1, 7], [3, 5, 9, 2, 8]]
PAIRS, GROUPS = 9, 13


def comb(rows, v):
    ja, jb, mode, _ = PROD[v]
    return rows[ja] if mode == 0 else (rows[ja] + rows[jb] if mode > 0 else rows[ja] - rows[jb])


def test_polyphase_products_and_packed_filters_reproduce_the_convolution():
    from humaniflow_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    H = W = 32                                                    # 16 x 16 outputs: two tiles of 8 x 16
    x = torch.randn(1, 18, H, W, generator=g)
    w = (torch.randn(64, 18, 7, 7, generator=g) / 30).contiguous()
    ref = F.conv2d(x.double(), w.double(), None, 2, 3)[0].numpy()            # (64, 16, 16)
    n = int(lib.hmf_stem_wino_packed_floats())
    assert n == 4 * PAIRS * GROUPS * 64 * 4
    packed = torch.empty(n)
    _lib.check(lib.hmf_stem_wino_pack_host(w.data_ptr(), packed.data_ptr()), 'pack')
    pk = packed.numpy().reshape(4, PAIRS * GROUPS, 64, 4)                    # [role][group][lane][slot]
    xp = np.zeros((18, H + 16, W + 16), np.float32)
    xp[:, 3:3 + H, 3:3 + W] = x[0].numpy()                                   # xp[c, r, q] = x[c, r - 3, q - 3]
    out = np.zeros((64, 16, 16), np.float64)
    for t0 in (0, 8):                                                        # tile rows
        oh0, ow0 = t0, 0
        part = {}
        for nh in range(2):
            for vh in range(2):
                acc = np.zeros((2, 4, 32, 32), np.float32)                   # [vertical class][horizontal class][tile][channel]
                for s in range(PAIRS):
                    for vi in range(5):
                        v = VLIST[vh][vi]
                        for h in range(10):
                            p = vi * 10 + h
                            gq, u = s * GROUPS + p // 4, p % 4
                            bw = pk[nh * 2 + vh, gq, :, u]                    # lane -> B[k = lane >> 5][n = lane & 31]
                            for k in range(2):
                                ch = 2 * s + k
                                a = np.zeros(32, np.float32)
                                for i in range(32):
                                    tr, tc = i >> 3, i & 7
                                    r0, c0 = 2 * oh0 + 4 * tr, 2 * ow0 + 4 * tc                       # strip origin = input coordinate 2 o - 3, stored at + 3
                                    win = xp[ch, r0:r0 + 9, c0:c0 + 9]
                                    t = comb(win, v)                                                 # (9,) vertical combination
                                    a[i] = comb(t, h)
                                acc[0 if vi < 3 else 1, PROD[h][3]] += np.outer(a, bw[32 * k:32 * k + 32]).astype(np.float32)
                cc = []
                for vc in range(2):
                    cc.append([acc[vc, 0] + acc[vc, 2] + acc[vc, 3], acc[vc, 1] + acc[vc, 2] - acc[vc, 3]])
                if vh == 0:
                    part[(nh, vh)] = [[cc[0][c] + cc[1][c] for c in range(2)], [cc[1][c] for c in range(2)]]
                else:
                    part[(nh, vh)] = [[cc[1][c] for c in range(2)], [cc[0][c] - cc[1][c] for c in range(2)]]
        for nh in range(2):
            for rr in range(2):
                for c in range(2):
                    y = part[(nh, 0)][rr][c] + part[(nh, 1)][rr][c]          # (tile, channel)
                    for i in range(32):
                        out[32 * nh:32 * nh + 32, oh0 + 2 * (i >> 3) + rr, ow0 + 2 * (i & 7) + c] = y[i]
    assert np.abs(out - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())
