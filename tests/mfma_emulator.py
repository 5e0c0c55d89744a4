"""NumPy emulation of the wave-level data flow of the MFMA kernels (test infrastructure).

It replays, lane by lane, what k_lbs_mfma / k_flow_ctx0 / k_flow_chain do with the PACKED operand arenas
produced by the C++ packers (hmf_smpl_pack_blend_host / hmf_flow_pack_host), using the documented
v_mfma_f32_32x32x2_f32 operand/result lane mapping (guides/cdna_hip_programming.md section 3):
    A operand: lane l supplies A[i = l & 31][k = l >> 5]
    B operand: lane l supplies B[k = l >> 5][j = l & 31]
    C/D      : lane l, register r holds D[i = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][j = l & 31]
Runs on CPU, so operand-layout mistakes are caught before any GPU time is spent.
"""
import numpy as np
import torch

LANES = np.arange(64)
L31 = LANES & 31
Q = LANES >> 5
# D row index held by (reg, lane)
D_ROW = (np.arange(16)[:, None] & 3) + 8 * (np.arange(16)[:, None] >> 2) + 4 * Q[None, :]     # (16, 64)

PART_FIELDS = [('n_anc', 1), ('anc', 7), ('kas', 1), ('kag', 1), ('o_wc0', 1), ('o_bctx', 1), ('o_wa', 1), ('o_w1', 1),
               ('o_w1e', 1), ('o_w2', 2), ('o_w3', 2), ('o_w4', 2), ('o_we', 2), ('o_wctxT', 1), ('o_nnT', 8), ('o_nnb', 8),
               ('kas16', 1), ('kag16', 1), ('o16_wa', 1), ('o16_w1', 1), ('o16_w1e', 1), ('o16_w2', 2), ('o16_w3', 2), ('o16_w4', 2), ('o16_we', 2)]
LEVEL_INTS = 1 + 24 + 23
PART_INTS = sum(n for _, n in PART_FIELDS)


def mfma(a, b, acc):
    """acc (16, 64) += A(32x2) . B(2x32) with a, b the per-lane operand values (64,)."""
    A = a.reshape(2, 32).T.astype(np.float32)          # [i][k]
    Bm = b.reshape(2, 32).astype(np.float32)           # [k][j]
    D = (A[:, 0:1] * Bm[0:1, :]).astype(np.float32)
    D = (D + A[:, 1:2] * Bm[1:2, :]).astype(np.float32)
    acc += D[D_ROW, L31[None, :]]
    return acc


def parse_flow_pack(buf):
    """Split hmf_flow_pack_host output -> (arena float32, list of part dicts, anc_off (23,64), globals dict)."""
    ints = buf.view(np.int32)
    n_arena = int(ints[-1])
    arena = buf[:n_arena]
    tab_raw = ints[n_arena:n_arena + 23 * PART_INTS].reshape(23, PART_INTS)
    parts = []
    for j in range(23):
        d, o = {}, 0
        for name, n in PART_FIELDS:
            v = tab_raw[j, o:o + n]
            d[name] = int(v[0]) if n == 1 else [int(x) for x in v]
            o += n
        parts.append(d)
    anc_off = ints[n_arena + 23 * PART_INTS:n_arena + 23 * PART_INTS + 23 * 64].reshape(23, 64)
    lev = ints[n_arena + 23 * PART_INTS + 23 * 64:n_arena + 23 * PART_INTS + 23 * 64 + LEVEL_INTS]
    nlev, start, order = int(lev[0]), lev[1:25], lev[25:48]
    rounds = [[int(order[i]) for i in range(start[r], start[r + 1])] for r in range(nlev)]
    glob = {'o_wisgc': int(ints[-4]), 'o_bisgc': int(ints[-3]), 'o_wbeta': int(ints[-2]), 'rounds': rounds}
    return arena, parts, anc_off, glob


def a_block(arena, off, ntiles, g4):
    """View of a packed A-operand block as [tile][g4][lane][4]."""
    return arena[off:off + ntiles * g4 * 64 * 4].reshape(ntiles, g4, 64, 4)


def bias_init(arena, off, ntiles):
    """acc[t][4g + r][lane] = bias[32 t + 8 g + 4 q + r]"""
    acc = np.zeros((ntiles, 16, 64), np.float32)
    for t in range(ntiles):
        for g in range(4):
            for r in range(4):
                acc[t, 4 * g + r] = arena[off + 32 * t + 8 * g + 4 * Q + r]
    return acc


def dense_layer(wblk, x_in, acc):
    """wblk [NTO][NKT*4][64][4]; x_in (NKT, 16, 64) in ARL; acc (NTO, 16, 64)."""
    nto, ns4 = wblk.shape[0], wblk.shape[1]
    for s4 in range(ns4):
        for u in range(4):
            s = s4 * 4 + u
            for to in range(nto):
                mfma(wblk[to, s4, :, u], x_in[s >> 4, s & 15], acc[to])
    return acc


def elu(x):
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0))).astype(np.float32)


def spline_lane(a4, x, bound, inverse=False):
    """Per-lane spline from the 32 packed parameter registers a4 (2, 16, 64) using the ORACLE's spline code:
    slots [0,8) widths, [8,16) heights, [16,23) derivatives, [24,32) lambdas of dimension q = lane >> 5."""
    from oracle.spline import rational_linear_spline
    import torch.nn.functional as F
    rw = torch.from_numpy(a4[0, 0:8].T.copy())      # (64, 8)
    rh = torch.from_numpy(a4[0, 8:16].T.copy())
    rd = torch.from_numpy(a4[1, 0:7].T.copy())
    rl = torch.from_numpy(a4[1, 8:16].T.copy())
    w, h, d, l = F.softmax(rw, -1), F.softmax(rh, -1), F.softplus(rd), torch.sigmoid(rl)
    y, ld = rational_linear_spline(torch.from_numpy(x.astype(np.float32)), w, h, d, l, bound, inverse=inverse)
    return y.numpy(), ld.numpy()


def emulate_flow_tile(arena, parts, anc_off, glob, levels, nb, G, shape_mode, shape_log_std, eps_shape, eps_pose, B, N, pe,
                      tile, radius, base_std):
    """One 32-row tile through k_flow_ctx0 + k_flow_chain.  Returns dict rowid -> (R (23,9), v (23,3), beta (nb,))."""
    from oracle import so3
    Mtot = B * N + (B if pe else 0)
    rowid = tile * 32 + L31
    live = rowid < Mtot
    rs = np.where(live, rowid, 0)
    is_point = rs >= B * N
    b = np.where(is_point, rs - B * N, rs // max(N, 1))
    n = np.where(is_point, 0, rs - b * N)
    # ---- k_flow_ctx0 prologue ----
    beta = np.zeros((16, 64), np.float32)
    for i in range(nb):
        v = shape_mode[b, i].astype(np.float32)
        if eps_shape is not None:
            v = np.where(is_point, v, (v + eps_shape[n, b, i] * np.exp(shape_log_std[b, i])).astype(np.float32))
        beta[i] = np.where(live, v, 0)
    h = np.zeros((8, 16, 64), np.float32)
    for t in range(8):
        for g in range(4):
            for r in range(4):
                h[t, 4 * g + r] = G[b, 32 * t + 8 * g + 4 * Q + r]
    g4 = (((nb + 1) >> 1) + 3) >> 2
    wb = a_block(arena, glob['o_wbeta'], 8, g4)
    for s4 in range(g4):
        for t in range(8):
            for u in range(4):
                s = s4 * 4 + u
                bop = np.where(Q == 1, beta[2 * s + 1], beta[2 * s])
                mfma(wb[t, s4, :, u], bop, h[t])
    h = np.where(live[None, None, :], elu(h), 0).astype(np.float32)
    C0 = {}
    for j in range(23):
        pt = parts[j]
        acc = bias_init(arena, pt['o_bctx'], 2)
        C0[j] = dense_layer(a_block(arena, pt['o_wc0'], 2, 32), h, acc)
    # ---- k_flow_chain ----
    sAnc = np.zeros((32, 209), np.float32)
    out_R = np.zeros((64, 23, 9), np.float32)
    out_v = np.zeros((64, 23, 3), np.float32)
    for lev in levels:
        for j in lev:
            pt = parts[j]
            cx = C0[j].copy()
            if pt['kag'] > 0:
                wa = a_block(arena, pt['o_wa'], 2, pt['kag'])
                for s4 in range(pt['kag']):
                    for u in range(4):
                        s = s4 * 4 + u
                        off = anc_off[j, 2 * s + Q]
                        bval = sAnc[L31, off]
                        mfma(wa[0, s4, :, u], bval, cx[0])
                        mfma(wa[1, s4, :, u], bval, cx[1])
            cx = elu(cx)
            x = np.zeros((3, 64), np.float32)
            if N > 0:
                e = eps_pose[j, b, n]                                   # (64, 3)
                x = np.where((live & ~is_point)[None, :], (e * np.float32(base_std)).T, 0).astype(np.float32)
            x0, x1, x2 = x
            one_q0 = np.where(Q == 0, 1.0, 0.0).astype(np.float32)
            w1e = arena[pt['o_w1e']:pt['o_w1e'] + 256].reshape(4, 64)
            a1 = np.zeros((4, 16, 64), np.float32)
            bx0 = np.where(Q == 0, x0, 1.0).astype(np.float32)
            mfma(w1e[0], bx0, a1[0])
            mfma(w1e[1], bx0, a1[1])
            dense_layer(a_block(arena, pt['o_w1'], 4, 8), cx, a1)

            def tail(c, h1, xin):
                we = arena[pt['o_we'][c]:pt['o_we'][c] + 256].reshape(4, 64)
                a2 = np.zeros((1, 16, 64), np.float32)
                mfma(we[0], one_q0, a2[0])
                dense_layer(a_block(arena, pt['o_w2'][c], 1, 8), h1, a2)
                a2 = np.maximum(a2, 0)
                a3 = np.zeros((1, 16, 64), np.float32)
                mfma(we[1], one_q0, a3[0])
                dense_layer(a_block(arena, pt['o_w3'][c], 1, 4), a2, a3)
                a3 = np.maximum(a3, 0)
                a4 = np.zeros((2, 16, 64), np.float32)
                mfma(we[2], one_q0, a4[0])
                mfma(we[3], one_q0, a4[1])
                dense_layer(a_block(arena, pt['o_w4'][c], 2, 4), a3, a4)
                y, _ = spline_lane(a4, xin, radius)
                return y
            h1 = np.maximum(a1[0:2], 0).astype(np.float32)
            xn = tail(0, h1, np.where(Q == 1, x2, x1))
            x1n, x2n = xn[L31], xn[32 + L31]
            y0 = x1n
            by0 = np.where(Q == 0, y0, 1.0).astype(np.float32)
            mfma(w1e[2], by0, a1[2])
            mfma(w1e[3], by0, a1[3])
            h1 = np.maximum(a1[2:4], 0).astype(np.float32)
            yn = tail(1, h1, np.where(Q == 1, x0, x2n))
            y1, y2 = yn[L31], yn[32 + L31]
            yv = torch.from_numpy(np.stack([y0, y1, y2], -1).astype(np.float32))
            from oracle.flow import radial_tanh_forward
            v = radial_tanh_forward(yv, radius)
            R_s = so3.so3_exp(v.double()).float().reshape(64, 9).numpy()
            R_p = so3.batch_rodrigues(v).reshape(64, 9).numpy()
            R = np.where(is_point[:, None], R_p, R_s)
            sAnc[L31[:32], j * 9:(j + 1) * 9] = R[:32]
            out_R[:, j] = R
            out_v[:, j] = v.numpy()
    res = {}
    for lane in range(32):
        if live[lane]:
            res[int(rowid[lane])] = (out_R[lane], out_v[lane], beta[:nb, lane].copy())
    return res


# ---- round 4: k_flow_chain16 -- 16-row tiles on v_mfma_f32_16x16x4_f32 (flow_common.h "ARL16") ------------------------------------------------------
J16 = LANES & 15
G16 = LANES >> 4


def mfma16(a, b, acc):
    """acc (4, 64) += A(16x4) . B(4x16): lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15]; D register r of lane (j, g) is row 4 g + r."""
    A = a.reshape(4, 16).T.astype(np.float32)          # [i][k]
    Bm = b.reshape(4, 16).astype(np.float32)           # [k][j]
    D = np.zeros((16, 16), np.float32)
    for k in range(4):
        D = (D + A[:, k:k + 1] * Bm[k:k + 1, :]).astype(np.float32)
    for r in range(4):
        acc[r] += D[4 * G16 + r, J16]
    return acc


def a_block16(arena, off, ntiles, g4):
    return arena[off:off + ntiles * g4 * 64 * 4].reshape(ntiles, g4, 64, 4)


def emulate_flow_tile16(arena, parts, anc_off, levels, C0_rows, z_rows, is_point_rows, radius):
    """One 16-row tile through k_flow_chain16, lane by lane on the packed `o16_*` blocks.  C0_rows (16, 23, 64): the image/shape half of every
    context layer (k_flow_ctx0's row-major output); z_rows (23, 16, 3): base samples; is_point_rows (16,) bool.  Returns (R (16,23,9), v (16,23,3))."""
    from oracle import so3
    from oracle.flow import radial_tanh_forward
    from oracle.spline import rational_linear_spline
    import torch.nn.functional as F
    sAnc = np.zeros((16, 209), np.float32)
    out_R = np.zeros((16, 23, 9), np.float32)
    out_v = np.zeros((16, 23, 3), np.float32)
    one_g0 = np.where(G16 == 0, 1.0, 0.0).astype(np.float32)
    for lev in levels:
        for j in lev:
            pt = parts[j]
            cx = np.zeros((4, 4, 64), np.float32)                      # [tile][reg][lane]: channel 16 t + 4 g + r
            for t in range(4):
                for r in range(4):
                    cx[t, r] = C0_rows[J16, j, 16 * t + 4 * G16 + r]
            kag = pt['kag16']
            if kag > 0:
                wa = a_block16(arena, pt['o16_wa'], 4, kag)
                for s4 in range(kag):
                    for u in range(4):
                        s = s4 * 4 + u
                        bval = sAnc[J16, anc_off[j, 4 * s + G16]]
                        for t in range(4):
                            mfma16(wa[t, s4, :, u], bval, cx[t])
            cx = elu(cx)
            x0, x1, x2 = z_rows[j, J16, 0], z_rows[j, J16, 1], z_rows[j, J16, 2]
            w1 = a_block16(arena, pt['o16_w1'], 8, 4)
            w1e = arena[pt['o16_w1e']:pt['o16_w1e'] + 512].reshape(8, 64)
            a1 = np.zeros((8, 4, 64), np.float32)
            bx0 = np.where(G16 == 0, x0, np.where(G16 == 1, 1.0, 0.0)).astype(np.float32)
            for to in range(4):
                mfma16(w1e[to], bx0, a1[to])
            for s in range(16):
                for to in range(8):
                    mfma16(w1[to, s >> 2, :, s & 3], cx[s >> 2, s & 3], a1[to])

            def tail(c, h1, xin):
                we = arena[pt['o16_we'][c]:pt['o16_we'][c] + 512].reshape(8, 64)
                w2, w3, w4 = a_block16(arena, pt['o16_w2'][c], 2, 4), a_block16(arena, pt['o16_w3'][c], 2, 2), a_block16(arena, pt['o16_w4'][c], 4, 2)
                a2 = np.zeros((2, 4, 64), np.float32)
                for to in range(2):
                    mfma16(we[to], one_g0, a2[to])
                for s in range(16):
                    for to in range(2):
                        mfma16(w2[to, s >> 2, :, s & 3], h1[s >> 2, s & 3], a2[to])
                a2 = np.maximum(a2, 0)
                a3 = np.zeros((2, 4, 64), np.float32)
                for to in range(2):
                    mfma16(we[2 + to], one_g0, a3[to])
                for s in range(8):
                    for to in range(2):
                        mfma16(w3[to, s >> 2, :, s & 3], a2[s >> 2, s & 3], a3[to])
                a3 = np.maximum(a3, 0)
                a4 = np.zeros((4, 4, 64), np.float32)
                for to in range(4):
                    mfma16(we[4 + to], one_g0, a4[to])
                for s in range(8):
                    for to in range(4):
                        mfma16(w4[to, s >> 2, :, s & 3], a3[s >> 2, s & 3], a4[to])
                p = a4.reshape(16, 64)                                     # slot m = 4 t + r
                # the even lane group of a coordinate holds widths | heights, its partner (lane ^ 16) derivatives | pad | lambdas
                partner = LANES ^ 16
                rw, rh = p[0:8].T, p[8:16].T                               # (64, 8) each: meaningful in the even groups
                rd, rl = p[0:7][:, partner].T, p[8:16][:, partner].T       # the partner's registers as seen from the even group
                w, h, d, l = F.softmax(torch.from_numpy(rw.copy()), -1), F.softmax(torch.from_numpy(rh.copy()), -1), F.softplus(torch.from_numpy(rd.copy())), \
                    torch.sigmoid(torch.from_numpy(rl.copy()))
                y, _ = rational_linear_spline(torch.from_numpy(xin.astype(np.float32)), w, h, d, l, radius)
                return y.numpy()
            h1 = np.maximum(a1[0:4], 0).astype(np.float32)
            xn = tail(0, h1, np.where((G16 >> 1) == 1, x2, x1))
            x1n, x2n = xn[J16], xn[32 + J16]
            y0 = x1n
            by0 = np.where(G16 == 0, y0, np.where(G16 == 1, 1.0, 0.0)).astype(np.float32)
            for to in range(4, 8):
                mfma16(w1e[to], by0, a1[to])
            h1 = np.maximum(a1[4:8], 0).astype(np.float32)
            yn = tail(1, h1, np.where((G16 >> 1) == 1, x0, x2n))
            y1, y2 = yn[J16], yn[32 + J16]
            yv = torch.from_numpy(np.stack([y0, y1, y2], -1).astype(np.float32)[:16])
            v = radial_tanh_forward(yv, radius)
            R_s = so3.so3_exp(v.double()).float().reshape(16, 9).numpy()
            R_p = so3.batch_rodrigues(v).reshape(16, 9).numpy()
            R = np.where(is_point_rows[:, None], R_p, R_s)
            sAnc[:, j * 9:(j + 1) * 9] = R
            out_R[:, j] = R
            out_v[:, j] = v.numpy()
    return out_R, out_v
