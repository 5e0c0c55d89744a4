// Shared definitions of the flow kernels: packed-weight arena layout and per-lane spline math.
//
// "Activation register layout" (ARL).  Every dense layer of the flow is evaluated for a tile of 32
// sample rows by one wave as  D(32 out-channels x 32 rows) += A(32 x 2) . B(2 x 32 rows)  on
// v_mfma_f32_32x32x2_f32 with the WEIGHTS as the A operand and the ACTIVATIONS as the B operand.
// The MFMA result layout then puts, in lane (row = l & 31, q = l >> 5), register reg = 4g + r of
// output tile t, the channel
//        ch(t, reg, q) = 32 t + 8 g + 4 q + r .
// A B operand is "one register across the wave = two k values (q = 0, 1) for all 32 rows", so the
// registers of one layer's output ARE the B operands of the next layer, k-step s = 16 t + reg, as long
// as the next layer's weights were packed with k index  kidx(s, q) = ch(s / 16, s % 16, q).
// No activation ever crosses lanes or touches LDS between layers.
#pragma once
#include "hmf_common.h"

namespace hmf {

constexpr int kCtx = 64;
constexpr int kIsgc = 256;
constexpr int kBins = 8;
constexpr int kMaxAnc = 7;
constexpr int kAncRow = 209;   // floats per row of the ancestor LDS block: 23*9 + zero slot, odd stride
constexpr int kAncZero = 207;

struct PartTab {
    int n_anc, anc[kMaxAnc];
    int kas, kag;          // ancestor k-steps ceil(9a/2) and float4 groups ceil(kas/4)
    int o_wc0, o_bctx, o_wa;
    int o_w1, o_w1e;       // first hyper-net layer of both couplings [4 tiles]; extra k-step [4][64]: (x1 column | bias)
    int o_w2[2], o_w3[2], o_w4[2];
    int o_we[2];           // bias k-steps of layers 2,3,4a,4b of coupling c: [4][64], lane q=0 -> bias, q=1 -> 0
    int o_wctxT;           // [(256 + 9a)][64]  natural transposed, for the teacher-forced kernels
    int o_nnT[2][4];       // [K][N] transposed hyper-network layers
    int o_nnb[2][4];
    // ---- round 4: the same layers packed for 16-row tiles on v_mfma_f32_16x16x4_f32 (k_flow_chain16, "ARL16" below) ----
    int kas16, kag16;      // ancestor k-steps ceil(9a/4) and float4 groups ceil(kas16/4)
    int o16_wa;            // ancestor half of the context layer: [4 tiles][kag16][64][4]
    int o16_w1, o16_w1e;   // first hyper-net layer of both couplings [8 tiles][4][64][4]; extra k-step [8][64]: (x1 column | bias | 0 | 0)
    int o16_w2[2], o16_w3[2], o16_w4[2];     // [2][4][64][4], [2][2][64][4], [4][2][64][4]
    int o16_we[2];         // bias k-steps of layers 2 (2 tiles), 3 (2 tiles), 4 (4 tiles): [8][64], lane group 0 -> bias, others 0
};
constexpr int kPartTabInts = sizeof(PartTab) / sizeof(int);

// Rounds of the kinematic schedule: parts of one round are mutually independent and all their ancestors belong to
// earlier rounds; at most kChainWaves parts per round (greedy list schedule by height, flow_pack.hip).
struct Levels {
    int nlev;
    int start[kP + 1];
    int parts[kP];
};
constexpr int kChainWaves = 4;   // one wave per SIMD: the full 512-VGPR budget for the prefetched weights

struct FlowDev {
    int feat_dim, nb, in_dim;        // in_dim = feat_dim + nb + 12
    float radius, base_std;
    const float* arena;
    const PartTab* tab;              // [23] device copy
    const int* anc_off;              // [23][64] float offset of flattened ancestor entry k (kAncZero = pad)
    int o_wisgc, o_bisgc, o_wbeta;   // natural [256][in_dim], [256]; o_wbeta: W[:, F:F+nb] in MFMA A-operand order (k_flow_ctx0)
    Levels lev;
};

// ---- non-default flow variants (flow_generic.hip): lane-per-channel kernels, every dense layer stored transposed [K][64] -----------
constexpr int kMaxT = 8;          // transforms per part
constexpr int kMaxLayers = 5;     // up to 4 hidden layers + output
struct GenNet {
    int nl;                       // number of layers (0 = absent)
    int K[kMaxLayers], N[kMaxLayers];
    int o_w[kMaxLayers], o_b[kMaxLayers];     // arena offsets: W^T [K][64] (columns >= N zero) followed by the natural copy [64 outputs][64 inputs], bias [64]
};
struct GenPartTab {
    GenNet tr[kMaxT];             // transform hyper-network ConditionalDenseNN([context, x1]) -> 62 (spline) | 2 (additive)
    GenNet plu[kMaxT];            // conditional_linear_plu: DenseNN(context) -> 9
    int o_lu[kMaxT];              // linear_plu: 9 floats, row-major LU
};
struct GenDev {
    int enabled, T, transform_type, permute_type;
    const GenPartTab* tab;        // [23] device copy
};

__host__ __device__ inline int arl_channel(int t, int reg, int q) { return 32 * t + 8 * (reg >> 2) + 4 * q + (reg & 3); }

// "ARL16": the same idea on v_mfma_f32_16x16x4_f32 for tiles of 16 sample rows (round 4: twice the work-groups, half the MFMA time per part).
// D(16 out-channels x 16 rows) += A(16 x 4) . B(4 x 16 rows); lane (row = l & 15, g = l >> 4), register r of output tile t holds channel
//        ch16(t, r, g) = 16 t + 4 g + r ,
// and that register across the wave is the B operand of k-step s = 4 t + r of the next layer (k index within the step = g).
__host__ __device__ inline int arl16_channel(int s, int g) { return 16 * (s >> 2) + 4 * g + (s & 3); }
// Last hyper-network layer, 16-row form: output (tile t, A row i) lands in lane group g = i >> 2, register r = i & 3.  Lane groups 0, 1 hold the
// spline parameters of coordinate 0, groups 2, 3 those of coordinate 1; within a coordinate the even group holds slot m = 4 t + r: widths [0,8),
// heights [8,16), the odd group derivatives [0,7), pad 7, lambdas [8,16).  Returns the index into pyro's 62-vector (w 16 | h 16 | d 14 | l 16), -1 = pad.
__host__ __device__ inline int spline_orig_index16(int t, int i) {
    const int g = i >> 2, r = i & 3, q = g >> 1, m = 4 * t + r;
    if ((g & 1) == 0) return m < 8 ? q * 8 + m : 16 + q * 8 + (m - 8);
    if (m < 7) return 32 + q * 7 + m;
    if (m == 7) return -1;
    return 46 + q * 8 + (m - 8);
}

// packed spline-parameter channel -> index into pyro's 62-vector (w 16 | h 16 | d 14 | l 16), -1 = pad.
// Lane q owns dimension q; its local slot m = 16 t + reg: [0,8) widths, [8,16) heights, [16,23) derivatives,
// 23 pad, [24,32) lambdas.
__host__ __device__ inline int spline_orig_index(int packed_channel) {
    const int t = packed_channel >> 5, g = (packed_channel & 31) >> 3, q = (packed_channel & 7) >> 2, r = packed_channel & 3;
    const int m = 16 * t + 4 * g + r;
    if (m < 8) return q * 8 + m;
    if (m < 16) return 16 + q * 8 + (m - 8);
    if (m < 23) return 32 + q * 7 + (m - 16);
    if (m == 23) return -1;
    return 46 + q * 8 + (m - 24);
}

__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : expm1f(x); }
// sampling-path variant: hardware exp2 (v_exp_f32) instead of the correctly rounded expm1f; abs error <= 2e-7 on (-1, 0]
__device__ __forceinline__ float elu_fast(float x) { return x > 0.f ? x : __expf(x) - 1.f; }
__device__ __forceinline__ float softplus1(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoid1(float x) { return 1.f / (1.f + expf(-x)); }

// pyro _monotonic_rational_spline(order='linear'), K = 8 bins, one scalar per lane.
// rw/rh/rl: 8 raw logits, rd: 7 raw derivatives.  INVERSE also returns log|d out / d in| of the inverse.
template <bool INVERSE>
__device__ __forceinline__ float rational_linear_spline(float x, const float (&rw)[8], const float (&rh)[8],
                                                        const float (&rd)[7], const float (&rl)[8], float bound,
                                                        float* logabsdet) {
    float w[8], h[8], cw[9], chh[9], dv[9];
    {
        float m = rw[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) m = fmaxf(m, rw[i]);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { w[i] = expf(rw[i] - m); s += w[i]; }
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = 1e-3f + (1.0f - 1e-3f * 8) * (w[i] / s);
        m = rh[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) m = fmaxf(m, rh[i]);
        s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { h[i] = expf(rh[i] - m); s += h[i]; }
#pragma unroll
        for (int i = 0; i < 8; ++i) h[i] = 1e-3f + (1.0f - 1e-3f * 8) * (h[i] / s);
    }
    {   // _calculate_knots: cumsum, affine map to [-B, B], pin the end knots, lengths = diff
        float a = 0.f, b = 0.f;
        cw[0] = -bound; chh[0] = -bound;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            a += w[i]; b += h[i];
            cw[i + 1] = (2.f * bound) * a + (-bound);
            chh[i + 1] = (2.f * bound) * b + (-bound);
        }
        cw[8] = bound; chh[8] = bound;
#pragma unroll
        for (int i = 0; i < 8; ++i) { w[i] = cw[i + 1] - cw[i]; h[i] = chh[i + 1] - chh[i]; }
    }
    dv[0] = 1.0f - 1e-3f; dv[8] = 1.0f - 1e-3f;
#pragma unroll
    for (int i = 0; i < 7; ++i) dv[i + 1] = 1e-3f + softplus1(rd[i]);
    int bin = -1;
#pragma unroll
    for (int i = 0; i < 9; ++i) bin += (x >= (INVERSE ? chh[i] : cw[i]) + 1e-6f) ? 1 : 0;
    bin = min(max(bin, 0), 7);
    float in_w = w[0], in_cw = cw[0], in_ch = chh[0], in_h = h[0], in_d = dv[0], in_d1 = dv[1], lam = rl[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) {
        const bool sel = bin == i;
        in_w = sel ? w[i] : in_w; in_cw = sel ? cw[i] : in_cw; in_ch = sel ? chh[i] : in_ch; in_h = sel ? h[i] : in_h;
        in_d = sel ? dv[i] : in_d; in_d1 = sel ? dv[i + 1] : in_d1; lam = sel ? rl[i] : lam;
    }
    lam = (1.f - 2.f * 0.025f) * sigmoid1(lam) + 0.025f;
    const float delta = in_h / in_w;
    const float wb = sqrtf(in_d / in_d1);
    const float wc = (lam * in_d + (1.f - lam) * wb * in_d1) / delta;
    const float ya = in_ch, yb = in_h + in_ch;
    const float yc = ((1.f - lam) * ya + lam * wb * yb) / ((1.f - lam) + lam * wb);
    float out, num, den, dnum;
    if (INVERSE) {
        const bool lo = x <= yc;
        num = lo ? (lam * (ya - x)) : ((wc - lam * wb) * x + lam * wb * yb - wc * yc);
        den = lo ? ((wc - 1.f) * x + ya - wc * yc) : ((wc - wb) * x + wb * yb - wc * yc);
        out = (num / den) * in_w + in_cw;
        dnum = (lo ? (wc * lam * (yc - ya)) : (wb * wc * (1.f - lam) * (yb - yc))) * in_w;
    } else {
        const float th = (x - in_cw) / in_w;
        const bool lo = th <= lam;
        num = lo ? (ya * (lam - th) + wc * yc * th) : (wc * yc * (1.f - th) + wb * yb * (th - lam));
        den = lo ? ((lam - th) + wc * th) : (wc * (1.f - th) + wb * (th - lam));
        out = num / den;
        dnum = (lo ? (wc * lam * (yc - ya)) : (wb * wc * (1.f - lam) * (yb - yc))) / in_w;
    }
    const bool inside = (x >= -bound) && (x <= bound);
    if (logabsdet) *logabsdet = inside ? (logf(dnum) - 2.f * logf(fabsf(den))) : 0.f;
    return inside ? out : x;
}

// Forward (sampling) direction of the same spline, arranged for latency: softmax through v_exp_f32 and one reciprocal,
// softplus only for the two derivatives of the selected bin, sigmoid only for the selected lambda, no log-det.
__device__ __forceinline__ float spline_forward_fast(float x, const float (&rw)[8], const float (&rh)[8], const float (&rd)[7],
                                                     const float (&rl)[8], float bound) {
    float w[8], h[8], cw[9], chh[9];
    {
        float m = rw[0], n = rh[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) { m = fmaxf(m, rw[i]); n = fmaxf(n, rh[i]); }
        float s = 0.f, t = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { w[i] = __expf(rw[i] - m); s += w[i]; h[i] = __expf(rh[i] - n); t += h[i]; }
        const float is = (1.0f - 1e-3f * 8) / s, it = (1.0f - 1e-3f * 8) / t;
#pragma unroll
        for (int i = 0; i < 8; ++i) { w[i] = fmaf(w[i], is, 1e-3f); h[i] = fmaf(h[i], it, 1e-3f); }
    }
    {
        float a = 0.f, b = 0.f;
        cw[0] = -bound; chh[0] = -bound;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            a += w[i]; b += h[i];
            cw[i + 1] = fmaf(2.f * bound, a, -bound);
            chh[i + 1] = fmaf(2.f * bound, b, -bound);
        }
        cw[8] = bound; chh[8] = bound;
    }
    int bin = -1;
#pragma unroll
    for (int i = 0; i < 9; ++i) bin += (x >= cw[i] + 1e-6f) ? 1 : 0;
    bin = min(max(bin, 0), 7);
    float k_lo = cw[0], k_hi = cw[1], c_lo = chh[0], c_hi = chh[1], lam = rl[0];
    float rd0 = 0.f, rd1 = rd[0];          // raw derivatives left/right of the bin (the end knots' are fixed, below)
#pragma unroll
    for (int i = 1; i < 8; ++i) {
        const bool sel = bin == i;
        k_lo = sel ? cw[i] : k_lo; k_hi = sel ? cw[i + 1] : k_hi; c_lo = sel ? chh[i] : c_lo; c_hi = sel ? chh[i + 1] : c_hi;
        lam = sel ? rl[i] : lam; rd0 = sel ? rd[i - 1] : rd0;
        if (i < 7) rd1 = sel ? rd[i] : rd1;
    }
    const float in_w = k_hi - k_lo, in_h = c_hi - c_lo;
    const float in_d = bin == 0 ? 1.0f - 1e-3f : 1e-3f + softplus1(rd0);
    const float in_d1 = bin == 7 ? 1.0f - 1e-3f : 1e-3f + softplus1(rd1);
    lam = (1.f - 2.f * 0.025f) * sigmoid1(lam) + 0.025f;
    const float delta = in_h / in_w;
    const float wb = sqrtf(in_d / in_d1);
    const float wc = (lam * in_d + (1.f - lam) * wb * in_d1) / delta;
    const float ya = c_lo, yb = c_hi;
    const float yc = ((1.f - lam) * ya + lam * wb * yb) / ((1.f - lam) + lam * wb);
    const float th = (x - k_lo) / in_w;
    const bool lo = th <= lam;
    const float num = lo ? (ya * (lam - th) + wc * yc * th) : (wc * yc * (1.f - th) + wb * yb * (th - lam));
    const float den = lo ? ((lam - th) + wc * th) : (wc * (1.f - th) + wb * (th - lam));
    const bool inside = (x >= -bound) && (x <= bound);
    return inside ? num / den : x;
}

}  // namespace hmf
