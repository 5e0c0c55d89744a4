// Ancestor-conditioned SO(3) flow on gfx950: sampling / point-estimate pass and teacher-forced log-prob.
//
// hmf_flow_forward launch sequence (replaces the 23-iteration Python loop of
// models/humaniflow_model.py:286-311 and its O(10^3) eager launches):
//   k_isgc        G[b] = W_isgc[:, feats|R|cam] . [f_b, vec(R_b), cam_b] + bias      (per image, no ELU)
//   k_flow_ctx0   rows = B*N samples (+ B point-estimate rows), tiles of 32 rows, one wave per
//                 (tile, 3 parts): h = ELU(G[b] + W_beta beta_row) built straight into MFMA B-operand
//                 registers; C0[part] = b_ctx + W_ctx[:, :256] h for ALL 23 parts (no kinematic dependency)
//   k_flow_chain  one workgroup per 32-row tile, 5 waves; the 23 parts run as 8 dependency levels, the parts
//                 of a level on different waves.  A wave takes its part through ancestor context -> ELU ->
//                 2 x (hyper-network 65->64->32->32->62 + rational-linear spline) -> radial tanh -> exp,
//                 entirely in registers (flow_common.h "ARL"); ancestor rotations live in LDS.
// Teacher-forced side: k_flow_context (all 23 contexts from given rotations) and k_flow_logprob (log-map,
// 2*pi pre-images, inverse flow, logsumexp; local_diffeo_transformed_distribution.py:91-142).
#include "flow_common.h"
#include "flow_handle.h"
#include "so3_math.h"
#include "rng_philox.h"

namespace hmf {

// ---------------------------------------------------------------------------------------------------
// k_isgc: out[b][o] = bias[o] + W[o][0:F].f_b (+ W[o][F:F+nb].shape_b) + W[o][F+nb:].[vec(R_b), cam_b]; optional ELU.
// One wave per (output o, chunk of 8 images): the weight row sits in registers, lanes split the F axis.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_isgc(FlowDev d, const float* __restrict__ feats, const float* __restrict__ shape,
                                              const float* __restrict__ glob_R, const float* __restrict__ cam, int B,
                                              int apply_elu, float* __restrict__ out, RngPass rng_imm, unsigned long long* __restrict__ rng_state,
                                              unsigned long long rng_inc, RngPass* __restrict__ rng_pass,
                                              unsigned long long* __restrict__ rng_mirror) {
    // First kernel of a sampling pass: fix the pass' position in the generator's stream for the kernels that follow.  Eager calls carry
    // (seed, offset) as immediates (the host read them from the device generator and advances it); a pass captured into a HIP graph takes
    // them from a device-resident state that every replay advances atomically (concurrent replays on other streams get disjoint ranges).
    // An eager pass also leaves the generator's state AFTER itself in `rng_mirror` (that same device-resident state), so that a pass captured
    // later continues the stream where the eager passes stopped instead of repeating what a warm-up pass already drew.
    if (rng_pass && blockIdx.x == 0 && threadIdx.x == 0) *rng_pass = rng_begin_pass(rng_imm, rng_state, rng_inc, rng_mirror);
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nchunk = (B + 7) / 8;
    const int o = gw / nchunk, b0 = (gw % nchunk) * 8;
    if (o >= kIsgc) return;
    const int F = d.feat_dim, nb = d.nb;
    const float* wrow = d.arena + d.o_wisgc + (size_t)o * d.in_dim;
    const int per = F / 256;                          // float4 per lane (F multiple of 256)
    for (int bb = b0; bb < min(b0 + 8, B); ++bb) {
        const f32x4* fr = (const f32x4*)(feats + (size_t)bb * F);
        const f32x4* wr = (const f32x4*)wrow;
        float acc = 0.f;
        for (int i = 0; i < per; ++i) {
            const f32x4 a = wr[i * 64 + lane], x = fr[i * 64 + lane];
            acc = fmaf(a[0], x[0], acc); acc = fmaf(a[1], x[1], acc); acc = fmaf(a[2], x[2], acc); acc = fmaf(a[3], x[3], acc);
        }
        if (lane < 9) acc = fmaf(wrow[F + nb + lane], glob_R[(size_t)bb * 9 + lane], acc);
        else if (lane < 12) acc = fmaf(wrow[F + nb + lane], cam[(size_t)bb * 3 + (lane - 9)], acc);
        else if (shape && lane >= 16 && lane < 16 + nb) acc = fmaf(wrow[F + lane - 16], shape[(size_t)bb * nb + (lane - 16)], acc);
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) acc += __shfl_xor(acc, s);
        if (lane == 0) {
            float v = acc + d.arena[d.o_bisgc + o];
            out[(size_t)bb * kIsgc + o] = apply_elu ? elu1(v) : v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// The isgc layer under autograd (training step; models/humaniflow_model.py:116-150 differentiated by the reference's autograd).
// Forward: pre[u][b] = W.[f_b, 0, vec(R^u_b), cam_b] + bias (k_isgc without shape / ELU) for the two uses of a training pass -- u = 0 the sample path with
// the PREDICTED global rotation, u = 1 the teacher-forced path with the TARGET one (:279-283) -- then one row kernel for both:
//   H[row] = ELU(pre[use of row][image of row] + W_beta . shape[row]),  rows = the B N per-sample shapes, then the B teacher-forced shapes.
// Backward: delta = g_H * ELU'(H) per row (stored for the dense weight-gradient reductions), g_pre_s[b] = sum of an image's sample rows of delta (fixed
// order), g_pre_f = g_pre_s + delta_ll (what reaches the shared image-feature product), g_shape[row] = W_beta^T delta[row].  The dense products with the
// image features and W stay calls of hmf_rows_atb_batched.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_isgc_rows(FlowDev d, const float* __restrict__ pre, const float* __restrict__ shape_rows, const float* __restrict__ shape_ll,
                                                   int B, int N, float* __restrict__ H) {
    const int row = blockIdx.x, o = threadIdx.x, BN = B * N, nb = d.nb;
    const int b = row < BN ? row / N : row - BN;
    const float* sh = row < BN ? shape_rows + (size_t)row * nb : shape_ll + (size_t)(row - BN) * nb;      // wave-uniform
    const float* wb = d.arena + d.o_wisgc + (size_t)o * d.in_dim + d.feat_dim;       // W[o][F : F + nb] of the natural copy (o_wbeta is in MFMA operand order)
    float acc = pre[((size_t)(row < BN ? 0 : B) + b) * kIsgc + o];
    for (int k = 0; k < nb; ++k) acc = fmaf(wb[k], sh[k], acc);
    H[(size_t)row * kIsgc + o] = elu1(acc);
}
// thread (image b, channel o): deltas of the image's N sample rows and of its teacher-forced row.  `delta` has (B N + 2 B) rows: sample rows, teacher-forced
// rows, then g_pre_s -- so that rows [B N, B N + 2 B) are one contiguous operand for the [vec(R), cam] weight gradient.
__global__ __launch_bounds__(256) void k_isgc_bwd_delta(const float* __restrict__ H, const float* __restrict__ gH_rows, const float* __restrict__ gH_ll, int B, int N,
                                                        float* __restrict__ delta, float* __restrict__ g_pre_f, float* __restrict__ g_pre_f_t,
                                                        float* __restrict__ g_pre_s_t) {
    const int b = blockIdx.x, o = threadIdx.x, BN = B * N;
    float acc = 0.f;
    for (int n = 0; n < N; ++n) {
        const size_t row = (size_t)b * N + n;
        float dl = 0.f;
        if (gH_rows) {
            const float g = gH_rows[row * kIsgc + o], h = H[row * kIsgc + o];
            dl = h > 0.f ? g : g * (h + 1.f);                          // ELU' = 1 (pre > 0) or exp(pre) = H + 1
        }
        delta[row * kIsgc + o] = dl;
        acc += dl;
    }
    float dll = 0.f;
    if (gH_ll) {
        const float g = gH_ll[(size_t)b * kIsgc + o], h = H[((size_t)BN + b) * kIsgc + o];
        dll = h > 0.f ? g : g * (h + 1.f);
    }
    delta[((size_t)BN + b) * kIsgc + o] = dll;
    delta[((size_t)BN + B + b) * kIsgc + o] = acc;                       // g_pre_s
    g_pre_s_t[(size_t)o * B + b] = acc;
    g_pre_f[(size_t)b * kIsgc + o] = acc + dll;
    g_pre_f_t[(size_t)o * B + b] = acc + dll;
}
// thread (row, k): g_shape[row][k] = sum_o delta[row][o] W_beta[o][k]
__global__ __launch_bounds__(256) void k_isgc_bwd_shape(FlowDev d, const float* __restrict__ delta, int rows, float* __restrict__ g_shape) {
    const int gid = blockIdx.x * 256 + threadIdx.x, nb = d.nb;
    if (gid >= rows * nb) return;
    const int row = gid / nb, k = gid - row * nb;
    const float* dl = delta + (size_t)row * kIsgc;
    const float* wb = d.arena + d.o_wisgc + d.feat_dim + k;          // W[o][F + k] of the natural copy
    const size_t ld = d.in_dim;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int o = 0; o < kIsgc; o += 4) {
        a0 = fmaf(dl[o], wb[(size_t)o * ld], a0);
        a1 = fmaf(dl[o + 1], wb[(size_t)(o + 1) * ld], a1);
        a2 = fmaf(dl[o + 2], wb[(size_t)(o + 2) * ld], a2);
        a3 = fmaf(dl[o + 3], wb[(size_t)(o + 3) * ld], a3);
    }
    g_shape[gid] = (a0 + a1) + (a2 + a3);
}

// The pass' base noise as tensors (training keeps them for the reverse sweep; tests compare them with torch.randn): the same generator calls the
// sampling kernels make in registers, materialised by one launch.
__global__ void k_rng_snapshot(RngPass rng_imm, unsigned long long* __restrict__ rng_state, unsigned long long rng_inc, RngPass* __restrict__ rng_pass,
                               unsigned long long* __restrict__ rng_mirror) {
    *rng_pass = rng_begin_pass(rng_imm, rng_state, rng_inc, rng_mirror);
}
__global__ __launch_bounds__(256) void k_draw_noise(NoiseSrc ns, unsigned int n_shape, unsigned int n_part, float* __restrict__ eps_shape,
                                                    float* __restrict__ eps_pose) {
    const unsigned int gid = blockIdx.x * 256u + threadIdx.x;
    if (gid < n_shape) {
        eps_shape[gid] = shape_noise(ns, gid);
    } else if (gid - n_shape < (unsigned int)kP * n_part) {
        const unsigned int r = gid - n_shape, j = r / n_part, e = r - j * n_part;
        eps_pose[r] = pose_noise(ns, (int)j, e, n_part);
    }
}

// ---------------------------------------------------------------------------------------------------
// row bookkeeping shared by the two forward kernels
// ---------------------------------------------------------------------------------------------------
struct RowInfo {
    bool live, is_point;
    int b, n;
};
__device__ __forceinline__ RowInfo row_info(int rowid, int B, int N, int Mtot) {
    RowInfo r;
    r.live = rowid < Mtot;
    const int rs = r.live ? rowid : 0;
    r.is_point = rs >= B * N;
    r.b = r.is_point ? rs - B * N : rs / max(N, 1);
    r.n = r.is_point ? 0 : rs - r.b * N;
    return r;
}

template <int NTO>
__device__ __forceinline__ void bias_init(const float* __restrict__ bias, int q, f32x16 (&acc)[NTO]) {
#pragma unroll
    for (int t = 0; t < NTO; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b = *(const f32x4*)(bias + 32 * t + 8 * g + 4 * q);
            acc[t][4 * g + 0] = b[0]; acc[t][4 * g + 1] = b[1]; acc[t][4 * g + 2] = b[2]; acc[t][4 * g + 3] = b[3];
        }
}

// acc[to] += W(tile to) . in   with `in` in ARL over NKT tiles; wp = packed block [NTO][NKT*4][64] float4, lane-adjusted.
template <int NTO, int NKT>
__device__ __forceinline__ void dense_layer(const f32x4* __restrict__ wp, const f32x16 (&in)[NKT], f32x16 (&acc)[NTO]) {
    constexpr int NS4 = NKT * 4;
#pragma unroll
    for (int s4 = 0; s4 < NS4; ++s4) {
        f32x4 w[NTO];
#pragma unroll
        for (int to = 0; to < NTO; ++to) w[to] = wp[(to * NS4 + s4) * 64];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int s = s4 * 4 + u;
#pragma unroll
            for (int to = 0; to < NTO; ++to)
                acc[to] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[to][u], in[s >> 4][s & 15], acc[to], 0, 0, 0);
        }
    }
}

// Same contraction with the weight stream software-pipelined DEPTH float4 groups ahead (the plain version above issues
// each group's loads right before its MFMAs and eats an L2 round trip per group).
template <int NTO, int NKT, int DEPTH>
__device__ __forceinline__ void dense_layer_pf(const f32x4* __restrict__ wp, const f32x16 (&in)[NKT], f32x16 (&acc)[NTO]) {
    constexpr int NS4 = NKT * 4;
    f32x4 w[DEPTH][NTO];
#pragma unroll
    for (int d0 = 0; d0 < DEPTH; ++d0)
#pragma unroll
        for (int to = 0; to < NTO; ++to) w[d0][to] = wp[(to * NS4 + d0) * 64];
#pragma unroll
    for (int s4 = 0; s4 < NS4; ++s4) {
        f32x4 cur[NTO];
#pragma unroll
        for (int to = 0; to < NTO; ++to) cur[to] = w[s4 % DEPTH][to];
        if (s4 + DEPTH < NS4) {
#pragma unroll
            for (int to = 0; to < NTO; ++to) w[s4 % DEPTH][to] = wp[(to * NS4 + s4 + DEPTH) * 64];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int s = s4 * 4 + u;
#pragma unroll
            for (int to = 0; to < NTO; ++to)
                acc[to] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[to][u], in[s >> 4][s & 15], acc[to], 0, 0, 0);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// k_flow_ctx0: image/shape half of every part's context layer.
// ---------------------------------------------------------------------------------------------------
// PPW parts per wave, four waves per work-group.  PPW = 3 (two work-groups per 32-row tile: 808 waves at B = 32, N = 100) is the throughput form;
// PPW = 1 (six work-groups per tile) the latency form for a handful of tiles (predict: one image, 11 rows -> 25 us instead of 49 us).  At full
// size one part per wave loses, 65 us vs 52-59 us: every wave rebuilds h, 64 MFMAs + 128 ELUs, for a 256-MFMA layer (round 4, measured).
constexpr int ctx0_groups(int ppw) { return ((kP + ppw - 1) / ppw + 3) / 4; }     // work-groups per tile

template <int PPW>
__global__ __launch_bounds__(256, 1) void k_flow_ctx0(FlowDev d, const float* __restrict__ G, const float* __restrict__ shape_mode,
                                                      const float* __restrict__ shape_log_std, NoiseSrc ns,
                                                      int B, int N, int Mtot, int ntiles, int packed_rows,
                                                      float* __restrict__ shape_samples, float* __restrict__ C0) {
    constexpr int kPartsPerWave0 = PPW, kGroups0 = ctx0_groups(PPW);
    const int lane = threadIdx.x & 63, l31 = lane & 31, q = lane >> 5;
    const int tile = blockIdx.x / kGroups0, pg = (blockIdx.x % kGroups0) * 4 + (threadIdx.x >> 6);      // pg = the wave's part (>= kP: idle after the barrier)
    const int nb = d.nb;
    const RowInfo ri = row_info(tile * 32 + l31, B, N, Mtot);
    // shape vectors of the tile's 32 rows (models/humaniflow_model.py:250-258), computed ONCE per work-group: its four waves work on the same
    // tile (block = (tile, 4 consecutive parts)), so thread (wave, q) takes the elements i = 2 * wave + q (+ 8) of its lane's row
    // and the rows are exchanged through LDS (row pitch 17: conflict-free).  Round 3 kept `float beta[16]` per thread and indexed it with a
    // run-time bound, which the compiler turned into a 16 KB LDS array with a 64-byte lane stride: the 93.6 % LDS bank-conflict cycles of the
    // round-1..3 profiles.  With the noise drawn in the kernel a lane now also pays for 1-2 Philox blocks instead of 10.
    __shared__ float sbeta[32][17];
    const int wv = threadIdx.x >> 6;
    for (int i = 2 * wv + q; i < 16; i += 8) {
        float v = 0.f;
        if (i < nb && ri.live) {
            v = shape_mode[(size_t)ri.b * nb + i];
            if (!ri.is_point && has_shape_noise(ns))
                v = v + shape_noise(ns, (unsigned int)((ri.n * B + ri.b) * nb + i)) * expf(shape_log_std[(size_t)ri.b * nb + i]);
        }
        sbeta[l31][i] = v;
    }
    __syncthreads();
    float beta[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) beta[i] = sbeta[l31][i];
    // row id = b*N + n for samples, B*N + b for point-estimate rows: with HMF_FLOW_PACKED_ROWS the shape buffer carries both
    if (pg == 0 && q == 0 && ri.live && shape_samples && (!ri.is_point || packed_rows)) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (i < nb) shape_samples[((size_t)tile * 32 + l31) * nb + i] = beta[i];
    }
    // h = ELU(G[b] + W_beta . beta) for the 256 isgc channels, produced directly in ARL registers: the rank-nb
    // update is itself an MFMA (A = packed W_beta, B = this lane's beta pair), G[b] is gathered as float4.
    f32x16 h[8];
    {
        const float* Gb = G + (size_t)ri.b * kIsgc;
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = *(const f32x4*)(Gb + 32 * t + 8 * g + 4 * q);
                h[t][4 * g] = v[0]; h[t][4 * g + 1] = v[1]; h[t][4 * g + 2] = v[2]; h[t][4 * g + 3] = v[3];
            }
        const int g4 = (((nb + 1) >> 1) + 3) >> 2;
        const f32x4* wp = (const f32x4*)(d.arena + d.o_wbeta) + lane;
#pragma unroll
        for (int s4 = 0; s4 < 2; ++s4) {
            if (s4 < g4) {
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const f32x4 w = wp[(t * g4 + s4) * 64];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int s = s4 * 4 + u;
                        h[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[u], q ? beta[2 * s + 1] : beta[2 * s], h[t], 0, 0, 0);
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) h[t][e] = ri.live ? elu_fast(h[t][e]) : 0.f;
    }
    for (int pi = 0; pi < kPartsPerWave0; ++pi) {
        const int j = pg * kPartsPerWave0 + pi;
        if (j >= kP) break;
        const PartTab& pt = d.tab[j];
        f32x16 acc[2];
        bias_init<2>(d.arena + pt.o_bctx, q, acc);
        dense_layer_pf<2, 8, 6>((const f32x4*)(d.arena + pt.o_wc0) + lane, h, acc);
        // C0 is row-major [row][part][64 channels] (round 4: the 16-row chain kernel and the 32-row one read it through their own register
        // layouts): register 4 g + r of tile t of lane (row, q) is channel 32 t + 8 g + 4 q + r
        float* out = C0 + (((size_t)tile * 32 + l31) * kP + j) * kCtx + 4 * q;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
                *(f32x4*)(out + 32 * t + 8 * g) = v;
            }
    }
}

// ---------------------------------------------------------------------------------------------------
// k_flow_chain
// ---------------------------------------------------------------------------------------------------
// Hyper-network tail (layers 2..4) of one coupling.  Its 20 weight float4 + 4 bias dwords are fetched by tail_fetch()
// one stage before tail_compute() consumes them, so the L2 latency of this dependent chain of small layers hides behind
// the previous stage's MFMAs.  Biases ride in one extra MFMA k-step per tile (A = (bias | 0), B = (1 | 0)): no bias
// registers, accumulators start at zero.
struct TailRegs {
    f32x4 w2[8], w3[4], w4[2][4];
    float we[4];
};

__device__ __forceinline__ void tail_fetch(const FlowDev& d, const PartTab& pt, int c, int lane, TailRegs& t) {
    const f32x4* p2 = (const f32x4*)(d.arena + pt.o_w2[c]) + lane;
    const f32x4* p3 = (const f32x4*)(d.arena + pt.o_w3[c]) + lane;
    const f32x4* p4 = (const f32x4*)(d.arena + pt.o_w4[c]) + lane;
    const float* pe = d.arena + pt.o_we[c] + lane;
#pragma unroll
    for (int i = 0; i < 8; ++i) t.w2[i] = p2[i * 64];
#pragma unroll
    for (int i = 0; i < 4; ++i) t.w3[i] = p3[i * 64];
#pragma unroll
    for (int to = 0; to < 2; ++to)
#pragma unroll
        for (int i = 0; i < 4; ++i) t.w4[to][i] = p4[(to * 4 + i) * 64];
#pragma unroll
    for (int i = 0; i < 4; ++i) t.we[i] = pe[i * 64];
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int e = 0; e < 16; ++e) z[e] = 0.f;
    return z;
}

__device__ __forceinline__ float tail_compute(const TailRegs& t, const f32x16 (&h1)[2], float x, float bound, float one_q0) {
    f32x16 a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(t.we[0], one_q0, zero16(), 0, 0, 0);
#pragma unroll
    for (int s4 = 0; s4 < 8; ++s4)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int s = s4 * 4 + u;
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(t.w2[s4][u], h1[s >> 4][s & 15], a2, 0, 0, 0);
        }
#pragma unroll
    for (int e = 0; e < 16; ++e) a2[e] = fmaxf(a2[e], 0.f);
    f32x16 a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(t.we[1], one_q0, zero16(), 0, 0, 0);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int u = 0; u < 4; ++u) a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(t.w3[s4][u], a2[s4 * 4 + u], a3, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 16; ++e) a3[e] = fmaxf(a3[e], 0.f);
    f32x16 a40 = __builtin_amdgcn_mfma_f32_32x32x2f32(t.we[2], one_q0, zero16(), 0, 0, 0);
    f32x16 a41 = __builtin_amdgcn_mfma_f32_32x32x2f32(t.we[3], one_q0, zero16(), 0, 0, 0);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a40 = __builtin_amdgcn_mfma_f32_32x32x2f32(t.w4[0][s4][u], a3[s4 * 4 + u], a40, 0, 0, 0);
            a41 = __builtin_amdgcn_mfma_f32_32x32x2f32(t.w4[1][s4][u], a3[s4 * 4 + u], a41, 0, 0, 0);
        }
    float rw[8], rh[8], rd[7], rl[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { rw[i] = a40[i]; rh[i] = a40[8 + i]; rl[i] = a41[8 + i]; }
#pragma unroll
    for (int i = 0; i < 7; ++i) rd[i] = a41[i];
    return spline_forward_fast(x, rw, rh, rd, rl, bound);
}

__global__ __launch_bounds__(kChainWaves * 64, 1) void k_flow_chain(FlowDev d, const float* __restrict__ C0,
                                                                    NoiseSrc ns, int B, int N, int Mtot,
                                                                    float* __restrict__ rot_samples, float* __restrict__ rot_point,
                                                                    float* __restrict__ so3_point, long long* __restrict__ timeline) {
    __shared__ float sAnc[32 * kAncRow];
    __shared__ int sOff[kP * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, q = lane >> 5;
    const int tile = blockIdx.x;
    for (int i = tid; i < 32 * kAncRow; i += kChainWaves * 64) sAnc[i] = 0.f;
    for (int i = tid; i < kP * 64; i += kChainWaves * 64) sOff[i] = d.anc_off[i];
    __syncthreads();
    const int rowid = tile * 32 + l31;
    const RowInfo ri = row_info(rowid, B, N, Mtot);
    const float bound = d.radius;
    const float one_q0 = q == 0 ? 1.f : 0.f;
    float* myAnc = sAnc + l31 * kAncRow;

    for (int lev = 0; lev < d.lev.nlev; ++lev) {
        const int pi = d.lev.start[lev] + wave;
        if (pi < d.lev.start[lev + 1]) {
            const int j = d.lev.parts[pi];
            const PartTab& pt = d.tab[j];
            // diagnostic timeline (hmf_debug_set_timeline): shader-clock stamps of tile 0's phases, 8 per part
            long long* tl = (timeline && tile == 0 && lane == 0) ? timeline + j * 8 : nullptr;
#define HMF_STAMP(i) do { if (tl) tl[i] = (long long)__builtin_readcyclecounter(); } while (0)
            HMF_STAMP(0);
            // ================= stage 0: issue every load the first two stages need =================
            // ancestor entries of this lane's row (B operands of the ancestor half): 32 LDS reads up front
            float bv[32];
            {
                const int* off = sOff + j * 64 + q;
#pragma unroll
                for (int s = 0; s < 32; ++s) bv[s] = myAnc[off[2 * s]];
            }
            f32x16 cx[2];                                   // context accumulators <- C0 (image/shape half, k_flow_ctx0)
            {
                const float* src = C0 + (((size_t)tile * 32 + l31) * kP + j) * kCtx + 4 * q;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 v = *(const f32x4*)(src + 32 * t + 8 * g);
                        cx[t][4 * g] = v[0]; cx[t][4 * g + 1] = v[1]; cx[t][4 * g + 2] = v[2]; cx[t][4 * g + 3] = v[3];
                    }
            }
            const int kag = pt.kag;
            f32x4 wa[2][8];
            {
                const f32x4* wp = (const f32x4*)(d.arena + pt.o_wa) + lane;
#pragma unroll
                for (int s4 = 0; s4 < 8; ++s4)
                    if (s4 < kag) { wa[0][s4] = wp[s4 * 64]; wa[1][s4] = wp[(kag + s4) * 64]; }
            }
            const f32x4* w1p = (const f32x4*)(d.arena + pt.o_w1) + lane;
            f32x4 w1a[4][4], w1b[4][4];                     // first layer of both couplings: k-step groups 0..3 / 4..7
#pragma unroll
            for (int to = 0; to < 4; ++to)
#pragma unroll
                for (int i = 0; i < 4; ++i) w1a[to][i] = w1p[(to * 8 + i) * 64];
            float w1e[4];
#pragma unroll
            for (int to = 0; to < 4; ++to) w1e[to] = d.arena[pt.o_w1e + to * 64 + lane];
            float x0 = 0.f, x1 = 0.f, x2 = 0.f;            // base sample z = base_std * eps; point-estimate rows push z = 0
            if (ri.live && !ri.is_point) {
                const unsigned int e = (unsigned int)((ri.b * N + ri.n) * 3);
                const unsigned long long per_part = (unsigned long long)B * N * 3;
                x0 = pose_noise(ns, j, e, per_part) * d.base_std; x1 = pose_noise(ns, j, e + 1, per_part) * d.base_std;
                x2 = pose_noise(ns, j, e + 2, per_part) * d.base_std;
            }
            TailRegs tr0, tr1;
            tail_fetch(d, pt, 0, lane, tr0);
            __builtin_amdgcn_sched_barrier(0);
            HMF_STAMP(1);
            // ================= stage 1: ancestor half of the context layer, ELU =================
#pragma unroll
            for (int s4 = 0; s4 < 8; ++s4) {
                if (s4 < kag) {                                           // wave-uniform
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        cx[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[0][s4][u], bv[s4 * 4 + u], cx[0], 0, 0, 0);
                        cx[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[1][s4][u], bv[s4 * 4 + u], cx[1], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int to = 0; to < 4; ++to)                                  // second half of the first-layer weights
#pragma unroll
                for (int i = 0; i < 4; ++i) w1b[to][i] = w1p[(to * 8 + 4 + i) * 64];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) cx[t][e] = elu_fast(cx[t][e]);
            HMF_STAMP(2);
            // ================= stage 2: first hyper-network layer of both couplings =================
            // extra k-step: (x1 column | bias) . (x0 | 1) for coupling 0 now; coupling 1's needs y0 and comes later
            const float bx0 = q == 0 ? x0 : 1.f;
            f32x16 a1[4];
            a1[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1e[0], bx0, zero16(), 0, 0, 0);
            a1[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1e[1], bx0, zero16(), 0, 0, 0);
            a1[2] = zero16();
            a1[3] = zero16();
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int to = 0; to < 4; ++to)
                        a1[to] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1a[to][s4][u], cx[0][s4 * 4 + u], a1[to], 0, 0, 0);
            tail_fetch(d, pt, 1, lane, tr1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int to = 0; to < 4; ++to)
                        a1[to] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1b[to][s4][u], cx[1][s4 * 4 + u], a1[to], 0, 0, 0);
            HMF_STAMP(3);
            // ================= stage 3: coupling 0 (Permute [0,1,2]): x0 passes, lane q transforms x_{1+q} =================
            f32x16 h1[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) h1[t][e] = fmaxf(a1[t][e], 0.f);
            const float xn = tail_compute(tr0, h1, q ? x2 : x1, bound, one_q0);
            const float x1n = __shfl(xn, l31), x2n = __shfl(xn, 32 + l31);
            HMF_STAMP(4);
            // ================= stage 4: coupling 1 (Permute [1,2,0]): y = (x1', x2', x0); y0 passes =================
            const float y0 = x1n;
            const float by0 = q == 0 ? y0 : 1.f;
            a1[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1e[2], by0, a1[2], 0, 0, 0);
            a1[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1e[3], by0, a1[3], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) h1[t][e] = fmaxf(a1[2 + t][e], 0.f);
            const float yn = tail_compute(tr1, h1, q ? x0 : x2n, bound, one_q0);
            const float y1 = __shfl(yn, l31), y2 = __shfl(yn, 32 + l31);
            HMF_STAMP(5);
            // ================= stage 5: ScaledRadialTanh (scaled_radial_tanh_transform.py:28-35) + exp map =================
            float v[3] = {y0, y1, y2};
            {
                const float nrm = sqrtf(y0 * y0 + y1 * y1 + y2 * y2);
                if (nrm > 1e-7f) {
                    const float tt = tanhf(nrm / bound);
                    v[0] = tt * (y0 / nrm) * bound; v[1] = tt * (y1 / nrm) * bound; v[2] = tt * (y2 / nrm) * bound;
                }
            }
            // fp64 Rodrigues rounded to fp32 for samples (:311), smplx Rodrigues for point rows (:299)
            float R[9];
            if (ri.is_point) {
                batch_rodrigues_f32(v, R);
            } else {
                const double vd[3] = {(double)v[0], (double)v[1], (double)v[2]};
                double Rd[9];
                so3_exp_f64(vd, Rd);
#pragma unroll
                for (int e = 0; e < 9; ++e) R[e] = (float)Rd[e];
            }
            HMF_STAMP(6);
            if (q == 0) {
#pragma unroll
                for (int e = 0; e < 9; ++e) myAnc[j * 9 + e] = R[e];
                if (ri.live) {
                    if (ri.is_point) {
                        float* o = rot_point + ((size_t)ri.b * kP + j) * 9;
#pragma unroll
                        for (int e = 0; e < 9; ++e) o[e] = R[e];
                        float* a = so3_point + ((size_t)ri.b * kP + j) * 3;
                        a[0] = v[0]; a[1] = v[1]; a[2] = v[2];
                    } else {
                        float* o = rot_samples + ((size_t)rowid * kP + j) * 9;
#pragma unroll
                        for (int e = 0; e < 9; ++e) o[e] = R[e];
                    }
                }
            }
            HMF_STAMP(7);
#undef HMF_STAMP
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// k_flow_chain16 (round 4): the same kinematic loop on 16-row tiles and v_mfma_f32_16x16x4_f32 (flow_common.h "ARL16").  k_flow_chain ran 101
// work-groups on 256 CUs at B = 32, N = 100, each a latency-bound stream of ~430 64-cycle MFMAs per part and round: half the rows per tile =
// the same number of 32-cycle MFMAs = half the time per part, on 202 work-groups; a single image (predict: one tile either way) halves its
// chain latency.  Lane (row j = l & 15, g = l >> 4).  Spline parameters of coordinate q sit in lane groups 2q (widths | heights) and 2q + 1
// (derivatives | lambdas); the two halves of a spline evaluation meet through four cross-lane reads.
// ---------------------------------------------------------------------------------------------------
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

struct TailRegs16 {
    f32x4 w2[2][4], w3[2][2], w4[4][2];
    float we[8];
};
__device__ __forceinline__ void tail_fetch16(const FlowDev& d, const PartTab& pt, int c, int lane, TailRegs16& t) {
    const f32x4* p2 = (const f32x4*)(d.arena + pt.o16_w2[c]) + lane;
    const f32x4* p3 = (const f32x4*)(d.arena + pt.o16_w3[c]) + lane;
    const f32x4* p4 = (const f32x4*)(d.arena + pt.o16_w4[c]) + lane;
    const float* pe = d.arena + pt.o16_we[c] + lane;
#pragma unroll
    for (int to = 0; to < 2; ++to)
#pragma unroll
        for (int i = 0; i < 4; ++i) t.w2[to][i] = p2[(to * 4 + i) * 64];
#pragma unroll
    for (int to = 0; to < 2; ++to)
#pragma unroll
        for (int i = 0; i < 2; ++i) t.w3[to][i] = p3[(to * 2 + i) * 64];
#pragma unroll
    for (int to = 0; to < 4; ++to)
#pragma unroll
        for (int i = 0; i < 2; ++i) t.w4[to][i] = p4[(to * 2 + i) * 64];
#pragma unroll
    for (int i = 0; i < 8; ++i) t.we[i] = pe[i * 64];
}

// layers 2..4 of one coupling + the spline of this lane's coordinate; h1: the four ReLU'd tiles of layer 1; returns the transformed coordinate
// (valid in the even lane group of the coordinate)
__device__ __forceinline__ float tail_compute16(const TailRegs16& t, const f32x4 (&h1)[4], float x, float bound, float one_g0, int g) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 a2[2] = {mfma16(t.we[0], one_g0, z), mfma16(t.we[1], one_g0, z)};
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int to = 0; to < 2; ++to) a2[to] = mfma16(t.w2[to][s >> 2][s & 3], h1[s >> 2][s & 3], a2[to]);
#pragma unroll
    for (int to = 0; to < 2; ++to)
#pragma unroll
        for (int e = 0; e < 4; ++e) a2[to][e] = fmaxf(a2[to][e], 0.f);
    f32x4 a3[2] = {mfma16(t.we[2], one_g0, z), mfma16(t.we[3], one_g0, z)};
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int to = 0; to < 2; ++to) a3[to] = mfma16(t.w3[to][s >> 2][s & 3], a2[s >> 2][s & 3], a3[to]);
#pragma unroll
    for (int to = 0; to < 2; ++to)
#pragma unroll
        for (int e = 0; e < 4; ++e) a3[to][e] = fmaxf(a3[to][e], 0.f);
    f32x4 a4[4] = {mfma16(t.we[4], one_g0, z), mfma16(t.we[5], one_g0, z), mfma16(t.we[6], one_g0, z), mfma16(t.we[7], one_g0, z)};
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int to = 0; to < 4; ++to) a4[to] = mfma16(t.w4[to][s >> 2][s & 3], a3[s >> 2][s & 3], a4[to]);
    // slots m = 4 t + r.  Even lane group: widths [0,8) | heights [8,16).  Odd group: derivatives [0,7) | pad | lambdas [8,16).
    float p[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) p[m] = a4[m >> 2][m & 3];
    // ---- role A (meaningful in the even group): knots and the bin of x  (spline_forward_fast, first half) ----
    float w[8], h[8], cw[9], chh[9];
    {
        float mx = p[0], nx = p[8];
#pragma unroll
        for (int i = 1; i < 8; ++i) { mx = fmaxf(mx, p[i]); nx = fmaxf(nx, p[8 + i]); }
        float sw = 0.f, sh = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { w[i] = __expf(p[i] - mx); sw += w[i]; h[i] = __expf(p[8 + i] - nx); sh += h[i]; }
        const float is = (1.0f - 1e-3f * 8) / sw, it = (1.0f - 1e-3f * 8) / sh;
#pragma unroll
        for (int i = 0; i < 8; ++i) { w[i] = fmaf(w[i], is, 1e-3f); h[i] = fmaf(h[i], it, 1e-3f); }
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
    float k_lo = cw[0], k_hi = cw[1], c_lo = chh[0], c_hi = chh[1];
#pragma unroll
    for (int i = 1; i < 8; ++i) {
        const bool sel = bin == i;
        k_lo = sel ? cw[i] : k_lo; k_hi = sel ? cw[i + 1] : k_hi; c_lo = sel ? chh[i] : c_lo; c_hi = sel ? chh[i + 1] : c_hi;
    }
    // ---- role B (meaningful in the odd group): the selected bin's raw derivatives and lambda, from the partner's bin ----
    const int lane_id = (int)(threadIdx.x & 63);
    const int pbin = __shfl(bin, lane_id ^ 16);
    float lamr = p[8], rd0 = 0.f, rd1 = p[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) {
        const bool sel = pbin == i;
        lamr = sel ? p[8 + i] : lamr; rd0 = sel ? p[i - 1] : rd0;
        if (i < 7) rd1 = sel ? p[i] : rd1;
    }
    float in_d = pbin == 0 ? 1.0f - 1e-3f : 1e-3f + softplus1(rd0);
    float in_d1 = pbin == 7 ? 1.0f - 1e-3f : 1e-3f + softplus1(rd1);
    float lam = (1.f - 2.f * 0.025f) * sigmoid1(lamr) + 0.025f;
    // back to the even group
    in_d = __shfl(in_d, lane_id ^ 16); in_d1 = __shfl(in_d1, lane_id ^ 16); lam = __shfl(lam, lane_id ^ 16);
    (void)g;
    const float in_w = k_hi - k_lo, in_h = c_hi - c_lo;
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

__global__ __launch_bounds__(kChainWaves * 64, 1) void k_flow_chain16(FlowDev d, const float* __restrict__ C0, NoiseSrc ns, int B, int N, int Mtot,
                                                                      float* __restrict__ rot_samples, float* __restrict__ rot_point,
                                                                      float* __restrict__ so3_point, long long* __restrict__ timeline) {
    __shared__ float sAnc[16 * kAncRow];
    __shared__ int sOff[kP * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j16 = lane & 15, g = lane >> 4;
    const int tile = blockIdx.x;
    // The base samples z = base_std * eps of all 23 parts x 16 rows, once per work-group: 1,104 generator elements over 256 threads (4-5 Philox
    // blocks + Box-Muller each) instead of 3 per lane and part inside the dependent rounds (timeline: ~2.4 k of a part's ~30 k cycles).
    __shared__ float sZ[kP * 16 * 3];
    for (int i = tid; i < 16 * kAncRow; i += kChainWaves * 64) sAnc[i] = 0.f;
    for (int i = tid; i < kP * 64; i += kChainWaves * 64) sOff[i] = d.anc_off[i];
    for (int i = tid; i < kP * 16 * 3; i += kChainWaves * 64) {
        const int jp = i / 48, rem = i - jp * 48, rw = rem / 3, c = rem - rw * 3;
        const RowInfo rr = row_info(tile * 16 + rw, B, N, Mtot);
        float zv = 0.f;
        if (rr.live && !rr.is_point) zv = pose_noise(ns, jp, (unsigned int)((rr.b * N + rr.n) * 3 + c), (unsigned long long)B * N * 3) * d.base_std;
        sZ[i] = zv;
    }
    __syncthreads();
    const int rowid = tile * 16 + j16;
    const RowInfo ri = row_info(rowid, B, N, Mtot);
    const float bound = d.radius;
    const float one_g0 = g == 0 ? 1.f : 0.f;
    float* myAnc = sAnc + j16 * kAncRow;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};

    // Global operands of a part's first stages: requested for the wave's part of round r + 1 BEFORE round r's tanh / exp / store phase and the
    // round barrier (timeline: a round used to start with 6-8 k cycles of load issue + first-use latency; the registers are free by then).
    f32x4 cx[4], wa[4][4], w1[8][4];
    float w1e[8];
    TailRegs16 tr0, tr1;
    auto load_part = [&](int j) {
        const PartTab& pt = d.tab[j];
        const float* src = C0 + ((size_t)rowid * kP + j) * kCtx + 4 * g;     // context accumulators <- C0 (image/shape half, k_flow_ctx0): channel 16 t + 4 g + r
#pragma unroll
        for (int t = 0; t < 4; ++t) cx[t] = *(const f32x4*)(src + 16 * t);
        const int kag = pt.kag16;
        const f32x4* wp = (const f32x4*)(d.arena + pt.o16_wa) + lane;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
                if (s4 < kag) wa[t][s4] = wp[(t * kag + s4) * 64];
        const f32x4* w1p = (const f32x4*)(d.arena + pt.o16_w1) + lane;      // first layer of both couplings: 8 tiles x 16 k-steps
#pragma unroll
        for (int to = 0; to < 8; ++to)
#pragma unroll
            for (int i = 0; i < 4; ++i) w1[to][i] = w1p[(to * 4 + i) * 64];
#pragma unroll
        for (int to = 0; to < 8; ++to) w1e[to] = d.arena[pt.o16_w1e + to * 64 + lane];
        tail_fetch16(d, pt, 0, lane, tr0);
    };
    auto part_of = [&](int lev, int& j) -> bool {
        if (lev >= d.lev.nlev) return false;
        const int pi = d.lev.start[lev] + wave;
        if (pi >= d.lev.start[lev + 1]) return false;
        j = d.lev.parts[pi];
        return true;
    };
    int j = 0, jn = 0;
    bool have = part_of(0, j);
    if (have) load_part(j);

    for (int lev = 0; lev < d.lev.nlev; ++lev) {
        const bool have_n = part_of(lev + 1, jn);
        if (have) {
            const PartTab& pt = d.tab[j];
            // diagnostic timeline (hmf_debug_set_timeline, experiments build): shader-clock stamps of tile 0's phases, 8 per part
            long long* tl = (timeline && tile == 0 && lane == 0) ? timeline + j * 8 : nullptr;
#define HMF_STAMP(i) do { if (tl) tl[i] = (long long)__builtin_readcyclecounter(); } while (0)
            HMF_STAMP(0);
            // ================= stage 0: ancestor entries 4 s + g of this lane's row (B operands of the ancestor half), base sample =================
            float bv[16];
            {
                const int* off = sOff + j * 64 + g;
#pragma unroll
                for (int s = 0; s < 16; ++s) bv[s] = myAnc[off[4 * s]];
            }
            const int kag = pt.kag16;
            // base sample z = base_std * eps; point-estimate rows push z = 0
            const float x0 = sZ[(j * 16 + j16) * 3], x1 = sZ[(j * 16 + j16) * 3 + 1], x2 = sZ[(j * 16 + j16) * 3 + 2];
            __builtin_amdgcn_sched_barrier(0);
            HMF_STAMP(1);
            // ================= stage 1: ancestor half of the context layer, ELU =================
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                if (s4 < kag) {                                           // wave-uniform
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int t = 0; t < 4; ++t) cx[t] = mfma16(wa[t][s4][u], bv[s4 * 4 + u], cx[t]);
                }
            }
            tail_fetch16(d, pt, 1, lane, tr1);
            __builtin_amdgcn_sched_barrier(0);
            HMF_STAMP(2);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) cx[t][e] = elu_fast(cx[t][e]);
            HMF_STAMP(3);
            // ================= stage 2: first hyper-network layer of both couplings =================
            // extra k-step: (x1 column | bias | 0 | 0) . (x0 | 1 | 0 | 0) for coupling 0 now; coupling 1's needs y0 and comes later
            const float bx0 = g == 0 ? x0 : (g == 1 ? 1.f : 0.f);
            f32x4 a1[8];
#pragma unroll
            for (int to = 0; to < 4; ++to) a1[to] = mfma16(w1e[to], bx0, z4);
#pragma unroll
            for (int to = 4; to < 8; ++to) a1[to] = z4;
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int to = 0; to < 8; ++to) a1[to] = mfma16(w1[to][s >> 2][s & 3], cx[s >> 2][s & 3], a1[to]);
            HMF_STAMP(4);
            // ================= stage 3: coupling 0 (Permute [0,1,2]): x0 passes, the lane's coordinate x_{1 + (g >> 1)} is transformed =================
            f32x4 h1[4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) h1[t][e] = fmaxf(a1[t][e], 0.f);
            const float xn = tail_compute16(tr0, h1, (g >> 1) ? x2 : x1, bound, one_g0, g);
            const float x1n = __shfl(xn, j16), x2n = __shfl(xn, 32 + j16);
            HMF_STAMP(5);
            // ================= stage 4: coupling 1 (Permute [1,2,0]): y = (x1', x2', x0); y0 passes =================
            const float y0 = x1n;
            const float by0 = g == 0 ? y0 : (g == 1 ? 1.f : 0.f);
#pragma unroll
            for (int to = 4; to < 8; ++to) a1[to] = mfma16(w1e[to], by0, a1[to]);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) h1[t][e] = fmaxf(a1[4 + t][e], 0.f);
            const float yn = tail_compute16(tr1, h1, (g >> 1) ? x0 : x2n, bound, one_g0, g);
            const float y1 = __shfl(yn, j16), y2 = __shfl(yn, 32 + j16);
            HMF_STAMP(6);
            if (have_n) load_part(jn);                      // next round's operands: in flight during stage 5 and the barrier
            // ================= stage 5: ScaledRadialTanh (scaled_radial_tanh_transform.py:28-35) + exp map =================
            float v[3] = {y0, y1, y2};
            {
                const float nrm = sqrtf(y0 * y0 + y1 * y1 + y2 * y2);
                if (nrm > 1e-7f) {
                    const float tt = tanhf(nrm / bound);
                    v[0] = tt * (y0 / nrm) * bound; v[1] = tt * (y1 / nrm) * bound; v[2] = tt * (y2 / nrm) * bound;
                }
            }
            if (g == 0) {
                // fp64 Rodrigues rounded to fp32 for samples (:311), smplx Rodrigues for point rows (:299)
                float R[9];
                if (ri.is_point) {
                    batch_rodrigues_f32(v, R);
                } else {
                    const double vd[3] = {(double)v[0], (double)v[1], (double)v[2]};
                    double Rd[9];
                    so3_exp_f64(vd, Rd);
#pragma unroll
                    for (int e = 0; e < 9; ++e) R[e] = (float)Rd[e];
                }
#pragma unroll
                for (int e = 0; e < 9; ++e) myAnc[j * 9 + e] = R[e];
                if (ri.live) {
                    if (ri.is_point) {
                        float* o = rot_point + ((size_t)ri.b * kP + j) * 9;
#pragma unroll
                        for (int e = 0; e < 9; ++e) o[e] = R[e];
                        float* a = so3_point + ((size_t)ri.b * kP + j) * 3;
                        a[0] = v[0]; a[1] = v[1]; a[2] = v[2];
                    } else {
                        float* o = rot_samples + ((size_t)rowid * kP + j) * 9;
#pragma unroll
                        for (int e = 0; e < 9; ++e) o[e] = R[e];
                    }
                }
            }
            HMF_STAMP(7);
#undef HMF_STAMP
        } else if (have_n) {
            load_part(jn);
        }
        have = have_n;
        j = jn;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// teacher-forced side (B rows only): contexts from given rotations, then log-probabilities.
// One wave per (row, part); lane = output channel; inputs broadcast with readlane.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_flow_context(FlowDev d, const float* __restrict__ isgc, const float* __restrict__ pose_R,
                                                      int B, float* __restrict__ ctx) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gw >= B * kP) return;
    const int b = gw / kP, j = gw % kP;
    const PartTab& pt = d.tab[j];
    const float* WT = d.arena + pt.o_wctxT;
    float acc = d.arena[pt.o_bctx + lane];
    const float* x = isgc + (size_t)b * kIsgc;
    for (int k = 0; k < kIsgc; ++k) acc = fmaf(WT[(size_t)k * kCtx + lane], x[k], acc);
    for (int a = 0; a < pt.n_anc; ++a) {
        const float* r = pose_R + ((size_t)b * kP + pt.anc[a]) * 9;
        for (int e = 0; e < 9; ++e) acc = fmaf(WT[(size_t)(kIsgc + a * 9 + e) * kCtx + lane], r[e], acc);
    }
    ctx[((size_t)b * kP + j) * kCtx + lane] = elu1(acc);
}

// hyper-network of coupling c evaluated on (ctx, x1) for one (row, part) by one wave -> raw[lane] (62 valid)
__device__ __forceinline__ float hypernet(const FlowDev& d, const PartTab& pt, int c, int lane, float ctx_lane, float x1) {
    const float* W = d.arena + pt.o_nnT[c][0];
    float a = d.arena[pt.o_nnb[c][0] + lane];
    for (int k = 0; k < 64; ++k) a = fmaf(W[k * 64 + lane], __shfl(ctx_lane, k), a);
    a = fmaf(W[64 * 64 + lane], x1, a);
    float h = fmaxf(a, 0.f);
    W = d.arena + pt.o_nnT[c][1];
    a = d.arena[pt.o_nnb[c][1] + lane];
    for (int k = 0; k < 64; ++k) a = fmaf(W[k * 64 + lane], __shfl(h, k), a);
    h = fmaxf(a, 0.f);                       // lanes >= 32 hold relu(0 + 0) = 0 (padded columns are zero)
    W = d.arena + pt.o_nnT[c][2];
    a = d.arena[pt.o_nnb[c][2] + lane];
    for (int k = 0; k < 32; ++k) a = fmaf(W[k * 64 + lane], __shfl(h, k), a);
    h = fmaxf(a, 0.f);
    W = d.arena + pt.o_nnT[c][3];
    a = d.arena[pt.o_nnb[c][3] + lane];
    for (int k = 0; k < 32; ++k) a = fmaf(W[k * 64 + lane], __shfl(h, k), a);
    return a;
}

// inverse spline of coupling c on the two transformed coordinates; every lane computes dimension (lane & 1)
__device__ __forceinline__ void coupling_inverse(const FlowDev& d, const PartTab& pt, int c, int lane, float ctx_lane,
                                                 float y1, float& z2a, float& z2b, float& ld_sum) {
    const float raw = hypernet(d, pt, c, lane, ctx_lane, y1);
    const int dim = lane & 1;
    float rw[8], rh[8], rd[7], rl[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        rw[i] = __shfl(raw, dim * 8 + i);
        rh[i] = __shfl(raw, 16 + dim * 8 + i);
        rl[i] = __shfl(raw, 46 + dim * 8 + i);
    }
#pragma unroll
    for (int i = 0; i < 7; ++i) rd[i] = __shfl(raw, 32 + dim * 7 + i);
    float ld;
    const float in = dim ? z2b : z2a;
    const float out = rational_linear_spline<true>(in, rw, rh, rd, rl, d.radius, &ld);
    z2a = __shfl(out, 0); z2b = __shfl(out, 1);
    ld_sum = __shfl(ld, 0) + __shfl(ld, 1);
}

// log p(v | ctx) on the algebra: torch TransformedDistribution.log_prob through
// [Permute, Spline, Permute, Spline, ScaledRadialTanh] (pyro_conditional_norm_flow.py:55-113)
__device__ __forceinline__ float log_prob_so3(const FlowDev& d, const PartTab& pt, int lane, float ctx_lane, const float (&v)[3]) {
    const float r = d.radius;
    // ScaledRadialTanh inverse in fp64 (scaled_radial_tanh_transform.py:37-46) and its log-det in fp32 (:48-59)
    float x[3] = {v[0], v[1], v[2]};
    {
        const double yd[3] = {(double)v[0], (double)v[1], (double)v[2]};
        const double n = sqrt(yd[0] * yd[0] + yd[1] * yd[1] + yd[2] * yd[2]);
        if (n > 1e-7) {
            const double s = atanh(n / (double)r);
            x[0] = (float)(s * (yd[0] / n) * (double)r); x[1] = (float)(s * (yd[1] / n) * (double)r); x[2] = (float)(s * (yd[2] / n) * (double)r);
        }
    }
    const float yn = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    const float xn = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    float lp = 0.f;
    if (yn > 1e-7f) lp = -(2.f * (logf(yn) - logf(xn)) + log1pf(-((yn / r) * (yn / r))));
    // coupling 1 inverse (log_abs_det_jacobian = -sum(inverse log-dets)), then Permute([1,2,0]) inverse
    float a = x[1], b = x[2], ld;
    coupling_inverse(d, pt, 1, lane, ctx_lane, x[0], a, b, ld);
    lp += ld;
    const float u0 = b, u1 = x[0], u2 = a;          // x_pre = y[argsort(perm)] = (y2, y0, y1)
    // coupling 0 inverse, Permute([0,1,2]) is the identity
    a = u1; b = u2;
    coupling_inverse(d, pt, 0, lane, ctx_lane, u0, a, b, ld);
    lp += ld;
    const float z[3] = {u0, a, b};
    // Independent(Normal(0, base_std)) log-density
    const float var = d.base_std * d.base_std, lsc = logf(d.base_std), hl2pi = 0.91893853320467274178f;
    float base = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) base += -(z[i] * z[i]) / (2.f * var) - lsc - hl2pi;
    return lp + base;
}

// algebra density of v (fp32, B x P x 3): one wave per (row, part)
__global__ __launch_bounds__(256) void k_flow_logprob(FlowDev d, const float* __restrict__ ctx, const int* __restrict__ part_ids,
                                                      int P, const float* __restrict__ vin, int B, float* __restrict__ logp) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gw >= B * P) return;
    const int p = gw % P;
    const int j = part_ids ? part_ids[p] : p;
    const PartTab& pt = d.tab[j];
    const float ctx_lane = ctx[(size_t)gw * kCtx + lane];
    const float v[3] = {vin[(size_t)gw * 3], vin[(size_t)gw * 3 + 1], vin[(size_t)gw * 3 + 2]};
    const float lp = log_prob_so3(d, pt, lane, ctx_lane, v);
    if (lane == 0) logp[gw] = lp;
}

// SO(3) density of R (fp64, B x P x 9), candidate-parallel form (round 5): one work-group of THREE waves per (row, part), wave c = pre-image c of
// so3_xset (utils/rigid_transform_utils.py:282-295) with k = 0, -1, +1 and the support mask (so3_exp_transform.py:36-41; NaN at x = 0 fails the
// comparison, as in the reference, quirk 2).  The single-wave form of rounds 1-4 walked the three pre-images one after the other (three dependent passes of the two hyper-networks: ~200 us at B = 72 whatever the grid,
// since every wave is resident at once and the launch lasts one wave's lifetime); here a wave evaluates one, masked pre-images (outside the support:
// usually two of the three) skip the evaluation and wait at the barrier, and the log-sum-exp meets in LDS.  Same arithmetic per term, same sum order.
__global__ __launch_bounds__(192) void k_flow_logprob_SO3_par(FlowDev d, const float* __restrict__ ctx, const int* __restrict__ part_ids, int P,
                                                              const double* __restrict__ Rin, float* __restrict__ logp) {
    __shared__ float s_term[3];
    const int lane = threadIdx.x & 63, c = threadIdx.x >> 6;
    const int gw = blockIdx.x;
    const int p = gw % P;
    const int j = part_ids ? part_ids[p] : p;
    const PartTab& pt = d.tab[j];
    double R[9], x[3];
#pragma unroll
    for (int e = 0; e < 9; ++e) R[e] = Rin[(size_t)gw * 9 + e];
    so3_log_f64(R, x);
    const double kTwoPi = 6.28318530717958647692;
    const double n = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    double cand[3] = {x[0], x[1], x[2]};
    bool ok = true;
    if (c > 0) {
        const double sc = n + kTwoPi * (c == 1 ? -1.0 : 1.0);
        cand[0] = x[0] / n * sc; cand[1] = x[1] / n * sc; cand[2] = x[2] / n * sc;
        ok = sqrt(cand[0] * cand[0] + cand[1] * cand[1] + cand[2] * cand[2]) < (double)d.radius;
    }
    // `ok` is wave-uniform.  A masked pre-image (outside the support: usually two of the three) skips both hyper-network passes but stays alive
    // to the barrier: returning before it relied on ended waves dropping out of s_barrier, which the HIP model does not promise.
    float term = -INFINITY;
    if (ok) {
        const float ctx_lane = ctx[(size_t)gw * kCtx + lane];
        const float ld = -(float)so3_logdet_f64(cand);
        const float v[3] = {(float)cand[0], (float)cand[1], (float)cand[2]};
        term = ld + log_prob_so3(d, pt, lane, ctx_lane, v);
    }
    if (lane == 0) s_term[c] = term;
    __syncthreads();
    if (c == 0 && lane == 0) {
        const float t0 = s_term[0], t1 = s_term[1], t2 = s_term[2];
        const float m = fmaxf(t0, fmaxf(t1, t2));
        logp[gw] = m + logf(expf(t0 - m) + expf(t1 - m) + expf(t2 - m));
    }
}

}  // namespace hmf

namespace hmf {
// flow_generic.hip: the non-default flow variants
int generic_forward(const hmf_flow* h, const float* G, const float* shape_mode, const float* shape_log_std, const NoiseSrc& ns,
                    int B, int N, int Mtot, int packed_rows, float* shape_samples, float* rot_samples, float* rot_point, float* so3_point, hipStream_t st);
int generic_logprob(const hmf_flow* h, const float* ctx, const int* dev_ids, int P, const double* R, const float* v, int B, int mode, float* logp, hipStream_t st);
}  // namespace hmf

using namespace hmf;

static long long* g_timeline = nullptr;
#ifdef HMF_EXPERIMENTS
static int g_flow_whatif = 0;
#endif

extern "C" {

#ifdef HMF_EXPERIMENTS
/* diagnostic (tools/flow_timeline.py): device buffer of 23 x 8 int64 shader-clock stamps written by tile 0, or NULL */
int hmf_debug_set_timeline(void* dev_ptr) { g_timeline = (long long*)dev_ptr; return HMF_OK; }
int hmf_debug_set_flow_whatif(int mask) { g_flow_whatif = mask; return HMF_OK; }
#endif

static int n_tiles(int B, int N, int flags) { return ((B * N + ((flags & HMF_FLOW_POINT_EST) ? B : 0)) + 31) / 32; }

size_t hmf_flow_forward_workspace_bytes(const hmf_flow_t* h, int B, int N, int flags) {
    (void)h;
    const size_t g = align_up((size_t)std::max(B, 1) * kIsgc * 4, 256);
    const size_t c0 = (size_t)std::max(n_tiles(B, N, flags), 1) * 32 * kP * kCtx * 4;          // [rows padded to 32][part][64]
    return g + c0 + 256;                 // + the pass' RngPass snapshot
}

unsigned long long hmf_flow_rng_increment(int B, int N, int num_betas, int draw_shape, unsigned int cap_blocks) {
    if (B <= 0 || N <= 0) return 0;
    return (draw_shape ? rng_call_increment((unsigned long long)N * B * num_betas, cap_blocks) : 0ull) +
           (unsigned long long)kP * rng_call_increment((unsigned long long)B * N * 3, cap_blocks);
}

int hmf_flow_forward(const hmf_flow_t* h, const float* feats, const float* glob_R, const float* cam, const float* shape_mode,
                     const float* shape_log_std, const float* eps_shape, const float* eps_pose, const hmf_rng_t* rng, int B, int N,
                     int flags, float* shape_samples, float* rotmats_samples, float* rotmats_point, float* so3_point,
                     void* workspace, size_t workspace_bytes, void* stream) {
    if (!h || !feats || !glob_R || !cam || !shape_mode) return fail(HMF_ERR_INVALID, "hmf_flow_forward: null input");
    if (B <= 0 || N < 0) return fail(HMF_ERR_INVALID, "hmf_flow_forward: B=%d N=%d", B, N);
    if (N > 0 && !rotmats_samples) return fail(HMF_ERR_INVALID, "hmf_flow_forward: N>0 needs rotmats_samples");
    if (N > 0 && !eps_pose && !rng) return fail(HMF_ERR_INVALID, "hmf_flow_forward: N>0 needs eps_pose or rng (the base noise: injected or drawn in the kernels)");
    const int shape_mode_samples = (flags & HMF_FLOW_SHAPE_MODE) ? 1 : 0;
    if (shape_mode_samples) eps_shape = nullptr;
    // shape noise: injected (eps_shape), drawn (rng given, no eps_shape, not HMF_FLOW_SHAPE_MODE), or none (use_shape_mode_for_samples)
    const int draw_shape = (N > 0 && rng && !eps_shape && !shape_mode_samples) ? 1 : 0;
    const int draw_pose = (N > 0 && !eps_pose) ? 1 : 0;
    if (N > 0 && (eps_shape || draw_shape) && !shape_log_std) return fail(HMF_ERR_INVALID, "hmf_flow_forward: shape noise needs shape_log_std");
    const int compute_point_est = flags & HMF_FLOW_POINT_EST;
    const int packed_rows = (flags & HMF_FLOW_PACKED_ROWS) ? 1 : 0;
    if (compute_point_est && (!rotmats_point || !so3_point)) return fail(HMF_ERR_INVALID, "hmf_flow_forward: point-estimate outputs missing");
    if (packed_rows && (!compute_point_est || !shape_samples)) return fail(HMF_ERR_INVALID, "hmf_flow_forward: HMF_FLOW_PACKED_ROWS needs the point estimate and shape_samples");
    if (N == 0 && !compute_point_est) return HMF_OK;
    if ((long long)B * N + B > (1LL << 30)) return fail(HMF_ERR_INVALID, "hmf_flow_forward: B*N too large");
    if ((draw_shape || draw_pose) && ((long long)B * N * std::max(h->d.nb, 3) >= (1LL << 32) || rng->grid_cap_blocks < 1))
        return fail(HMF_ERR_INVALID, "hmf_flow_forward: in-kernel noise needs B*N*max(nb,3) < 2^32 and rng->grid_cap_blocks >= 1");
    const size_t need = hmf_flow_forward_workspace_bytes(h, B, N, flags);
    if (!workspace || workspace_bytes < need) return fail(HMF_ERR_WORKSPACE, "hmf_flow_forward: workspace %zu < %zu bytes", workspace_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    float* G = (float*)workspace;
    float* C0 = (float*)((char*)workspace + align_up((size_t)B * kIsgc * 4, 256));
    const int Mtot = B * N + (compute_point_est ? B : 0);
    const int ntiles = (Mtot + 31) / 32;
    RngPass* pass = (RngPass*)((char*)workspace + need - 256);
    NoiseSrc ns{};
    ns.eps_shape = eps_shape; ns.eps_pose = eps_pose;
    RngPass imm{};
    unsigned long long inc = 0;
    const bool draws = draw_shape || draw_pose;
    if (draws) {
        const unsigned int cap = (unsigned int)rng->grid_cap_blocks;
        ns.pass = pass;
        ns.draw_shape = draw_shape;
        ns.T_shape = rng_threads((unsigned long long)N * B * h->d.nb, cap);
        ns.T_pose = rng_threads((unsigned long long)B * N * 3, cap);
        ns.inc_shape = rng_call_increment((unsigned long long)N * B * h->d.nb, cap);
        ns.inc_pose = rng_call_increment((unsigned long long)B * N * 3, cap);
        imm.seed = rng->seed; imm.offset = rng->offset; imm.cap_blocks = cap;
        if (rng->offset & 3) return fail(HMF_ERR_INVALID, "hmf_flow_forward: rng->offset must be a multiple of 4 (as the device generator keeps it)");
        inc = (draw_shape ? ns.inc_shape : 0ull) + (draw_pose ? (unsigned long long)kP * ns.inc_pose : 0ull);
    }
    {
        const int nw = kIsgc * ((B + 7) / 8);
        ProfScope ps(7, B, st);
        hipLaunchKernelGGL(k_isgc, dim3((nw + 3) / 4), dim3(256), 0, st, h->d, feats, (const float*)nullptr, glob_R, cam, B, 0, G, imm,
                           draws ? (unsigned long long*)rng->state_dev : (unsigned long long*)nullptr, inc, draws ? pass : (RngPass*)nullptr,
                           draws ? (unsigned long long*)rng->mirror_dev : (unsigned long long*)nullptr);
        HMF_LAUNCH_CHECK();
    }
    if (h->g.enabled)
        return generic_forward(h, G, shape_mode, shape_log_std, ns, B, N, Mtot, packed_rows, shape_samples, rotmats_samples, rotmats_point,
                               so3_point, st);
    {
        ProfScope ps(5, Mtot, st);
        if (ntiles <= 16)
            hipLaunchKernelGGL(k_flow_ctx0<1>, dim3(ntiles * ctx0_groups(1)), dim3(256), 0, st, h->d, G, shape_mode, shape_log_std, ns, B, N, Mtot,
                               ntiles, packed_rows, shape_samples, C0);
        else
            hipLaunchKernelGGL(k_flow_ctx0<3>, dim3(ntiles * ctx0_groups(3)), dim3(256), 0, st, h->d, G, shape_mode, shape_log_std, ns, B, N, Mtot,
                               ntiles, packed_rows, shape_samples, C0);
        HMF_LAUNCH_CHECK();
    }
    {
        ProfScope ps(6, Mtot, st);
        // Two forms of the same loop.  16-row tiles (k_flow_chain16) finish a tile in 0.73x the time -- 117 vs 159 us alone at B = 32, N = 100, and
        // for a single image -- but hold twice the SIMDs (a wave of either kernel owns its SIMD's register file): 808 x 117 us = 94 k SIMD-us against
        // 404 x 159 us = 64 k.  With other batches' encoders co-running (bench.py: three streams) the step went 3.14 -> 3.18 ms with the 16-row form.
        // So: the latency form up to 1,024 rows (predict: one image x 10 samples; small evaluation batches), the 32-row form beyond;
        // HMF_FLOW_CHAIN32 / HMF_FLOW_CHAIN16 force one (cross-checks, tools/flow_timeline.py).
        const bool use32 = (flags & HMF_FLOW_CHAIN32) || (!(flags & HMF_FLOW_CHAIN16) && Mtot > 1024);
#ifdef HMF_EXPERIMENTS
        if (g_flow_whatif & 1) {                               // tools/whatif_1x1.py: the step WITHOUT the kinematic loop (wrong results): the ceiling of any faster chain kernel
        } else
#endif
        if (use32)
            hipLaunchKernelGGL(k_flow_chain, dim3(ntiles), dim3(kChainWaves * 64), 0, st, h->d, C0, ns, B, N, Mtot, rotmats_samples,
                               rotmats_point, so3_point, g_timeline);
        else
            hipLaunchKernelGGL(k_flow_chain16, dim3((Mtot + 15) / 16), dim3(kChainWaves * 64), 0, st, h->d, C0, ns, B, N, Mtot, rotmats_samples,
                               rotmats_point, so3_point, g_timeline);
    }
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

int hmf_flow_draw_noise(const hmf_rng_t* rng, int B, int N, int num_betas, float* eps_shape, float* eps_pose, void* workspace, size_t workspace_bytes,
                        void* stream) {
    if (!rng || !eps_pose || B <= 0 || N <= 0 || num_betas < 1 || num_betas > 16) return fail(HMF_ERR_INVALID, "hmf_flow_draw_noise: bad args");
    if (!workspace || workspace_bytes < 256) return fail(HMF_ERR_WORKSPACE, "hmf_flow_draw_noise: workspace of >= 256 bytes");
    if ((long long)B * N * 3 * kP >= (1LL << 32) || (long long)B * N * num_betas >= (1LL << 31) || rng->grid_cap_blocks < 1 || (rng->offset & 3))
        return fail(HMF_ERR_INVALID, "hmf_flow_draw_noise: B*N too large, grid_cap_blocks < 1 or offset not a multiple of 4");
    const unsigned int cap = (unsigned int)rng->grid_cap_blocks;
    NoiseSrc ns{};
    ns.pass = (RngPass*)workspace;
    ns.draw_shape = eps_shape ? 1 : 0;
    ns.T_shape = rng_threads((unsigned long long)N * B * num_betas, cap);
    ns.T_pose = rng_threads((unsigned long long)B * N * 3, cap);
    ns.inc_shape = rng_call_increment((unsigned long long)N * B * num_betas, cap);
    ns.inc_pose = rng_call_increment((unsigned long long)B * N * 3, cap);
    RngPass imm{};
    imm.seed = rng->seed; imm.offset = rng->offset; imm.cap_blocks = cap;
    const unsigned long long inc = (ns.draw_shape ? ns.inc_shape : 0ull) + (unsigned long long)kP * ns.inc_pose;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_rng_snapshot, dim3(1), dim3(1), 0, st, imm, (unsigned long long*)rng->state_dev, inc, (RngPass*)workspace,
                       (unsigned long long*)rng->mirror_dev);
    HMF_LAUNCH_CHECK();
    const unsigned int n_shape = ns.draw_shape ? (unsigned int)(N * B * num_betas) : 0u, n_part = (unsigned int)(B * N * 3);
    const unsigned long long total = (unsigned long long)n_shape + (unsigned long long)kP * n_part;
    hipLaunchKernelGGL(k_draw_noise, dim3((unsigned int)((total + 255) / 256)), dim3(256), 0, st, ns, n_shape, n_part, eps_shape, eps_pose);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

int hmf_flow_isgc(const hmf_flow_t* h, const float* feats, const float* shape, const float* glob_R, const float* cam, int B,
                  float* isgc, void* stream) {
    if (!h || !feats || !shape || !glob_R || !cam || !isgc || B <= 0) return fail(HMF_ERR_INVALID, "hmf_flow_isgc: bad args");
    const int nw = kIsgc * ((B + 7) / 8);
    hipLaunchKernelGGL(k_isgc, dim3((nw + 3) / 4), dim3(256), 0, (hipStream_t)stream, h->d, feats, shape, glob_R, cam, B, 1, isgc, RngPass{},
                       (unsigned long long*)nullptr, 0ull, (RngPass*)nullptr, (unsigned long long*)nullptr);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

int hmf_flow_isgc_train(const hmf_flow_t* h, const float* feats, const float* glob_R, const float* glob_R_ll, const float* cam, const float* shape_rows,
                        const float* shape_ll, int B, int N, float* pre, float* H, void* stream) {
    if (!h || !feats || !glob_R || !glob_R_ll || !cam || !pre || !H || B <= 0 || N < 1 || !shape_rows || !shape_ll)
        return fail(HMF_ERR_INVALID, "hmf_flow_isgc_train: bad args");
    hipStream_t st = (hipStream_t)stream;
    const int nw = kIsgc * ((B + 7) / 8);
    for (int u = 0; u < 2; ++u) {
        hipLaunchKernelGGL(k_isgc, dim3((nw + 3) / 4), dim3(256), 0, st, h->d, feats, (const float*)nullptr, u ? glob_R_ll : glob_R, cam, B, 0,
                           pre + (size_t)u * B * kIsgc, RngPass{}, (unsigned long long*)nullptr, 0ull, (RngPass*)nullptr, (unsigned long long*)nullptr);
        HMF_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_isgc_rows, dim3(B * N + B), dim3(256), 0, st, h->d, pre, shape_rows, shape_ll, B, N, H);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

int hmf_flow_isgc_train_backward(const hmf_flow_t* h, const float* H, const float* g_H_rows, const float* g_H_ll, int B, int N, float* delta, float* g_pre_f,
                                 float* g_pre_f_t, float* g_pre_s_t, float* g_shape, void* stream) {
    if (!h || !H || !delta || !g_pre_f || !g_pre_f_t || !g_pre_s_t || !g_shape || B <= 0 || N < 1)
        return fail(HMF_ERR_INVALID, "hmf_flow_isgc_train_backward: bad args");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_isgc_bwd_delta, dim3(B), dim3(256), 0, st, H, g_H_rows, g_H_ll, B, N, delta, g_pre_f, g_pre_f_t, g_pre_s_t);
    HMF_LAUNCH_CHECK();
    const int rows = B * N + B, n = rows * h->d.nb;
    hipLaunchKernelGGL(k_isgc_bwd_shape, dim3((n + 255) / 256), dim3(256), 0, st, h->d, delta, rows, g_shape);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

int hmf_flow_context(const hmf_flow_t* h, const float* isgc, const float* pose_R, int B, float* ctx, void* stream) {
    if (!h || !isgc || !pose_R || !ctx || B <= 0) return fail(HMF_ERR_INVALID, "hmf_flow_context: bad args");
    const int nw = B * kP;
    hipLaunchKernelGGL(k_flow_context, dim3((nw + 3) / 4), dim3(256), 0, (hipStream_t)stream, h->d, isgc, pose_R, B, ctx);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

static int logprob_common(const hmf_flow_t* h, const float* ctx, const int32_t* part_ids_host, int P, const double* R, const float* v,
                          int B, int mode, float* logp, void* stream) {
    if (!h || !ctx || !logp || B <= 0 || P <= 0 || P > kP) return fail(HMF_ERR_INVALID, "hmf_flow_log_prob: bad args");
    if (!part_ids_host && P != kP) return fail(HMF_ERR_INVALID, "hmf_flow_log_prob: part_ids NULL requires P == 23");
    const int* dev_ids = nullptr;
    if (part_ids_host) {
        for (int i = 0; i < P; ++i)
            if (part_ids_host[i] < 0 || part_ids_host[i] >= kP) return fail(HMF_ERR_INVALID, "hmf_flow_log_prob: part id %d", part_ids_host[i]);
        for (int i = 1; i < P; ++i)
            if (part_ids_host[i] != part_ids_host[0] + i)
                return fail(HMF_ERR_UNSUPPORTED, "hmf_flow_log_prob: part ids must be a contiguous ascending range");
        dev_ids = h->d_ids + part_ids_host[0];   // identity table 0..22 on the device
    }
    if (h->g.enabled) return generic_logprob(h, ctx, dev_ids, P, R, v, B, mode, logp, (hipStream_t)stream);
    const int nw = B * P;
    if (mode == 0) {
        hipLaunchKernelGGL(k_flow_logprob_SO3_par, dim3(nw), dim3(192), 0, (hipStream_t)stream, h->d, ctx, dev_ids, P, R, logp);
        HMF_LAUNCH_CHECK();
        return HMF_OK;
    }
    hipLaunchKernelGGL(k_flow_logprob, dim3((nw + 3) / 4), dim3(256), 0, (hipStream_t)stream, h->d, ctx, dev_ids, P, v, B, logp);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

int hmf_flow_log_prob_SO3(const hmf_flow_t* h, const float* ctx, const int32_t* part_ids_host, int P, const double* R, int B,
                          float* logp, void* stream) {
    if (!R) return fail(HMF_ERR_INVALID, "hmf_flow_log_prob_SO3: null R");
    return logprob_common(h, ctx, part_ids_host, P, R, nullptr, B, 0, logp, stream);
}

int hmf_flow_log_prob_so3(const hmf_flow_t* h, const float* ctx, const int32_t* part_ids_host, int P, const float* v, int B,
                          float* logp, void* stream) {
    if (!v) return fail(HMF_ERR_INVALID, "hmf_flow_log_prob_so3: null v");
    return logprob_common(h, ctx, part_ids_host, P, nullptr, v, B, 1, logp, stream);
}

}  // extern "C"
