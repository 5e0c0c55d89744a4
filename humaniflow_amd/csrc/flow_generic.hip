// Non-default flow variants: every configuration create_conditional_norm_flow can build from the options HumaniflowModel
// passes (models/norm_flows/pyro_conditional_norm_flow.py:21-117, models/humaniflow_model.py:92-104) that is not the default:
//   PERMUTE_TYPE   'permute' | 'linear_plu' | 'conditional_linear_plu'      (transforms/linearPLU_transform.py:12-150)
//   TRANSFORM_TYPE 'spline_coupling' | 'additive_coupling'                  (transforms/additive_coupling_transform.py:12-85)
//   NUM_TRANSFORMS 1..8, TRANSFORM_NN_HIDDEN_DIMS / PERMUTE_NN_HIDDEN_DIMS: 1..4 layers of <= 64 channels.
// The default configuration has its own MFMA kernels (flow_kernels.hip); these variants are rarely-used ablations of the reference, so
// they run on simple lane-per-channel kernels: one wave per sample row walks the 23 parts in index order (an ancestor always has a smaller
// index), the row's rotations live in LDS, every dense layer is W^T [K][64] with the inputs broadcast by shuffles.
// Note on 'additive_coupling': in the reference this option raises TypeError at the first .condition() (conditional_additive_coupling
// forwards bound= / count_bins= into AdditiveCoupling.__init__, additive_coupling_transform.py:80-85,119-137); the arithmetic of the
// transform itself (_call/_inverse/log_abs_det_jacobian, :28-66) is what is built here.
#include "flow_common.h"
#include "flow_handle.h"
#include "so3_math.h"
#include "rng_philox.h"
#include "flow_grad_common.h"

namespace hmf {

// dense network, lane = output channel.  in_lane: lane k holds input k (k < 64); `extra` = input 64 (the coupling's x1), used when K[0] == 65
__device__ __forceinline__ float gen_net(const float* __restrict__ arena, const GenNet& n, int lane, float in_lane, float extra) {
    float h = in_lane, a = 0.f;
    for (int l = 0; l < n.nl; ++l) {
        const float* W = arena + n.o_w[l];
        a = arena[n.o_b[l] + lane];
        const int K = n.K[l], kk = K > 64 ? 64 : K;
        for (int k = 0; k < kk; ++k) a = fmaf(W[k * 64 + lane], __shfl(h, k), a);
        if (K > 64) a = fmaf(W[64 * 64 + lane], extra, a);
        h = (lane < n.N[l]) ? fmaxf(a, 0.f) : 0.f;
    }
    return a;                                                   // last layer: no activation
}

// torch.nn.functional.softplus(x, beta=0.75) (threshold 20 on beta * x), linearPLU_transform.py:146-147
__device__ __forceinline__ float softplus_b075(float x) { return 0.75f * x > 20.f ? x : log1pf(expf(0.75f * x)) / 0.75f; }

// LU (9 floats, row-major: L strictly below the diagonal with unit diagonal, U on and above) of transform t for this wave's context
__device__ __forceinline__ void plu_matrix(const FlowDev& d, const GenDev& g, const GenPartTab& gt, int t, int lane, float ctx_lane, float (&LU)[9]) {
    if (g.permute_type == 1) {
#pragma unroll
        for (int e = 0; e < 9; ++e) LU[e] = d.arena[gt.o_lu[t] + e];
    } else {
        const float raw = gen_net(d.arena, gt.plu[t], lane, ctx_lane, 0.f);
#pragma unroll
        for (int e = 0; e < 9; ++e) LU[e] = __shfl(raw, e);
        LU[0] = softplus_b075(LU[0]); LU[4] = softplus_b075(LU[4]); LU[8] = softplus_b075(LU[8]);
    }
}

// y = P L U x, P = eye[perm] i.e. (P v)[m] = v[perm[m]], perm = cyclic shift by s (pyro_conditional_norm_flow.py:46-47)
__device__ __forceinline__ void plu_forward(const float (&LU)[9], int s, float (&x)[3]) {
    const float u0 = LU[0] * x[0] + LU[1] * x[1] + LU[2] * x[2], u1 = LU[4] * x[1] + LU[5] * x[2], u2 = LU[8] * x[2];
    const float v[3] = {u0, LU[3] * u0 + u1, LU[6] * u0 + LU[7] * u1 + u2};
#pragma unroll
    for (int m = 0; m < 3; ++m) x[m] = v[(m + s) % 3];
}
// x = U^-1 L^-1 P^T y (torch.triangular_solve twice, :56-74); returns log|det| of the FORWARD map = sum log|u_ii| (:76-82)
__device__ __forceinline__ float plu_inverse(const float (&LU)[9], int s, float (&y)[3]) {
    float v[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) v[(m + s) % 3] = y[m];          // P^T y
    const float w0 = v[0], w1 = v[1] - LU[3] * w0, w2 = v[2] - LU[6] * w0 - LU[7] * w1;
    const float x2 = w2 / LU[8], x1 = (w1 - LU[5] * x2) / LU[4], x0 = (w0 - LU[1] * x1 - LU[2] * x2) / LU[0];
    y[0] = x0; y[1] = x1; y[2] = x2;
    return logf(fabsf(LU[0])) + logf(fabsf(LU[4])) + logf(fabsf(LU[8]));
}

__device__ __forceinline__ void gather62(float raw, int dim, float (&rw)[8], float (&rh)[8], float (&rd)[7], float (&rl)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        rw[i] = __shfl(raw, dim * 8 + i);
        rh[i] = __shfl(raw, 16 + dim * 8 + i);
        rl[i] = __shfl(raw, 46 + dim * 8 + i);
    }
#pragma unroll
    for (int i = 0; i < 7; ++i) rd[i] = __shfl(raw, 32 + dim * 7 + i);
}

// one transform in the sampling direction: [permute-like] then [coupling]
__device__ __forceinline__ void gen_transform_forward(const FlowDev& d, const GenDev& g, const GenPartTab& gt, int t, int lane, float ctx_lane, float (&x)[3]) {
    const int s = t % 3;
    if (g.permute_type == 0) {
        const float v[3] = {x[0], x[1], x[2]};
#pragma unroll
        for (int m = 0; m < 3; ++m) x[m] = v[(m + s) % 3];       // pyro Permute: y = x.index_select(-1, perm)
    } else {
        float LU[9];
        plu_matrix(d, g, gt, t, lane, ctx_lane, LU);
        plu_forward(LU, s, x);
    }
    const float raw = gen_net(d.arena, gt.tr[t], lane, ctx_lane, x[0]);
    if (g.transform_type == 0) {
        const int dim = lane & 1;
        float rw[8], rh[8], rd[7], rl[8];
        gather62(raw, dim, rw, rh, rd, rl);
        const float o = rational_linear_spline<false>(dim ? x[2] : x[1], rw, rh, rd, rl, d.radius, nullptr);
        x[1] = __shfl(o, 0); x[2] = __shfl(o, 1);
    } else {
        x[1] += __shfl(raw, 0); x[2] += __shfl(raw, 1);
    }
}

// inverse of one transform; returns the contribution to log p: -(log|det| of the forward map)
__device__ __forceinline__ float gen_transform_inverse(const FlowDev& d, const GenDev& g, const GenPartTab& gt, int t, int lane, float ctx_lane, float (&y)[3]) {
    float lp = 0.f;
    const float raw = gen_net(d.arena, gt.tr[t], lane, ctx_lane, y[0]);
    if (g.transform_type == 0) {
        const int dim = lane & 1;
        float rw[8], rh[8], rd[7], rl[8], ld;
        gather62(raw, dim, rw, rh, rd, rl);
        const float o = rational_linear_spline<true>(dim ? y[2] : y[1], rw, rh, rd, rl, d.radius, &ld);
        y[1] = __shfl(o, 0); y[2] = __shfl(o, 1);
        lp += __shfl(ld, 0) + __shfl(ld, 1);                    // log-det of the inverse = -(log-det of the forward map)
    } else {
        y[1] -= __shfl(raw, 0); y[2] -= __shfl(raw, 1);
    }
    const int s = t % 3;
    if (g.permute_type == 0) {
        float v[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) v[(m + s) % 3] = y[m];
        y[0] = v[0]; y[1] = v[1]; y[2] = v[2];
    } else {
        float LU[9];
        plu_matrix(d, g, gt, t, lane, ctx_lane, LU);
        lp -= plu_inverse(LU, s, y);
    }
    return lp;
}

// ---------------------------------------------------------------------------------------------------
// sampling / point estimate: one wave per row (rows = B*N samples, then B point-estimate rows), 4 rows per work-group
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_flow_generic_sample(FlowDev d, GenDev g, const float* __restrict__ G, const float* __restrict__ shape_mode,
                                                             const float* __restrict__ shape_log_std, NoiseSrc ns,
                                                             int B, int N, int Mtot, int packed_rows,
                                                             float* __restrict__ shape_samples, float* __restrict__ rot_samples,
                                                             float* __restrict__ rot_point, float* __restrict__ so3_point) {
    __shared__ float sH[4][kIsgc];
    __shared__ float sR[4][kP * 9];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    const bool live = row < Mtot;
    const int rs = live ? row : 0;
    const bool is_point = rs >= B * N;
    const int b = is_point ? rs - B * N : rs / max(N, 1);
    const int n = is_point ? 0 : rs - b * N;
    const int nb = d.nb;
    // shape vector of the row (models/humaniflow_model.py:250-258), isgc features h = ELU(G[b] + W_beta beta) (:116-150)
    float beta = 0.f;
    if (lane < nb) {
        beta = shape_mode[(size_t)b * nb + lane];
        if (!is_point && has_shape_noise(ns)) beta += shape_noise(ns, (unsigned int)((n * B + b) * nb + lane)) * expf(shape_log_std[(size_t)b * nb + lane]);
        if (live && shape_samples && (!is_point || packed_rows)) shape_samples[(size_t)row * nb + lane] = beta;
    }
    for (int i = 0; i < kIsgc / 64; ++i) {
        const int k = lane + 64 * i;
        float a = G[(size_t)b * kIsgc + k];
        const float* wrow = d.arena + d.o_wisgc + (size_t)k * d.in_dim + d.feat_dim;
        for (int c = 0; c < nb; ++c) a = fmaf(wrow[c], __shfl(beta, c), a);
        sH[wave][k] = elu1(a);
    }
    for (int i = lane; i < kP * 9; i += 64) sR[wave][i] = 0.f;
    __syncthreads();
    for (int j = 0; j < kP; ++j) {
        const PartTab& pt = d.tab[j];
        const GenPartTab& gt = g.tab[j];
        const float* WT = d.arena + pt.o_wctxT;
        float pre = d.arena[pt.o_bctx + lane];
        for (int k = 0; k < kIsgc; ++k) pre = fmaf(WT[(size_t)k * kCtx + lane], sH[wave][k], pre);
        for (int a = 0; a < pt.n_anc; ++a)
            for (int e = 0; e < 9; ++e) pre = fmaf(WT[(size_t)(kIsgc + a * 9 + e) * kCtx + lane], sR[wave][pt.anc[a] * 9 + e], pre);
        const float ctx = elu1(pre);
        float x[3] = {0.f, 0.f, 0.f};
        if (!is_point) {
            const unsigned int e = (unsigned int)((b * N + n) * 3);
            const unsigned long long per_part = (unsigned long long)B * N * 3;
            x[0] = pose_noise(ns, j, e, per_part) * d.base_std; x[1] = pose_noise(ns, j, e + 1, per_part) * d.base_std;
            x[2] = pose_noise(ns, j, e + 2, per_part) * d.base_std;
        }
        for (int t = 0; t < g.T; ++t) gen_transform_forward(d, g, gt, t, lane, ctx, x);
        float v[3] = {x[0], x[1], x[2]};
        {
            const float nrm = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
            if (nrm > 1e-7f) {
                const float tt = tanhf(nrm / d.radius);
                v[0] = tt * (x[0] / nrm) * d.radius; v[1] = tt * (x[1] / nrm) * d.radius; v[2] = tt * (x[2] / nrm) * d.radius;
            }
        }
        float R[9];
        if (is_point) {
            batch_rodrigues_f32(v, R);
        } else {
            const double vd[3] = {(double)v[0], (double)v[1], (double)v[2]};
            double Rd[9];
            so3_exp_f64(vd, Rd);
#pragma unroll
            for (int e = 0; e < 9; ++e) R[e] = (float)Rd[e];
        }
        __syncthreads();                                        // every wave done reading sR for this part
        if (lane == 0) {
#pragma unroll
            for (int e = 0; e < 9; ++e) sR[wave][j * 9 + e] = R[e];
            if (live) {
                if (is_point) {
#pragma unroll
                    for (int e = 0; e < 9; ++e) rot_point[((size_t)b * kP + j) * 9 + e] = R[e];
                    so3_point[((size_t)b * kP + j) * 3 + 0] = v[0]; so3_point[((size_t)b * kP + j) * 3 + 1] = v[1]; so3_point[((size_t)b * kP + j) * 3 + 2] = v[2];
                } else {
#pragma unroll
                    for (int e = 0; e < 9; ++e) rot_samples[((size_t)row * kP + j) * 9 + e] = R[e];
                }
            }
        }
        __syncthreads();
    }
}

// log p(v | ctx) on the algebra for the generic variants (torch TransformedDistribution.log_prob through the transform list)
__device__ __forceinline__ float gen_log_prob_so3(const FlowDev& d, const GenDev& g, const GenPartTab& gt, int lane, float ctx_lane, const float (&v)[3]) {
    const float r = d.radius;
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
    for (int t = g.T - 1; t >= 0; --t) lp += gen_transform_inverse(d, g, gt, t, lane, ctx_lane, x);
    const float var = d.base_std * d.base_std, lsc = logf(d.base_std), hl2pi = 0.91893853320467274178f;
    return lp - (x[0] * x[0] + x[1] * x[1] + x[2] * x[2]) / (2.f * var) - 3.f * (lsc + hl2pi);
}

__global__ __launch_bounds__(256) void k_flow_generic_logprob(FlowDev d, GenDev g, const float* __restrict__ ctx, const int* __restrict__ part_ids, int P,
                                                              const double* __restrict__ Rin, const float* __restrict__ vin, int B, int mode,
                                                              float* __restrict__ logp) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gw >= B * P) return;
    const int p = gw % P;
    const int j = part_ids ? part_ids[p] : p;
    const GenPartTab& gt = g.tab[j];
    const float ctx_lane = ctx[(size_t)gw * kCtx + lane];
    if (mode == 1) {
        const float v[3] = {vin[(size_t)gw * 3], vin[(size_t)gw * 3 + 1], vin[(size_t)gw * 3 + 2]};
        const float lp = gen_log_prob_so3(d, g, gt, lane, ctx_lane, v);
        if (lane == 0) logp[gw] = lp;
        return;
    }
    double R[9], x[3];
#pragma unroll
    for (int e = 0; e < 9; ++e) R[e] = Rin[(size_t)gw * 9 + e];
    so3_log_f64(R, x);
    const double kTwoPi = 6.28318530717958647692;
    const double n = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    float terms[3];
    for (int c = 0; c < 3; ++c) {
        double cand[3] = {x[0], x[1], x[2]};
        bool ok = true;
        if (c > 0) {
            const double k = c == 1 ? -1.0 : 1.0;
            const double sc = (n + kTwoPi * k);
            cand[0] = x[0] / n * sc; cand[1] = x[1] / n * sc; cand[2] = x[2] / n * sc;
            const double cn = sqrt(cand[0] * cand[0] + cand[1] * cand[1] + cand[2] * cand[2]);
            ok = cn < (double)d.radius;
            if (!ok) { cand[0] = 0.0; cand[1] = 0.0; cand[2] = 0.0; }
        }
        const float ld = -(float)so3_logdet_f64(cand);
        const float v[3] = {(float)cand[0], (float)cand[1], (float)cand[2]};
        const float lp = gen_log_prob_so3(d, g, gt, lane, ctx_lane, v);
        terms[c] = ok ? (ld + lp) : -INFINITY;
    }
    const float m = fmaxf(terms[0], fmaxf(terms[1], terms[2]));
    const float s = expf(terms[0] - m) + expf(terms[1] - m) + expf(terms[2] - m);
    if (lane == 0) logp[gw] = m + logf(s);
}

// ---------------------------------------------------------------------------------------------------
// Reverse sweeps of the variants (round 5; SURVEY 8f-4 under autograd: the reference trains these flows through the same
// loss.backward(), models/norm_flows/pyro_conditional_norm_flow.py:60-106).  Same design as flow_grad.hip: one wave per (row, part),
// lane = channel; the wave recomputes the part's forward pass, keeps the stage inputs of the T transforms in LDS, walks the stages in
// reverse (each stage: forward with activations, then its pull-back) and writes per-(part, transform, row) deltas / activations of every
// dense layer; the weight gradients are dense reductions over the row axis on hmf_rows_atb_batched.
// ---------------------------------------------------------------------------------------------------
struct GenAct { float h[kMaxLayers - 1]; };     // post-ReLU output of layer l, this lane's channel (0 beyond N[l])
struct GenDelta { float d[kMaxLayers]; };       // pre-activation gradient of layer l, this lane's channel

// gen_net with the activations kept (same operation order: bit-identical outputs)
__device__ __forceinline__ float gen_net_keep(const float* __restrict__ arena, const GenNet& n, int lane, float in_lane, float extra, GenAct& A) {
    float h = in_lane, a = 0.f;
#pragma unroll
    for (int l = 0; l < kMaxLayers; ++l) {
        if (l < n.nl) {
            const float* W = arena + n.o_w[l];
            a = arena[n.o_b[l] + lane];
            const int K = n.K[l], kk = K > 64 ? 64 : K;
            for (int k = 0; k < kk; ++k) a = fmaf(W[k * 64 + lane], __shfl(h, k), a);
            if (K > 64) a = fmaf(W[64 * 64 + lane], extra, a);
            h = (lane < n.N[l]) ? fmaxf(a, 0.f) : 0.f;
        }
        if (l < kMaxLayers - 1) A.h[l < kMaxLayers - 1 ? l : 0] = l < n.nl ? h : 0.f;
    }
    return a;
}

// g_out: d L / d (output of the last layer), lane < N[nl - 1] -> deltas of every layer, d L / d input (lane = input k < 64), d L / d extra input.
// g_in[k] = sum_o W[o][k] delta[o] walks the NATURAL copy of a layer (flow_pack.hip: behind the transposed one, rows = outputs, lane = input).
__device__ __forceinline__ void gen_net_bwd(const float* __restrict__ arena, const GenNet& n, int lane, const GenAct& A, float g_out, GenDelta& D,
                                            float& g_in, float& g_extra) {
    float up = 0.f;
    g_in = 0.f; g_extra = 0.f;
#pragma unroll
    for (int l = kMaxLayers - 1; l >= 0; --l) {
        float dl = 0.f;
        if (l < n.nl) {
            const float hl = A.h[l < kMaxLayers - 1 ? l : 0];
            dl = (l == n.nl - 1) ? (lane < n.N[l] ? g_out : 0.f) : ((lane < n.N[l] && hl > 0.f) ? up : 0.f);
            const int K = n.K[l], kk = K > 64 ? 64 : K;
            const float* Wn = arena + n.o_w[l] + (size_t)K * 64;
            float a = 0.f;
            for (int o = 0; o < n.N[l]; ++o) a = fmaf(Wn[o * 64 + lane], __shfl(dl, o), a);
            up = lane < kk ? a : 0.f;
            if (l == 0) {
                g_in = up;
                if (K > 64) g_extra = wave_sum(arena[n.o_w[0] + 64 * 64 + lane] * dl);
            }
        }
        D.d[l] = dl;
    }
}

// Per-(part, transform, row) records.  Zero-filled by the caller: rows of masked pre-images are not written.
struct GenRec {
    float* D;      // [P][T][L][R][64]    deltas of the transform hyper-network
    float* A;      // [P][T][L][R][64]    inputs of layers 1 .. L-1 (slot 0 unused: layer 0 reads [ACTX | XIN])
    float* XIN;    // [P][T][R]
    float* DP;     // [P][T][LP][R][64]   conditional_linear_plu: deltas of the PLU network
    float* AP;     // [P][T][LP][R][64]
    float* GLU;    // [P][T][R][9]        linear_plu: d L / d LU of the row
    float* ACTX;   // [P][R][64]
    float* DCTX;   // [P][R][64]          sample path
    int R, L, LP, T;
};

__device__ __forceinline__ void gen_store(float* Dbuf, float* Abuf, int L, int R, int T, const GenNet& n, int p, int t, int r, int lane, const GenDelta& D,
                                          const GenAct& A) {
#pragma unroll
    for (int l = 0; l < kMaxLayers; ++l)
        if (l < n.nl) {
            const size_t o = ((((size_t)p * T + t) * L + l) * R + r) * 64 + lane;
            Dbuf[o] = D.d[l];
            if (l > 0) Abuf[o] = A.h[l - 1 < kMaxLayers - 1 ? (l > 0 ? l - 1 : 0) : 0];
        }
}

// LU of transform t with what the pull-back needs (conditional: the network's activations and raw outputs)
__device__ __forceinline__ void plu_matrix_keep(const FlowDev& d, const GenDev& g, const GenPartTab& gt, int t, int lane, float ctx_lane, float (&LU)[9],
                                                float& rawP, GenAct& AP) {
    rawP = 0.f;
    if (g.permute_type == 1) {
#pragma unroll
        for (int e = 0; e < 9; ++e) LU[e] = d.arena[gt.o_lu[t] + e];
    } else {
        rawP = gen_net_keep(d.arena, gt.plu[t], lane, ctx_lane, 0.f, AP);
#pragma unroll
        for (int e = 0; e < 9; ++e) LU[e] = __shfl(rawP, e);
        LU[0] = softplus_b075(LU[0]); LU[4] = softplus_b075(LU[4]); LU[8] = softplus_b075(LU[8]);
    }
}

// d L / d LU (wave-uniform) -> the LU parameter's per-row record (linear_plu) or the PLU network's reverse sweep (conditional_linear_plu)
__device__ __forceinline__ void plu_param_bwd(const FlowDev& d, const GenDev& g, const GenPartTab& gt, int t, int lane, const float (&gLU)[9], float rawP,
                                              const GenAct& AP, float& g_ctx_acc, const GenRec& G, int p, int row) {
    float mine = 0.f;
#pragma unroll
    for (int e = 0; e < 9; ++e) mine = lane == e ? gLU[e] : mine;
    if (g.permute_type == 1) {
        if (lane < 9) G.GLU[(((size_t)p * G.T + t) * G.R + row) * 9 + lane] = mine;
    } else {
        // softplus(beta = 0.75)' = sigmoid(0.75 x) on the diagonal entries (linearPLU_transform.py:146-147)
        const bool diag = lane == 0 || lane == 4 || lane == 8;
        const float sg = 0.75f * rawP > 20.f ? 1.f : 1.f / (1.f + expf(-0.75f * rawP));
        const float g_raw = lane < 9 ? (diag ? mine * sg : mine) : 0.f;
        GenDelta DP;
        float g_ctx, g_dummy;
        gen_net_bwd(d.arena, gt.plu[t], lane, AP, g_raw, DP, g_ctx, g_dummy);
        g_ctx_acc += g_ctx;
        gen_store(G.DP, G.AP, G.LP, G.R, G.T, gt.plu[t], p, t, row, lane, DP, AP);
    }
}

// Stage t in the SAMPLING direction, x_in -> x_out = coupling_t(permute_t(x_in)), and its pull-back: gx holds d L / d x_out on entry, d L / d x_in on exit.
__device__ __forceinline__ void gen_stage_sample_bwd(const FlowDev& d, const GenDev& g, const GenPartTab& gt, int t, int lane, float ctx_lane,
                                                     const float (&xin)[3], float (&gx)[3], float& g_ctx_acc, const GenRec& G, int p, int row) {
    const int s = t % 3;
    float pm[3], LU[9], rawP = 0.f;
    GenAct AP;
    if (g.permute_type == 0) {
#pragma unroll
        for (int m = 0; m < 3; ++m) pm[m] = xin[(m + s) % 3];
    } else {
        plu_matrix_keep(d, g, gt, t, lane, ctx_lane, LU, rawP, AP);
        pm[0] = xin[0]; pm[1] = xin[1]; pm[2] = xin[2];
        plu_forward(LU, s, pm);
    }
    GenAct A;
    const float raw = gen_net_keep(d.arena, gt.tr[t], lane, ctx_lane, pm[0], A);
    float g_pm[3] = {gx[0], gx[1], gx[2]};
    float g_raw;
    if (g.transform_type == 0) {
        const int dim = lane & 1;
        float rw[8], rh[8], rd[7], rl[8], o, ld;
        gather_raw(raw, dim, rw, rh, rd, rl);
        SplineGrad S;
        spline_vjp<false>(dim ? pm[2] : pm[1], rw, rh, rd, rl, d.radius, dim ? gx[2] : gx[1], 0.f, o, ld, S);
        g_pm[1] = __shfl(S.g_x, 0); g_pm[2] = __shfl(S.g_x, 1);
        g_raw = scatter_raw_grad(S, lane);
    } else {
        g_raw = lane == 0 ? gx[1] : (lane == 1 ? gx[2] : 0.f);       // y = x + nn(.): additive_coupling_transform.py:28-40
    }
    GenDelta D;
    float g_ctx_n, g_extra;
    gen_net_bwd(d.arena, gt.tr[t], lane, A, g_raw, D, g_ctx_n, g_extra);
    g_pm[0] += g_extra;
    g_ctx_acc += g_ctx_n;
    gen_store(G.D, G.A, G.L, G.R, G.T, gt.tr[t], p, t, row, lane, D, A);
    if (lane == 0) G.XIN[((size_t)p * G.T + t) * G.R + row] = pm[0];
    if (g.permute_type == 0) {
#pragma unroll
        for (int m = 0; m < 3; ++m) gx[(m + s) % 3] = g_pm[m];
    } else {
        // y[m] = v[(m + s) % 3], v = L u, u = U x
        float gv[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) gv[(m + s) % 3] = g_pm[m];
        const float u0 = LU[0] * xin[0] + LU[1] * xin[1] + LU[2] * xin[2], u1 = LU[4] * xin[1] + LU[5] * xin[2];
        const float gu2 = gv[2], gu1 = gv[1] + LU[7] * gv[2], gu0 = gv[0] + LU[3] * gv[1] + LU[6] * gv[2];
        const float gLU[9] = {gu0 * xin[0], gu0 * xin[1], gu0 * xin[2], gv[1] * u0, gu1 * xin[1], gu1 * xin[2], gv[2] * u0, gv[2] * u1, gu2 * xin[2]};
        gx[0] = LU[0] * gu0; gx[1] = LU[1] * gu0 + LU[4] * gu1; gx[2] = LU[2] * gu0 + LU[5] * gu1 + LU[8] * gu2;
        plu_param_bwd(d, g, gt, t, lane, gLU, rawP, AP, g_ctx_acc, G, p, row);
    }
}

// Stage t in the DENSITY direction (gen_transform_inverse: y_in -> y_out, lp += its log-det terms) and its pull-back: glp = d L / d lp (wave-uniform),
// gy holds d L / d y_out on entry and d L / d y_in on exit.
__device__ __forceinline__ void gen_stage_logprob_bwd(const FlowDev& d, const GenDev& g, const GenPartTab& gt, int t, int lane, float ctx_lane,
                                                      const float (&yin)[3], float glp, float (&gy)[3], float& g_ctx_acc, const GenRec& G, int p, int row) {
    GenAct A;
    const float raw = gen_net_keep(d.arena, gt.tr[t], lane, ctx_lane, yin[0], A);
    const int dim = lane & 1;
    float rw[8], rh[8], rd[7], rl[8];
    float mid[3] = {yin[0], yin[1], yin[2]};
    if (g.transform_type == 0) {
        gather_raw(raw, dim, rw, rh, rd, rl);
        float ldu;
        const float o = rational_linear_spline<true>(dim ? yin[2] : yin[1], rw, rh, rd, rl, d.radius, &ldu);
        mid[1] = __shfl(o, 0); mid[2] = __shfl(o, 1);
    } else {
        mid[1] -= __shfl(raw, 0); mid[2] -= __shfl(raw, 1);
    }
    const int s = t % 3;
    float g_mid[3];
    if (g.permute_type == 0) {
#pragma unroll
        for (int m = 0; m < 3; ++m) g_mid[m] = gy[(m + s) % 3];        // y_out[(m + s) % 3] = mid[m]
    } else {
        float LU[9], rawP;
        GenAct AP;
        plu_matrix_keep(d, g, gt, t, lane, ctx_lane, LU, rawP, AP);
        float v[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) v[(m + s) % 3] = mid[m];
        const float w0 = v[0], w1 = v[1] - LU[3] * w0, w2 = v[2] - LU[6] * w0 - LU[7] * w1;
        const float x2 = w2 / LU[8], x1 = (w1 - LU[5] * x2) / LU[4], x0 = (w0 - LU[1] * x1 - LU[2] * x2) / LU[0];
        // x = U^-1 w: g_w = U^-T g_x, g_U = -g_w x^T; w = L^-1 v: g_v = L^-T g_w, g_L = -g_v w^T; lp -= sum log|u_ii|
        const float gw0 = gy[0] / LU[0], gw1 = (gy[1] - LU[1] * gw0) / LU[4], gw2 = (gy[2] - LU[2] * gw0 - LU[5] * gw1) / LU[8];
        const float gv2 = gw2, gv1 = gw1 - LU[7] * gv2, gv0 = gw0 - LU[3] * gv1 - LU[6] * gv2;
        const float gLU[9] = {-gw0 * x0 - glp / LU[0], -gw0 * x1, -gw0 * x2, -gv1 * w0, -gw1 * x1 - glp / LU[4], -gw1 * x2, -gv2 * w0, -gv2 * w1,
                              -gw2 * x2 - glp / LU[8]};
        const float gvv[3] = {gv0, gv1, gv2};
#pragma unroll
        for (int m = 0; m < 3; ++m) g_mid[m] = gvv[(m + s) % 3];
        plu_param_bwd(d, g, gt, t, lane, gLU, rawP, AP, g_ctx_acc, G, p, row);
    }
    float g_raw;
    if (g.transform_type == 0) {
        float o, ld;
        SplineGrad S;
        spline_vjp<true>(dim ? yin[2] : yin[1], rw, rh, rd, rl, d.radius, dim ? g_mid[2] : g_mid[1], glp, o, ld, S);
        gy[1] = __shfl(S.g_x, 0); gy[2] = __shfl(S.g_x, 1);
        g_raw = scatter_raw_grad(S, lane);
    } else {
        gy[1] = g_mid[1]; gy[2] = g_mid[2];
        g_raw = lane == 0 ? -g_mid[1] : (lane == 1 ? -g_mid[2] : 0.f);
    }
    GenDelta D;
    float g_ctx_n, g_extra;
    gen_net_bwd(d.arena, gt.tr[t], lane, A, g_raw, D, g_ctx_n, g_extra);
    gy[0] = g_mid[0] + g_extra;
    g_ctx_acc += g_ctx_n;
    gen_store(G.D, G.A, G.L, G.R, G.T, gt.tr[t], p, t, row, lane, D, A);
    if (lane == 0) G.XIN[((size_t)p * G.T + t) * G.R + row] = yin[0];
}

// Teacher-forced NLL of the variants, reverse sweep.  Records: R = 3 B rows per part, row = pre-image term * B + b.
__global__ __launch_bounds__(256) void k_flow_generic_logprob_bwd(FlowDev d, GenDev g, const float* __restrict__ ctx, const int* __restrict__ part_ids, int P,
                                                                  const double* __restrict__ Rin, int B, const float* __restrict__ g_logp,
                                                                  float* __restrict__ g_ctx, GenRec G) {
    __shared__ float sY[4][kMaxT + 1][3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = blockIdx.x * 4 + wave;
    if (gw >= B * P) return;
    const int b = gw / P, p = gw % P;
    const int j = part_ids ? part_ids[p] : p;
    const GenPartTab& gt = g.tab[j];
    const float ctx_lane = ctx[(size_t)gw * kCtx + lane];
    double R[9], x[3];
#pragma unroll
    for (int e = 0; e < 9; ++e) R[e] = Rin[(size_t)gw * 9 + e];
    so3_log_f64(R, x);
    const double kTwoPi = 6.28318530717958647692;
    const double n = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    float terms[3];
    bool ok[3];
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
        double cand[3] = {x[0], x[1], x[2]};
        bool okc = true;
        if (c > 0) {
            const double sc = n + kTwoPi * (c == 1 ? -1.0 : 1.0);
            cand[0] = x[0] / n * sc; cand[1] = x[1] / n * sc; cand[2] = x[2] / n * sc;
            okc = sqrt(cand[0] * cand[0] + cand[1] * cand[1] + cand[2] * cand[2]) < (double)d.radius;
            if (!okc) { cand[0] = 0.0; cand[1] = 0.0; cand[2] = 0.0; }
        }
        const float ld = -(float)so3_logdet_f64(cand);
        const float v[3] = {(float)cand[0], (float)cand[1], (float)cand[2]};
        const float lp = gen_log_prob_so3(d, g, gt, lane, ctx_lane, v);
        const float tc = okc ? (ld + lp) : -INFINITY;
        terms[0] = c == 0 ? tc : terms[0]; terms[1] = c == 1 ? tc : terms[1]; terms[2] = c == 2 ? tc : terms[2];
        ok[0] = c == 0 ? okc : ok[0]; ok[1] = c == 1 ? okc : ok[1]; ok[2] = c == 2 ? okc : ok[2];
    }
    const float m = fmaxf(terms[0], fmaxf(terms[1], terms[2]));
    const float ssum = expf(terms[0] - m) + expf(terms[1] - m) + expf(terms[2] - m);
    const float gup = g_logp[gw];
    float g_acc = 0.f;
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
        const bool okc = c == 0 ? ok[0] : (c == 1 ? ok[1] : ok[2]);
        const float tc = c == 0 ? terms[0] : (c == 1 ? terms[1] : terms[2]);
        const int row = c * B + b;
        G.ACTX[((size_t)p * G.R + row) * 64 + lane] = ctx_lane;
        if (!okc) continue;                                            // masked pre-image: its (zero-filled) records stay zero
        const float gc = gup * (expf(tc - m) / ssum);                  // d logsumexp / d term_c
        double cand[3] = {x[0], x[1], x[2]};
        if (c > 0) {
            const double sc = n + kTwoPi * (c == 1 ? -1.0 : 1.0);
            cand[0] = x[0] / n * sc; cand[1] = x[1] / n * sc; cand[2] = x[2] / n * sc;
        }
        // forward walk (gen_log_prob_so3), stage inputs kept: sY[T] = radial-tanh inverse of the pre-image, sY[t] = inverse_t(sY[t + 1])
        float y[3] = {(float)cand[0], (float)cand[1], (float)cand[2]};
        {
            const double nn = sqrt(cand[0] * cand[0] + cand[1] * cand[1] + cand[2] * cand[2]);
            if (nn > 1e-7) {
                const double sa = atanh(nn / (double)d.radius);
                y[0] = (float)(sa * (cand[0] / nn) * (double)d.radius); y[1] = (float)(sa * (cand[1] / nn) * (double)d.radius);
                y[2] = (float)(sa * (cand[2] / nn) * (double)d.radius);
            }
        }
        if (lane < 3) sY[wave][g.T][lane] = lane == 0 ? y[0] : (lane == 1 ? y[1] : y[2]);
        for (int t = g.T - 1; t >= 0; --t) {
            (void)gen_transform_inverse(d, g, gt, t, lane, ctx_lane, y);
            if (lane < 3) sY[wave][t][lane] = lane == 0 ? y[0] : (lane == 1 ? y[1] : y[2]);
        }
        const float var = d.base_std * d.base_std;
        float gy[3] = {-gc * y[0] / var, -gc * y[1] / var, -gc * y[2] / var};      // base density -|y^(0)|^2 / (2 var)
        for (int t = 0; t < g.T; ++t) {
            const float yin[3] = {sY[wave][t + 1][0], sY[wave][t + 1][1], sY[wave][t + 1][2]};
            gen_stage_logprob_bwd(d, g, gt, t, lane, ctx_lane, yin, gc, gy, g_acc, G, p, row);
        }
    }
    g_ctx[(size_t)gw * kCtx + lane] = g_acc;
}

// Sampling path of the variants, reverse sweep of one round of the kinematic schedule (rounds nlev-1 .. 0 on one stream, as k_flow_chain_bwd).
__global__ __launch_bounds__(256) void k_flow_generic_chain_bwd(FlowDev d, GenDev g, int lev, const float* __restrict__ H, const float* __restrict__ rot,
                                                                const float* __restrict__ eps, int B, int N, float* __restrict__ g_rot, GenRec G) {
    __shared__ float sX[4][kMaxT + 1][3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = blockIdx.x * 4 + wave;
    const int rows = B * N;
    const int nparts = d.lev.start[lev + 1] - d.lev.start[lev];
    if (gw >= rows * nparts) return;
    const int row = gw / nparts;
    const int j = d.lev.parts[d.lev.start[lev] + gw % nparts];
    const PartTab& pt = d.tab[j];
    const GenPartTab& gt = g.tab[j];
    // forward recompute: context layer in k_flow_generic_sample's order
    const float* WT = d.arena + pt.o_wctxT;
    float pre = d.arena[pt.o_bctx + lane];
    {
        const float* h = H + (size_t)row * kIsgc;
        for (int k = 0; k < kIsgc; ++k) pre = fmaf(WT[(size_t)k * kCtx + lane], h[k], pre);
        const float* rr = rot + (size_t)row * kP * 9;
        for (int a = 0; a < pt.n_anc; ++a)
            for (int e = 0; e < 9; ++e) pre = fmaf(WT[(size_t)(kIsgc + a * 9 + e) * kCtx + lane], rr[pt.anc[a] * 9 + e], pre);
    }
    const float ctx = elu1(pre);
    const int bi = row / N, ni = row - bi * N;
    const float* e3 = eps + (((size_t)j * B + bi) * N + ni) * 3;
    float x[3] = {e3[0] * d.base_std, e3[1] * d.base_std, e3[2] * d.base_std};
    for (int t = 0; t < g.T; ++t) {
        if (lane < 3) sX[wave][t][lane] = lane == 0 ? x[0] : (lane == 1 ? x[1] : x[2]);
        gen_transform_forward(d, g, gt, t, lane, ctx, x);
    }
    float gx[3];
    exp_tanh_vjp(x[0], x[1], x[2], d.radius, g_rot + ((size_t)row * kP + j) * 9, gx);
    float g_ctx_acc = 0.f;
    for (int t = g.T - 1; t >= 0; --t) {
        const float xin[3] = {sX[wave][t][0], sX[wave][t][1], sX[wave][t][2]};
        gen_stage_sample_bwd(d, g, gt, t, lane, ctx, xin, gx, g_ctx_acc, G, j, row);
    }
    const float dctx = g_ctx_acc * (pre > 0.f ? 1.f : ctx + 1.f);                   // ELU'
    {
        const int ka = 9 * pt.n_anc;
        const float* Wn = WT + (size_t)(kIsgc + ka) * kCtx;
        float ga = 0.f;
        for (int o = 0; o < 64; ++o) ga = fmaf(Wn[o * 64 + lane], __shfl(dctx, o), ga);
        if (lane < ka) atomicAdd(g_rot + ((size_t)row * kP + pt.anc[lane / 9]) * 9 + lane % 9, ga);
    }
    G.ACTX[((size_t)j * G.R + row) * 64 + lane] = ctx;
    G.DCTX[((size_t)j * G.R + row) * 64 + lane] = dctx;
}

// entry points used by flow_kernels.hip's dispatch
int generic_forward(const hmf_flow* h, const float* G, const float* shape_mode, const float* shape_log_std, const NoiseSrc& ns,
                    int B, int N, int Mtot, int packed_rows, float* shape_samples, float* rot_samples, float* rot_point, float* so3_point, hipStream_t st) {
    hipLaunchKernelGGL(k_flow_generic_sample, dim3((Mtot + 3) / 4), dim3(256), 0, st, h->d, h->g, G, shape_mode, shape_log_std, ns, B, N, Mtot,
                       packed_rows, shape_samples, rot_samples, rot_point, so3_point);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

int generic_logprob(const hmf_flow* h, const float* ctx, const int* dev_ids, int P, const double* R, const float* v, int B, int mode, float* logp, hipStream_t st) {
    const int nw = B * P;
    hipLaunchKernelGGL(k_flow_generic_logprob, dim3((nw + 3) / 4), dim3(256), 0, st, h->d, h->g, ctx, dev_ids, P, R, v, B, mode, logp);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

static int gen_rec_from(const hmf_flow* h, const hmf_flow_generic_grad_records_t* r, GenRec& G, bool need_dctx) {
    if (!r || !r->D || !r->A || !r->XIN || !r->ACTX || r->rows <= 0 || (need_dctx && !r->DCTX))
        return fail(HMF_ERR_INVALID, "hmf_flow_generic_*_backward: incomplete gradient-record struct");
    const int L = h->gen_layers_tr, LP = h->gen_layers_plu;
    if (r->L != L) return fail(HMF_ERR_INVALID, "hmf_flow_generic_*_backward: records say L = %d, the handle's transform networks have %d layers", r->L, L);
    if (h->g.permute_type == 2 && (!r->DP || !r->AP || r->LP != LP))
        return fail(HMF_ERR_INVALID, "hmf_flow_generic_*_backward: conditional_linear_plu needs DP / AP with LP = %d", LP);
    if (h->g.permute_type == 1 && !r->GLU) return fail(HMF_ERR_INVALID, "hmf_flow_generic_*_backward: linear_plu needs GLU");
    G.D = r->D; G.A = r->A; G.XIN = r->XIN; G.DP = r->DP; G.AP = r->AP; G.GLU = r->GLU; G.ACTX = r->ACTX; G.DCTX = r->DCTX;
    G.R = r->rows; G.L = r->L; G.LP = r->LP; G.T = h->g.T;
    return HMF_OK;
}

}  // namespace hmf

using namespace hmf;

extern "C" {

int hmf_flow_generic_log_prob_SO3_backward(const hmf_flow_t* h, const float* ctx, const int32_t* part_ids_host, int P, const double* R, int B,
                                           const float* g_logp, float* g_ctx, const hmf_flow_generic_grad_records_t* rec, void* stream) {
    if (!h || !ctx || !R || !g_logp || !g_ctx || B <= 0 || P <= 0 || P > kP) return fail(HMF_ERR_INVALID, "hmf_flow_generic_log_prob_SO3_backward: bad args");
    if (!h->g.enabled) return fail(HMF_ERR_UNSUPPORTED, "hmf_flow_generic_log_prob_SO3_backward: the handle runs the default kernels (hmf_flow_log_prob_SO3_backward)");
    if (!part_ids_host && P != kP) return fail(HMF_ERR_INVALID, "hmf_flow_generic_log_prob_SO3_backward: part_ids NULL requires P == 23");
    const int* dev_ids = nullptr;
    if (part_ids_host) {
        for (int i = 0; i < P; ++i)
            if (part_ids_host[i] < 0 || part_ids_host[i] >= kP || part_ids_host[i] != part_ids_host[0] + i)
                return fail(HMF_ERR_UNSUPPORTED, "hmf_flow_generic_log_prob_SO3_backward: part ids must be a contiguous ascending range inside [0, 23)");
        dev_ids = h->d_ids + part_ids_host[0];
    }
    GenRec G;
    int rc = gen_rec_from(h, rec, G, false);
    if (rc) return rc;
    if (G.R != 3 * B) return fail(HMF_ERR_INVALID, "hmf_flow_generic_log_prob_SO3_backward: records need rows == 3 * B");
    const int nw = B * P;
    hipLaunchKernelGGL(k_flow_generic_logprob_bwd, dim3((nw + 3) / 4), dim3(256), 0, (hipStream_t)stream, h->d, h->g, ctx, dev_ids, P, R, B, g_logp, g_ctx, G);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

int hmf_flow_generic_sample_backward(const hmf_flow_t* h, const float* isgc_rows, const float* rotmats_samples, const float* eps_pose, int B, int N,
                                     float* g_rotmats, const hmf_flow_generic_grad_records_t* rec, void* stream) {
    if (!h || !isgc_rows || !rotmats_samples || !eps_pose || !g_rotmats || B <= 0 || N <= 0)
        return fail(HMF_ERR_INVALID, "hmf_flow_generic_sample_backward: bad args");
    if (!h->g.enabled) return fail(HMF_ERR_UNSUPPORTED, "hmf_flow_generic_sample_backward: the handle runs the default kernels (hmf_flow_sample_backward)");
    GenRec G;
    int rc = gen_rec_from(h, rec, G, true);
    if (rc) return rc;
    if (G.R != B * N) return fail(HMF_ERR_INVALID, "hmf_flow_generic_sample_backward: records need rows == B * N");
    for (int lev = h->d.lev.nlev - 1; lev >= 0; --lev) {
        const int nw = B * N * (h->d.lev.start[lev + 1] - h->d.lev.start[lev]);
        hipLaunchKernelGGL(k_flow_generic_chain_bwd, dim3((nw + 3) / 4), dim3(256), 0, (hipStream_t)stream, h->d, h->g, lev, isgc_rows, rotmats_samples,
                           eps_pose, B, N, g_rotmats, G);
        HMF_LAUNCH_CHECK();
    }
    return HMF_OK;
}

}  // extern "C"
