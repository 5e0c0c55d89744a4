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

}  // namespace hmf
