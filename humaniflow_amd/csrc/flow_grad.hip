// Backward (vector-Jacobian) kernels of the ancestor-conditioned SO(3) flow: the training step of BASELINE configs[4] / SURVEY 8f-3.
//
// What the reference differentiates (train/train_humaniflow.py:248-330, losses/humaniflow_loss.py:23-79):
//   * the teacher-forced pose NLL: sum_j log p(R_j | ctx_j) through LocalDiffeoTransformedDistribution.log_prob
//     (local_diffeo_transformed_distribution.py:84-142) -> d/d ctx and d/d hyper-network weights        k_flow_logprob_bwd
//   * the reparameterised samples pose_SO3_samples[:, :, j] (models/humaniflow_model.py:302-311) that feed the 2D reprojection loss:
//     d/d ctx-layer inputs, d/d ancestors' rotations (the kinematic recursion, reversed), d/d weights      k_flow_chain_bwd
//
// Design: one wave per (row, part), lane = channel, exactly the op order of the teacher-forced forward kernels of flow_kernels.hip.
// The wave recomputes the part's forward pass from the saved inputs, runs the reverse sweep in registers and writes, per (row, part),
// the pre-activation gradients ("deltas") and the activations of every dense layer.  The weight gradients are then plain dense
// reductions over the row axis, dW_l = delta_l^T . act_{l-1}, which the host side does with library GEMMs (torch.bmm): small
// (K = rows), order-independent of the kinematic tree, and exactly the kind of GEMM hipBLASLt is for.
// The spline's derivatives with respect to its 31 raw parameters come from forward-mode dual numbers through the ~20 scalar
// operations of the selected bin (one evaluation, 8 tangents), followed by the closed-form softmax / cumsum / softplus / sigmoid
// pull-backs -- no hand-derived rational-spline Jacobian to get wrong.
#include "flow_common.h"
#include "flow_handle.h"
#include "so3_math.h"
#include "flow_grad_common.h"

namespace hmf {

struct NetAct {
    float h1, h2, h3;          // post-ReLU activations of this lane's channel (h2, h3: lanes < 32, else 0)
};

// pyro ConditionalDenseNN([context, x1]) of coupling c (65 -> 64 -> 32 -> 32 -> 62, ReLU): raw[lane], activations kept for the reverse sweep
__device__ __forceinline__ float hypernet_fwd(const FlowDev& d, const PartTab& pt, int c, int lane, float ctx_lane, float x1, NetAct& A) {
    const float* W = d.arena + pt.o_nnT[c][0];
    float a = d.arena[pt.o_nnb[c][0] + lane];
#pragma unroll 32
    for (int k = 0; k < 64; ++k) a = fmaf(W[k * 64 + lane], __shfl(ctx_lane, k), a);      // unrolled: the weight loads of 16 steps in flight together
    a = fmaf(W[64 * 64 + lane], x1, a);
    A.h1 = fmaxf(a, 0.f);
    W = d.arena + pt.o_nnT[c][1];
    a = d.arena[pt.o_nnb[c][1] + lane];
#pragma unroll 32
    for (int k = 0; k < 64; ++k) a = fmaf(W[k * 64 + lane], __shfl(A.h1, k), a);
    A.h2 = lane < 32 ? fmaxf(a, 0.f) : 0.f;
    W = d.arena + pt.o_nnT[c][2];
    a = d.arena[pt.o_nnb[c][2] + lane];
#pragma unroll 32
    for (int k = 0; k < 32; ++k) a = fmaf(W[k * 64 + lane], __shfl(A.h2, k), a);
    A.h3 = lane < 32 ? fmaxf(a, 0.f) : 0.f;
    W = d.arena + pt.o_nnT[c][3];
    a = d.arena[pt.o_nnb[c][3] + lane];
#pragma unroll 32
    for (int k = 0; k < 32; ++k) a = fmaf(W[k * 64 + lane], __shfl(A.h3, k), a);
    return a;
}

struct NetDelta {
    float d1, d2, d3, d4;      // pre-activation gradients of the four layers, this lane's channel
};

// reverse sweep of the hyper-network: g_raw (lane < 62) -> deltas, d L / d context (lane = context channel), d L / d x1 (wave-uniform).
// g_in[k] = sum_o W[o][k] delta[o] runs on the NATURAL copy of each layer (row o, lane = input k: flow_pack.hip places it behind the transposed
// copy): a lane-parallel walk over the outputs with delta[o] broadcast by a lane read -- the same shape as the forward sweep.  (The first version
// reduced W^T[k][lane] delta[lane] across lanes with a 63-shuffle transpose-reduce per layer: 2.3 of the 3.2 ms of the sample-chain backward.)
__device__ __forceinline__ void hypernet_bwd(const FlowDev& d, const PartTab& pt, int c, int lane, const NetAct& A, float g_raw, NetDelta& D,
                                             float& g_ctx, float& g_x1) {
    D.d4 = lane < 62 ? g_raw : 0.f;
    {
        const float* Wn = d.arena + pt.o_nnT[c][3] + 32 * 64;
        float a = 0.f;
#pragma unroll 31
        for (int o = 0; o < 62; ++o) a = fmaf(Wn[o * 64 + lane], __shfl(D.d4, o), a);
        D.d3 = (lane < 32 && A.h3 > 0.f) ? a : 0.f;
    }
    {
        const float* Wn = d.arena + pt.o_nnT[c][2] + 32 * 64;
        float a = 0.f;
#pragma unroll
        for (int o = 0; o < 32; ++o) a = fmaf(Wn[o * 64 + lane], __shfl(D.d3, o), a);
        D.d2 = (lane < 32 && A.h2 > 0.f) ? a : 0.f;
    }
    {
        const float* Wn = d.arena + pt.o_nnT[c][1] + 64 * 64;
        float a = 0.f;
#pragma unroll
        for (int o = 0; o < 32; ++o) a = fmaf(Wn[o * 64 + lane], __shfl(D.d2, o), a);
        D.d1 = A.h1 > 0.f ? a : 0.f;
    }
    {
        const float* Wn = d.arena + pt.o_nnT[c][0] + 65 * 64;
        float a = 0.f;
#pragma unroll 32
        for (int o = 0; o < 64; ++o) a = fmaf(Wn[o * 64 + lane], __shfl(D.d1, o), a);
        g_ctx = a;
        g_x1 = wave_sum(d.arena[pt.o_nnT[c][0] + 64 * 64 + lane] * D.d1);
    }
}

// Per-(part, row) records for the host-side weight-gradient GEMMs.  R = rows per part; every array is [P][...][R][width].
struct GradRec {
    float* D1;      // [P][2][R][64]   delta of hyper-net layer 1, coupling c
    float* D2;      // [P][2][R][32]
    float* D3;      // [P][2][R][32]
    float* D4;      // [P][2][R][64]   (62 valid)
    float* A1;      // [P][2][R][64]   ReLU output of layer 1 (input of layer 2)
    float* A2;      // [P][2][R][32]
    float* A3;      // [P][2][R][32]
    float* XIN;     // [P][2][R]       scalar input x1 of the hyper-network
    float* ACTX;    // [P][R][64]      context vector (input of layer 1, both couplings)
    float* DCTX;    // [P][R][64]      delta of the context layer (sampling path only, else NULL)
    int R;
};

__device__ __forceinline__ void store_net(const GradRec& G, int p, int c, int r, int lane, const NetDelta& D, const NetAct& A, float x1) {
    const size_t pc = (size_t)(p * 2 + c) * G.R + r;
    G.D1[pc * 64 + lane] = D.d1;
    G.D4[pc * 64 + lane] = D.d4;
    G.A1[pc * 64 + lane] = A.h1;
    if (lane < 32) {
        G.D2[pc * 32 + lane] = D.d2;
        G.D3[pc * 32 + lane] = D.d3;
        G.A2[pc * 32 + lane] = A.h2;
        G.A3[pc * 32 + lane] = A.h3;
    }
    if (lane == 0) G.XIN[pc] = x1;
}

// ---------------------------------------------------------------------------------------------------
// log p(v | ctx) on the algebra (mirrors log_prob_so3 of flow_kernels.hip) in two halves: the forward keeps what the reverse sweep needs in
// registers (AlgState), so that a wave evaluates its pre-image ONCE and the three pre-images of a (row, part) run on three waves side by side.
// ---------------------------------------------------------------------------------------------------
struct AlgState {
    NetAct A0, A1;
    float raw0, raw1;             // hyper-network outputs (lane = spline parameter)
    float x0, u0, in0, in1, z1, z2;
};

__device__ __forceinline__ float algebra_fwd(const FlowDev& d, const PartTab& pt, int lane, float ctx_lane, const float (&v)[3], AlgState& S) {
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
    const int dim = lane & 1;
    // coupling 1 inverse: hyper-net on (ctx, x0); dims (x1, x2) -> (a, b)
    S.x0 = x[0];
    S.raw1 = hypernet_fwd(d, pt, 1, lane, ctx_lane, x[0], S.A1);
    float w1[8], h1[8], d1[7], l1[8];
    gather_raw(S.raw1, dim, w1, h1, d1, l1);
    S.in1 = dim ? x[2] : x[1];
    float ld1;
    const float o1 = rational_linear_spline<true>(S.in1, w1, h1, d1, l1, r, &ld1);
    const float a = __shfl(o1, 0), b = __shfl(o1, 1);
    lp += __shfl(ld1, 0) + __shfl(ld1, 1);
    // Permute([1,2,0]) inverse: u = (b, x0, a); coupling 0 inverse: hyper-net on (ctx, u0); dims (u1, u2) -> (z1, z2)
    const float u0 = b, u1 = x[0], u2 = a;
    S.u0 = u0;
    S.raw0 = hypernet_fwd(d, pt, 0, lane, ctx_lane, u0, S.A0);
    float w0[8], h0[8], d0[7], l0[8];
    gather_raw(S.raw0, dim, w0, h0, d0, l0);
    S.in0 = dim ? u2 : u1;
    float ld0;
    const float o0 = rational_linear_spline<true>(S.in0, w0, h0, d0, l0, r, &ld0);
    S.z1 = __shfl(o0, 0); S.z2 = __shfl(o0, 1);
    lp += __shfl(ld0, 0) + __shfl(ld0, 1);
    const float var = d.base_std * d.base_std, lsc = logf(d.base_std), hl2pi = 0.91893853320467274178f;
    lp += -(u0 * u0 + S.z1 * S.z1 + S.z2 * S.z2) / (2.f * var) - 3.f * (lsc + hl2pi);
    return lp;
}

// g: upstream gradient of this term (wave-uniform).  Returns d L / d ctx of this lane's channel; writes the records of row `row` (REC).
// gx (round 6, non-null): d L / d x, the gradient w.r.t. the INPUT of the inverse couplings (x = the pre-image of v under the radial tanh): the hyper-networks'
// scalar inputs, the splines' inputs (value and log-derivative both depend on them) and the base density, wave-uniform.
template <bool REC = true>
__device__ __forceinline__ float algebra_bwd(const FlowDev& d, const PartTab& pt, int lane, float g, const AlgState& S, const GradRec& G, int p, int row,
                                             float* gx = nullptr) {
    const float r = d.radius;
    const int dim = lane & 1;
    const float var = d.base_std * d.base_std;
    const float g_z0 = -g * S.u0 / var, g_z1 = -g * S.z1 / var, g_z2 = -g * S.z2 / var;
    float w0[8], h0[8], d0[7], l0[8];
    gather_raw(S.raw0, dim, w0, h0, d0, l0);
    SplineGrad S0;
    float t0, t1;
    spline_vjp<true>(S.in0, w0, h0, d0, l0, r, dim ? g_z2 : g_z1, g, t0, t1, S0);
    NetDelta D0;
    float g_ctx0, g_x1_0;
    hypernet_bwd(d, pt, 0, lane, S.A0, scatter_raw_grad(S0, lane), D0, g_ctx0, g_x1_0);
    const float g_u0 = g_z0 + g_x1_0;
    const float g_u2 = __shfl(S0.g_x, 1);                       // u2 is the input of dimension 1 of spline 0
    float w1[8], h1[8], d1[7], l1[8];
    gather_raw(S.raw1, dim, w1, h1, d1, l1);
    SplineGrad S1;
    spline_vjp<true>(S.in1, w1, h1, d1, l1, r, dim ? g_u0 : g_u2, g, t0, t1, S1);     // a = out of dim 0 -> u2; b = out of dim 1 -> u0
    NetDelta D1;
    float g_ctx1, g_x1_1;
    hypernet_bwd(d, pt, 1, lane, S.A1, scatter_raw_grad(S1, lane), D1, g_ctx1, g_x1_1);
    if (REC) {
        store_net(G, p, 0, row, lane, D0, S.A0, S.u0);
        store_net(G, p, 1, row, lane, D1, S.A1, S.x0);
    }
    if (gx) {
        // x0 enters three times: as u1 (input of dimension 0 of spline 0), as the scalar input of hyper-network 1 -- and nowhere else (it passes
        // through coupling 1 unchanged); x1 / x2 are the inputs of dimensions 0 / 1 of spline 1
        gx[0] = __shfl(S0.g_x, 0) + g_x1_1;
        gx[1] = __shfl(S1.g_x, 0);
        gx[2] = __shfl(S1.g_x, 1);
    }
    return g_ctx0 + g_ctx1;
}

// Teacher-forced NLL, reverse sweep.  Records: R = 3 * B rows per part, row = term * B + b (terms of out-of-support pre-images: zeros).
// One work-group of three waves per (row, part), wave c = pre-image c (round 5; the single-wave form walked the three forwards, then forward + reverse
// again per live pre-image: 5-6 dependent passes of the hyper-networks per wave, 395 us at B = 72; here one forward and one reverse per wave).
__global__ __launch_bounds__(192) void k_flow_logprob_bwd(FlowDev d, const float* __restrict__ ctx, const int* __restrict__ part_ids, int P,
                                                          const double* __restrict__ Rin, int B, const float* __restrict__ g_logp,
                                                          float* __restrict__ g_ctx, GradRec G) {
    __shared__ float s_term[3];
    __shared__ float s_gctx[3][64];
    const int lane = threadIdx.x & 63, c = threadIdx.x >> 6;
    const int gw = blockIdx.x;
    const int b = gw / P, p = gw % P;
    const int j = part_ids ? part_ids[p] : p;
    const PartTab& pt = d.tab[j];
    const float ctx_lane = ctx[(size_t)gw * kCtx + lane];
    double R[9], x[3];
#pragma unroll
    for (int e = 0; e < 9; ++e) R[e] = Rin[(size_t)gw * 9 + e];
    so3_log_f64(R, x);
    const double kTwoPi = 6.28318530717958647692;
    const double n = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    double cd[3] = {x[0], x[1], x[2]};
    bool okc = true;
    if (c > 0) {
        const double sc = n + kTwoPi * (c == 1 ? -1.0 : 1.0);
        cd[0] = x[0] / n * sc; cd[1] = x[1] / n * sc; cd[2] = x[2] / n * sc;
        okc = sqrt(cd[0] * cd[0] + cd[1] * cd[1] + cd[2] * cd[2]) < (double)d.radius;
    }
    const int row = c * B + b;
    G.ACTX[((size_t)p * G.R + row) * 64 + lane] = ctx_lane;
    if (!okc) {
        // masked pre-image (wave-uniform; usually two of the three waves): -inf into the log-sum-exp, zero records, no context gradient -- and the wave
        // ENDS here instead of idling at the two barriers below with its registers held (the hardware barrier counts live waves only; the fence makes
        // its LDS writes land first).  Wave 0 (the principal pre-image) is never masked.
        if (lane == 0) s_term[c] = -INFINITY;
        s_gctx[c][lane] = 0.f;
        NetDelta Z = {0.f, 0.f, 0.f, 0.f};
        NetAct ZA = {0.f, 0.f, 0.f};
        store_net(G, p, 0, row, lane, Z, ZA, 0.f);
        store_net(G, p, 1, row, lane, Z, ZA, 0.f);
        __threadfence_block();
        return;
    }
    AlgState S;
    const float ldc = -(float)so3_logdet_f64(cd);
    const float cc[3] = {(float)cd[0], (float)cd[1], (float)cd[2]};
    const float term = ldc + algebra_fwd(d, pt, lane, ctx_lane, cc, S);
    if (lane == 0) s_term[c] = term;
    __syncthreads();
    const float t0 = s_term[0], t1 = s_term[1], t2 = s_term[2];
    const float m = fmaxf(t0, fmaxf(t1, t2));
    const float ssum = expf(t0 - m) + expf(t1 - m) + expf(t2 - m);
    const float wgt = expf(term - m) / ssum;                          // d logsumexp / d term_c
    s_gctx[c][lane] = algebra_bwd(d, pt, lane, g_logp[gw] * wgt, S, G, p, row);
    __syncthreads();
    if (c == 0) g_ctx[(size_t)gw * kCtx + lane] = (s_gctx[0][lane] + s_gctx[1][lane]) + s_gctx[2][lane];
}

// Round 6: d log p(v | ctx) / d v and / d ctx of the ALGEBRA density (hmf_flow_log_prob_so3's reverse sweep w.r.t. its argument): one wave per (row, part),
// no weight-gradient records.  The flow-as-prior call pattern (optimise/optimise_humaniflow.py:96-109) differentiates the SO(3) density w.r.t. the rotation it
// is evaluated at: host code composes this with the fp64 pull-back through so3_log / the 2 pi k pre-images / the exp map's log-determinant (autograd.py).
// v -> x = r atanh(|v| / r) v / |v| (scaled_radial_tanh_transform.py inverse) and its log-determinant term -(2 (log |v| - log |x|) + log1p(-(|v| / r)^2)):
//   d x / d v = (f / n)(I - e e^T) + f' e e^T,  f = r atanh(n / r), f' = 1 / (1 - (n / r)^2), e = v / n;   d lp / d n = -(2 (1 / n - f' / f) - 2 n f' / r^2)
__global__ __launch_bounds__(256) void k_flow_logprob_so3_vjp(FlowDev d, const float* __restrict__ ctx, const int* __restrict__ part_ids, int P,
                                                              const float* __restrict__ vin, int B, const float* __restrict__ g_logp,
                                                              float* __restrict__ g_ctx, float* __restrict__ g_v) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gw >= B * P) return;
    const int p = gw % P;
    const int j = part_ids ? part_ids[p] : p;
    const PartTab& pt = d.tab[j];
    const float ctx_lane = ctx[(size_t)gw * kCtx + lane];
    const float v[3] = {vin[(size_t)gw * 3], vin[(size_t)gw * 3 + 1], vin[(size_t)gw * 3 + 2]};
    AlgState S;
    (void)algebra_fwd(d, pt, lane, ctx_lane, v, S);
    const float g = g_logp[gw];
    float gx[3];
    GradRec none = {};
    const float gc = algebra_bwd<false>(d, pt, lane, g, S, none, 0, 0, gx);
    g_ctx[(size_t)gw * kCtx + lane] = gc;
    if (lane == 0) {
        const double r = (double)d.radius;
        const double vd[3] = {(double)v[0], (double)v[1], (double)v[2]};
        const double n = sqrt(vd[0] * vd[0] + vd[1] * vd[1] + vd[2] * vd[2]);
        double out[3] = {(double)gx[0], (double)gx[1], (double)gx[2]};                  // |v| <= 1e-7: x = v, no log-determinant term (algebra_fwd)
        if (n > 1e-7) {
            const double e[3] = {vd[0] / n, vd[1] / n, vd[2] / n};
            const double f = r * atanh(n / r), fp = 1.0 / (1.0 - (n / r) * (n / r));
            const double eg = e[0] * gx[0] + e[1] * gx[1] + e[2] * gx[2];
            const double dlp = (sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) > 1e-7f) ? -(2.0 * (1.0 / n - fp / f) - 2.0 * n * fp / (r * r)) : 0.0;
#pragma unroll
            for (int i = 0; i < 3; ++i) out[i] = (f / n) * ((double)gx[i] - eg * e[i]) + fp * eg * e[i] + (double)g * dlp * e[i];
        }
        g_v[(size_t)gw * 3] = (float)out[0]; g_v[(size_t)gw * 3 + 1] = (float)out[1]; g_v[(size_t)gw * 3 + 2] = (float)out[2];
    }
}

// ---------------------------------------------------------------------------------------------------
// sampling path, reverse sweep of one round of the kinematic schedule.  Launched for rounds nlev-1 .. 0 on one stream:
// when part j runs, g_rot[row][j] already holds the loss gradient plus every descendant's contribution.
// rows = B*N sample rows (point-estimate rows carry no gradient: torch.set_grad_enabled(grad_for_pose_point_est), :294).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_flow_chain_bwd(FlowDev d, int lev, const float* __restrict__ H, const float* __restrict__ rot,
                                                        const float* __restrict__ eps, int B, int N, float* __restrict__ g_rot, GradRec G) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);   // launched with ONE wave per work-group: 1,700 waves of a round spread evenly over the CUs
    const int rows = B * N;
    const int nparts = d.lev.start[lev + 1] - d.lev.start[lev];
    if (gw >= rows * nparts) return;
    const int row = gw / nparts;
    const int j = d.lev.parts[d.lev.start[lev] + gw % nparts];
    const PartTab& pt = d.tab[j];
    const float bound = d.radius;
    // ---- forward recompute: context layer (models/humaniflow_model.py:152-186) ----
    const float* WT = d.arena + pt.o_wctxT;
    float pre = d.arena[pt.o_bctx + lane];
    {
        const float* h = H + (size_t)row * kIsgc;
#pragma unroll 32
        for (int k = 0; k < kIsgc; ++k) pre = fmaf(WT[(size_t)k * kCtx + lane], h[k], pre);
        // ancestors: a fixed, fully unrolled walk over the (at most 7 x 9) rotation entries with the loads predicated, so that all of them are in
        // flight together (a runtime-bounded loop waited for one L2 round trip per entry: up to 63 dependent round trips per wave)
        const int ka = 9 * pt.n_anc;
        const float* rr = rot + (size_t)row * kP * 9;
#pragma unroll
        for (int k = 0; k < 63; ++k) {
            const int kc = k < ka ? k : 0;
            const float w = WT[(size_t)(kIsgc + kc) * kCtx + lane];
            const float r = rr[(k < ka ? pt.anc[kc / 9] : 0) * 9 + kc % 9];          // (a root part has no valid anc[] entry at all)
            pre = fmaf(k < ka ? w : 0.f, r, pre);
        }
    }
    const float ctx = elu1(pre);
    const int bi = row / N, ni = row - bi * N;
    const float* e3 = eps + (((size_t)j * B + bi) * N + ni) * 3;
    const float z0 = e3[0] * d.base_std, z1 = e3[1] * d.base_std, z2 = e3[2] * d.base_std;
    const int dim = lane & 1;
    // coupling 0 (Permute [0,1,2]): z0 passes, (z1, z2) -> (x1n, x2n)
    NetAct A0;
    const float raw0 = hypernet_fwd(d, pt, 0, lane, ctx, z0, A0);
    float w0[8], h0[8], d0[7], l0[8];
    gather_raw(raw0, dim, w0, h0, d0, l0);
    const float in0 = dim ? z2 : z1;
    const float o0 = rational_linear_spline<false>(in0, w0, h0, d0, l0, bound, nullptr);
    const float x1n = __shfl(o0, 0), x2n = __shfl(o0, 1);
    // coupling 1 (Permute [1,2,0]): y = (x1n, x2n, z0); y0 passes, (y1, y2) -> (y1', y2')
    NetAct A1;
    const float raw1 = hypernet_fwd(d, pt, 1, lane, ctx, x1n, A1);
    float w1[8], h1[8], d1[7], l1[8];
    gather_raw(raw1, dim, w1, h1, d1, l1);
    const float in1 = dim ? z0 : x2n;
    const float o1 = rational_linear_spline<false>(in1, w1, h1, d1, l1, bound, nullptr);
    const float y1 = __shfl(o1, 0), y2 = __shfl(o1, 1);
    // ---- reverse: exp map (fp64), radial tanh ----
    float gpre[3];
    exp_tanh_vjp(x1n, y1, y2, bound, g_rot + ((size_t)row * kP + j) * 9, gpre);
    // ---- reverse: coupling 1, coupling 0 ----
    SplineGrad S1;
    float t0, t1;
    spline_vjp<false>(in1, w1, h1, d1, l1, bound, dim ? gpre[2] : gpre[1], 0.f, t0, t1, S1);
    const float g_x2n = __shfl(S1.g_x, 0);                         // x2n is the input of dimension 0 of spline 1
    NetDelta D1;
    float g_ctx1, g_y0;
    hypernet_bwd(d, pt, 1, lane, A1, scatter_raw_grad(S1, lane), D1, g_ctx1, g_y0);
    const float g_x1n = gpre[0] + g_y0;
    SplineGrad S0;
    spline_vjp<false>(in0, w0, h0, d0, l0, bound, dim ? g_x2n : g_x1n, 0.f, t0, t1, S0);
    NetDelta D0;
    float g_ctx0, g_z0;
    hypernet_bwd(d, pt, 0, lane, A0, scatter_raw_grad(S0, lane), D0, g_ctx0, g_z0);
    const float dctx = (g_ctx0 + g_ctx1) * (pre > 0.f ? 1.f : ctx + 1.f);          // ELU'
    // ---- ancestors: d L / d R_anc += W_ctx[:, 256 + k]^T . delta (natural copy of the ancestor columns, lane = entry k) ----
    {
        const int ka = 9 * pt.n_anc;
        const float* Wn = WT + (size_t)(kIsgc + ka) * kCtx;
        float ga = 0.f;
#pragma unroll 32
        for (int o = 0; o < 64; ++o) ga = fmaf(Wn[o * 64 + lane], __shfl(dctx, o), ga);
        if (lane < ka) atomicAdd(g_rot + ((size_t)row * kP + pt.anc[lane / 9]) * 9 + lane % 9, ga);
    }
    store_net(G, j, 0, row, lane, D0, A0, z0);
    store_net(G, j, 1, row, lane, D1, A1, x1n);
    G.ACTX[((size_t)j * G.R + row) * 64 + lane] = ctx;
    G.DCTX[((size_t)j * G.R + row) * 64 + lane] = dctx;
}

// ---------------------------------------------------------------------------------------------------
// Weight-gradient reductions of the flow: C[b] = A[b]^T . B[b] over the row axis (dW_l = delta_l^T . a_{l-1}; train/train_humaniflow.py:351
// reaches them through loss.backward()), plus the column sums of A (the bias gradients).  A[b] is (rows, m) with row pitch lda, B[b] (rows, n)
// with pitch ldb, both fp32; a batch stride of 0 shares one operand across the batch (the context layers all multiply the same input rows).
// One wave per 32 x 32 tile of C: MFMA row i / column j of lane (i, q) are operand COLUMNS, so a k-step reads one row pair of A and of B with
// lane = column -- 128-byte coalesced loads, no transpose, no LDS.  rows <= a few hundred: a launch is latency-sized (tens of us), it replaces
// the library GEMMs (torch.matmul / einsum) the round-2 backward called here.
// ---------------------------------------------------------------------------------------------------
// Round 5: FOUR waves per tile, each a quarter of the row axis (the launch is latency-sized: one wave walking 576 rows is a chain of 288 dependent MFMAs
// behind their loads, 15-35 us x ~18 launches per training step); the four partial tiles meet in LDS and are summed in a fixed order.
__global__ __launch_bounds__(256) void k_atb_batched(const float* __restrict__ A, const float* __restrict__ Bm, float* __restrict__ Cm, float* __restrict__ asum,
                                                     int rows, int m, int n, int lda, int ldb, long long sA, long long sB, int mt, int nt) {
    __shared__ float part[3][16][64];
    __shared__ float psum[3][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lq = lane >> 5;
    int t = blockIdx.x;
    const int jn = t % nt; t /= nt;
    const int im = t % mt;
    const int b = t / mt;
    const float* a = A + (long long)b * sA;
    const float* bb = Bm + (long long)b * sB;
    const int ca = im * 32 + li, cb = jn * 32 + li;
    const bool va = ca < m, vb = cb < n;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float s = 0.f;
    constexpr int U = 8;                                         // k-steps (row pairs) in flight
    const int per = ((rows + 3) / 4 + 1) & ~1;                   // rows per wave, even: a k-step is a row pair
    const int r_begin = wave * per, r_end = min(rows, r_begin + per);
    for (int r0 = r_begin; r0 < r_end; r0 += 2 * U) {
        float fa[U], fb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = r0 + 2 * u + lq;
            fa[u] = (va && r < r_end) ? a[(long long)r * lda + ca] : 0.f;
            fb[u] = (vb && r < r_end) ? bb[(long long)r * ldb + cb] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[u], fb[u], acc, 0, 0, 0);
            s += fa[u];
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) part[wave - 1][r][lane] = acc[r];
        psum[wave - 1][lane] = s;
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = ((acc[r] + part[0][r][lane]) + part[1][r][lane]) + part[2][r][lane];
    s = ((s + psum[0][lane]) + psum[1][lane]) + psum[2][lane];
    // accumulator register r of lane (j = li, q): C row (r & 3) + 8 (r >> 2) + 4 q of the tile, column j
    float* c = Cm + (long long)b * m * n;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = im * 32 + (r & 3) + 8 * (r >> 2) + 4 * lq;
        if (row < m && vb) c[(long long)row * n + cb] = acc[r];
    }
    if (asum && jn == 0) {                                       // column sums of A: the two row parities of a column sit in lanes li and li + 32
        s += __shfl_xor(s, 32);
        if (lq == 0 && va) asum[(long long)b * m + ca] = s;
    }
}

}  // namespace hmf

using namespace hmf;

extern "C" {

static int rec_from(const hmf_flow_grad_records_t* r, GradRec& G, bool need_dctx) {
    if (!r || !r->D1 || !r->D2 || !r->D3 || !r->D4 || !r->A1 || !r->A2 || !r->A3 || !r->XIN || !r->ACTX || r->rows <= 0 || (need_dctx && !r->DCTX))
        return fail(HMF_ERR_INVALID, "hmf_flow_*_backward: incomplete gradient-record struct");
    G.D1 = r->D1; G.D2 = r->D2; G.D3 = r->D3; G.D4 = r->D4; G.A1 = r->A1; G.A2 = r->A2; G.A3 = r->A3; G.XIN = r->XIN; G.ACTX = r->ACTX; G.DCTX = r->DCTX;
    G.R = r->rows;
    return HMF_OK;
}

int hmf_flow_log_prob_SO3_backward(const hmf_flow_t* h, const float* ctx, const int32_t* part_ids_host, int P, const double* R, int B,
                                   const float* g_logp, float* g_ctx, const hmf_flow_grad_records_t* rec, void* stream) {
    if (!h || !ctx || !R || !g_logp || !g_ctx || B <= 0 || P <= 0 || P > kP) return fail(HMF_ERR_INVALID, "hmf_flow_log_prob_SO3_backward: bad args");
    if (!part_ids_host && P != kP) return fail(HMF_ERR_INVALID, "hmf_flow_log_prob_SO3_backward: part_ids NULL requires P == 23");
    const int* dev_ids = nullptr;
    if (part_ids_host) {
        for (int i = 0; i < P; ++i)
            if (part_ids_host[i] < 0 || part_ids_host[i] >= kP || part_ids_host[i] != part_ids_host[0] + i)
                return fail(HMF_ERR_UNSUPPORTED, "hmf_flow_log_prob_SO3_backward: part ids must be a contiguous ascending range inside [0, 23)");
        dev_ids = h->d_ids + part_ids_host[0];
    }
    if (h->g.enabled) return fail(HMF_ERR_UNSUPPORTED, "hmf_flow_log_prob_SO3_backward: gradients are built for the default flow configuration only");
    GradRec G;
    int rc = rec_from(rec, G, false);
    if (rc) return rc;
    if (G.R != 3 * B) return fail(HMF_ERR_INVALID, "hmf_flow_log_prob_SO3_backward: records need rows == 3 * B");
    const int nw = B * P;
    hipLaunchKernelGGL(k_flow_logprob_bwd, dim3(nw), dim3(192), 0, (hipStream_t)stream, h->d, ctx, dev_ids, P, R, B, g_logp, g_ctx, G);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

int hmf_flow_log_prob_so3_vjp(const hmf_flow_t* h, const float* ctx, const int32_t* part_ids_host, int P, const float* v, int B, const float* g_logp,
                              float* g_ctx, float* g_v, void* stream) {
    if (!h || !ctx || !v || !g_logp || !g_ctx || !g_v || B <= 0 || P <= 0 || P > kP) return fail(HMF_ERR_INVALID, "hmf_flow_log_prob_so3_vjp: bad args");
    if (!part_ids_host && P != kP) return fail(HMF_ERR_INVALID, "hmf_flow_log_prob_so3_vjp: part_ids NULL requires P == 23");
    const int* dev_ids = nullptr;
    if (part_ids_host) {
        for (int i = 0; i < P; ++i)
            if (part_ids_host[i] < 0 || part_ids_host[i] >= kP || part_ids_host[i] != part_ids_host[0] + i)
                return fail(HMF_ERR_UNSUPPORTED, "hmf_flow_log_prob_so3_vjp: part ids must be a contiguous ascending range inside [0, 23)");
        dev_ids = h->d_ids + part_ids_host[0];
    }
    if (h->g.enabled) return fail(HMF_ERR_UNSUPPORTED, "hmf_flow_log_prob_so3_vjp: built for the default flow configuration only");
    const int nw = B * P;
    hipLaunchKernelGGL(k_flow_logprob_so3_vjp, dim3((nw + 3) / 4), dim3(256), 0, (hipStream_t)stream, h->d, ctx, dev_ids, P, v, B, g_logp, g_ctx, g_v);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

int hmf_flow_sample_backward(const hmf_flow_t* h, const float* isgc_rows, const float* rotmats_samples, const float* eps_pose, int B, int N,
                             float* g_rotmats, const hmf_flow_grad_records_t* rec, void* stream) {
    if (!h || !isgc_rows || !rotmats_samples || !eps_pose || !g_rotmats || B <= 0 || N <= 0)
        return fail(HMF_ERR_INVALID, "hmf_flow_sample_backward: bad args");
    if (h->g.enabled) return fail(HMF_ERR_UNSUPPORTED, "hmf_flow_sample_backward: gradients are built for the default flow configuration only");
    GradRec G;
    int rc = rec_from(rec, G, true);
    if (rc) return rc;
    if (G.R != B * N) return fail(HMF_ERR_INVALID, "hmf_flow_sample_backward: records need rows == B * N");
    for (int lev = h->d.lev.nlev - 1; lev >= 0; --lev) {
        const int nw = B * N * (h->d.lev.start[lev + 1] - h->d.lev.start[lev]);
        hipLaunchKernelGGL(k_flow_chain_bwd, dim3(nw), dim3(64), 0, (hipStream_t)stream, h->d, lev, isgc_rows, rotmats_samples, eps_pose, B, N,
                           g_rotmats, G);
        HMF_LAUNCH_CHECK();
    }
    return HMF_OK;
}

int hmf_rows_atb_batched(const float* A, const float* B, float* C, float* a_colsum, int batch, int rows, int m, int n, int lda, int ldb,
                         long long batch_stride_a, long long batch_stride_b, void* stream) {
    if (!A || !B || !C || batch <= 0 || rows <= 0 || m <= 0 || n <= 0 || lda < m || ldb < n) return fail(HMF_ERR_INVALID, "hmf_rows_atb_batched: bad args");
    const int mt = (m + 31) / 32, nt = (n + 31) / 32;
    const long long blocks = (long long)batch * mt * nt;
    if (blocks >= (1LL << 31)) return fail(HMF_ERR_UNSUPPORTED, "hmf_rows_atb_batched: too many tiles");
    hipLaunchKernelGGL(k_atb_batched, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, A, B, C, a_colsum, rows, m, n, lda, ldb, batch_stride_a,
                       batch_stride_b, mt, nt);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

}  // extern "C"
