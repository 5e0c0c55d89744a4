// The heads between the encoder and the flow (models/humaniflow_model.py:232-252, models/resnet.py:214-215):
//
//   hmf_avgpool_nhwc   F.adaptive_avg_pool2d(x, 1) + flatten for maps whose pooling is not already done in the last convolution's epilogue
//                      (hmf_conv1x1_gemm with a pool output covers the 8 x 8 maps of the 256 x 256 input)
//   hmf_heads          x1 = ELU(fc1 f);  cam = fc_cam x1 + init_cam;  glob_R = rot6d_to_rotmat(fc_glob x1 + init_glob);
//                      (shape_mode | shape_log_std) = fc_shape x1;  shape_std = exp(shape_log_std)
//
// fc1 is the only layer with work in it: a (B x F) . (F x N1) product, F x N1 = 2048 x 1024 for ResNet-50: 8 MB of weights streamed once,
// 134 MFLOP at B = 32 -- bound by how fast ~128 work-groups can pull 64 KB each out of HBM, not by arithmetic.  k_heads_fc1 runs it on the fp32
// MFMA with the WEIGHT rows as the A operand straight from their natural (N1, F) layout (no packed copy: the parameters change under training
// and the kernel reads the live tensors) and the feature rows as the B operand; the reduction axis is split over (4 waves x KS work-groups)
// and combined in a fixed order, so results do not depend on scheduling.  k_heads_out finishes one image per work-group.
#include <algorithm>
#include "hmf_common.h"
#include "flow_common.h"
#include "so3_math.h"

namespace hmf {

// partial[ks][b][n] = sum over the ks-th K range of feats[b][k] * w1[n][k];   grid (N1 / 32, KS, ceil(B / 32)), 256 threads
__global__ __launch_bounds__(256) void k_heads_fc1(const float* __restrict__ feats, const float* __restrict__ w1, int B, int F, int N1, int Kw,
                                                   float* __restrict__ partial) {
    __shared__ float sred[4][32][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31, q = lane >> 5;
    const int n0 = blockIdx.x * 32, ks = blockIdx.y, b0 = blockIdx.z * 32;
    const int k0 = (ks * 4 + wave) * Kw;
    const float* wr = w1 + (size_t)(n0 + i) * F + k0 + 4 * q;
    const bool brow = b0 + i < B;
    const float* fr = feats + (size_t)(brow ? b0 + i : 0) * F + k0 + 4 * q;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int ng = Kw / 8;
    f32x4 a = *(const f32x4*)wr, x = brow ? *(const f32x4*)fr : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int g = 0; g < ng; ++g) {
        f32x4 an = a, xn = x;
        if (g + 1 < ng) {
            an = *(const f32x4*)(wr + (g + 1) * 8);
            if (brow) xn = *(const f32x4*)(fr + (g + 1) * 8);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], x[u], acc, 0, 0, 0);
        a = an; x = xn;
    }
    // acc[r] of lane (b = i, q) is output channel n = 8 (r >> 2) + 4 q + (r & 3)
#pragma unroll
    for (int r = 0; r < 16; ++r) sred[wave][8 * (r >> 2) + 4 * q + (r & 3)][i] = acc[r];
    __syncthreads();
    const int n = tid & 31;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int b = (tid >> 5) + 8 * s;
        if (b0 + b < B) {
            const float v = ((sred[0][n][b] + sred[1][n][b]) + sred[2][n][b]) + sred[3][n][b];
            partial[((size_t)ks * B + b0 + b) * N1 + n0 + n] = v;
        }
    }
}

struct HeadsOut {
    const float *w_cam, *b_cam, *init_cam, *w_glob, *b_glob, *init_glob, *w_shape, *b_shape;
    float *cam, *glob_R, *shape_mode, *shape_log_std, *shape_std;
    float *x1, *glob6;             // training forward only (else NULL): ELU(fc1) (B, N1) and the 6-D vector before Gram-Schmidt (B, 6), kept for hmf_heads_backward
};

// one work-group per image: x1 = ELU(sum of the K-split partials + b1) in LDS, then the 3 + 6 + 2 nb small outputs, one wave per output
__global__ __launch_bounds__(256) void k_heads_out(const float* __restrict__ partial, const float* __restrict__ b1, int B, int N1, int KS, int nb, HeadsOut o) {
    // the eight lanes of an output read eight contiguous eighths of x1: each eighth starts 4 floats later than its length, so that the eight 16-byte reads of
    // a step hit eight different bank groups (unpadded: 84 % of the kernel's LDS cycles were bank conflicts, round-4 PMC)
    __shared__ __attribute__((aligned(16))) float sx[1024 + 32];
    __shared__ float so[48];
    const int tid = threadIdx.x, b = blockIdx.x;
    for (int n = tid; n < N1; n += 256) {
        float v = partial[(size_t)b * N1 + n];
        for (int ks = 1; ks < KS; ++ks) v += partial[((size_t)ks * B + b) * N1 + n];
        const float xv = elu1(v + b1[n]);
        sx[n + 4 * (n / (N1 / 8))] = xv;
        if (o.x1) o.x1[(size_t)b * N1 + n] = xv;
    }
    __syncthreads();
    // 3 + 6 + 2 nb outputs: eight lanes per output, each lane a contiguous eighth of the N1 axis with all its loads independent (round 4, first
    // version: one wave per output, four dependent 16-byte iterations + a 6-step shuffle reduction each, outputs one after the other: 21 us for
    // 29 dot products of length 1024)
    const int NO = 9 + 2 * nb;
    for (int base = 0; base < NO; base += 32) {
        const int oi = base + (tid >> 3), part = tid & 7;
        float acc = 0.f;
        if (oi < NO) {
            const float* wrow = oi < 3 ? o.w_cam + (size_t)oi * N1 : (oi < 9 ? o.w_glob + (size_t)(oi - 3) * N1 : o.w_shape + (size_t)(oi - 9) * N1);
            const int n8 = N1 / 8;                                 // N1 % 32 == 0: a multiple of 4 floats per lane
            const f32x4* wp = (const f32x4*)(wrow + part * n8);
            const f32x4* xp = (const f32x4*)(sx + part * (n8 + 4));
            float a4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
            for (int k4 = 0; k4 < n8 / 4; ++k4) {
                const f32x4 w = wp[k4], x = xp[k4];
                a4[0] = fmaf(w[0], x[0], a4[0]); a4[1] = fmaf(w[1], x[1], a4[1]); a4[2] = fmaf(w[2], x[2], a4[2]); a4[3] = fmaf(w[3], x[3], a4[3]);
            }
            acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
        }
        acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 4);
        if (oi < NO && part == 0) so[oi] = acc;
    }
    __syncthreads();
    if (tid < 3) o.cam[(size_t)b * 3 + tid] = (so[tid] + o.b_cam[tid]) + o.init_cam[tid];                    // :237-238
    if (tid == 64) {                                                                                         // :243-245
        float g6[6], R[9];
#pragma unroll
        for (int e = 0; e < 6; ++e) g6[e] = (so[3 + e] + o.b_glob[e]) + o.init_glob[e];
        rot6d_to_rotmat_f32(g6, R);
#pragma unroll
        for (int e = 0; e < 9; ++e) o.glob_R[(size_t)b * 9 + e] = R[e];
        if (o.glob6)
#pragma unroll
            for (int e = 0; e < 6; ++e) o.glob6[(size_t)b * 6 + e] = g6[e];
    }
    if (tid >= 128 && tid < 128 + 2 * nb) {                                                                  // :250-253
        const int e = tid - 128;
        const float v = so[9 + e] + o.b_shape[e];
        if (e < nb) {
            o.shape_mode[(size_t)b * nb + e] = v;
        } else {
            o.shape_log_std[(size_t)b * nb + (e - nb)] = v;
            o.shape_std[(size_t)b * nb + (e - nb)] = expf(v);
        }
    }
}

// (B, HW, C) NHWC -> (B, C) mean over the HW pixels.  A work-group = 64 channel quads x 4 pixel quarters of one image (the quarters are added in
// a fixed order: deterministic); one thread per (image, 4 channels) walking all HW pixels was ONE work-group and 17 us for a single 8 x 8 x 2048
// map -- 3 % of the predict-shape latency.
__global__ __launch_bounds__(256) void k_avgpool_nhwc(const float* __restrict__ x, int B, int HW, int C, float* __restrict__ out) {
    __shared__ f32x4 part[4][64];
    const int c4n = C / 4, gpi = (c4n + 63) / 64;                       // work-groups per image
    const int b = blockIdx.x / gpi, c4 = (blockIdx.x - b * gpi) * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (c4 < c4n) {
        const f32x4* p = (const f32x4*)(x + (size_t)b * HW * C) + c4;
        const int i0 = (int)((long long)HW * q / 4), i1 = (int)((long long)HW * (q + 1) / 4);
        for (int i = i0; i < i1; ++i) s += p[(size_t)i * c4n];
    }
    part[q][threadIdx.x & 63] = s;
    __syncthreads();
    if (q == 0 && c4 < c4n) {
        const int l = threadIdx.x;
        ((f32x4*)(out + (size_t)b * C))[c4] = (((part[0][l] + part[1][l]) + part[2][l]) + part[3][l]) / (float)HW;
    }
}


// ---------------------------------------------------------------------------------------------------
// Backward of the heads for the training step (the reference differentiates models/humaniflow_model.py:232-253 with autograd: ~40 small ATen
// launches and a vendor GEMM per step).  k_heads_bwd: one thread per fc1 output channel n, the gradients of the 3 + 6 + 2 nb small outputs of
// every image in LDS (the 6-D -> rotation pull-back applied on the way in):
//   g_x1[b][n] = sum_o g_out[b][o] W_o[o][n];   g_pre[b][n] = g_x1 * ELU'(pre) (= x1 + 1 where x1 <= 0);   dW_o[o][n] = sum_b g_out[b][o] x1[b][n]
// g_pre leaves in both layouts -- (B, N1) for dW1 = g_pre^T feats and (N1, B) for g_feats = g_pre W1 -- so that both dense products are calls of
// hmf_rows_atb_batched without a transpose launch.  Images are summed in order: deterministic.
// ---------------------------------------------------------------------------------------------------
struct HeadsBwd {
    const float *x1, *glob6, *w_cam, *w_glob, *w_shape;
    const float *g_cam, *g_glob_R, *g_mode, *g_lstd;          // any may be NULL (= zero gradient)
    float *g_pre, *g_pre_t, *g_w_cam, *g_b_cam, *g_w_glob, *g_b_glob, *g_w_shape, *g_b_shape;
};
constexpr int kHbMaxB = 128, kHbMaxO = 9 + 32;
// block = 64 channels x 4 image quarters (wave q takes images b = q, q + 4, ...); the four partial dW_o meet in LDS and are added in a fixed order
__global__ __launch_bounds__(256) void k_heads_bwd(int B, int N1, int nb, HeadsBwd a) {
    __shared__ float go[kHbMaxB * kHbMaxO];
    __shared__ float pdw[3][kHbMaxO][64];
    const int NO = 9 + 2 * nb, tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
    for (int b = tid; b < B; b += 256) {
        float* g = go + b * NO;
#pragma unroll
        for (int e = 0; e < 3; ++e) g[e] = a.g_cam ? a.g_cam[b * 3 + e] : 0.f;
        float gx[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (a.g_glob_R) rot6d_to_rotmat_bwd_f32(a.glob6 + b * 6, a.g_glob_R + b * 9, gx);
#pragma unroll
        for (int e = 0; e < 6; ++e) g[3 + e] = gx[e];
        for (int e = 0; e < nb; ++e) {
            g[9 + e] = a.g_mode ? a.g_mode[b * nb + e] : 0.f;
            g[9 + nb + e] = a.g_lstd ? a.g_lstd[b * nb + e] : 0.f;
        }
    }
    __syncthreads();
    const int n = blockIdx.x * 64 + lane;
    if (blockIdx.x == 0)                                               // bias gradients of the small layers: column sums of g_out, images in order
        for (int o = tid; o < NO; o += 256) {
            float s = 0.f;
            for (int b = 0; b < B; ++b) s += go[b * NO + o];
            (o < 3 ? a.g_b_cam + o : (o < 9 ? a.g_b_glob + (o - 3) : a.g_b_shape + (o - 9)))[0] = s;
        }
    const bool live = n < N1;
    float w[kHbMaxO], dw[kHbMaxO];
#pragma unroll
    for (int o = 0; o < kHbMaxO; ++o) {
        dw[o] = 0.f;
        w[o] = (live && o < NO) ? (o < 3 ? a.w_cam[(size_t)o * N1 + n] : (o < 9 ? a.w_glob[(size_t)(o - 3) * N1 + n] : a.w_shape[(size_t)(o - 9) * N1 + n])) : 0.f;
    }
    if (live)
        for (int b = q; b < B; b += 4) {
            const float x = a.x1[(size_t)b * N1 + n];
            const float* g = go + b * NO;                               // wave-uniform: LDS broadcast
            float gx0 = 0.f, gx1 = 0.f;
#pragma unroll
            for (int o = 0; o < kHbMaxO; ++o)
                if (o < NO) {
                    const float gv = g[o];
                    if (o & 1) gx1 = fmaf(gv, w[o], gx1); else gx0 = fmaf(gv, w[o], gx0);
                    dw[o] = fmaf(gv, x, dw[o]);
                }
            const float gx = gx0 + gx1;
            const float gp = x > 0.f ? gx : gx * (x + 1.f);             // ELU' = 1 (pre > 0) or exp(pre) = x1 + 1
            a.g_pre[(size_t)b * N1 + n] = gp;
            a.g_pre_t[(size_t)n * B + b] = gp;
        }
    if (q > 0)
#pragma unroll
        for (int o = 0; o < kHbMaxO; ++o)
            if (o < NO) pdw[q - 1][o][lane] = dw[o];
    __syncthreads();
    if (q == 0 && live)
#pragma unroll
        for (int o = 0; o < kHbMaxO; ++o)
            if (o < NO)
                (o < 3 ? a.g_w_cam + (size_t)o * N1 : (o < 9 ? a.g_w_glob + (size_t)(o - 3) * N1 : a.g_w_shape + (size_t)(o - 9) * N1))[n] =
                    ((dw[o] + pdw[0][o][lane]) + pdw[1][o][lane]) + pdw[2][o][lane];
}

static int heads_ksplit(int F) {
    for (int ks = 4; ks >= 1; ks >>= 1)
        if (F % (ks * 4 * 8) == 0) return ks;
    return 0;
}

}  // namespace hmf

using namespace hmf;

extern "C" {

size_t hmf_heads_workspace_bytes(int B, int F, int N1) {
    const int ks = heads_ksplit(F);
    return (size_t)std::max(ks, 1) * std::max(B, 1) * N1 * sizeof(float);
}

static int heads_impl(const float* feats, int B, int F, const float* w1, const float* b1, int N1, const float* w_cam, const float* b_cam, const float* init_cam,
              const float* w_glob, const float* b_glob, const float* init_glob, const float* w_shape, const float* b_shape, int nb, float* cam,
              float* glob_R, float* shape_mode, float* shape_log_std, float* shape_std, float* x1, float* glob6, void* workspace, size_t workspace_bytes, void* stream) {
    if (!feats || !w1 || !b1 || !w_cam || !b_cam || !init_cam || !w_glob || !b_glob || !init_glob || !w_shape || !b_shape || !cam || !glob_R || !shape_mode ||
        !shape_log_std || !shape_std)
        return fail(HMF_ERR_INVALID, "hmf_heads: null argument");
    if (B <= 0 || nb < 1 || nb > 16) return fail(HMF_ERR_INVALID, "hmf_heads: B=%d nb=%d", B, nb);
    const int KS = heads_ksplit(F);
    if (KS == 0 || N1 % 32 || N1 > 1024 || N1 % 4)
        return fail(HMF_ERR_UNSUPPORTED, "hmf_heads: F=%d must be a multiple of 32 and N1=%d a multiple of 32, <= 1024", F, N1);
    if (((uintptr_t)feats & 15) || ((uintptr_t)w1 & 15) || ((uintptr_t)w_cam & 15) || ((uintptr_t)w_glob & 15) || ((uintptr_t)w_shape & 15))
        return fail(HMF_ERR_INVALID, "hmf_heads: feats and the weight matrices must be 16-byte aligned");
    const size_t need = hmf_heads_workspace_bytes(B, F, N1);
    if (!workspace || workspace_bytes < need) return fail(HMF_ERR_WORKSPACE, "hmf_heads: workspace %zu < %zu bytes", workspace_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)workspace;
    ProfScope ps(15, B, st);
    hipLaunchKernelGGL(k_heads_fc1, dim3(N1 / 32, KS, (B + 31) / 32), dim3(256), 0, st, feats, w1, B, F, N1, F / (KS * 4), partial);
    HMF_LAUNCH_CHECK();
    HeadsOut o{w_cam, b_cam, init_cam, w_glob, b_glob, init_glob, w_shape, b_shape, cam, glob_R, shape_mode, shape_log_std, shape_std, x1, glob6};
    hipLaunchKernelGGL(k_heads_out, dim3(B), dim3(256), 0, st, partial, b1, B, N1, KS, nb, o);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

int hmf_heads(const float* feats, int B, int F, const float* w1, const float* b1, int N1, const float* w_cam, const float* b_cam, const float* init_cam,
              const float* w_glob, const float* b_glob, const float* init_glob, const float* w_shape, const float* b_shape, int nb, float* cam,
              float* glob_R, float* shape_mode, float* shape_log_std, float* shape_std, void* workspace, size_t workspace_bytes, void* stream) {
    return heads_impl(feats, B, F, w1, b1, N1, w_cam, b_cam, init_cam, w_glob, b_glob, init_glob, w_shape, b_shape, nb, cam, glob_R, shape_mode, shape_log_std,
                      shape_std, nullptr, nullptr, workspace, workspace_bytes, stream);
}

int hmf_heads_train(const float* feats, int B, int F, const float* w1, const float* b1, int N1, const float* w_cam, const float* b_cam, const float* init_cam,
                    const float* w_glob, const float* b_glob, const float* init_glob, const float* w_shape, const float* b_shape, int nb, float* cam,
                    float* glob_R, float* shape_mode, float* shape_log_std, float* shape_std, float* x1, float* glob6, void* workspace, size_t workspace_bytes,
                    void* stream) {
    if (!x1 || !glob6) return fail(HMF_ERR_INVALID, "hmf_heads_train: null x1 / glob6");
    return heads_impl(feats, B, F, w1, b1, N1, w_cam, b_cam, init_cam, w_glob, b_glob, init_glob, w_shape, b_shape, nb, cam, glob_R, shape_mode, shape_log_std,
                      shape_std, x1, glob6, workspace, workspace_bytes, stream);
}

int hmf_heads_backward(const float* x1, const float* glob6, int B, int N1, int nb, const float* w_cam, const float* w_glob, const float* w_shape,
                       const float* g_cam, const float* g_glob_R, const float* g_shape_mode, const float* g_shape_log_std, float* g_pre, float* g_pre_t,
                       float* g_w_cam, float* g_b_cam, float* g_w_glob, float* g_b_glob, float* g_w_shape, float* g_b_shape, void* stream) {
    if (!x1 || !glob6 || !w_cam || !w_glob || !w_shape || !g_pre || !g_pre_t || !g_w_cam || !g_b_cam || !g_w_glob || !g_b_glob || !g_w_shape || !g_b_shape)
        return fail(HMF_ERR_INVALID, "hmf_heads_backward: null argument");
    if (B <= 0 || B > kHbMaxB || nb < 1 || nb > 16 || N1 <= 0) return fail(HMF_ERR_UNSUPPORTED, "hmf_heads_backward: B=%d (<= %d), nb=%d, N1=%d", B, kHbMaxB, nb, N1);
    HeadsBwd a{x1, glob6, w_cam, w_glob, w_shape, g_cam, g_glob_R, g_shape_mode, g_shape_log_std, g_pre, g_pre_t, g_w_cam, g_b_cam, g_w_glob, g_b_glob, g_w_shape, g_b_shape};
    hipLaunchKernelGGL(k_heads_bwd, dim3((N1 + 63) / 64), dim3(256), 0, (hipStream_t)stream, B, N1, nb, a);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

int hmf_avgpool_nhwc(const float* x, int B, int HW, int C, float* out, void* stream) {
    if (!x || !out || B <= 0 || HW <= 0 || C <= 0 || C % 4) return fail(HMF_ERR_INVALID, "hmf_avgpool_nhwc: bad args (C %% 4 == 0)");
    if (((uintptr_t)x & 15) || ((uintptr_t)out & 15)) return fail(HMF_ERR_INVALID, "hmf_avgpool_nhwc: pointers must be 16-byte aligned");
    const int gpi = (C / 4 + 63) / 64;
    hipLaunchKernelGGL(k_avgpool_nhwc, dim3(B * gpi), dim3(256), 0, (hipStream_t)stream, x, B, HW, C, out);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

}  // extern "C"
