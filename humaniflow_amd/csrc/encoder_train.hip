// Training-mode pieces of the ResNet encoder (models/resnet.py with nn.BatchNorm2d in train(): batch statistics, running-stat update)
// and the backward passes the training step needs (train/train_humaniflow.py:248-347 back-propagates through the whole encoder).
//
//   forward   conv (the inference kernels, bias-free, no ReLU)  ->  hmf_bn_stats (+ hmf_bn_finalize)  ->  hmf_bn_apply (+ residual, ReLU)
//   backward  hmf_bn_bwd_reduce (+ hmf_bn_bwd_finalize)  ->  hmf_bn_bwd_apply  ->  hmf_conv_wgrad  and  dgrad
//
// dgrad needs no kernel of its own: the gradient w.r.t. a convolution's input is a convolution of the output gradient with the
// flipped / transposed filter, i.e. hmf_conv2d_nhwc again (stride-2 layers through a zero-inserted gradient map); the host side
// (humaniflow_amd/encoder_train.py) prepares those filters.  The weight gradient is a GEMM whose reduction axis is the PIXEL axis
// (dW[co][tap][ci] = sum_m gy[m][co] x[m @ tap][ci]): with NHWC tensors both MFMA operands are rows of 32 consecutive floats per pixel,
// so they stream straight from L2 into the operand registers (no LDS, no transposition); a wave owns a 64 x 64 tile of dW and a slice
// of the pixel axis, slices are combined with float atomics.
#include <algorithm>
#include "hmf_common.h"

namespace hmf {

// ---------------------------------------------------------------------------------------------------
// BatchNorm, training mode.  y: [M][C] (NHWC rows), C % 64 == 0.
// ---------------------------------------------------------------------------------------------------
constexpr int kBnRows = 512;      // rows per block of the reductions

// acc[0][C] += sum_m y, acc[1][C] += sum_m y^2 (double).  Block = 16 row lanes x 16 float4 lanes = 64 channels.
__global__ __launch_bounds__(256) void k_bn_stats(const float* __restrict__ y, int M, int C, double* __restrict__ acc) {
    __shared__ float red[2][16][64 + 1];
    const int c16 = threadIdx.x & 15, r16 = threadIdx.x >> 4;
    const int c0 = blockIdx.x * 64 + c16 * 4;
    const int m0 = blockIdx.y * kBnRows, m1 = min(m0 + kBnRows, M);
    float s[4] = {0.f, 0.f, 0.f, 0.f}, ss[4] = {0.f, 0.f, 0.f, 0.f};
    for (int m = m0 + r16; m < m1; m += 16) {
        const f32x4 v = *(const f32x4*)(y + (size_t)m * C + c0);
#pragma unroll
        for (int e = 0; e < 4; ++e) { s[e] += v[e]; ss[e] = fmaf(v[e], v[e], ss[e]); }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[0][r16][c16 * 4 + e] = s[e]; red[1][r16][c16 * 4 + e] = ss[e]; }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int which = threadIdx.x >> 6, c = threadIdx.x & 63;
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += (double)red[which][r][c];
        atomicAdd(acc + (size_t)which * C + blockIdx.x * 64 + c, t);
    }
}

// mean / biased variance -> scale = gamma * rstd, shift = beta - mean * scale; running statistics as torch (momentum, unbiased variance)
__global__ void k_bn_finalize(const double* __restrict__ acc, int M, int C, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                              float momentum, float* __restrict__ running_mean, float* __restrict__ running_var, float* __restrict__ mean,
                              float* __restrict__ rstd, float* __restrict__ scale, float* __restrict__ shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double mu = acc[c] / M;
    const double var = fmax(acc[C + c] / M - mu * mu, 0.0);
    const float rs = (float)(1.0 / sqrt(var + (double)eps));
    mean[c] = (float)mu;
    rstd[c] = rs;
    const float sc = gamma[c] * rs;
    scale[c] = sc;
    shift[c] = beta[c] - (float)mu * sc;
    if (running_mean) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(var * ((double)M / (double)max(M - 1, 1)));
    }
}

// out = relu?(y * scale[c] + shift[c] (+ res)), float4 per thread
__global__ __launch_bounds__(256) void k_bn_apply(const f32x4* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift,
                                                  const f32x4* __restrict__ res, f32x4* __restrict__ out, long long n4, int C4, int relu) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C4) * 4;
        const f32x4 sc = *(const f32x4*)(scale + c), sh = *(const f32x4*)(shift + c);
        f32x4 v = y[i] * sc + sh;
        if (res) v += res[i];
        if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        out[i] = v;
    }
}

// acc[0][C] += sum_m gz, acc[1][C] += sum_m gz * xhat,  gz = g * (out > 0 if relu), xhat = (y - mean) * rstd
__global__ __launch_bounds__(256) void k_bn_bwd_reduce(const float* __restrict__ g, const float* __restrict__ out, const float* __restrict__ y,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd, int M, int C, int relu,
                                                       double* __restrict__ acc) {
    __shared__ float red[2][16][64 + 1];
    const int c16 = threadIdx.x & 15, r16 = threadIdx.x >> 4;
    const int c0 = blockIdx.x * 64 + c16 * 4;
    const int m0 = blockIdx.y * kBnRows, m1 = min(m0 + kBnRows, M);
    const f32x4 mu = *(const f32x4*)(mean + c0), rs = *(const f32x4*)(rstd + c0);
    float s[4] = {0.f, 0.f, 0.f, 0.f}, sx[4] = {0.f, 0.f, 0.f, 0.f};
    for (int m = m0 + r16; m < m1; m += 16) {
        const size_t o = (size_t)m * C + c0;
        f32x4 gv = *(const f32x4*)(g + o);
        const f32x4 yv = *(const f32x4*)(y + o);
        if (relu) {
            const f32x4 ov = *(const f32x4*)(out + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) gv[e] = ov[e] > 0.f ? gv[e] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { s[e] += gv[e]; sx[e] = fmaf(gv[e], (yv[e] - mu[e]) * rs[e], sx[e]); }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[0][r16][c16 * 4 + e] = s[e]; red[1][r16][c16 * 4 + e] = sx[e]; }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int which = threadIdx.x >> 6, c = threadIdx.x & 63;
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += (double)red[which][r][c];
        atomicAdd(acc + (size_t)which * C + blockIdx.x * 64 + c, t);
    }
}

// g_gamma = sum gz xhat, g_beta = sum gz; coefficients of the input gradient: gy = a * gz + b * y + c0 with
// a = gamma rstd, b = -gamma rstd^2 sgx / M, c0 = -a sg / M - b mean      (gy = gamma rstd (gz - sg/M - xhat sgx/M))
__global__ void k_bn_bwd_finalize(const double* __restrict__ acc, int M, int C, const float* __restrict__ gamma, const float* __restrict__ mean,
                                  const float* __restrict__ rstd, float* __restrict__ g_gamma, float* __restrict__ g_beta, float* __restrict__ ca,
                                  float* __restrict__ cb, float* __restrict__ cc) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double sg = acc[c], sgx = acc[C + c];
    g_beta[c] = (float)sg;
    g_gamma[c] = (float)sgx;
    const double a = (double)gamma[c] * rstd[c];
    const double b = -a * rstd[c] * sgx / M;
    ca[c] = (float)a;
    cb[c] = (float)b;
    cc[c] = (float)(-a * sg / M - b * mean[c]);
}

// gy = a gz + b y + c (per channel); gz_out (optional) = the ReLU-masked gradient, which is also the identity branch's gradient
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const f32x4* __restrict__ g, const f32x4* __restrict__ out, const f32x4* __restrict__ y,
                                                      const float* __restrict__ ca, const float* __restrict__ cb, const float* __restrict__ cc,
                                                      f32x4* __restrict__ gy, f32x4* __restrict__ gz_out, long long n4, int C4, int relu) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C4) * 4;
        const f32x4 a = *(const f32x4*)(ca + c), b = *(const f32x4*)(cb + c), k = *(const f32x4*)(cc + c);
        f32x4 gv = g[i];
        if (relu) {
            const f32x4 ov = out[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) gv[e] = ov[e] > 0.f ? gv[e] : 0.f;
        }
        if (gz_out) gz_out[i] = gv;
        gy[i] = a * gv + b * y[i] + k;
    }
}

// ---------------------------------------------------------------------------------------------------
// max-pool 3x3 / stride 2 / pad 1 backward (NHWC): the gradient goes to the FIRST maximum of each window in (kh, kw) scan order
// (what torch's forward records with its strict `>` comparison).  Gather form: every input pixel checks the <= 4 windows covering it.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_maxpool_bwd(const f32x4* __restrict__ x, const f32x4* __restrict__ gout, f32x4* __restrict__ gx, int B, int H,
                                                     int W, int C4, int OH, int OW) {
    const long long n = (long long)B * H * W * C4;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const int c = (int)(e % C4);
        long long p = e / C4;
        const int iw = (int)(p % W);
        p /= W;
        const int ih = (int)(p % H), b = (int)(p / H);
        const f32x4 xv = x[e];
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int oh = (ih - 1 + 1) / 2; oh <= (ih + 1) / 2; ++oh) {          // windows with |2 oh - ih| <= 1
            if (oh < 0 || oh >= OH) continue;
            for (int ow = (iw - 1 + 1) / 2; ow <= (iw + 1) / 2; ++ow) {
                if (ow < 0 || ow >= OW) continue;
                // first maximum of window (oh, ow) per channel
                f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                int bi[4] = {-1, -1, -1, -1};
                for (int dh = -1; dh <= 1; ++dh) {
                    const int jh = 2 * oh + dh;
                    if ((unsigned)jh >= (unsigned)H) continue;
                    for (int dw = -1; dw <= 1; ++dw) {
                        const int jw = 2 * ow + dw;
                        if ((unsigned)jw >= (unsigned)W) continue;
                        const f32x4 v = x[(((long long)b * H + jh) * W + jw) * C4 + c];
                        const int id = (dh + 1) * 3 + (dw + 1);
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (v[k] > best[k]) { best[k] = v[k]; bi[k] = id; }
                    }
                }
                const int me = (ih - 2 * oh + 1) * 3 + (iw - 2 * ow + 1);
                const f32x4 gv = gout[(((long long)b * OH + oh) * OW + ow) * C4 + c];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (bi[k] == me) acc[k] += gv[k];
            }
        }
        (void)xv;
        gx[e] = acc;
    }
}

// ---------------------------------------------------------------------------------------------------
// weight gradient.  gy [M][Cout] (M = B OH OW), x NHWC [B][H][W][Cin] (STEM: the NCHW proxy input [B][Cin][H][W]),
// dw [Cout][K]: K = KS*KS*Cin in (kh, kw, ci) order = channels_last filter layout (STEM: (ci, kh, kw) order = torch's native layout).
// One work-group of 4 waves per (64 x 64 tile of dw, slice of the pixel axis): each wave multiplies a quarter of the slice into its own
// accumulators, the four partial tiles are summed through LDS and added to dw with one atomic pass; grid = (tiles, slices).
// ---------------------------------------------------------------------------------------------------
struct WgradArgs {
    const float* gy;
    const float* x;
    float* dw;
    int B, H, W, Cin, OH, OW, Cout, KS, stride, pad, M, K, ktiles, rows_per_slice;
};

template <bool STEM>
__global__ __launch_bounds__(256) void k_conv_wgrad(const WgradArgs a) {
    __shared__ float part[3][64 * 64];                       // partial tiles of waves 1..3 (wave 0 keeps its own in registers)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lq = lane >> 5;
    const int ct = blockIdx.x / a.ktiles, kt = blockIdx.x - ct * a.ktiles;
    const int co0 = ct * 64, k0 = kt * 64;
    // the block owns rows_per_slice pixels, each of its 4 waves a quarter of them (multiples of 8 pixels)
    const int q4 = a.rows_per_slice / 4;
    const int m_begin = blockIdx.y * a.rows_per_slice + wave * q4, m_end = min(m_begin + q4, a.M);
    // ---- per-lane description of its two B-operand columns (kcol = k0 + 32 tb + li) ----
    int kh[2], kw[2], cbase[2];          // filter tap and the float offset of the channel inside a pixel (NHWC) / plane (STEM)
    bool kvalid[2];
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
        const int kcol = k0 + 32 * tb + li;
        kvalid[tb] = kcol < a.K;
        const int kc = kvalid[tb] ? kcol : 0;
        if (STEM) {
            const int ci = kc / (a.KS * a.KS), r = kc - ci * a.KS * a.KS;
            kh[tb] = r / a.KS; kw[tb] = r - kh[tb] * a.KS;
            cbase[tb] = ci * a.H * a.W;
        } else {
            const int tap = kc / a.Cin;
            kh[tb] = tap / a.KS; kw[tb] = tap - kh[tb] * a.KS;
            cbase[tb] = kc - tap * a.Cin;
        }
    }
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.B * a.H * a.W * a.Cin * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)a.gy, 0, a.M * a.Cout * 4, 0x00020000);
    constexpr int kOOB = (int)0x80000000;
    // this lane's pixel m = ms + lq; decode once, then advance by 2 per k-step
    int m = m_begin + lq;
    int b = m / (a.OH * a.OW), r0 = m - b * a.OH * a.OW;
    int oh = r0 / a.OW, ow = r0 - oh * a.OW;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    constexpr int U = 4;                   // k-steps per group: 16 operand loads in flight ahead of 16 MFMAs
    float fa[2][U][2], fb[2][U][2];
    auto load_group = [&](int slot) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool live = m < m_end;
            const int goff = live ? (m * a.Cout + co0 + li) * 4 : kOOB;
            fa[slot][u][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, goff, 0, 0));
            fa[slot][u][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, live ? goff + 128 : kOOB, 0, 0));
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) {
                const int ih = oh * a.stride + kh[tb] - a.pad, iw = ow * a.stride + kw[tb] - a.pad;
                const bool ok = live & kvalid[tb] & ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W);
                int off;
                if (STEM) off = (b * a.Cin * a.H * a.W + cbase[tb] + ih * a.W + iw) * 4;
                else off = (((b * a.H + ih) * a.W + iw) * a.Cin + cbase[tb]) * 4;
                fb[slot][u][tb] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, ok ? off : kOOB, 0, 0));
            }
            m += 2;
            ow += 2;
            while (ow >= a.OW) {
                ow -= a.OW;
                if (++oh >= a.OH) { oh = 0; ++b; }
            }
        }
    };
    const int ngroups = m_end > m_begin ? (m_end - m_begin + 2 * U - 1) / (2 * U) : 0;
    if (ngroups > 0) load_group(0);
    for (int g = 0; g < ngroups; ++g) {
        const int slot = g & 1;
        if (g + 1 < ngroups) load_group(slot ^ 1);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                for (int tb = 0; tb < 2; ++tb)
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[slot][u][ta], fb[slot][u][tb], acc[ta][tb], 0, 0, 0);
    }
    // ---- combine: the four waves of the block through LDS, then ONE atomic pass per block over the pixel slices (128-byte runs) ----
    // tile element (ta, tb, r) of lane (li, lq) lives at part[..][((ta*2 + tb)*16 + r)*64 + lane]
    if (wave > 0) {
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                for (int r = 0; r < 16; ++r) part[wave - 1][((ta * 2 + tb) * 16 + r) * 64 + lane] = acc[ta][tb][r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) {
                const int kcol = k0 + 32 * tb + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int e = ((ta * 2 + tb) * 16 + r) * 64 + lane;
                    const float v = acc[ta][tb][r] + part[0][e] + part[1][e] + part[2][e];
                    const int co = co0 + 32 * ta + (r & 3) + 8 * (r >> 2) + 4 * lq;
                    if (kcol < a.K && co < a.Cout) atomicAdd(a.dw + (size_t)co * a.K + kcol, v);
                }
            }
    }
}

}  // namespace hmf

using namespace hmf;

extern "C" {

int hmf_bn_stats(const float* y, int M, int C, double* acc, void* stream) {
    if (!y || !acc || M <= 0 || C <= 0 || C % 64) return fail(HMF_ERR_INVALID, "hmf_bn_stats: bad args (C %% 64 == 0 required)");
    hipLaunchKernelGGL(k_bn_stats, dim3(C / 64, (M + kBnRows - 1) / kBnRows), dim3(256), 0, (hipStream_t)stream, y, M, C, acc);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

int hmf_bn_finalize(const double* acc, int M, int C, const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                    float* running_var, float* mean, float* rstd, float* scale, float* shift, void* stream) {
    if (!acc || !gamma || !beta || !mean || !rstd || !scale || !shift || M <= 0 || C <= 0) return fail(HMF_ERR_INVALID, "hmf_bn_finalize: bad args");
    hipLaunchKernelGGL(k_bn_finalize, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, acc, M, C, gamma, beta, eps, momentum, running_mean,
                       running_var, mean, rstd, scale, shift);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

int hmf_bn_apply(const float* y, const float* scale, const float* shift, const float* residual, float* out, int M, int C, int relu, void* stream) {
    if (!y || !scale || !shift || !out || M <= 0 || C <= 0 || C % 4) return fail(HMF_ERR_INVALID, "hmf_bn_apply: bad args");
    const long long n4 = (long long)M * C / 4;
    long long blocks = std::min<long long>((n4 + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(k_bn_apply, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const f32x4*)y, scale, shift, (const f32x4*)residual,
                       (f32x4*)out, n4, C / 4, relu);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

int hmf_bn_bwd_reduce(const float* g, const float* out, const float* y, const float* mean, const float* rstd, int M, int C, int relu, double* acc,
                      void* stream) {
    if (!g || !y || !mean || !rstd || !acc || (relu && !out) || M <= 0 || C <= 0 || C % 64) return fail(HMF_ERR_INVALID, "hmf_bn_bwd_reduce: bad args");
    hipLaunchKernelGGL(k_bn_bwd_reduce, dim3(C / 64, (M + kBnRows - 1) / kBnRows), dim3(256), 0, (hipStream_t)stream, g, out, y, mean, rstd, M, C, relu, acc);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

int hmf_bn_bwd_finalize(const double* acc, int M, int C, const float* gamma, const float* mean, const float* rstd, float* g_gamma, float* g_beta,
                        float* ca, float* cb, float* cc, void* stream) {
    if (!acc || !gamma || !mean || !rstd || !g_gamma || !g_beta || !ca || !cb || !cc || M <= 0 || C <= 0) return fail(HMF_ERR_INVALID, "hmf_bn_bwd_finalize: bad args");
    hipLaunchKernelGGL(k_bn_bwd_finalize, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, acc, M, C, gamma, mean, rstd, g_gamma, g_beta, ca, cb, cc);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

int hmf_bn_bwd_apply(const float* g, const float* out, const float* y, const float* ca, const float* cb, const float* cc, float* gy, float* gz_out,
                     int M, int C, int relu, void* stream) {
    if (!g || !y || !ca || !cb || !cc || !gy || (relu && !out) || M <= 0 || C <= 0 || C % 4) return fail(HMF_ERR_INVALID, "hmf_bn_bwd_apply: bad args");
    const long long n4 = (long long)M * C / 4;
    long long blocks = std::min<long long>((n4 + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(k_bn_bwd_apply, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const f32x4*)g, (const f32x4*)out, (const f32x4*)y, ca, cb,
                       cc, (f32x4*)gy, (f32x4*)gz_out, n4, C / 4, relu);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

int hmf_maxpool3x3s2_nhwc_backward(const float* x, const float* g_out, float* g_x, int B, int H, int W, int C, void* stream) {
    if (!x || !g_out || !g_x || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4) return fail(HMF_ERR_INVALID, "hmf_maxpool3x3s2_nhwc_backward: bad args");
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    const long long n = (long long)B * H * W * (C / 4);
    long long blocks = std::min<long long>((n + 255) / 256, 256 * 64);
    hipLaunchKernelGGL(k_maxpool_bwd, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const f32x4*)x, (const f32x4*)g_out, (f32x4*)g_x, B, H, W,
                       C / 4, OH, OW);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

int hmf_conv_wgrad(const float* gy, const float* x, float* dw, int B, int H, int W, int Cin, int Cout, int ks, int stride, int stem_nchw, void* stream) {
    if (!gy || !x || !dw || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return fail(HMF_ERR_INVALID, "hmf_conv_wgrad: bad args");
    if (!stem_nchw && Cin % 32) return fail(HMF_ERR_UNSUPPORTED, "hmf_conv_wgrad: Cin=%d must be a multiple of 32 (NHWC path)", Cin);
    if (Cout % 32) return fail(HMF_ERR_UNSUPPORTED, "hmf_conv_wgrad: Cout=%d must be a multiple of 32", Cout);
    WgradArgs a;
    a.gy = gy; a.x = x; a.dw = dw; a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.KS = ks; a.stride = stride; a.pad = ks / 2;
    a.OH = (H + 2 * a.pad - ks) / stride + 1;
    a.OW = (W + 2 * a.pad - ks) / stride + 1;
    const long long M = (long long)B * a.OH * a.OW;
    if (M * Cout >= (1LL << 29) || (long long)B * H * W * Cin >= (1LL << 29)) return fail(HMF_ERR_UNSUPPORTED, "hmf_conv_wgrad: tensor exceeds 2^29 elements");
    a.M = (int)M;
    a.K = ks * ks * Cin;
    a.ktiles = (a.K + 63) / 64;
    const int tiles = a.ktiles * ((Cout + 63) / 64);
    // ~6 work-groups (4 waves each, 48 KB of LDS) per CU; a block's pixel slice is 4 wave quarters of a multiple of 8 pixels
    int slices = std::max(1, std::min((int)((M + 255) / 256), (256 * 6 + tiles - 1) / tiles));
    a.rows_per_slice = (int)(((M + slices - 1) / slices + 31) / 32 * 32);
    slices = (int)((M + a.rows_per_slice - 1) / a.rows_per_slice);
    hipStream_t st = (hipStream_t)stream;
    HMF_HIP(hipMemsetAsync(dw, 0, (size_t)Cout * a.K * sizeof(float), st));
    if (stem_nchw) hipLaunchKernelGGL(k_conv_wgrad<true>, dim3(tiles, slices), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_conv_wgrad<false>, dim3(tiles, slices), dim3(256), 0, st, a);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

}  // extern "C"
