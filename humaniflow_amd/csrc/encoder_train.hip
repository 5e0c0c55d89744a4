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
    // eight independent 16-byte loads per thread in flight: with one (round 2) the 576 work-groups of a layer1 map held ~9 KB per CU in flight and
    // the reduction ran at 2.8-3.2 TB/s
    constexpr int U = 8;
    int m = m0 + r16;
    for (; m + 16 * (U - 1) < m1; m += 16 * U) {
        f32x4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = *(const f32x4*)(y + (size_t)(m + 16 * k) * C + c0);
#pragma unroll
        for (int k = 0; k < U; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) { s[e] += v[k][e]; ss[e] = fmaf(v[k][e], v[k][e], ss[e]); }
    }
    for (; m < m1; m += 16) {
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
        if (relu) {                                    // NaN-propagating, like torch.relu (fmaxf alone would turn a NaN into 0 and hide it from nan_policy='rollback')
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = (v[k] != v[k]) ? v[k] : fmaxf(v[k], 0.f);
        }
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
    constexpr int U = 4;                                              // 8 - 12 independent loads per thread in flight (see k_bn_stats)
    int m = m0 + r16;
    for (; m + 16 * (U - 1) < m1; m += 16 * U) {
        f32x4 gv[U], yv[U], ov[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const size_t o = (size_t)(m + 16 * k) * C + c0;
            gv[k] = *(const f32x4*)(g + o);
            yv[k] = *(const f32x4*)(y + o);
            if (relu) ov[k] = *(const f32x4*)(out + o);
        }
#pragma unroll
        for (int k = 0; k < U; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gg = (relu && !(ov[k][e] > 0.f)) ? 0.f : gv[k][e];
                s[e] += gg;
                sx[e] = fmaf(gg, (yv[k][e] - mu[e]) * rs[e], sx[e]);
            }
    }
    for (; m < m1; m += 16) {
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
// (what torch's forward records with its strict `>` comparison).  Gather form, one thread per 2 x 2 block of input pixels (x 4 channels): the four
// windows (a, b), (a, b+1), (a+1, b), (a+1, b+1) that cover the block live in a 5 x 5 neighbourhood held in registers, so an input element costs
// 6.25 loads instead of the 20 of a per-pixel gather (every pixel re-deriving the arg-max of each of its <= 4 windows); deterministic, no atomics.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_maxpool_bwd(const f32x4* __restrict__ x, const f32x4* __restrict__ gout, f32x4* __restrict__ gx, int B, int H,
                                                     int W, int C4, int OH, int OW) {
    const int HB = (H + 1) / 2, WB = (W + 1) / 2;
    const long long n = (long long)B * HB * WB * C4;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const int c = (int)(e % C4);
        long long p = e / C4;
        const int bb = (int)(p % WB);
        p /= WB;
        const int a = (int)(p % HB), b = (int)(p / HB);
        // neighbourhood rows 2a-1 .. 2a+3, columns 2bb-1 .. 2bb+3
        f32x4 v[5][5];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int jh = 2 * a - 1 + i;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int jw = 2 * bb - 1 + j;
                const bool in = ((unsigned)jh < (unsigned)H) & ((unsigned)jw < (unsigned)W);
                const f32x4 ninf = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                v[i][j] = in ? x[(((long long)b * H + jh) * W + jw) * C4 + c] : ninf;
            }
        }
        f32x4 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int da = 0; da < 2; ++da)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const int oh = a + da, ow = bb + db;
                if (oh >= OH || ow >= OW) continue;
                const f32x4 gv = gout[(((long long)b * OH + oh) * OW + ow) * C4 + c];
                // first maximum of window (oh, ow): local rows 2 da .. 2 da + 2, columns 2 db .. 2 db + 2
                f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                int bi[4] = {-1, -1, -1, -1};
#pragma unroll
                for (int dh = 0; dh < 3; ++dh)
#pragma unroll
                    for (int dw = 0; dw < 3; ++dw) {
                        const f32x4 u = v[2 * da + dh][2 * db + dw];
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (u[k] > best[k]) { best[k] = u[k]; bi[k] = (2 * da + dh) * 5 + (2 * db + dw); }
                    }
                // the block's own pixels are local (1,1), (1,2), (2,1), (2,2)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int me = (1 + i) * 5 + (1 + j);
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (bi[k] == me) acc[i][j][k] += gv[k];
                    }
            }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ih = 2 * a + i, iw = 2 * bb + j;
                if (ih < H && iw < W) gx[(((long long)b * H + ih) * W + iw) * C4 + c] = acc[i][j];
            }
    }
}

// ---------------------------------------------------------------------------------------------------
// Stem tail, training: BatchNorm affine + ReLU + max-pool 3x3 / stride 2 / pad 1 in ONE pass over the convolution's output (round 5).
// models/resnet.py:148-151 (bn1, relu, maxpool) on the 302 MB stem map of the training batch: as three kernels the map is written once more
// (bn_apply), read back by the pool, and in the backward re-read by the pool's backward, the BatchNorm reduction and the BatchNorm input
// gradient with a 302 MB pool gradient written in between.  Here the activated map never exists: the forward reads y and writes the pooled
// map + one byte per pooled element (which of the 9 taps was the FIRST maximum, torch's strict `>` scan), the backward reduction walks the
// pooled elements (each routes its gradient to exactly one input element: sum_m gz = sum over pooled elements with a positive maximum) and the
// input-gradient pass is per input element (<= 4 covering windows, matched against the recorded tap).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bn_relu_pool(const f32x4* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift,
                                                     f32x4* __restrict__ out, unsigned int* __restrict__ idx, int B, int H, int W, int C4, int OH, int OW) {
    const long long n = (long long)B * OH * OW * C4;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const int c = (int)(e % C4);
        long long p = e / C4;
        const int ow = (int)(p % OW);
        p /= OW;
        const int oh = (int)(p % OH), b = (int)(p / OH);
        const f32x4 sc = *(const f32x4*)(scale + 4 * c), sh = *(const f32x4*)(shift + 4 * c);
        f32x4 v[9];
        bool in[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ih = 2 * oh - 1 + t / 3, iw = 2 * ow - 1 + t % 3;
            in[t] = ((unsigned)ih < (unsigned)H) & ((unsigned)iw < (unsigned)W);
            v[t] = in[t] ? y[(((long long)b * H + ih) * W + iw) * C4 + c] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        unsigned int bi[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float z = fmaf(v[t][k], sc[k], sh[k]);
                const float a = in[t] ? ((z != z) ? z : fmaxf(z, 0.f)) : -INFINITY;          // relu and max both propagate NaN, as torch.relu / max_pool2d do
                if (a > best[k] || ((a != a) && !(best[k] != best[k]))) { best[k] = a; bi[k] = (unsigned)t; }
            }
        }
        out[e] = best;
        idx[e] = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
    }
}

// acc[0][C] += sum gz, acc[1][C] += sum gz xhat over the INPUT elements of the fused tail = over the pooled elements whose maximum is positive
// (gz of the arg-max element = the pooled gradient, every other element of the window gets nothing from it).  Block = 16 row lanes x 16 float4 lanes.
__global__ __launch_bounds__(256) void k_bn_relu_pool_bwd_reduce(const float* __restrict__ g, const unsigned int* __restrict__ idx, const float* __restrict__ pooled,
                                                                const float* __restrict__ y, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                int B, int H, int W, int C, int OH, int OW, double* __restrict__ acc) {
    __shared__ float red[2][16][64 + 1];
    const int c16 = threadIdx.x & 15, r16 = threadIdx.x >> 4;
    const int c0 = blockIdx.x * 64 + c16 * 4;
    const int M = B * OH * OW;
    const int m0 = blockIdx.y * kBnRows, m1 = min(m0 + kBnRows, M);
    const f32x4 mu = *(const f32x4*)(mean + c0), rs = *(const f32x4*)(rstd + c0);
    float s[4] = {0.f, 0.f, 0.f, 0.f}, sx[4] = {0.f, 0.f, 0.f, 0.f};
    for (int m = m0 + r16; m < m1; m += 16) {
        const size_t o = (size_t)m * C + c0;
        const f32x4 gv = *(const f32x4*)(g + o), pv = *(const f32x4*)(pooled + o);
        const unsigned int iv = idx[o >> 2];
        const int ow = m % OW, oh = (m / OW) % OH, b = m / (OW * OH);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int t = (iv >> (8 * k)) & 0xff;
            const int ih = 2 * oh - 1 + t / 3, iw = 2 * ow - 1 + t % 3;
            const float yv = y[(((size_t)b * H + ih) * W + iw) * C + c0 + k];
            const float gg = pv[k] > 0.f ? gv[k] : 0.f;
            s[k] += gg;
            sx[k] = fmaf(gg, (yv - mu[k]) * rs[k], sx[k]);
        }
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

// gy = a gz + b y + c per input element; gz = sum over the <= 4 pooled windows that cover it of [it is the window's recorded arg-max and the maximum
// is positive] * pooled gradient.  Window (oh, ow) covers input row h with tap row dh = h - 2 oh + 1: h even -> oh = h / 2 (dh = 1); h odd ->
// oh = (h - 1) / 2 (dh = 2) and oh = (h + 1) / 2 (dh = 0).
__global__ __launch_bounds__(256) void k_bn_relu_pool_bwd_apply(const f32x4* __restrict__ g, const unsigned int* __restrict__ idx, const f32x4* __restrict__ pooled,
                                                               const f32x4* __restrict__ y, const float* __restrict__ ca, const float* __restrict__ cb,
                                                               const float* __restrict__ cc, f32x4* __restrict__ gy, int B, int H, int W, int C4, int OH, int OW) {
    const long long n = (long long)B * H * W * C4;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const int c = (int)(e % C4);
        long long p = e / C4;
        const int w = (int)(p % W);
        p /= W;
        const int h = (int)(p % H), b = (int)(p / H);
        const int oh0 = h >> 1, ow0 = w >> 1;                          // first covering window; odd coordinates have a second one at + 1
        const int nh = 1 + (h & 1), nw = 1 + (w & 1);
        f32x4 gz = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int oh = oh0 + i, ow = ow0 + j;
                if (i < nh && j < nw && oh < OH && ow < OW) {
                    const unsigned int me = (unsigned)((h - 2 * oh + 1) * 3 + (w - 2 * ow + 1));
                    const long long o = (((long long)b * OH + oh) * OW + ow) * C4 + c;
                    const unsigned int iv = idx[o];
                    const f32x4 gv = g[o], pv = pooled[o];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (((iv >> (8 * k)) & 0xff) == me && pv[k] > 0.f) gz[k] += gv[k];
                }
            }
        const f32x4 a = *(const f32x4*)(ca + 4 * c), bb = *(const f32x4*)(cb + 4 * c), k0 = *(const f32x4*)(cc + 4 * c);
        gy[e] = a * gz + bb * y[e] + k0;
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

// ---------------------------------------------------------------------------------------------------
// weight gradient of a 3x3 layer, all nine taps resident (round 2).  The kernel above streams gy and x once per 64 x 64 tile of dw: a 3x3
// layer re-reads both tensors 9 x Cin/64 times (16 flop per byte: it ran at 45-59 TFLOP/s, memory-bound).  Here a work-group owns
// 64 output channels x (9 taps x 32 input channels) of dw and walks chunks of <= 64 output pixels (whole rows of one image): the chunk of gy
// [64 px][64 co] and the input patch [(rows-1) s + 3][(OW-1) s + 3][32 ci] are copied to LDS once and serve all nine taps (87 flop per byte).
// Wave (h, tg) owns output-channel half h and taps {0..4} or {5..8} (alternating per work-group so the SIMDs balance): 5 (4) accumulator
// tiles, per k-step (2 pixels) one A operand (gy) and one B operand per tap, ds_read_b32 with lane = channel (conflict-free).  No
// cross-wave reduction; one atomic pass per work-group (pixel slices).
// ---------------------------------------------------------------------------------------------------
struct Wgrad3Args {
    const float* gy;      // [B][OH][OW][Cout]
    const float* x;       // [B][H][W][Cin]
    float* dw;            // [Cout][3][3][Cin], zeroed
    int B, H, W, Cin, OH, OW, Cout, rows, chunks_per_img, nchunks, chunks_per_slice, ci_tiles;
};

template <int ST>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ST == 1 ? 3 : 2))) void k_conv_wgrad3(const Wgrad3Args a) {
    constexpr int kMaxPatch = ST == 1 ? 200 : 328;           // patch pixels: 3 x 66 (one 64-pixel row), 5 x 65 (two 32-pixel rows, stride 2)
    constexpr int NX = (kMaxPatch * 8 + 255) / 256;
    __shared__ float gyP[64 * 64];
    __shared__ float xP[kMaxPatch * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 31, lq = lane >> 5;
    const int h = wave & 1, tg = ((wave >> 1) ^ (int)(blockIdx.x + blockIdx.y)) & 1;
    const int tap0 = tg ? 5 : 0, ntap = tg ? 4 : 5;
    const int ct = blockIdx.x / a.ci_tiles, it = blockIdx.x - ct * a.ci_tiles;
    const int co0 = ct * 64, ci0 = it * 32;
    const int PH = (a.rows - 1) * ST + 3, PW = (a.OW - 1) * ST + 3, NP = PH * PW;
    const int P = a.rows * a.OW, nsteps = (P + 1) / 2;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.B * a.H * a.W * a.Cin * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)a.gy, 0, a.B * a.OH * a.OW * a.Cout * 4, 0x00020000);
    constexpr int kOOB = (int)0x80000000;
    // patch element e = tid + 256 i: pixel (pa, pb) of the patch, 16-byte channel group e & 7 (chunk-invariant: decoded once)
    int pab[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int pix = (tid + 256 * i) >> 3;
        const int pa = pix / PW;
        pab[i] = pix < NP ? (pa << 16) | (pix - pa * PW) : -1;
    }
    f32x4 pg[4], px[NX];
    auto fetch = [&](int chunk) {
        const int b = chunk / a.chunks_per_img, oh0 = (chunk - b * a.chunks_per_img) * a.rows;
        const int pvalid = min(a.rows, a.OH - oh0) * a.OW;
        const int m0 = (b * a.OH + oh0) * a.OW;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i, p = e >> 4, c4 = e & 15;
            pg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, p < pvalid ? ((m0 + p) * a.Cout + co0 + 4 * c4) * 4 : kOOB, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int ih = oh0 * ST - 1 + (pab[i] >> 16), iw = (pab[i] & 0xffff) - 1;
            const bool ok = (pab[i] >= 0) & ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W);
            px[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? (((b * a.H + ih) * a.W + iw) * a.Cin + ci0 + 4 * ((tid + 256 * i) & 7)) * 4 : kOOB, 0, 0));
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) *(f32x4*)&gyP[(tid + 256 * i) * 4] = pg[i];
#pragma unroll
        for (int i = 0; i < NX; ++i)
            if (pab[i] >= 0) *(f32x4*)&xP[(tid + 256 * i) * 4] = px[i];
    };

    f32x16 acc[5];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    int toff[5];                                             // float offset of the wave's taps inside the patch
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const int tap = min(tap0 + t, 8);
        toff[t] = ((tap / 3) * PW + (tap % 3)) * 32;
    }
    const float* ga = gyP + lq * 64 + 32 * h + li;

    const int c_begin = blockIdx.y * a.chunks_per_slice, c_end = min(c_begin + a.chunks_per_slice, a.nchunks);
    if (c_begin < c_end) fetch(c_begin);
    for (int c = c_begin; c < c_end; ++c) {
        if (c > c_begin) __syncthreads();                    // the previous chunk's operands have been read
        stash();
        __syncthreads();
        if (c + 1 < c_end) fetch(c + 1);
        int pr = 0, pc = lq;                                 // this lane's pixel of the k-step: (row, column) inside the chunk
        float fa = ga[0], fb[5];
        {
            const float* xb = xP + (pr * ST * PW + pc * ST) * 32 + li;
#pragma unroll
            for (int t = 0; t < 5; ++t) fb[t] = xb[toff[t]];
        }
        for (int s = 0; s < nsteps; ++s) {
            pc += 2;
            if (pc >= a.OW) { pc -= a.OW; ++pr; }
            const int sn = min(s + 1, nsteps - 1);
            const bool live = 2 * (s + 1) + lq < P;          // a pixel beyond the chunk has a zero gy row: point it at patch pixel (0, 0), never at unwritten LDS
            const int prn = live ? pr : 0, pcn = live ? pc : 0;
            const float fan = ga[sn * 128];
            float fbn[5];
            const float* xb = xP + (prn * ST * PW + pcn * ST) * 32 + li;
#pragma unroll
            for (int t = 0; t < 5; ++t) fbn[t] = xb[toff[t]];
#pragma unroll
            for (int t = 0; t < 5; ++t)
                if (t < ntap) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb[t], acc[t], 0, 0, 0);
            fa = fan;
#pragma unroll
            for (int t = 0; t < 5; ++t) fb[t] = fbn[t];
        }
    }
    // ---- one atomic pass: dw[co][tap][ci0 + li] (128-byte runs) ----
    const int K = 9 * a.Cin;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        if (t < ntap) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + 32 * h + (r & 3) + 8 * (r >> 2) + 4 * lq;
                atomicAdd(a.dw + (size_t)co * K + (tap0 + t) * a.Cin + ci0 + li, acc[t][r]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// weight gradient of the 7x7/2 stem (NCHW input, dw in torch's native (co, ci, kh, kw) order; 133 GFLOP at B = 72, the largest single kernel
// of the training step).  The generic kernel gathers x per lane from global memory (a wave-level load touches ~18 short runs: texture-address
// bound, 46 TFLOP/s).  Here a work-group owns 64 output channels x 224 consecutive k = (ci, kh, kw) (7 MFMA tiles: 28 tiles = 896 >= 882 in
// total, 4 k-blocks) and walks chunks of <= 64 output pixels of one output row: gy [64 px][64 co] and the <= 6 input channels x 7 rows x 133
// columns the chunk touches are copied to LDS with coalesced loads; the B operand of lane k is patch[koff(k) + 2 p] -- with a row pitch of
// 7 (mod 32) floats and a channel pitch of 17 (mod 32) the bank of lane k is k mod 32: conflict-free.  Wave (h, kg): channel half h, 4 or 3 k-tiles.
// ---------------------------------------------------------------------------------------------------
struct StemWgradArgs {
    const float* gy;      // [B][OH][OW][64]
    const float* x;       // [B][Cin][H][W]
    float* dw;            // [64][Cin * 49], zeroed
    int B, Cin, H, W, OH, OW, K, segs_per_row, nchunks, chunks_per_slice;
};

constexpr int kSwRP = 135, kSwCP = 7 * kSwRP, kSwCols = 133, kSwCh = 6;      // 135 = 7 (mod 32), 945 = 17 (mod 32)

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void k_stem_wgrad(const StemWgradArgs a) {
    constexpr int NE = kSwCh * 7 * kSwCols, NX = (NE + 255) / 256;          // 5586 patch floats, 22 per thread
    __shared__ float gyP[64 * 64];
    __shared__ float xP[kSwCh * kSwCP];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 31, lq = lane >> 5;
    const int h = wave & 1, kg = ((wave >> 1) ^ (int)blockIdx.y) & 1;
    const int kb = blockIdx.x, t0 = kg ? 4 : 0, nt = kg ? 3 : 4;
    const int ci_base = (224 * kb) / 49;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.B * a.Cin * a.H * a.W * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)a.gy, 0, a.B * a.OH * a.OW * 64 * 4, 0x00020000);
    constexpr int kOOB = (int)0x80000000;
    int koff[4];
    bool kval[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int k = 224 * kb + 32 * (t0 + t) + li;
        kval[t] = (t < nt) & (k < a.K);
        const int kc = kval[t] ? k : 224 * kb;
        const int ci = kc / 49, r = kc - ci * 49, kh = r / 7, kw = r - kh * 7;
        koff[t] = (ci - ci_base) * kSwCP + kh * kSwRP + kw + 2 * lq;
    }
    // Patch walk: the 42 (channel, row) lines of 133 floats.  Waves 0-1 take the even lines, waves 2-3 the odd ones, lane = column 0..127 (coalesced
    // 256-byte runs; the line index is wave-uniform: scalar address arithmetic); the last 5 columns of all lines are one extra load of threads 0..209.
    static_assert(NX == 22 && kSwCols == 133, "patch walk");
    const int half = wave >> 1, colA = tid & 127;
    const int tl = tid / 5, tc = 128 + (tid - tl * 5), tlc = tl / 7, tlr = tl - tlc * 7;       // tail element: line tl (< 42), column tc
    f32x4 pg[4];
    float px[NX];
    auto fetch = [&](int chunk) {
        const int seg = chunk % a.segs_per_row, row = chunk / a.segs_per_row;          // row = b * OH + oh
        const int b = row / a.OH, oh = row - b * a.OH, ow0 = seg * 64;
        const int pvalid = min(64, a.OW - ow0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i, p = e >> 4, c4 = e & 15;
            pg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, p < pvalid ? ((row * a.OW + ow0 + p) * 64 + 4 * c4) * 4 : kOOB, 0, 0));
        }
        const int iwA = 2 * ow0 - 3 + colA;
        const int vA = (unsigned)iwA < (unsigned)a.W ? iwA * 4 : kOOB;
#pragma unroll
        for (int i = 0; i < 21; ++i) {
            const int line = 2 * i + half, c = line / 7, r = line - c * 7;              // wave-uniform
            const int ci = ci_base + c, ih = 2 * oh - 3 + r;
            const bool ok = (ci < a.Cin) & ((unsigned)ih < (unsigned)a.H);
            px[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, (ok && vA != kOOB) ? vA + (((b * a.Cin + ci) * a.H + ih) * a.W) * 4 : kOOB, 0, 0));
        }
        {
            const int ci = ci_base + tlc, ih = 2 * oh - 3 + tlr, iw = 2 * ow0 - 3 + tc;
            const bool ok = (tid < 210) & (ci < a.Cin) & ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W);
            px[21] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, ok ? (((b * a.Cin + ci) * a.H + ih) * a.W + iw) * 4 : kOOB, 0, 0));
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) *(f32x4*)&gyP[(tid + 256 * i) * 4] = pg[i];
#pragma unroll
        for (int i = 0; i < 21; ++i) {
            const int line = 2 * i + half, c = line / 7, r = line - c * 7;
            xP[c * kSwCP + r * kSwRP + colA] = px[i];
        }
        if (tid < 210) xP[tlc * kSwCP + tlr * kSwRP + tc] = px[21];
    };
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const float* ga = gyP + lq * 64 + 32 * h + li;
    const int nsteps = (min(64, a.OW) + 1) / 2;
    const int c_begin = blockIdx.y * a.chunks_per_slice, c_end = min(c_begin + a.chunks_per_slice, a.nchunks);
    if (c_begin < c_end) fetch(c_begin);
    for (int c = c_begin; c < c_end; ++c) {
        if (c > c_begin) __syncthreads();
        stash();
        __syncthreads();
        if (c + 1 < c_end) fetch(c + 1);
        float fa = ga[0], fb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) fb[t] = xP[koff[t]];
        // (round 5, measured and rejected: the 32 k-steps fully unrolled so that every LDS operand read carries its step as an immediate -- no address
        // VALU in the loop, but 228 instead of 80 VGPRs: 1.42 ms against 1.29-1.37 ms; round 6, measured and rejected: every wave owning all seven k-tiles of
        // its channel half and every second pixel step -- balanced waves instead of 4 + 3 tiles, but 203 VGPRs = two work-groups per CU and 8 LDS reads per 7
        // MFMAs: 1.67 ms, training step 12.34 vs 11.87 ms)
        for (int s = 0; s < nsteps; ++s) {
            const int sn = min(s + 1, nsteps - 1);
            const float fan = ga[sn * 128];
            float fbn[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) fbn[t] = xP[koff[t] + 4 * sn];
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (t < nt) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb[t], acc[t], 0, 0, 0);
            fa = fan;
#pragma unroll
            for (int t = 0; t < 4; ++t) fb[t] = fbn[t];
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (kval[t]) {
            const int k = 224 * kb + 32 * (t0 + t) + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = 32 * h + (r & 3) + 8 * (r >> 2) + 4 * lq;
                atomicAdd(a.dw + (size_t)co * a.K + k, acc[t][r]);
            }
        }
    }
}

}  // namespace hmf

namespace hmf {
struct I64Ptrs { long long* p[64]; int n; };
__global__ void k_i64_increment(const I64Ptrs a) {
    if ((int)threadIdx.x < a.n) *a.p[threadIdx.x] += 1;
}
}  // namespace hmf

using namespace hmf;

extern "C" {

// *counters[i] += 1 for n device int64 scalars in one launch per 64 (the BatchNorm num_batches_tracked buffers of one training pass, nn.BatchNorm2d
// semantics; torch._foreach_add_ on int64 tensors issues one launch per tensor)
int hmf_i64_increment_batch(long long* const* counters, int n, void* stream) {
    if (!counters || n < 1) return fail(HMF_ERR_INVALID, "hmf_i64_increment_batch: bad args");
    for (int base = 0; base < n; base += 64) {
        I64Ptrs a;
        a.n = (n - base) < 64 ? (n - base) : 64;
        for (int i = 0; i < a.n; ++i) {
            if (!counters[base + i] || ((uintptr_t)counters[base + i] & 7)) return fail(HMF_ERR_INVALID, "hmf_i64_increment_batch: counter %d is null or misaligned", base + i);
            a.p[i] = counters[base + i];
        }
        hipLaunchKernelGGL(k_i64_increment, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
        HMF_LAUNCH_CHECK();
    }
    return HMF_OK;
}

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

int hmf_bn_relu_maxpool3x3s2_nhwc(const float* y, const float* scale, const float* shift, float* out, unsigned char* argmax, int B, int H, int W, int C,
                                  void* stream) {
    if (!y || !scale || !shift || !out || !argmax || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 || ((uintptr_t)argmax & 3))
        return fail(HMF_ERR_INVALID, "hmf_bn_relu_maxpool3x3s2_nhwc: bad args (C %% 4 == 0, argmax 4-byte aligned)");
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    const long long n = (long long)B * OH * OW * (C / 4);
    long long blocks = std::min<long long>((n + 255) / 256, 256 * 64);
    hipLaunchKernelGGL(k_bn_relu_pool, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const f32x4*)y, scale, shift, (f32x4*)out,
                       (unsigned int*)argmax, B, H, W, C / 4, OH, OW);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

int hmf_bn_relu_maxpool3x3s2_nhwc_bwd_reduce(const float* g_out, const unsigned char* argmax, const float* out, const float* y, const float* mean,
                                             const float* rstd, int B, int H, int W, int C, double* acc, void* stream) {
    if (!g_out || !argmax || !out || !y || !mean || !rstd || !acc || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 64)
        return fail(HMF_ERR_INVALID, "hmf_bn_relu_maxpool3x3s2_nhwc_bwd_reduce: bad args (C %% 64 == 0 required)");
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    const long long M = (long long)B * OH * OW;
    if (M * C >= (1LL << 31)) return fail(HMF_ERR_UNSUPPORTED, "hmf_bn_relu_maxpool3x3s2_nhwc_bwd_reduce: pooled map exceeds 2^31 elements");
    hipLaunchKernelGGL(k_bn_relu_pool_bwd_reduce, dim3(C / 64, (unsigned)((M + kBnRows - 1) / kBnRows)), dim3(256), 0, (hipStream_t)stream, g_out,
                       (const unsigned int*)argmax, out, y, mean, rstd, B, H, W, C, OH, OW, acc);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

int hmf_bn_relu_maxpool3x3s2_nhwc_bwd_apply(const float* g_out, const unsigned char* argmax, const float* out, const float* y, const float* ca,
                                            const float* cb, const float* cc, float* gy, int B, int H, int W, int C, void* stream) {
    if (!g_out || !argmax || !out || !y || !ca || !cb || !cc || !gy || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4)
        return fail(HMF_ERR_INVALID, "hmf_bn_relu_maxpool3x3s2_nhwc_bwd_apply: bad args");
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    const long long n = (long long)B * H * W * (C / 4);
    long long blocks = std::min<long long>((n + 255) / 256, 256 * 64);
    hipLaunchKernelGGL(k_bn_relu_pool_bwd_apply, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const f32x4*)g_out, (const unsigned int*)argmax,
                       (const f32x4*)out, (const f32x4*)y, ca, cb, cc, (f32x4*)gy, B, H, W, C / 4, OH, OW);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

int hmf_maxpool3x3s2_nhwc_backward(const float* x, const float* g_out, float* g_x, int B, int H, int W, int C, void* stream) {
    if (!x || !g_out || !g_x || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4) return fail(HMF_ERR_INVALID, "hmf_maxpool3x3s2_nhwc_backward: bad args");
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    const long long n = (long long)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
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
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(13, a.M, st);                  // weight-gradient family (bench.py --train)
    if (stem_nchw && ks == 7 && stride == 2 && Cout == 64 && (a.OW % 2) == 0) {
        StemWgradArgs w;
        w.gy = gy; w.x = x; w.dw = dw; w.B = B; w.Cin = Cin; w.H = H; w.W = W; w.OH = a.OH; w.OW = a.OW; w.K = Cin * 49;
        w.segs_per_row = (a.OW + 63) / 64;
        w.nchunks = B * a.OH * w.segs_per_row;
        const int kblocks = (w.K + 223) / 224;
        int slices = std::max(1, std::min(w.nchunks, (256 * 3 + kblocks - 1) / kblocks));
        w.chunks_per_slice = (w.nchunks + slices - 1) / slices;
        slices = (w.nchunks + w.chunks_per_slice - 1) / w.chunks_per_slice;
        if (int zr = zero_async(dw, (size_t)Cout * w.K * sizeof(float), st)) return zr;
        hipLaunchKernelGGL(k_stem_wgrad, dim3(kblocks, slices), dim3(256), 0, st, w);
        HMF_LAUNCH_CHECK();
        return HMF_OK;
    }
    if (!stem_nchw && ks == 3 && Cout % 64 == 0 && a.OW >= 2 && a.OW <= 64) {
        // all nine taps resident: chunks of whole output rows, <= 64 pixels
        Wgrad3Args w;
        w.gy = gy; w.x = x; w.dw = dw; w.B = B; w.H = H; w.W = W; w.Cin = Cin; w.OH = a.OH; w.OW = a.OW; w.Cout = Cout;
        w.rows = std::max(1, std::min(64 / a.OW, a.OH));
        const int PH = (w.rows - 1) * stride + 3, PW = (a.OW - 1) * stride + 3;
        if (PH * PW <= (stride == 1 ? 200 : 328)) {
            w.chunks_per_img = (a.OH + w.rows - 1) / w.rows;
            w.nchunks = B * w.chunks_per_img;
            w.ci_tiles = Cin / 32;
            const int tiles = (Cout / 64) * w.ci_tiles;
            const int per_cu = stride == 1 ? 3 : 2;                     // work-groups that fit a CU (LDS 42 / 58 KB); measured best WG count = one full round
            int slices = std::max(1, std::min(w.nchunks, (256 * per_cu + tiles - 1) / tiles));
            w.chunks_per_slice = (w.nchunks + slices - 1) / slices;
            slices = (w.nchunks + w.chunks_per_slice - 1) / w.chunks_per_slice;
            if (int zr = zero_async(dw, (size_t)Cout * 9 * Cin * sizeof(float), st)) return zr;
            if (stride == 1) hipLaunchKernelGGL(k_conv_wgrad3<1>, dim3(tiles, slices), dim3(256), 0, st, w);
            else hipLaunchKernelGGL(k_conv_wgrad3<2>, dim3(tiles, slices), dim3(256), 0, st, w);
            HMF_LAUNCH_CHECK();
            return HMF_OK;
        }
    }
    a.K = ks * ks * Cin;
    a.ktiles = (a.K + 63) / 64;
    const int tiles = a.ktiles * ((Cout + 63) / 64);
    // ~6 work-groups (4 waves each, 48 KB of LDS) per CU; a block's pixel slice is 4 wave quarters of a multiple of 8 pixels
    int slices = std::max(1, std::min((int)((M + 255) / 256), (256 * 6 + tiles - 1) / tiles));
    a.rows_per_slice = (int)(((M + slices - 1) / slices + 31) / 32 * 32);
    slices = (int)((M + a.rows_per_slice - 1) / a.rows_per_slice);
    if (int zr = zero_async(dw, (size_t)Cout * a.K * sizeof(float), st)) return zr;
    if (stem_nchw) hipLaunchKernelGGL(k_conv_wgrad<true>, dim3(tiles, slices), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_conv_wgrad<false>, dim3(tiles, slices), dim3(256), 0, st, a);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

}  // extern "C"
