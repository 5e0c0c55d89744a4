// Proxy-representation front end: Canny edge image + 2D-joint Gaussian heat-maps -> (B, 18, S, S) model input in ONE launch.
//
// Replaces the op chains of models/canny_edge_detector.py:104-166 (per channel: two zero-padded 1-D Gaussian
// cross-correlations, two Sobel convolutions; magnitude; orientation binned to 45 degrees; 8 directional-difference
// convolutions + 4 masked passes of non-maximum suppression; threshold -- ~40 eager launches) and
// utils/label_conversions.py:106-125 (17 heat-maps), as called at predict/predict_humaniflow.py:101-109 and
// evaluate/evaluate_humaniflow.py:71-73.
// One workgroup per 32x32 output tile: the RGB halo (40x40x3), the horizontally blurred rows, the blurred image and the
// gradient field live in LDS (66 KB); every intermediate is produced exactly once per tile and nothing but the 18 output
// channels goes to HBM.  HBM-bound: (3 in + 18 out) x 4 B per pixel.
#include "hmf_common.h"

namespace hmf {

constexpr int kT = 32;                 // output tile
constexpr int kImgW = kT + 8;          // rgb halo +-4
constexpr int kHbH = kT + 8, kHbW = kT + 4;
constexpr int kBlW = kT + 4;           // blurred +-2
constexpr int kMgW = kT + 2;           // gradient field +-1

struct ProxyArgs {
    const float* img;        // B x 3 x S x S
    const float* joints;     // B x 17 x 2 (u = column, v = row)   (nullable: no heat-maps)
    const float* visib;      // B x 17 (nullable)
    float* proxy;            // B x 18 x S x S (nullable)
    float* grad_mag;         // B x 1 x S x S  (nullable, diagnostics / the reference's extra dict entries)
    float* grad_ori;
    float* thin;
    int B, S, nms, njoints;
    float thr, hm_std;
    float g[5];
};

__global__ __launch_bounds__(256) void k_proxy_rep(ProxyArgs a) {
    __shared__ float sImg[3][kImgW][kImgW];
    __shared__ float sHb[3][kHbH][kHbW];
    __shared__ float sBl[3][kBlW][kBlW];
    __shared__ float sGx[kMgW][kMgW], sGy[kMgW][kMgW], sMg[kMgW][kMgW];
    __shared__ float sHr[32][kT], sHc[32][kT];      // per joint: exp(-dr^2/2) of the tile's rows (times visibility), exp(-dc^2/2) of its columns
    const int S = a.S, tid = threadIdx.x;
    const int b = blockIdx.z, ty0 = blockIdx.y * kT, tx0 = blockIdx.x * kT;
    const float* img = a.img + (size_t)b * 3 * S * S;
    // 0. heat-maps are separable: exp(-dr^2/2 - dc^2/2) = exp(-dr^2/2) exp(-dc^2/2) -> 2 x 32 exps per joint and tile instead
    //    of 1024 (differs from the one-exp form by <= 2 ulp)
    if (a.joints) {
        for (int i = tid; i < a.njoints * kT * 2; i += 256) {
            const int j = i / (2 * kT), k = i % (2 * kT);
            const float u = a.joints[((size_t)b * a.njoints + j) * 2], v = a.joints[((size_t)b * a.njoints + j) * 2 + 1];
            if (k < kT) {
                const float dr = ((float)(ty0 + k) - v) / a.hm_std;
                sHr[j][k] = expf(-(dr * dr) / 2.f) * (a.visib ? a.visib[(size_t)b * a.njoints + j] : 1.f);
            } else {
                const float dc = ((float)(tx0 + k - kT) - u) / a.hm_std;
                sHc[j][k - kT] = expf(-(dc * dc) / 2.f);
            }
        }
    }
    // 1. RGB halo, zero outside the image (padding of the horizontal pass)
    for (int i = tid; i < 3 * kImgW * kImgW; i += 256) {
        const int c = i / (kImgW * kImgW), r = (i / kImgW) % kImgW, q = i % kImgW;
        const int y = ty0 - 4 + r, x = tx0 - 4 + q;
        sImg[c][r][q] = (y >= 0 && y < S && x >= 0 && x < S) ? img[((size_t)c * S + y) * S + x] : 0.f;
    }
    __syncthreads();
    // 2. horizontal Gaussian: rows -4..+35, cols -2..+33; zero outside the image (padding of the vertical pass)
    for (int i = tid; i < 3 * kHbH * kHbW; i += 256) {
        const int c = i / (kHbH * kHbW), r = (i / kHbW) % kHbH, q = i % kHbW;
        const int y = ty0 - 4 + r, x = tx0 - 2 + q;
        float v = 0.f;
        if (y >= 0 && y < S && x >= 0 && x < S) {
#pragma unroll
            for (int t = 0; t < 5; ++t) v = fmaf(a.g[t], sImg[c][r][q + t], v);        // img col x + t - 2 -> halo col q + t
        }
        sHb[c][r][q] = v;
    }
    __syncthreads();
    // 3. vertical Gaussian: rows/cols -2..+33; zero outside the image (padding of the Sobel filters)
    for (int i = tid; i < 3 * kBlW * kBlW; i += 256) {
        const int c = i / (kBlW * kBlW), r = (i / kBlW) % kBlW, q = i % kBlW;
        const int y = ty0 - 2 + r, x = tx0 - 2 + q;
        float v = 0.f;
        if (y >= 0 && y < S && x >= 0 && x < S) {
#pragma unroll
            for (int t = 0; t < 5; ++t) v = fmaf(a.g[t], sHb[c][r + t][q], v);         // hb row y + t - 2 -> r + t
        }
        sBl[c][r][q] = v;
    }
    __syncthreads();
    // 4. Sobel gradients averaged over the channels, magnitude: rows/cols -1..+32; zero outside (padding of the NMS filters)
    for (int i = tid; i < kMgW * kMgW; i += 256) {
        const int r = i / kMgW, q = i % kMgW;
        const int y = ty0 - 1 + r, x = tx0 - 1 + q;
        float gx = 0.f, gy = 0.f, m = 0.f;
        if (y >= 0 && y < S && x >= 0 && x < S) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float (*B_)[kBlW] = sBl[c];
                const int rr = r + 1, qq = q + 1;       // blurred index of (y, x)
                gx += (B_[rr - 1][qq - 1] + 2.f * B_[rr][qq - 1] + B_[rr + 1][qq - 1]) -
                      (B_[rr - 1][qq + 1] + 2.f * B_[rr][qq + 1] + B_[rr + 1][qq + 1]);
                gy += (B_[rr - 1][qq - 1] + 2.f * B_[rr - 1][qq] + B_[rr - 1][qq + 1]) -
                      (B_[rr + 1][qq - 1] + 2.f * B_[rr + 1][qq] + B_[rr + 1][qq + 1]);
            }
            gx = gx / 3.f; gy = gy / 3.f;
            m = sqrtf(gx * gx + gy * gy);
        }
        sGx[r][q] = gx; sGy[r][q] = gy; sMg[r][q] = m;
    }
    __syncthreads();
    // 5. per output pixel: orientation bin, non-maximum suppression, threshold, heat-maps
    const size_t plane = (size_t)S * S;
    for (int i = tid; i < kT * kT; i += 256) {
        const int r = i / kT, q = i % kT;
        const int y = ty0 + r, x = tx0 + q;
        if (y >= S || x >= S) continue;
        const int rr = r + 1, qq = q + 1;
        const float m = sMg[rr][qq];
        float ori = atan2f(sGy[rr][qq], sGx[rr][qq]) * (float)(180.0 / 3.14159265358979323846) + 180.0f;
        ori = rintf(ori / 45.0f) * 45.0f;                                    // torch.round: half to even
        float thin = m;
        if (a.nms) {
            const int idx = ((int)(ori / 45.0f)) & 7;                        // (ori / 45) % 8 with ori / 45 in {0..8}
            const int p = idx & 3;                                           // direction pair (p, p + 4)
            const int dy[4] = {0, 1, 1, 1}, dx[4] = {1, 1, 0, -1};
            const float dpos = m - sMg[rr + dy[p]][qq + dx[p]];
            const float dneg = m - sMg[rr - dy[p]][qq - dx[p]];
            if (!(fminf(dpos, dneg) > 0.0f)) thin = 0.f;
        }
        const size_t o = (size_t)y * S + x;
        if (a.grad_mag) a.grad_mag[(size_t)b * plane + o] = m;
        if (a.grad_ori) a.grad_ori[(size_t)b * plane + o] = ori;
        if (a.thin) a.thin[(size_t)b * plane + o] = thin;
        if (a.proxy) {
            float* out = a.proxy + (size_t)b * (1 + a.njoints) * plane + o;
            const float e = a.nms ? thin : m;
            out[0] = e < a.thr ? 0.f : e;
            if (a.joints) {
                for (int j = 0; j < a.njoints; ++j) out[(size_t)(1 + j) * plane] = sHr[j][r] * sHc[j][q];
            }
        }
    }
}

}  // namespace hmf

using namespace hmf;

extern "C" int hmf_proxy_rep(const float* img, const float* joints2D, const float* visib, int B, int S, int num_joints,
                             int edge_nms, float edge_threshold, float edge_gaussian_std, int edge_gaussian_size, float heatmap_std,
                             float* proxy, float* grad_magnitude, float* grad_orientation, float* thin_edges, void* stream) {
    if (!img || B <= 0 || S <= 0 || num_joints < 0 || num_joints > 32) return fail(HMF_ERR_INVALID, "hmf_proxy_rep: bad args (num_joints <= 32)");
    if (edge_gaussian_size != 5) return fail(HMF_ERR_UNSUPPORTED, "hmf_proxy_rep: EDGE_GAUSSIAN_SIZE=%d (only 5 is built)", edge_gaussian_size);
    if (!proxy && !grad_magnitude && !grad_orientation && !thin_edges) return fail(HMF_ERR_INVALID, "hmf_proxy_rep: no output requested");
    ProxyArgs a;
    a.img = img; a.joints = num_joints > 0 ? joints2D : nullptr; a.visib = visib; a.proxy = proxy;
    a.grad_mag = grad_magnitude; a.grad_ori = grad_orientation; a.thin = thin_edges;
    a.B = B; a.S = S; a.nms = edge_nms; a.njoints = a.joints ? num_joints : 0; a.thr = edge_threshold; a.hm_std = heatmap_std;
    // scipy.signal.windows.gaussian(5, std) / sum, in double, rounded to float (canny_edge_detector.py:23-24,31)
    double w[5], sum = 0.0;
    for (int t = 0; t < 5; ++t) { const double n = t - 2.0; w[t] = exp(-0.5 * (n / edge_gaussian_std) * (n / edge_gaussian_std)); sum += w[t]; }
    for (int t = 0; t < 5; ++t) a.g[t] = (float)(w[t] / sum);
    const dim3 grid((S + kT - 1) / kT, (S + kT - 1) / kT, B);
    hipLaunchKernelGGL(k_proxy_rep, grid, dim3(256), 0, (hipStream_t)stream, a);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}
