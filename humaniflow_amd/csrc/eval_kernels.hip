// Evaluation epilogue on the device: the point-set error metrics of metrics/eval_metrics_tracker.py:95-280 and the
// sample-diversity reductions :398-410, fused so that the (B,N,6890,3) sample tensors are read from HBM once instead of being copied
// to the host (the reference's 0.53 GB/batch D2H, evaluate_humaniflow.py:227-252) or swept by ~40 elementwise passes.
//
// k_pointset_errors: one work-group per (prediction, target) pair of npts 3-D points.
//   pass 1: 17 running sums (sum P, sum T, sum |P|^2, sum |T|^2, sum P T^T) in fp64 -> means, variances, cross-covariance K
//   thread 0: scale+translation correction (utils/eval_utils.py:107-125) and the similarity Procrustes alignment (:62-104):
//             R = V Z U^T from the SVD of K, here via the Jacobi eigen-decomposition of K^T K (fp64), scale = tr(R K) / var(P)
//   pass 2 (second read of P comes from L2): mean |P - T|, mean |sc(P) - T|, mean |pa(P) - T|
// k_sample_diversity: mean distance of every sample point from the per-point mean over the N samples of its image.
// HBM-bound by construction: 12 B per point read once (+ one L2 re-read).
#include "hmf_common.h"

namespace hmf {

__device__ inline double block_sum(double v, double* red) {
    // 256 threads: wave shuffle then 4-way LDS combine
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// N sums of a 256-thread block at once, result valid in thread 0 (same tree and order as block_sum, hence the same bits).  The N shuffle chains of a
// step are independent and pipeline; one barrier pair serves all of them.  (Twenty block_sum calls in a row -- 17 moments + 3 errors -- were ~11 us of
// dependent cross-lane permutes and barriers per work-group: the point-set kernels ran at 1.6 TB/s.)
template <int N>
__device__ inline void block_sum_n(double (&v)[N], double* red /* [4][N] */) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] += __shfl_down(v[i], o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int i = 0; i < N; ++i) red[(threadIdx.x >> 6) * N + i] = v[i];
    __syncthreads();
    if (threadIdx.x == 0)
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = red[i] + red[N + i] + red[2 * N + i] + red[3 * N + i];
}

// eigen-decomposition of a symmetric 3x3 (cyclic Jacobi, fp64): A = V diag(w) V^T, columns of V = eigenvectors
__device__ inline void jacobi3(double A[3][3], double V[3][3], double w[3]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) V[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 12; ++sweep) {
        // converged when the off-diagonal mass is below 1e-30 of the diagonal (far under the fp64 rounding of the eigenvalues; cyclic Jacobi converges
        // quadratically: 4-6 sweeps for these 3x3 Gram matrices instead of the fixed 12 -- the sweep loop is a serial fp64 div/sqrt chain on one lane)
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        if (off <= 1e-30 * (fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]))) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (fabs(A[p][q]) < 1e-300) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < 3; ++i) w[i] = A[i][i];
}

struct Align {          // y = sc_a * (x - mu_p) + mu_t   |   y = pa_s * R x + pa_t
    float mu_p[3], mu_t[3], sc_a, R[9], pa_s, pa_t[3];
};

// scale + translation correction and similarity Procrustes alignment from the 17 sums of a pair (one thread)
__device__ inline void solve_alignment(const double* s, int npts, int want_pa, Align& al) {
    const double n = npts;
    double mp[3] = {s[0] / n, s[1] / n, s[2] / n}, mt[3] = {s[3] / n, s[4] / n, s[5] / n};
    const double varp = s[6] - n * (mp[0] * mp[0] + mp[1] * mp[1] + mp[2] * mp[2]);
    const double vart = s[7] - n * (mt[0] * mt[0] + mt[1] * mt[1] + mt[2] * mt[2]);
    for (int i = 0; i < 3; ++i) { al.mu_p[i] = (float)mp[i]; al.mu_t[i] = (float)mt[i]; }
    al.sc_a = (float)(sqrt(vart / n) / sqrt(varp / n));
    if (want_pa) {
        double K[3][3];                                   // K = X1 X2^T (source x target), utils/eval_utils.py:80
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) K[i][j] = s[8 + 3 * i + j] - n * mp[i] * mt[j];
        double KtK[3][3], V[3][3], w[3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) KtK[i][j] = K[0][i] * K[0][j] + K[1][i] * K[1][j] + K[2][i] * K[2][j];
        jacobi3(KtK, V, w);
        int o[3] = {0, 1, 2};                              // descending eigenvalues
        if (w[o[0]] < w[o[1]]) { int x = o[0]; o[0] = o[1]; o[1] = x; }
        if (w[o[1]] < w[o[2]]) { int x = o[1]; o[1] = o[2]; o[2] = x; }
        if (w[o[0]] < w[o[1]]) { int x = o[0]; o[0] = o[1]; o[1] = x; }
        double v[3][3], u[3][3];                           // v[k] = k-th right singular vector, u[k] = K v[k] / sigma_k
        for (int k = 0; k < 3; ++k)
            for (int i = 0; i < 3; ++i) v[k][i] = V[i][o[k]];
        for (int k = 0; k < 2; ++k) {
            for (int i = 0; i < 3; ++i) u[k][i] = K[i][0] * v[k][0] + K[i][1] * v[k][1] + K[i][2] * v[k][2];
            if (k == 1) {
                const double d = u[1][0] * u[0][0] + u[1][1] * u[0][1] + u[1][2] * u[0][2];
                for (int i = 0; i < 3; ++i) u[1][i] -= d * u[0][i];
            }
            const double nn = sqrt(u[k][0] * u[k][0] + u[k][1] * u[k][1] + u[k][2] * u[k][2]);
            for (int i = 0; i < 3; ++i) u[k][i] /= (nn > 0 ? nn : 1.0);
        }
        u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1];
        u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2];
        u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
        const double detV = v[0][0] * (v[1][1] * v[2][2] - v[1][2] * v[2][1]) - v[0][1] * (v[1][0] * v[2][2] - v[1][2] * v[2][0]) +
                            v[0][2] * (v[1][0] * v[2][1] - v[1][1] * v[2][0]);
        double R[3][3];                                    // R = V Z U^T: Z33 u3_true = det(V) (u1 x u2)
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) R[i][j] = v[0][i] * u[0][j] + v[1][i] * u[1][j] + (detV < 0 ? -1.0 : 1.0) * v[2][i] * u[2][j];
        double tr = 0.0;
        for (int i = 0; i < 3; ++i) tr += R[i][0] * K[0][i] + R[i][1] * K[1][i] + R[i][2] * K[2][i];
        const double sc = tr / varp;
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) al.R[3 * i + j] = (float)R[i][j];
            al.pa_t[i] = (float)(mt[i] - sc * (R[i][0] * mp[0] + R[i][1] * mp[1] + R[i][2] * mp[2]));
        }
        al.pa_s = (float)sc;
    }
}

__global__ __launch_bounds__(256) void k_pointset_errors(const float* __restrict__ P, const float* __restrict__ T, int npts, int group,
                                                         int want_pa, float* __restrict__ out) {
    __shared__ double red[4 * 17];
    __shared__ Align al;
    const int pair = blockIdx.x;
    const float* p = P + (size_t)pair * npts * 3;
    const float* t = T + (size_t)(pair / group) * npts * 3;
    double s[17];
#pragma unroll
    for (int i = 0; i < 17; ++i) s[i] = 0.0;
    for (int i = threadIdx.x; i < npts; i += 256) {
        const float px = p[3 * i], py = p[3 * i + 1], pz = p[3 * i + 2];
        const float tx = t[3 * i], ty = t[3 * i + 1], tz = t[3 * i + 2];
        s[0] += px; s[1] += py; s[2] += pz;
        s[3] += tx; s[4] += ty; s[5] += tz;
        s[6] += (double)px * px + (double)py * py + (double)pz * pz;
        s[7] += (double)tx * tx + (double)ty * ty + (double)tz * tz;
        s[8] += (double)px * tx;  s[9] += (double)px * ty;  s[10] += (double)px * tz;
        s[11] += (double)py * tx; s[12] += (double)py * ty; s[13] += (double)py * tz;
        s[14] += (double)pz * tx; s[15] += (double)pz * ty; s[16] += (double)pz * tz;
    }
    block_sum_n<17>(s, red);
    if (threadIdx.x == 0) solve_alignment(s, npts, want_pa, al);
    __syncthreads();
    double e0 = 0.0, e1 = 0.0, e2 = 0.0;
    for (int i = threadIdx.x; i < npts; i += 256) {
        const float px = p[3 * i], py = p[3 * i + 1], pz = p[3 * i + 2];
        const float tx = t[3 * i], ty = t[3 * i + 1], tz = t[3 * i + 2];
        float dx = px - tx, dy = py - ty, dz = pz - tz;
        e0 += sqrtf(dx * dx + dy * dy + dz * dz);
        dx = (px - al.mu_p[0]) * al.sc_a + al.mu_t[0] - tx;
        dy = (py - al.mu_p[1]) * al.sc_a + al.mu_t[1] - ty;
        dz = (pz - al.mu_p[2]) * al.sc_a + al.mu_t[2] - tz;
        e1 += sqrtf(dx * dx + dy * dy + dz * dz);
        if (want_pa) {
            dx = al.pa_s * (al.R[0] * px + al.R[1] * py + al.R[2] * pz) + al.pa_t[0] - tx;
            dy = al.pa_s * (al.R[3] * px + al.R[4] * py + al.R[5] * pz) + al.pa_t[1] - ty;
            dz = al.pa_s * (al.R[6] * px + al.R[7] * py + al.R[8] * pz) + al.pa_t[2] - tz;
            e2 += sqrtf(dx * dx + dy * dy + dz * dz);
        }
    }
    double e[3] = {e0, e1, e2};
    block_sum_n<3>(e, red);
    if (threadIdx.x == 0) {
        out[3 * pair + 0] = (float)(e[0] / npts);
        out[3 * pair + 1] = (float)(e[1] / npts);
        out[3 * pair + 2] = want_pa ? (float)(e[2] / npts) : 0.f;
    }
}

__device__ inline void accumulate17(double* s, float px, float py, float pz, float tx, float ty, float tz) {
    s[0] += px; s[1] += py; s[2] += pz;
    s[3] += tx; s[4] += ty; s[5] += tz;
    s[6] += (double)px * px + (double)py * py + (double)pz * pz;
    s[7] += (double)tx * tx + (double)ty * ty + (double)tz * tz;
    s[8] += (double)px * tx;  s[9] += (double)px * ty;  s[10] += (double)px * tz;
    s[11] += (double)py * tx; s[12] += (double)py * ty; s[13] += (double)py * tz;
    s[14] += (double)pz * tx; s[15] += (double)pz * ty; s[16] += (double)pz * tz;
}

__device__ inline void errors3(const Align& al, int want_pa, float px, float py, float pz, float tx, float ty, float tz, double& e0, double& e1, double& e2) {
    float dx = px - tx, dy = py - ty, dz = pz - tz;
    e0 += sqrtf(dx * dx + dy * dy + dz * dz);
    dx = (px - al.mu_p[0]) * al.sc_a + al.mu_t[0] - tx;
    dy = (py - al.mu_p[1]) * al.sc_a + al.mu_t[1] - ty;
    dz = (pz - al.mu_p[2]) * al.sc_a + al.mu_t[2] - tz;
    e1 += sqrtf(dx * dx + dy * dy + dz * dz);
    if (want_pa) {
        dx = al.pa_s * (al.R[0] * px + al.R[1] * py + al.R[2] * pz) + al.pa_t[0] - tx;
        dy = al.pa_s * (al.R[3] * px + al.R[4] * py + al.R[5] * pz) + al.pa_t[1] - ty;
        dz = al.pa_s * (al.R[6] * px + al.R[7] * py + al.R[8] * pz) + al.pa_t[2] - tz;
        e2 += sqrtf(dx * dx + dy * dy + dz * dz);
    }
}

// Same arithmetic and summation order as k_pointset_errors, but a pair's points stay in REGISTERS between the two passes (NPT points per
// thread, npts <= 256 NPT): all loads of a pair are in flight at once and the prediction is read from HBM once instead of twice
// (6890-vertex meshes: NPT = 27, 81 VGPRs of points).  Round 1's kernel walked the points twice with 6 dependent scalar loads per step.
template <int NPT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void k_pointset_errors_reg(const float* __restrict__ P, const float* __restrict__ T, int npts, int group,
                                                             int want_pa, float* __restrict__ out) {
    __shared__ double red[4 * 17];
    __shared__ Align al;
    const int pair = blockIdx.x;
    const float* p = P + (size_t)pair * npts * 3;
    const float* t = T + (size_t)(pair / group) * npts * 3;
    // the prediction stays in registers; the target (shared by the `group` samples of an image: L2-resident) is read again in pass 2
    float pv[NPT][3];
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        const int i = threadIdx.x + 256 * k;
        const bool ok = i < npts;
#pragma unroll
        for (int c = 0; c < 3; ++c) pv[k][c] = ok ? p[3 * i + c] : 0.f;
    }
    double s[17];
#pragma unroll
    for (int i = 0; i < 17; ++i) s[i] = 0.0;
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        const int i = threadIdx.x + 256 * k;
        if (i < npts) accumulate17(s, pv[k][0], pv[k][1], pv[k][2], t[3 * i], t[3 * i + 1], t[3 * i + 2]);
    }
    block_sum_n<17>(s, red);
    if (threadIdx.x == 0) solve_alignment(s, npts, want_pa, al);
    __syncthreads();
    double e0 = 0.0, e1 = 0.0, e2 = 0.0;
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        const int i = threadIdx.x + 256 * k;
        if (i < npts) errors3(al, want_pa, pv[k][0], pv[k][1], pv[k][2], t[3 * i], t[3 * i + 1], t[3 * i + 2], e0, e1, e2);
    }
    double e[3] = {e0, e1, e2};
    block_sum_n<3>(e, red);
    if (threadIdx.x == 0) {
        out[3 * pair + 0] = (float)(e[0] / npts);
        out[3 * pair + 1] = (float)(e[1] / npts);
        out[3 * pair + 2] = want_pa ? (float)(e[2] / npts) : 0.f;
    }
}

// Small point sets (joints: 14 / 17 / 90 points <= 64): one WAVE per pair, four pairs per work-group, lane = point; no block reductions.
// (One 256-thread work-group per 14-point pair spent its time in 20 block-wide reductions: 111 us for 3,200 pairs.)
__global__ __launch_bounds__(256) void k_pointset_errors_wave(const float* __restrict__ P, const float* __restrict__ T, int npairs, int npts, int group,
                                                              int want_pa, float* __restrict__ out) {
    __shared__ Align als[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pair = min(blockIdx.x * 4 + wave, npairs - 1);
    const bool live = blockIdx.x * 4 + wave < npairs;
    const float* p = P + (size_t)pair * npts * 3;
    const float* t = T + (size_t)(pair / group) * npts * 3;
    const bool ok = lane < npts;
    const float px = ok ? p[3 * lane] : 0.f, py = ok ? p[3 * lane + 1] : 0.f, pz = ok ? p[3 * lane + 2] : 0.f;
    const float tx = ok ? t[3 * lane] : 0.f, ty = ok ? t[3 * lane + 1] : 0.f, tz = ok ? t[3 * lane + 2] : 0.f;
    double s[17];
#pragma unroll
    for (int i = 0; i < 17; ++i) s[i] = 0.0;
    if (ok) accumulate17(s, px, py, pz, tx, ty, tz);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int i = 0; i < 17; ++i) s[i] += __shfl_down(s[i], o, 64);
    if (lane == 0) solve_alignment(s, npts, want_pa, als[wave]);
    __syncthreads();
    double e0 = 0.0, e1 = 0.0, e2 = 0.0;
    if (ok) errors3(als[wave], want_pa, px, py, pz, tx, ty, tz, e0, e1, e2);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        e0 += __shfl_down(e0, o, 64);
        e1 += __shfl_down(e1, o, 64);
        e2 += __shfl_down(e2, o, 64);
    }
    if (lane == 0 && live) {
        out[3 * pair + 0] = (float)(e0 / npts);
        out[3 * pair + 1] = (float)(e1 / npts);
        out[3 * pair + 2] = want_pa ? (float)(e2 / npts) : 0.f;
    }
}

// S: (B, N, npts, 3); partial: (B, chunks) sums over n and the chunk's points of |S - mean_n S|
__global__ __launch_bounds__(256) void k_sample_diversity(const float* __restrict__ S, int N, int npts, int chunks, float* __restrict__ partial) {
    __shared__ double red[4];
    const int b = blockIdx.x / chunks, ch = blockIdx.x - b * chunks;
    const int i = ch * 256 + threadIdx.x;
    double acc = 0.0;
    if (i < npts) {
        const float* s = S + ((size_t)b * N * npts + i) * 3;
        float m0 = 0.f, m1 = 0.f, m2 = 0.f;
        for (int n = 0; n < N; ++n) {
            const float* q = s + (size_t)n * npts * 3;
            m0 += q[0]; m1 += q[1]; m2 += q[2];
        }
        m0 /= N; m1 /= N; m2 /= N;
        float l2 = 0.f;
        for (int n = 0; n < N; ++n) {
            const float* q = s + (size_t)n * npts * 3;
            const float dx = q[0] - m0, dy = q[1] - m1, dz = q[2] - m2;
            l2 += sqrtf(dx * dx + dy * dy + dz * dz);
        }
        acc = l2;
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = (float)acc;
}

}  // namespace hmf

using namespace hmf;

extern "C" int hmf_pointset_errors(const float* P, const float* T, int npairs, int group, int npts, int want_pa, float* out, void* stream) {
    if (npairs == 0) return HMF_OK;
    if (!P || !T || !out || npairs < 0 || group < 1 || npts < 1) return fail(HMF_ERR_INVALID, "hmf_pointset_errors: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (npts <= 64) hipLaunchKernelGGL(k_pointset_errors_wave, dim3((npairs + 3) / 4), dim3(256), 0, st, P, T, npairs, npts, group, want_pa, out);
    else if (npts <= 256 * 4) hipLaunchKernelGGL(k_pointset_errors_reg<4>, dim3(npairs), dim3(256), 0, st, P, T, npts, group, want_pa, out);
    else if (npts <= 256 * 27) hipLaunchKernelGGL(k_pointset_errors_reg<27>, dim3(npairs), dim3(256), 0, st, P, T, npts, group, want_pa, out);
    else hipLaunchKernelGGL(k_pointset_errors, dim3(npairs), dim3(256), 0, st, P, T, npts, group, want_pa, out);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

extern "C" int hmf_sample_diversity(const float* S, int B, int N, int npts, float* partial, void* stream) {
    if (B == 0) return HMF_OK;
    if (!S || !partial || B < 0 || N < 1 || npts < 1) return fail(HMF_ERR_INVALID, "hmf_sample_diversity: bad args");
    const int chunks = (npts + 255) / 256;
    hipLaunchKernelGGL(k_sample_diversity, dim3(B * chunks), dim3(256), 0, (hipStream_t)stream, S, N, npts, chunks, partial);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}
