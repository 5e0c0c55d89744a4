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

// Distances of the error and diversity sums: v_sqrt_f32 (1 ulp) instead of the correctly rounded sqrtf expansion (~20 instructions, three per point in
// the error pass: it was most of that pass' VALU time).  Mean distances of ~1e-2 m against a 2e-5 relative bar; sqrt(0) stays an exact 0.
__device__ __forceinline__ float dist_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

__device__ inline void errors3(const Align& al, int want_pa, float px, float py, float pz, float tx, float ty, float tz, double& e0, double& e1, double& e2) {
    float dx = px - tx, dy = py - ty, dz = pz - tz;
    e0 += dist_sqrt(dx * dx + dy * dy + dz * dz);
    dx = (px - al.mu_p[0]) * al.sc_a + al.mu_t[0] - tx;
    dy = (py - al.mu_p[1]) * al.sc_a + al.mu_t[1] - ty;
    dz = (pz - al.mu_p[2]) * al.sc_a + al.mu_t[2] - tz;
    e1 += dist_sqrt(dx * dx + dy * dy + dz * dz);
    if (want_pa) {
        dx = al.pa_s * (al.R[0] * px + al.R[1] * py + al.R[2] * pz) + al.pa_t[0] - tx;
        dy = al.pa_s * (al.R[3] * px + al.R[4] * py + al.R[5] * pz) + al.pa_t[1] - ty;
        dz = al.pa_s * (al.R[6] * px + al.R[7] * py + al.R[8] * pz) + al.pa_t[2] - tz;
        e2 += dist_sqrt(dx * dx + dy * dy + dz * dz);
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
    struct P3 { float x, y, z; };                     // one 12-byte load per point (three dword loads per point made the kernel VMEM-issue bound)
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        const int i = threadIdx.x + 256 * k;
        P3 v = {0.f, 0.f, 0.f};
        if (i < npts) v = ((const P3*)p)[i];
        pv[k][0] = v.x; pv[k][1] = v.y; pv[k][2] = v.z;
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

// S: (B, N, npts, 3); partial: (B, chunks) sums over n and the chunk's 64 points of |S - mean_n S|.
// One pass (round 5): a work-group owns 64 points of one image, wave w the samples n = w, w + 4, ... -- every sample of the chunk sits in
// registers (NPT x 3 per thread, all loads in flight at once), the four partial means meet in LDS, the distances come from the registers.
// The round-1 kernel walked the N samples twice per thread with dependent scalar loads: 550 MB read for a 267 MB tensor, and 54 us of pure
// latency for the 17-joint sets.
template <int NPT>
__global__ __launch_bounds__(256) void k_sample_diversity_reg(const float* __restrict__ S, int N, int npts, int chunks, float* __restrict__ partial) {
    __shared__ float sm[4][3][64];
    __shared__ float red[4];
    const int b = blockIdx.x / chunks, ch = blockIdx.x - b * chunks;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = ch * 64 + lane;
    const bool ok = i < npts;
    const float* s = S + ((size_t)b * N * npts + (ok ? i : 0)) * 3;
    float q[NPT][3];
    float m0 = 0.f, m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        const int n = w + 4 * k;
        const bool live = ok && n < N;
        const float* r = s + (size_t)(live ? n : 0) * npts * 3;
        q[k][0] = live ? r[0] : 0.f; q[k][1] = live ? r[1] : 0.f; q[k][2] = live ? r[2] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < NPT; ++k) { m0 += q[k][0]; m1 += q[k][1]; m2 += q[k][2]; }
    sm[w][0][lane] = m0; sm[w][1][lane] = m1; sm[w][2][lane] = m2;
    __syncthreads();
    m0 = (sm[0][0][lane] + sm[1][0][lane] + sm[2][0][lane] + sm[3][0][lane]) / N;
    m1 = (sm[0][1][lane] + sm[1][1][lane] + sm[2][1][lane] + sm[3][1][lane]) / N;
    m2 = (sm[0][2][lane] + sm[1][2][lane] + sm[2][2][lane] + sm[3][2][lane]) / N;
    float l2 = 0.f;
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        const float dx = q[k][0] - m0, dy = q[k][1] - m1, dz = q[k][2] - m2;
        l2 += (ok && w + 4 * k < N) ? dist_sqrt(dx * dx + dy * dy + dz * dz) : 0.f;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) l2 += __shfl_down(l2, o, 64);
    if (lane == 0) red[w] = l2;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// any N: one wave per 64 points, two passes over the samples (the second one from L2)
__global__ __launch_bounds__(64) void k_sample_diversity(const float* __restrict__ S, int N, int npts, int chunks, float* __restrict__ partial) {
    const int b = blockIdx.x / chunks, ch = blockIdx.x - b * chunks;
    const int i = ch * 64 + threadIdx.x;
    float l2 = 0.f;
    if (i < npts) {
        const float* s = S + ((size_t)b * N * npts + i) * 3;
        float m0 = 0.f, m1 = 0.f, m2 = 0.f;
        for (int n = 0; n < N; ++n) {
            const float* q = s + (size_t)n * npts * 3;
            m0 += q[0]; m1 += q[1]; m2 += q[2];
        }
        m0 /= N; m1 /= N; m2 /= N;
        for (int n = 0; n < N; ++n) {
            const float* q = s + (size_t)n * npts * 3;
            const float dx = q[0] - m0, dy = q[1] - m1, dz = q[2] - m2;
            l2 += dist_sqrt(dx * dx + dy * dy + dz * dz);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) l2 += __shfl_down(l2, o, 64);
    if (threadIdx.x == 0) partial[blockIdx.x] = l2;
}

// ---------------------------------------------------------------------------------------------------
// k_eval_joints: the joint-side glue of the evaluation step (evaluate/evaluate_humaniflow.py:138-143,176-183,197-206) in one launch --
// from the packed (M, 90, 3) joints of the SMPL launch: the 14 H36M evaluation joints, the 17 COCO joints and their 2-D projection
// undo_keypoint_normalisation(orthographic_project(flip about x, cam)).  Row r belongs to image r / N (sample rows) or r - B N (point estimates).
// ---------------------------------------------------------------------------------------------------
struct EvalJointsArgs {
    const float* joints; const float* cam;
    float *j14, *coco, *j2d;
    int B, N, M;
    float half_wh;
    int idx14[14], idx17[17];
};
__global__ __launch_bounds__(256) void k_eval_joints(const EvalJointsArgs a) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= a.M * 31) return;
    const int row = g / 31, j = g - row * 31;
    const float* src = a.joints + ((size_t)row * 90 + (j < 14 ? a.idx14[j] : a.idx17[j - 14])) * 3;
    const float x = src[0], y = src[1], z = src[2];
    if (j < 14) {
        float* o = a.j14 + ((size_t)row * 14 + j) * 3;
        o[0] = x; o[1] = y; o[2] = z;
    } else {
        const int jj = j - 14, img = row < a.B * a.N ? row / a.N : row - a.B * a.N;
        float* o = a.coco + ((size_t)row * 17 + jj) * 3;
        o[0] = x; o[1] = y; o[2] = z;
        const float s = a.cam[img * 3], tx = a.cam[img * 3 + 1], ty = a.cam[img * 3 + 2];
        float* p = a.j2d + ((size_t)row * 17 + jj) * 2;
        p[0] = (s * (x + tx) + 1.f) * a.half_wh;                       // flip about x: (x, -y, -z)
        p[1] = (s * (-y + ty) + 1.f) * a.half_wh;
    }
}

// ---------------------------------------------------------------------------------------------------
// hmf_eval_tracker_update: everything metrics/eval_metrics_tracker.py:95-433 does with one batch AFTER the point-set kernels, in two launches:
// k_eval_tracker_frames (one work-group per image: sample minima, diversity sums, 2-D joint errors -> the per-frame arrays) and
// k_eval_tracker_sums (one thread per metric adds the B per-frame values to the fp64 running sums in image order: deterministic).
// (As one work-group for the whole batch the first version took 135 us: one lane walking 108 diversity chunks per image.)
// Metric order = humaniflow_amd/evaluate.py METRICS (20) followed by the three counts.
// ---------------------------------------------------------------------------------------------------
}  // namespace hmf
#include "../../include/humaniflow_amd.h"
namespace hmf {
enum { M_PVE, M_PVE_SC, M_PVE_PA, M_PVET, M_PVET_SC, M_MPJPE, M_MPJPE_SC, M_MPJPE_PA, M_PVE_MIN, M_PVE_SC_MIN, M_PVE_PA_MIN, M_PVET_MIN, M_PVET_SC_MIN,
       M_MPJPE_MIN, M_MPJPE_SC_MIN, M_MPJPE_PA_MIN, M_J2D, M_J2DS, M_DIV_V, M_DIV_J, C_TOTAL, C_VIS, C_VIS_S, kEvalSlots };
static_assert(kEvalSlots == HMF_EVAL_NUM_SLOTS, "slot count of hmf_eval_batch_t");

__device__ inline float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// one work-group per image: wave 0 = point-estimate copies + vertex minima, wave 1 = T-pose and joint minima, wave 2 = diversity, wave 3 = 2-D joints
__global__ __launch_bounds__(256) void k_eval_tracker_frames(const hmf_eval_batch_t a, float* __restrict__ frames) {
    const int B = a.B, N = a.N, b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    auto F = [&](int m) -> float& { return frames[(size_t)m * B + b]; };
    auto minima = [&](const float* es, int mbase, int ncol) {      // per metric: the sample with the smallest mean error (:230-280)
        if (!es || N <= 0) return;
        float v[3] = {INFINITY, INFINITY, INFINITY};
        for (int n = lane; n < N; n += 64)
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = fminf(v[c], es[((size_t)b * N + n) * 3 + c]);
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = wave_min(v[c]);
        if (lane == 0)
            for (int c = 0; c < ncol; ++c) F(mbase + c) = v[c];
    };
    if (wave == 0) {
        if (lane == 0) {
            if (a.e_verts) { F(M_PVE) = a.e_verts[3 * b]; F(M_PVE_SC) = a.e_verts[3 * b + 1]; F(M_PVE_PA) = a.e_verts[3 * b + 2]; }
            if (a.e_tpose) { F(M_PVET) = a.e_tpose[3 * b]; F(M_PVET_SC) = a.e_tpose[3 * b + 1]; }
            if (a.e_joints) { F(M_MPJPE) = a.e_joints[3 * b]; F(M_MPJPE_SC) = a.e_joints[3 * b + 1]; F(M_MPJPE_PA) = a.e_joints[3 * b + 2]; }
        }
        minima(a.es_verts, M_PVE_MIN, 3);
    } else if (wave == 1) {
        minima(a.es_tpose, M_PVET_MIN, 2);
        minima(a.es_joints, M_MPJPE_MIN, 3);
    } else if (wave == 2) {
        // diversity: the chunk partial sums of the image, in fp64
        const float* dv[2] = {a.div_verts, a.div_joints};
        const int dch[2] = {a.chunks_v, a.chunks_j}, dm[2] = {M_DIV_V, M_DIV_J}, dn[2] = {a.num_verts, 17};
        for (int t = 0; t < 2; ++t)
            if (dv[t]) {
                double d = 0.0;
                for (int c = lane; c < dch[t]; c += 64) d += (double)dv[t][(size_t)b * dch[t] + c];
                d = wave_sum(d);
                if (lane == 0) {
                    F(dm[t]) = (float)(d / ((double)N * dn[t]));
                    a.scratch[(size_t)t * B + b] = d;                     // the fp64 sum itself goes into the running sum
                }
            }
    } else if (a.j2d_target) {
        // 2-D joints (eval_metrics_tracker.py:284-359): |pred - target| over the visible joints; lane = joint (+ 17 x sample slot)
        const int j = lane % 17, slot = lane / 17;                     // 3 sample slots of 17 lanes, lanes 51..63 idle
        const bool vis_l = a.j2d_vis ? a.j2d_vis[b * 17 + j] != 0 : true;
        const float visn = (float)__popcll(__ballot(vis_l && lane < 17));
        const float tx = a.j2d_target[(b * 17 + j) * 2], ty = a.j2d_target[(b * 17 + j) * 2 + 1];
        if (a.j2d_pred) {
            float e = 0.f;
            if (vis_l && lane < 17) {
                const float dx = a.j2d_pred[(b * 17 + j) * 2] - tx, dy = a.j2d_pred[(b * 17 + j) * 2 + 1] - ty;
                e = sqrtf(dx * dx + dy * dy);
            }
            const double sumd = wave_sum((double)e);
            if (lane == 0) { F(M_J2D) = (float)sumd / visn; a.scratch[(size_t)2 * B + b] = sumd; }
        }
        if (a.j2d_samples && N > 0) {
            double acc = 0.0;
            if (vis_l && slot < 3)
                for (int n = slot; n < N; n += 3) {
                    const float* p = a.j2d_samples + (((size_t)b * N + n) * 17 + j) * 2;
                    const float dx = p[0] - tx, dy = p[1] - ty;
                    acc += (double)sqrtf(dx * dx + dy * dy);
                }
            acc = wave_sum(acc);
            if (lane == 0) { F(M_J2DS) = (float)acc / (visn * N); a.scratch[(size_t)3 * B + b] = acc; }
        }
        if (lane == 0) a.scratch[(size_t)4 * B + b] = visn;
    }
}

// running sums: one wave per slot (lanes stride the images, fixed butterfly order: deterministic)
__global__ __launch_bounds__(1024) void k_eval_tracker_sums(const hmf_eval_batch_t a, double* __restrict__ sums, const float* __restrict__ frames) {
    const int B = a.B, N = a.N, lane = threadIdx.x & 63;
    for (int m = threadIdx.x >> 6; m < kEvalSlots; m += 16) {
        if (a.slot[m] < 0) continue;
        double s = 0.0;
        bool have = true;
        if (m == C_TOTAL) s = lane == 0 ? (double)B : 0.0;
        else if (m == C_VIS || m == C_VIS_S) {
            have = a.j2d_target && (m == C_VIS ? a.j2d_pred != nullptr : (a.j2d_samples != nullptr && N > 0));
            if (have) for (int b = lane; b < B; b += 64) s += a.scratch[(size_t)4 * B + b] * (m == C_VIS ? 1.0 : (double)N);
        } else if (m == M_DIV_V || m == M_DIV_J) {
            have = (m == M_DIV_V ? a.div_verts : a.div_joints) != nullptr;
            if (have) for (int b = lane; b < B; b += 64) s += a.scratch[(size_t)(m == M_DIV_V ? 0 : 1) * B + b];
        } else if (m == M_J2D || m == M_J2DS) {
            have = a.j2d_target && (m == M_J2D ? a.j2d_pred != nullptr : (a.j2d_samples != nullptr && N > 0));
            if (have) for (int b = lane; b < B; b += 64) s += a.scratch[(size_t)(m == M_J2D ? 2 : 3) * B + b];
        } else {
            const bool vert = m <= M_PVET_SC || (m >= M_PVE_MIN && m <= M_PVET_SC_MIN);
            const bool smp = m >= M_PVE_MIN;
            const bool tp = m == M_PVET || m == M_PVET_SC || m == M_PVET_MIN || m == M_PVET_SC_MIN;
            const float* src = smp ? (tp ? a.es_tpose : (vert ? a.es_verts : a.es_joints)) : (tp ? a.e_tpose : (vert ? a.e_verts : a.e_joints));
            have = src != nullptr && (!smp || N > 0);
            const double npts = vert ? a.num_verts : a.num_joints;
            if (have) for (int b = lane; b < B; b += 64) s += (double)frames[(size_t)m * B + b];
            s *= npts;
        }
        s = wave_sum(s);
        if (have && lane == 0) sums[a.slot[m]] += s;
    }
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

extern "C" int hmf_sample_diversity_chunks(int npts) { return (npts + 63) / 64; }

extern "C" int hmf_sample_diversity(const float* S, int B, int N, int npts, float* partial, void* stream) {
    if (B == 0) return HMF_OK;
    if (!S || !partial || B < 0 || N < 1 || npts < 1) return fail(HMF_ERR_INVALID, "hmf_sample_diversity: bad args");
    const int chunks = (npts + 63) / 64;
    hipStream_t st = (hipStream_t)stream;
    if (N <= 32) hipLaunchKernelGGL(k_sample_diversity_reg<8>, dim3(B * chunks), dim3(256), 0, st, S, N, npts, chunks, partial);
    else if (N <= 64) hipLaunchKernelGGL(k_sample_diversity_reg<16>, dim3(B * chunks), dim3(256), 0, st, S, N, npts, chunks, partial);
    else if (N <= 100) hipLaunchKernelGGL(k_sample_diversity_reg<25>, dim3(B * chunks), dim3(256), 0, st, S, N, npts, chunks, partial);
    else if (N <= 128) hipLaunchKernelGGL(k_sample_diversity_reg<32>, dim3(B * chunks), dim3(256), 0, st, S, N, npts, chunks, partial);
    else hipLaunchKernelGGL(k_sample_diversity, dim3(B * chunks), dim3(64), 0, st, S, N, npts, chunks, partial);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

extern "C" int hmf_eval_joints(const float* joints, const float* cam, int B, int N, const int* idx14_host, const int* idx17_host, float img_wh, float* j14,
                               float* coco3d, float* joints2d, void* stream) {
    const long long M = (long long)B * N + B;
    if (B == 0) return HMF_OK;
    if (!joints || !cam || !idx14_host || !idx17_host || !j14 || !coco3d || !joints2d || B < 0 || N < 0 || M * 31 >= (1LL << 31))
        return fail(HMF_ERR_INVALID, "hmf_eval_joints: bad args");
    EvalJointsArgs a;
    a.joints = joints; a.cam = cam; a.j14 = j14; a.coco = coco3d; a.j2d = joints2d; a.B = B; a.N = N; a.M = (int)M; a.half_wh = img_wh / 2.0f;
    for (int i = 0; i < 14; ++i) { if (idx14_host[i] < 0 || idx14_host[i] >= 90) return fail(HMF_ERR_INVALID, "hmf_eval_joints: joint index out of the 90 output joints"); a.idx14[i] = idx14_host[i]; }
    for (int i = 0; i < 17; ++i) { if (idx17_host[i] < 0 || idx17_host[i] >= 90) return fail(HMF_ERR_INVALID, "hmf_eval_joints: joint index out of the 90 output joints"); a.idx17[i] = idx17_host[i]; }
    hipLaunchKernelGGL(k_eval_joints, dim3((unsigned)((M * 31 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

extern "C" long long hmf_eval_tracker_scratch_bytes(int B) { return (long long)5 * (B > 0 ? B : 1) * (long long)sizeof(double); }

extern "C" int hmf_eval_tracker_update(const hmf_eval_batch_t* batch, double* sums, float* frames, void* stream) {
    if (!batch || !sums || !frames) return fail(HMF_ERR_INVALID, "hmf_eval_tracker_update: null argument");
    if (batch->B == 0) return HMF_OK;
    if (batch->B < 0 || batch->N < 0 || !batch->scratch || batch->num_verts < 1 || batch->num_joints < 1)
        return fail(HMF_ERR_INVALID, "hmf_eval_tracker_update: bad sizes or no scratch (hmf_eval_tracker_scratch_bytes)");
    if ((batch->div_verts && batch->chunks_v < 1) || (batch->div_joints && batch->chunks_j < 1) || ((batch->div_verts || batch->div_joints) && batch->N < 1))
        return fail(HMF_ERR_INVALID, "hmf_eval_tracker_update: diversity partials need their chunk counts and N >= 1");
    hipLaunchKernelGGL(k_eval_tracker_frames, dim3(batch->B), dim3(256), 0, (hipStream_t)stream, *batch, frames);
    HMF_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_eval_tracker_sums, dim3(1), dim3(1024), 0, (hipStream_t)stream, *batch, sums, (const float*)frames);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}
