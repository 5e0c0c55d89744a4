// Device-side SO(3)/so(3) closed forms shared by the standalone kernels and the fused flow kernels.
// Arithmetic follows utils/rigid_transform_utils.py of the reference, dtype hop for dtype hop
// (fp64 exp/log/log-det, fp32 6D->R and smplx-style Rodrigues); line numbers cited per function.
#pragma once
#include <hip/hip_runtime.h>

namespace hmf {

// rot6d_to_rotmat, utils/rigid_transform_utils.py:86-100.  x6 = [R11,R12,R21,R22,R31,R32].
__device__ __forceinline__ void rot6d_to_rotmat_f32(const float* __restrict__ x, float* __restrict__ R) {
    float a1[3] = {x[0], x[2], x[4]}, a2[3] = {x[1], x[3], x[5]};
    float n1 = fmaxf(sqrtf(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]), 1e-12f);  // F.normalize eps
    float b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
    float d = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
    float u[3] = {a2[0] - d * b1[0], a2[1] - d * b1[1], a2[2] - d * b1[2]};
    float n2 = fmaxf(sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), 1e-12f);
    float b2[3] = {u[0] / n2, u[1] / n2, u[2] / n2};
    float b3[3] = {b1[1] * b2[2] - b1[2] * b2[1], b1[2] * b2[0] - b1[0] * b2[2], b1[0] * b2[1] - b1[1] * b2[0]};
    // stack((b1,b2,b3), dim=-1): columns
    R[0] = b1[0]; R[1] = b2[0]; R[2] = b3[0];
    R[3] = b1[1]; R[4] = b2[1]; R[5] = b3[1];
    R[6] = b1[2]; R[7] = b2[2]; R[8] = b3[2];
}

// Pull-back of rot6d_to_rotmat_f32: G = dL/dR (row-major 3x3) -> gx = dL/dx (6), through the two F.normalize calls (gradient of a / max(|a|, eps))
// and the cross product: dL/db1 = G1 + b2 x G3, dL/db2 = G2 + G3 x b1.
__device__ __forceinline__ void rot6d_to_rotmat_bwd_f32(const float* __restrict__ x, const float* __restrict__ G, float* __restrict__ gx) {
    const float eps = 1e-12f;
    const float a1[3] = {x[0], x[2], x[4]}, a2[3] = {x[1], x[3], x[5]};
    const float n1r = sqrtf(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]), n1 = fmaxf(n1r, eps);
    const float b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
    const float d = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
    const float u[3] = {a2[0] - d * b1[0], a2[1] - d * b1[1], a2[2] - d * b1[2]};
    const float n2r = sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), n2 = fmaxf(n2r, eps);
    const float b2[3] = {u[0] / n2, u[1] / n2, u[2] / n2};
    const float G1[3] = {G[0], G[3], G[6]}, G2[3] = {G[1], G[4], G[7]}, G3[3] = {G[2], G[5], G[8]};
    float gb1[3] = {G1[0] + (b2[1] * G3[2] - b2[2] * G3[1]), G1[1] + (b2[2] * G3[0] - b2[0] * G3[2]), G1[2] + (b2[0] * G3[1] - b2[1] * G3[0])};
    const float gb2[3] = {G2[0] + (G3[1] * b1[2] - G3[2] * b1[1]), G2[1] + (G3[2] * b1[0] - G3[0] * b1[2]), G2[2] + (G3[0] * b1[1] - G3[1] * b1[0])};
    float gu[3];
    if (n2r > eps) {
        const float p = gb2[0] * b2[0] + gb2[1] * b2[1] + gb2[2] * b2[2];
#pragma unroll
        for (int k = 0; k < 3; ++k) gu[k] = (gb2[k] - p * b2[k]) / n2;
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) gu[k] = gb2[k] / eps;
    }
    const float gd = -(gu[0] * b1[0] + gu[1] * b1[1] + gu[2] * b1[2]);
    float ga2[3], ga1[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        gb1[k] += -d * gu[k] + gd * a2[k];
        ga2[k] = gu[k] + gd * b1[k];
    }
    if (n1r > eps) {
        const float p = gb1[0] * b1[0] + gb1[1] * b1[1] + gb1[2] * b1[2];
#pragma unroll
        for (int k = 0; k < 3; ++k) ga1[k] = (gb1[k] - p * b1[k]) / n1;
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) ga1[k] = gb1[k] / eps;
    }
    gx[0] = ga1[0]; gx[1] = ga2[0]; gx[2] = ga1[1]; gx[3] = ga2[1]; gx[4] = ga1[2]; gx[5] = ga2[2];
}

// so3_exp, :182-201.  R = I + alpha*X + (beta*X)@X with X = hat(v) (:142-165); the small-angle
// branch is evaluated on theta AFTER it was replaced by 1 -- harmless since X=0.
__device__ __forceinline__ void so3_exp_f64(const double v[3], double R[9]) {
    double th = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    bool big = th > 1e-10;
    double t = big ? th : 1.0;
    double s, c;
    sincos(t, &s, &c);
    double alpha = big ? s / t : 1.0 - t * t / 6.0;
    double beta = big ? (1.0 - c) / (t * t) : 0.5 - t * t / 24.0;
    double X[9] = {0.0, -v[2], v[1], v[2], 0.0, -v[0], -v[1], v[0], 0.0};
    double Y[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) Y[i] = beta * X[i];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double acc = Y[i * 3 + 0] * X[0 * 3 + j];
            acc = fma(Y[i * 3 + 1], X[1 * 3 + j], acc);
            acc = fma(Y[i * 3 + 2], X[2 * 3 + j], acc);
            R[i * 3 + j] = ((i == j) ? 1.0 : 0.0) + alpha * X[i * 3 + j] + acc;
        }
}

// so3_log_abs_det_jacobian, :298-314 (returns log(11/12) at |v| <= 1e-10: quirk 1).
__device__ __forceinline__ double so3_logdet_f64(const double v[3]) {
    double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    bool big = n > 1e-10;
    double t = big ? n : 1.0;
    double ratio = big ? (2.0 - 2.0 * cos(t)) / (t * t) : 1.0 - t * t / 12.0;
    return log(ratio);
}

// so3_vee(so3_log(R)), :204-279.  Near pi (|pi - theta| < 1e-2): magnitudes from the diagonal of the
// symmetric part, signs by re-exponentiating the 8 candidates and taking the first arg-min (:240-279).
__device__ inline void so3_log_f64(const double R[9], double v[3]) {
    const double kPi = 3.14159265358979323846;
    double tr = R[0] + R[4] + R[8];
    double ct = 0.5 * (tr - 1.0);
    ct = fmin(fmax(ct, -1.0), 1.0);
    double th = acos(ct);
    double ratio = th / sin(th);
    if (th < 1e-20) ratio = 1.0 + th * th / 6.0;
    // vee(ratio * 0.5 (R - R^T)) = (-L12, L02, -L01)
    v[0] = -(ratio * (0.5 * (R[5] - R[7])));
    v[1] = ratio * (0.5 * (R[2] - R[6]));
    v[2] = -(ratio * (0.5 * (R[1] - R[3])));
    if (fabs(kPi - th) < 1e-2) {
        double k = th * th / (1.0 - cos(th));
        double q1 = k * (R[0] - 1.0), q2 = k * (R[4] - 1.0), q3 = k * (R[8] - 1.0);
        double x[3] = {sqrt(fmax(q1 - q2 - q3, 1e-8) / 2.0), sqrt(fmax(-q1 + q2 - q3, 1e-8) / 2.0),
                       sqrt(fmax(-q1 - q2 + q3, 1e-8) / 2.0)};
        double best = 0.0;
        int best_i = -1;
        for (int s = 0; s < 8; ++s) {  // itertools.product([0,1], repeat=3): first factor slowest
            double c[3] = {((s >> 2) & 1) ? x[0] : -x[0], ((s >> 1) & 1) ? x[1] : -x[1], (s & 1) ? x[2] : -x[2]};
            double E[9];
            so3_exp_f64(c, E);
            double d = 0.0;
            // (r - r_stack).pow(2).sum(-1).sum(-1): rows summed first, then the three row sums
            for (int i = 0; i < 3; ++i) {
                double rs = 0.0;
                for (int j = 0; j < 3; ++j) {
                    double e = R[i * 3 + j] - E[i * 3 + j];
                    rs += e * e;
                }
                d += rs;
            }
            if (best_i < 0 || d < best) { best = d; best_i = s; }
        }
        v[0] = ((best_i >> 2) & 1) ? x[0] : -x[0];
        v[1] = ((best_i >> 1) & 1) ? x[1] : -x[1];
        v[2] = (best_i & 1) ? x[2] : -x[2];
    }
}

// smplx.lbs.batch_rodrigues (fp32; SURVEY App. B, quirk 3: angle = |v + 1e-8|).
__device__ __forceinline__ void batch_rodrigues_f32(const float v[3], float R[9]) {
    float a0 = v[0] + 1e-8f, a1 = v[1] + 1e-8f, a2 = v[2] + 1e-8f;
    float angle = sqrtf(a0 * a0 + a1 * a1 + a2 * a2);
    float rx = v[0] / angle, ry = v[1] / angle, rz = v[2] / angle;
    float s = sinf(angle), c = cosf(angle);
    float K[9] = {0.f, -rz, ry, rz, 0.f, -rx, -ry, rx, 0.f};
    float omc = 1.f - c;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float kk = K[i * 3 + 0] * K[0 * 3 + j];
            kk = fmaf(K[i * 3 + 1], K[1 * 3 + j], kk);
            kk = fmaf(K[i * 3 + 2], K[2 * 3 + j], kk);
            R[i * 3 + j] = ((i == j) ? 1.f : 0.f) + s * K[i * 3 + j] + omc * kk;
        }
}

}  // namespace hmf
