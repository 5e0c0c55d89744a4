// Device helpers shared by the reverse sweeps of the flow (flow_grad.hip: default configuration, MFMA-free wave-per-(row, part) kernels;
// flow_generic_grad.hip: the variants of flow_generic.hip): forward-mode dual numbers, the rational-linear spline with its pull-back to
// (x, 31 raw parameters), lane <-> spline-parameter gathers, and the pull-back of R = so3_exp(radial_tanh(x)).
#pragma once
#include "flow_common.h"
#include "so3_math.h"

namespace hmf {


// ---------------------------------------------------------------------------------------------------
// forward-mode dual numbers
// ---------------------------------------------------------------------------------------------------
template <class F, int N>
struct DualT {
    F v;
    F d[N];
};
#define HMF_DUAL_LOOP _Pragma("unroll") for (int i_ = 0; i_ < N; ++i_)
template <class F, int N> __device__ __forceinline__ DualT<F, N> dconst(F c) { DualT<F, N> r; r.v = c; HMF_DUAL_LOOP r.d[i_] = F(0); return r; }
template <class F, int N> __device__ __forceinline__ DualT<F, N> dvar(F c, int idx) { DualT<F, N> r; r.v = c; HMF_DUAL_LOOP r.d[i_] = (i_ == idx) ? F(1) : F(0); return r; }
template <class F, int N> __device__ __forceinline__ DualT<F, N> operator+(const DualT<F, N>& a, const DualT<F, N>& b) { DualT<F, N> r; r.v = a.v + b.v; HMF_DUAL_LOOP r.d[i_] = a.d[i_] + b.d[i_]; return r; }
template <class F, int N> __device__ __forceinline__ DualT<F, N> operator-(const DualT<F, N>& a, const DualT<F, N>& b) { DualT<F, N> r; r.v = a.v - b.v; HMF_DUAL_LOOP r.d[i_] = a.d[i_] - b.d[i_]; return r; }
template <class F, int N> __device__ __forceinline__ DualT<F, N> operator*(const DualT<F, N>& a, const DualT<F, N>& b) { DualT<F, N> r; r.v = a.v * b.v; HMF_DUAL_LOOP r.d[i_] = a.d[i_] * b.v + a.v * b.d[i_]; return r; }
template <class F, int N> __device__ __forceinline__ DualT<F, N> operator/(const DualT<F, N>& a, const DualT<F, N>& b) {
    DualT<F, N> r; const F inv = F(1) / b.v; r.v = a.v * inv; HMF_DUAL_LOOP r.d[i_] = (a.d[i_] - r.v * b.d[i_]) * inv; return r;
}
template <class F, int N> __device__ __forceinline__ DualT<F, N> operator+(const DualT<F, N>& a, F b) { DualT<F, N> r = a; r.v += b; return r; }
template <class F, int N> __device__ __forceinline__ DualT<F, N> operator-(const DualT<F, N>& a, F b) { DualT<F, N> r = a; r.v -= b; return r; }
template <class F, int N> __device__ __forceinline__ DualT<F, N> operator-(F a, const DualT<F, N>& b) { DualT<F, N> r; r.v = a - b.v; HMF_DUAL_LOOP r.d[i_] = -b.d[i_]; return r; }
template <class F, int N> __device__ __forceinline__ DualT<F, N> operator*(const DualT<F, N>& a, F b) { DualT<F, N> r; r.v = a.v * b; HMF_DUAL_LOOP r.d[i_] = a.d[i_] * b; return r; }
template <class F, int N> __device__ __forceinline__ DualT<F, N> operator*(F a, const DualT<F, N>& b) { return b * a; }
template <class F, int N> __device__ __forceinline__ DualT<F, N> dsqrt(const DualT<F, N>& a) { DualT<F, N> r; r.v = sqrt(a.v); const F k = F(0.5) / r.v; HMF_DUAL_LOOP r.d[i_] = a.d[i_] * k; return r; }
template <class F, int N> __device__ __forceinline__ DualT<F, N> dlog(const DualT<F, N>& a) { DualT<F, N> r; r.v = log(a.v); const F k = F(1) / a.v; HMF_DUAL_LOOP r.d[i_] = a.d[i_] * k; return r; }
template <class F, int N> __device__ __forceinline__ DualT<F, N> dabs(const DualT<F, N>& a) { DualT<F, N> r; const F s = a.v < F(0) ? F(-1) : F(1); r.v = a.v * s; HMF_DUAL_LOOP r.d[i_] = a.d[i_] * s; return r; }

// ---------------------------------------------------------------------------------------------------
// rational-linear spline: value, log|derivative| and their pull-back to (x, 31 raw parameters)
// ---------------------------------------------------------------------------------------------------
template <bool INVERSE, int N>
__device__ __forceinline__ void rls_core(const DualT<float, N>& x, const DualT<float, N>& k_lo, const DualT<float, N>& k_hi, const DualT<float, N>& c_lo,
                                         const DualT<float, N>& c_hi, const DualT<float, N>& d0, const DualT<float, N>& d1, const DualT<float, N>& lam,
                                         DualT<float, N>& out, DualT<float, N>& ld) {
    typedef DualT<float, N> D;
    const D in_w = k_hi - k_lo, in_h = c_hi - c_lo;
    const D delta = in_h / in_w;
    const D wb = dsqrt(d0 / d1);
    const D one_m = 1.f - lam;
    const D wc = (lam * d0 + one_m * wb * d1) / delta;
    const D ya = c_lo, yb = c_hi;
    const D yc = (one_m * ya + lam * wb * yb) / (one_m + lam * wb);
    D num, den, dnum;
    if (INVERSE) {
        if (x.v <= yc.v) {
            num = lam * (ya - x); den = (wc - 1.f) * x + ya - wc * yc; dnum = wc * lam * (yc - ya) * in_w;
        } else {
            num = (wc - lam * wb) * x + lam * wb * yb - wc * yc; den = (wc - wb) * x + wb * yb - wc * yc; dnum = wb * wc * one_m * (yb - yc) * in_w;
        }
        out = (num / den) * in_w + k_lo;
    } else {
        const D th = (x - k_lo) / in_w;
        if (th.v <= lam.v) {
            num = ya * (lam - th) + wc * yc * th; den = (lam - th) + wc * th; dnum = wc * lam * (yc - ya) / in_w;
        } else {
            num = wc * yc * (1.f - th) + wb * yb * (th - lam); den = wc * (1.f - th) + wb * (th - lam); dnum = wb * wc * one_m * (yb - yc) / in_w;
        }
        out = num / den;
    }
    ld = dlog(dnum) - 2.f * dlog(dabs(den));
}

struct SplineGrad {
    float g_x, g_rw[8], g_rh[8], g_rd[7], g_rl[8];
};

// out/ld as rational_linear_spline<INVERSE>; G = g_out * d out/d(.) + g_ld * d ld/d(.)
template <bool INVERSE>
__device__ __forceinline__ void spline_vjp(float x, const float (&rw)[8], const float (&rh)[8], const float (&rd)[7], const float (&rl)[8], float bound,
                                           float g_out, float g_ld, float& out, float& ld, SplineGrad& G) {
    constexpr float kScale = 1.0f - 1e-3f * 8;
    float smw[8], smh[8], cw[9], chh[9], dv[9];
    {
        float m = rw[0], n = rh[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) { m = fmaxf(m, rw[i]); n = fmaxf(n, rh[i]); }
        float s = 0.f, t = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { smw[i] = expf(rw[i] - m); s += smw[i]; smh[i] = expf(rh[i] - n); t += smh[i]; }
#pragma unroll
        for (int i = 0; i < 8; ++i) { smw[i] /= s; smh[i] /= t; }
        float a = 0.f, b = 0.f;
        cw[0] = -bound; chh[0] = -bound;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            a += 1e-3f + kScale * smw[i]; b += 1e-3f + kScale * smh[i];
            cw[i + 1] = (2.f * bound) * a + (-bound);
            chh[i + 1] = (2.f * bound) * b + (-bound);
        }
        cw[8] = bound; chh[8] = bound;
    }
    dv[0] = 1.0f - 1e-3f; dv[8] = 1.0f - 1e-3f;
#pragma unroll
    for (int i = 0; i < 7; ++i) dv[i + 1] = 1e-3f + softplus1(rd[i]);
    int bin = -1;
#pragma unroll
    for (int i = 0; i < 9; ++i) bin += (x >= (INVERSE ? chh[i] : cw[i]) + 1e-6f) ? 1 : 0;
    bin = min(max(bin, 0), 7);
    float k_lo = cw[0], k_hi = cw[1], c_lo = chh[0], c_hi = chh[1], d0 = dv[0], d1 = dv[1], rls = rl[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) {
        const bool sel = bin == i;
        k_lo = sel ? cw[i] : k_lo; k_hi = sel ? cw[i + 1] : k_hi; c_lo = sel ? chh[i] : c_lo; c_hi = sel ? chh[i + 1] : c_hi;
        d0 = sel ? dv[i] : d0; d1 = sel ? dv[i + 1] : d1; rls = sel ? rl[i] : rls;
    }
    const float sl = sigmoid1(rls);
    const float lamv = (1.f - 2.f * 0.025f) * sl + 0.025f;
    typedef DualT<float, 8> D;
    D O, L;
    rls_core<INVERSE, 8>(dvar<float, 8>(x, 0), dvar<float, 8>(k_lo, 1), dvar<float, 8>(k_hi, 2), dvar<float, 8>(c_lo, 3), dvar<float, 8>(c_hi, 4),
                         dvar<float, 8>(d0, 5), dvar<float, 8>(d1, 6), dvar<float, 8>(lamv, 7), O, L);
    const bool inside = (x >= -bound) && (x <= bound);
    out = inside ? O.v : x;
    ld = inside ? L.v : 0.f;
    float g[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) g[i] = inside ? (g_out * O.d[i] + g_ld * L.d[i]) : 0.f;
    G.g_x = inside ? g[0] : g_out;
    // knots -> bin lengths -> softmax logits.  cw[i] = 2B sum_{m<i} w'_m - B for 1 <= i <= 7 (the end knots are pinned constants)
    float g_wp[8], g_hp[8];
    {
        float accw = 0.f, acch = 0.f;
#pragma unroll
        for (int m = 7; m >= 0; --m) {
            g_wp[m] = 2.f * bound * accw; g_hp[m] = 2.f * bound * acch;
            // add the gradient of knot m (i = m) for the next (smaller) m; knot 0 is never needed
            accw += ((bin == m) ? g[1] : 0.f) + ((bin + 1 == m) ? g[2] : 0.f);
            acch += ((bin == m) ? g[3] : 0.f) + ((bin + 1 == m) ? g[4] : 0.f);
        }
        float dotw = 0.f, doth = 0.f;
#pragma unroll
        for (int m = 0; m < 8; ++m) { dotw += g_wp[m] * smw[m]; doth += g_hp[m] * smh[m]; }
#pragma unroll
        for (int m = 0; m < 8; ++m) { G.g_rw[m] = kScale * smw[m] * (g_wp[m] - dotw); G.g_rh[m] = kScale * smh[m] * (g_hp[m] - doth); }
    }
#pragma unroll
    for (int i = 0; i < 7; ++i) {                   // dv[i + 1] = 1e-3 + softplus(rd[i]); d0 = dv[bin], d1 = dv[bin + 1]
        const float gd = ((bin == i + 1) ? g[5] : 0.f) + ((bin == i) ? g[6] : 0.f);
        G.g_rd[i] = gd * sigmoid1(rd[i]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) G.g_rl[i] = (bin == i) ? g[7] * (1.f - 2.f * 0.025f) * sl * (1.f - sl) : 0.f;
}

// ---------------------------------------------------------------------------------------------------
// wave-level helpers (lane = channel)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s);
    return v;
}

// spline parameters of dimension `dim` out of the 62-vector raw[lane]: (w 16 | h 16 | d 14 | l 16)
__device__ __forceinline__ void gather_raw(float raw, int dim, float (&rw)[8], float (&rh)[8], float (&rd)[7], float (&rl)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        rw[i] = __shfl(raw, dim * 8 + i);
        rh[i] = __shfl(raw, 16 + dim * 8 + i);
        rl[i] = __shfl(raw, 46 + dim * 8 + i);
    }
#pragma unroll
    for (int i = 0; i < 7; ++i) rd[i] = __shfl(raw, 32 + dim * 7 + i);
}

// inverse of gather_raw for gradients: lanes 0 and 1 hold dimension 0 / 1's SplineGrad; returns d L / d raw[lane]
__device__ __forceinline__ float scatter_raw_grad(const SplineGrad& S, int lane) {
    float g = 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float a = __shfl(S.g_rw[i], q), b = __shfl(S.g_rh[i], q), c = __shfl(S.g_rl[i], q);
            g = (lane == q * 8 + i) ? a : g;
            g = (lane == 16 + q * 8 + i) ? b : g;
            g = (lane == 46 + q * 8 + i) ? c : g;
        }
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const float a = __shfl(S.g_rd[i], q);
            g = (lane == 32 + q * 7 + i) ? a : g;
        }
    }
    return g;
}


// R = so3_exp(v) in fp64 (utils/rigid_transform_utils.py:182-201), v = scaled radial tanh of x (scaled_radial_tanh_transform.py:28-35):
// gpre[i] = sum_e g_R[e] d R[e] / d x[i], by dual numbers over x (fp32, tanh) and over v (fp64, exp map).  gr: the 9 upstream gradients.
__device__ __forceinline__ void exp_tanh_vjp(float x0, float x1, float x2, float bound, const float* __restrict__ gr, float (&gpre)[3]) {
    float gv[3] = {0.f, 0.f, 0.f};          // d L / d v
    typedef DualT<float, 3> D3;
    const D3 a0 = dvar<float, 3>(x0, 0), a1 = dvar<float, 3>(x1, 1), a2 = dvar<float, 3>(x2, 2);
    D3 v[3] = {a0, a1, a2};
    const float nrm = sqrtf(x0 * x0 + x1 * x1 + x2 * x2);
    if (nrm > 1e-7f) {
        const D3 nd = dsqrt(a0 * a0 + a1 * a1 + a2 * a2);
        const D3 q = nd * (1.f / bound);
        D3 th;                                                 // tanh(q)
        th.v = tanhf(q.v);
#pragma unroll
        for (int i = 0; i < 3; ++i) th.d[i] = (1.f - th.v * th.v) * q.d[i];
        const D3 sc = th * bound / nd;
        v[0] = sc * a0; v[1] = sc * a1; v[2] = sc * a2;
    }
    typedef DualT<double, 3> E3;
    const E3 vd[3] = {dvar<double, 3>((double)v[0].v, 0), dvar<double, 3>((double)v[1].v, 1), dvar<double, 3>((double)v[2].v, 2)};
    const double th0 = sqrt(vd[0].v * vd[0].v + vd[1].v * vd[1].v + vd[2].v * vd[2].v);
    E3 alpha, beta;
    if (th0 > 1e-10) {
        const E3 t = dsqrt(vd[0] * vd[0] + vd[1] * vd[1] + vd[2] * vd[2]);
        double s, c;
        sincos(t.v, &s, &c);
        E3 sn, cs;
        sn.v = s; cs.v = c;
#pragma unroll
        for (int i = 0; i < 3; ++i) { sn.d[i] = c * t.d[i]; cs.d[i] = -s * t.d[i]; }
        alpha = sn / t;
        beta = (1.0 - cs) / (t * t);
    } else {                                                   // quirk 1: Taylor branch evaluated at theta = 1, constants
        alpha = dconst<double, 3>(1.0 - 1.0 / 6.0);
        beta = dconst<double, 3>(0.5 - 1.0 / 24.0);
    }
    const E3 zero = dconst<double, 3>(0.0);
    const E3 X[9] = {zero, 0.0 - vd[2], vd[1], vd[2], zero, 0.0 - vd[0], 0.0 - vd[1], vd[0], zero};
    double gvd[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            E3 acc = (beta * X[i * 3 + 0]) * X[0 * 3 + k] + (beta * X[i * 3 + 1]) * X[1 * 3 + k] + (beta * X[i * 3 + 2]) * X[2 * 3 + k];
            const E3 Rik = alpha * X[i * 3 + k] + acc;          // + identity: constant
            const double gg = (double)gr[i * 3 + k];
            gvd[0] += gg * Rik.d[0]; gvd[1] += gg * Rik.d[1]; gvd[2] += gg * Rik.d[2];
        }
    gv[0] = (float)gvd[0]; gv[1] = (float)gvd[1]; gv[2] = (float)gvd[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) gpre[i] = gv[0] * v[0].d[i] + gv[1] * v[1].d[i] + gv[2] * v[2].d[i];
}

}  // namespace hmf
