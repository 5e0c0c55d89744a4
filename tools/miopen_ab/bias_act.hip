// Fused convolution epilogue for the ResNet encoder: y = relu?(x + bias[channel] (+ residual)), in place, one HBM pass.
// The convolutions themselves stay on MIOpen (north star: the encoder is PyTorch-ROCm); eager PyTorch spends a separate
// elementwise pass on each of bias-add / residual-add / ReLU (7-8 passes per bottleneck), which this kernel folds into
// the single pass per convolution that is unavoidable without a fused conv epilogue.  HBM-bound: 16 B per lane,
// grid-stride, NCHW (one channel per float4, H*W % 4 == 0) or NHWC (C % 4 == 0, bias read as float4).
#include "hmf_common.h"

namespace hmf {

template <bool NHWC, bool RES, bool RELU>
__global__ __launch_bounds__(256) void k_bias_act(f32x4* __restrict__ x, const float* __restrict__ bias,
                                                  const f32x4* __restrict__ res, long long n4, int C, int hw4) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 v = x[i];
        if (NHWC) {
            const int c4 = (int)(i % (C / 4));
            v += *(const f32x4*)(bias + c4 * 4);
        } else {
            const int c = (int)((i / hw4) % C);
            v += bias[c];
        }
        if (RES) v += res[i];
        if (RELU) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        x[i] = v;
    }
}

}  // namespace hmf

using namespace hmf;

extern "C" int hmf_bias_act(float* x, const float* bias, const float* residual, int N, int C, int HW, int nhwc, int relu,
                            void* stream) {
    if (!x || !bias || N <= 0 || C <= 0 || HW <= 0) return fail(HMF_ERR_INVALID, "hmf_bias_act: bad args");
    if (nhwc ? (C % 4 != 0) : (HW % 4 != 0)) return fail(HMF_ERR_UNSUPPORTED, "hmf_bias_act: C=%d HW=%d not float4-tileable", C, HW);
    if (((uintptr_t)x & 15) || (residual && ((uintptr_t)residual & 15)) || (nhwc && ((uintptr_t)bias & 15)))
        return fail(HMF_ERR_INVALID, "hmf_bias_act: pointers must be 16-byte aligned");
    const long long n4 = (long long)N * C * HW / 4;
    const int hw4 = HW / 4;
    long long blocks = (n4 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipStream_t st = (hipStream_t)stream;
    f32x4* xv = (f32x4*)x;
    const f32x4* rv = (const f32x4*)residual;
#define LAUNCH(A, B, Cc) hipLaunchKernelGGL((k_bias_act<A, B, Cc>), dim3((unsigned)blocks), dim3(256), 0, st, xv, bias, rv, n4, C, hw4)
    if (nhwc) {
        if (residual) { if (relu) LAUNCH(true, true, true); else LAUNCH(true, true, false); }
        else          { if (relu) LAUNCH(true, false, true); else LAUNCH(true, false, false); }
    } else {
        if (residual) { if (relu) LAUNCH(false, true, true); else LAUNCH(false, true, false); }
        else          { if (relu) LAUNCH(false, false, true); else LAUNCH(false, false, false); }
    }
#undef LAUNCH
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}
