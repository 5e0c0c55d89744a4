// Standalone element-wise SO(3) kernels (one rotation per lane) + error plumbing of the C ABI.
// They back humaniflow_amd/rigid_transform_utils.py; the fused flow kernels inline the same
// device functions from so3_math.h.
#include <algorithm>
#include <cstdint>
#include "hmf_common.h"
#include "so3_math.h"

namespace hmf {

char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

__global__ __launch_bounds__(256) void k_zero(unsigned int* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0u;
}

int zero_async(void* ptr, size_t bytes, hipStream_t st) {
    if (bytes == 0) return HMF_OK;
    if (!ptr || (bytes & 3) || ((uintptr_t)ptr & 3)) return fail(HMF_ERR_INVALID, "zero_async: %zu bytes at %p (4-byte granularity)", bytes, ptr);
    const size_t n = bytes / 4;
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(k_zero, dim3(blocks), dim3(256), 0, st, (unsigned int*)ptr, n);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

// ---- per-launch event timing --------------------------------------------------------------------
namespace {
constexpr int kProfMax = 16384;
struct ProfRec { hipEvent_t a, b; int tag, rows; };
ProfRec g_prof[kProfMax];
int g_prof_n = 0;
int g_prof_pool = 0;               // events [0, g_prof_pool) exist (created outside any stream capture, reused across enable() calls)
bool g_prof_skip = false;
bool g_prof_on = false;
}  // namespace
bool prof_on() { return g_prof_on; }
// Launches issued while their stream is being captured into a HIP graph are not timed: a captured event is graph-internal (it cannot be
// synchronised or timed from the host), and the external-event-node form (hipEventRecordWithFlags(..., hipEventRecordExternal)) returns
// "invalid argument" under the HIP runtime PyTorch 2.10+rocm7.0 loads (it works against /opt/rocm 7.2: tools/micro/graph_event.hip).
static bool capturing(hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive;
}
void prof_begin(int tag, int rows, hipStream_t st) {
    g_prof_skip = true;
    if (g_prof_n >= kProfMax || capturing(st)) return;
    ProfRec& r = g_prof[g_prof_n];
    if (g_prof_n >= g_prof_pool) {                 // pool exhausted: create on demand
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
        g_prof_pool = g_prof_n + 1;
    }
    g_prof_skip = false;
    r.tag = tag; r.rows = rows;
    (void)hipEventRecord(r.a, st);
}
void prof_end(hipStream_t st) {
    if (g_prof_skip) return;
    (void)hipEventRecord(g_prof[g_prof_n].b, st);
    ++g_prof_n;
}

__global__ void k_rot6d(const float* __restrict__ x, int n, float* __restrict__ R) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float in[6], out[9];
#pragma unroll
    for (int k = 0; k < 6; ++k) in[k] = x[(size_t)i * 6 + k];
    rot6d_to_rotmat_f32(in, out);
#pragma unroll
    for (int k = 0; k < 9; ++k) R[(size_t)i * 9 + k] = out[k];
}

__global__ void k_rot6d_bwd(const float* __restrict__ x, const float* __restrict__ g, int n, float* __restrict__ gx) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float in[6], gr[9], out[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) in[k] = x[(size_t)i * 6 + k];
#pragma unroll
    for (int k = 0; k < 9; ++k) gr[k] = g[(size_t)i * 9 + k];
    rot6d_to_rotmat_bwd_f32(in, gr, out);
#pragma unroll
    for (int k = 0; k < 6; ++k) gx[(size_t)i * 6 + k] = out[k];
}

__global__ void k_so3_exp(const double* __restrict__ v, int n, double* __restrict__ R) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double in[3] = {v[(size_t)i * 3], v[(size_t)i * 3 + 1], v[(size_t)i * 3 + 2]}, out[9];
    so3_exp_f64(in, out);
#pragma unroll
    for (int k = 0; k < 9; ++k) R[(size_t)i * 9 + k] = out[k];
}

__global__ void k_so3_log(const double* __restrict__ R, int n, double* __restrict__ v) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double in[9], out[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) in[k] = R[(size_t)i * 9 + k];
    so3_log_f64(in, out);
    v[(size_t)i * 3] = out[0];
    v[(size_t)i * 3 + 1] = out[1];
    v[(size_t)i * 3 + 2] = out[2];
}

__global__ void k_so3_logdet(const double* __restrict__ v, int n, double* __restrict__ o) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double in[3] = {v[(size_t)i * 3], v[(size_t)i * 3 + 1], v[(size_t)i * 3 + 2]};
    o[i] = so3_logdet_f64(in);
}

__global__ void k_rodrigues(const float* __restrict__ v, int n, float* __restrict__ R) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float in[3] = {v[(size_t)i * 3], v[(size_t)i * 3 + 1], v[(size_t)i * 3 + 2]}, out[9];
    batch_rodrigues_f32(in, out);
#pragma unroll
    for (int k = 0; k < 9; ++k) R[(size_t)i * 9 + k] = out[k];
}

// the same for whole SMPL poses: row r of `aa` = 24 axis-angle vectors (global orientation first); the rotations leave in the two tensors the SMPL
// launch takes (smplx's global_orient / body_pose split), so that no slice has to be copied
__global__ void k_rodrigues_pose(const float* __restrict__ aa, int M, float* __restrict__ glob, float* __restrict__ body) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * 24) return;
    const int row = i / 24, j = i - row * 24;
    float in[3] = {aa[(size_t)i * 3], aa[(size_t)i * 3 + 1], aa[(size_t)i * 3 + 2]}, out[9];
    batch_rodrigues_f32(in, out);
    float* o = j == 0 ? glob + (size_t)row * 9 : body + ((size_t)row * 23 + (j - 1)) * 9;
#pragma unroll
    for (int k = 0; k < 9; ++k) o[k] = out[k];
}

}  // namespace hmf

namespace hmf {
__global__ void k_clock_probe(long long ticks, long long* out) {
    const long long w0 = (long long)wall_clock64(), c0 = (long long)__builtin_readcyclecounter();
    long long w1 = w0;
    while (w1 - w0 < ticks) {
        __builtin_amdgcn_s_sleep(32);
        w1 = (long long)wall_clock64();
    }
    const long long c1 = (long long)__builtin_readcyclecounter();
    if (threadIdx.x == 0) {
        out[0] = c1 - c0;
        out[1] = w1 - w0;
    }
}
}  // namespace hmf

using namespace hmf;

extern "C" {

int hmf_abi_version(void) { return HMF_ABI_VERSION; }
const char* hmf_last_error(void) { return err_buf(); }
int hmf_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int hmf_profile_enable(int enable) {
    g_prof_n = 0;
    g_prof_on = enable != 0;
    // events are pooled: hipEventCreate is not allowed while a stream capture is active, so a first block exists before any launch
    if (g_prof_on)
        for (; g_prof_pool < 2048; ++g_prof_pool)
            if (hipEventCreate(&g_prof[g_prof_pool].a) != hipSuccess || hipEventCreate(&g_prof[g_prof_pool].b) != hipSuccess) break;
    return HMF_OK;
}

int hmf_profile_collect(double* ms, int32_t* tag, int32_t* rows, int max) {
    if (!ms || !tag || !rows || max < 0) return fail(HMF_ERR_INVALID, "hmf_profile_collect: bad args");
    int n = g_prof_n < max ? g_prof_n : max;
    for (int i = 0; i < n; ++i) {
        HMF_HIP(hipEventSynchronize(g_prof[i].b));
        float t = 0.f;
        HMF_HIP(hipEventElapsedTime(&t, g_prof[i].a, g_prof[i].b));
        ms[i] = t; tag[i] = g_prof[i].tag; rows[i] = g_prof[i].rows;
    }
    return n;
}

/* One wave spins for ~spin_us of wall time and reports how far the shader-clock counter (s_memtime) and the constant 100 MHz counter
 * (s_memrealtime) advanced: out_dev[0] / out_dev[1] * 100 = the shader clock in MHz that the CU it ran on sustained meanwhile.  Launched on a
 * side stream beside the measured work, it occupies one wave slot of one CU: the clock the DVFS governor grants under THAT load (the
 * 157.3 TFLOP/s fp32 MFMA peak assumes 2400 MHz). */
int hmf_profile_clock_probe(int spin_us, long long* out_dev, void* stream) {
    if (!out_dev || spin_us < 1 || spin_us > 2000000) return fail(HMF_ERR_INVALID, "hmf_profile_clock_probe: bad args");
    hipLaunchKernelGGL(hmf::k_clock_probe, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long)spin_us * 100, out_dev);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

#define HMF_ELEMENTWISE(name, kern, TIN, TOUT)                                                  \
    int name(const TIN* in, int n, TOUT* out, void* stream) {                                   \
        if (n < 0 || (n > 0 && (!in || !out))) return fail(HMF_ERR_INVALID, #name ": bad args"); \
        if (n == 0) return HMF_OK;                                                              \
        hipLaunchKernelGGL(kern, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, in, n, out); \
        HMF_LAUNCH_CHECK();                                                                     \
        return HMF_OK;                                                                          \
    }

HMF_ELEMENTWISE(hmf_rot6d_to_rotmat, k_rot6d, float, float)
// dL/dx6 (n, 6) from dL/dR (n, 3, 3): the pull-back of hmf_rot6d_to_rotmat (one launch; round 2 re-traced the Gram-Schmidt map in ~45 torch launches)
int hmf_rot6d_to_rotmat_backward(const float* x6, const float* gR, int n, float* gx6, void* stream) {
    if (n < 0 || (n > 0 && (!x6 || !gR || !gx6))) return fail(HMF_ERR_INVALID, "hmf_rot6d_to_rotmat_backward: bad args");
    if (n == 0) return HMF_OK;
    hipLaunchKernelGGL(k_rot6d_bwd, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x6, gR, n, gx6);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}
HMF_ELEMENTWISE(hmf_so3_exp, k_so3_exp, double, double)
HMF_ELEMENTWISE(hmf_so3_log, k_so3_log, double, double)
HMF_ELEMENTWISE(hmf_so3_log_abs_det_jacobian, k_so3_logdet, double, double)
HMF_ELEMENTWISE(hmf_batch_rodrigues, k_rodrigues, float, float)

int hmf_batch_rodrigues_pose(const float* aa, int M, float* global_orient, float* body_pose, void* stream) {
    if (M < 0 || (M > 0 && (!aa || !global_orient || !body_pose))) return fail(HMF_ERR_INVALID, "hmf_batch_rodrigues_pose: bad args");
    if (M == 0) return HMF_OK;
    hipLaunchKernelGGL(k_rodrigues_pose, dim3((M * 24 + 255) / 256), dim3(256), 0, (hipStream_t)stream, aa, M, global_orient, body_pose);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

}  // extern "C"
