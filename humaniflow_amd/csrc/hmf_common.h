// Shared host/device helpers for libhumaniflow_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include "../../include/humaniflow_amd.h"

namespace hmf {

// thread-local last-error text returned by hmf_last_error()
char* err_buf();
int fail(int code, const char* fmt, ...);

#define HMF_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return hmf::fail(HMF_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                             __FILE__, __LINE__);                                              \
    } while (0)

#define HMF_LAUNCH_CHECK()                                                                     \
    do {                                                                                       \
        hipError_t e_ = hipGetLastError();                                                     \
        if (e_ != hipSuccess)                                                                  \
            return hmf::fail(HMF_ERR_HIP, "kernel launch failed: %s (%s:%d)",                   \
                             hipGetErrorString(e_), __FILE__, __LINE__);                       \
    } while (0)

// optional per-launch HIP-event timing (hmf_profile_enable / hmf_profile_collect)
bool prof_on();
void prof_begin(int tag, int rows, hipStream_t st);
void prof_end(hipStream_t st);
struct ProfScope {
    hipStream_t st;
    bool on;
    ProfScope(int tag, int rows, hipStream_t s) : st(s), on(prof_on()) { if (on) prof_begin(tag, rows, st); }
    ~ProfScope() { if (on) prof_end(st); }
};

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Work-group ids are dealt to the 8 XCDs round-robin (id & 7), each XCD with its own L2.  A GEMM-shaped launch of MT x NT tiles reads an
// activation panel per m-tile and a filter panel per n-tile; splitting the XCDs a x b (a b = 8) over (m, n) makes every activation byte cross into
// b L2s and every filter byte into a: HBM reads ~ b * act + a * filt.  The all-m split (b = 1: every XCD streams ALL filters) is right while
// activations dominate (layers 1-3) and 3-7x too much traffic for the deep layers on small maps (layer4: 138 MB read for 21 MB of operands).
// Returns b in {1, 2, 4, 8}; 1 keeps the contiguous-run order.
static inline int xcd_n_split(long long act_bytes, long long filt_bytes, int MT, int NT) {
    if ((((long long)MT * NT) & 7) != 0) return 1;
    int best = 1;
    long long cost = act_bytes + 8 * filt_bytes;
    for (int b = 2; b <= 8; b *= 2) {
        const int a = 8 / b;
        if (NT % b || MT % a) continue;
        const long long c = b * act_bytes + a * filt_bytes;
        if (c < cost) { cost = c; best = b; }
    }
    return best;
}

// Zero `bytes` (a multiple of 4, pointer 4-byte aligned) on `st` with a KERNEL.  hipMemsetAsync is not used by this library: captured into a HIP
// graph (torch.cuda.graph) its memset node was observed not to clear the buffer on later replays (ROCm 7.2) -- the pooled stem map and the
// weight-gradient accumulators then carried the previous replay's values into their atomic updates.
int zero_async(void* ptr, size_t bytes, hipStream_t st);

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kV = HMF_NUM_VERTS;          // 6890
constexpr int kJ = HMF_NUM_SMPL_JOINTS;    // 24
constexpr int kP = HMF_NUM_PARTS;          // 23
constexpr int kOutJ = HMF_NUM_OUT_JOINTS;  // 90

// Small launches (the predict shape: one image): with fewer than 128 work-groups on 256 CUs the reduction loop of a convolution (slabs of 64 input
// channels) is split over up to 8 work-groups per tile; the raw partial outputs are added in a fixed order by splitk_finish (conv_winograd.hip),
// which also applies bias / residual / ReLU.  Returns the number of slab ranges (a divisor of nslab; 1 = no split).
static inline int small_launch_ksplit(long long blocks, int nslab) {
    if (blocks >= 128 || nslab < 2) return 1;
    int best = 1;
    for (int k = 2; k <= nslab; ++k)
        if (nslab % k == 0 && blocks * k <= 512) best = k;
    return best;
}
// y[n] = relu?(sum_s part[s][n] + bias[n % Cout] + res[n]); n and Cout multiples of 4, all pointers 16-byte aligned
int splitk_finish(const float* part, int ksplit, long long n, const float* bias, const float* res, float* y, int Cout, int relu, hipStream_t st);

#ifdef __HIPCC__
// logical tile (t = mt * NT + nt, n fastest) of work-group `bid` under an a x xb split of the XCDs (xcd_n_split); xb = 1: one contiguous run per XCD
__device__ __forceinline__ int xcd_tile(int bid, int MT, int NT, int xb) {
    const int T = MT * NT;
    if ((T & 7) != 0) return bid;
    if (xb <= 1) return (bid & 7) * (T >> 3) + (bid >> 3);
    const int x = bid & 7, j = bid >> 3, a = 8 / xb;
    const int npg = NT / xb, mpg = MT / a;
    (void)mpg;
    return ((x / xb) * (MT / a) + j / npg) * NT + (x % xb) * npg + j % npg;
}
#endif

}  // namespace hmf
