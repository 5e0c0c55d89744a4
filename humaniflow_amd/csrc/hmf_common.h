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

}  // namespace hmf
