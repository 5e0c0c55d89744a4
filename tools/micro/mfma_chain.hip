// Microbenchmark: issue rate of v_mfma_f32_32x32x2_f32 with 1, 2, 4 independent accumulator chains per wave and 1, 2, 4 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_chain.hip -o /tmp/mfma_chain && /tmp/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = (float)c;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int CHAINS>
void run(int blocks_per_cu, float* out) {
    const int iters = 4096 / CHAINS;                     // same MFMA count per wave
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * blocks_per_cu;                // 4 waves per block: blocks_per_cu waves per SIMD
    hipLaunchKernelGGL(k<CHAINS>, dim3(grid), dim3(256), 0, 0, out, 16, 1.f, 0.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<CHAINS>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 0.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)grid * 4 * iters * 8 * CHAINS;
    printf("chains %d waves/SIMD %d: %.3f ms  %.1f TFLOP/s\n", CHAINS, blocks_per_cu, ms, mfmas * 4096 / ms / 1e9);
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int w : {1, 2, 4}) { run<1>(w, out); run<2>(w, out); run<4>(w, out); }
    return 0;
}
