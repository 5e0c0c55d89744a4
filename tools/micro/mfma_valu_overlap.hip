// Microbenchmark: do v_mfma_f32_32x32x2_f32 (fp32 MFMA) and fp32 VALU work of ANOTHER wave on the same SIMD overlap on gfx950?
// One work-group of 512 threads per CU: waves 0-3 (one per SIMD) issue NM MFMAs on 4 independent accumulators, waves 4-7 (the second wave of each
// SIMD) issue NV independent v_fma_f32 (and, separately, ds_read_b128 / nothing).  Reported: shader-clock cycles of the MFMA waves alone, of the
// VALU waves alone, and of both together.  Separate pipes -> together ~ max(alone); a shared datapath -> together ~ sum.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu_overlap.hip -o /tmp/mvo && /tmp/mvo
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// mode bit 0: waves 0-3 run MFMAs; bit 1: waves 4-7 run VALU fmas; bit 2: waves 4-7 run ds_read_b128; bit 3: waves 4-7 run bf16-free integer VALU (v_add_u32)
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int nm, int nv, int mode, float a, float b) {
    __shared__ float lds[4096];
    const int wave = threadIdx.x >> 6;
    lds[threadIdx.x] = a;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    float s = 0.f;
    if (wave < 4) {
        if (mode & 1) {
            f32x16 acc[4];
            for (int c = 0; c < 4; ++c)
                for (int r = 0; r < 16; ++r) acc[c][r] = (float)c;
            for (int i = 0; i < nm / 4; ++i) {
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
            }
            for (int c = 0; c < 4; ++c)
                for (int r = 0; r < 16; ++r) s += acc[c][r];
        }
    } else {
        if (mode & 2) {
            float v[8];
            for (int j = 0; j < 8; ++j) v[j] = a + j;
            for (int i = 0; i < nv / 8; ++i) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaf(v[j], b, a);
            }
            for (int j = 0; j < 8; ++j) s += v[j];
        }
        if (mode & 4) {
            f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
            for (int i = 0; i < nv / 8; ++i) {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc4 += *(const volatile f32x4*)&lds[((threadIdx.x & 63) * 4 + j * 256) & 4095];
            }
            s += acc4[0] + acc4[1] + acc4[2] + acc4[3];
        }
        if (mode & 8) {
            unsigned v[8];
            for (int j = 0; j < 8; ++j) v[j] = threadIdx.x + j;
            for (int i = 0; i < nv / 8; ++i) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = v[j] * 3u + (unsigned)i;
            }
            for (int j = 0; j < 8; ++j) s += (float)v[j];
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    long long h[256 * 8];
    const int nm = 4096, nv = 32768;
    const char* names[] = {"", "MFMA waves alone", "fp32 VALU waves alone", "MFMA + fp32 VALU", "ds_read_b128 waves alone", "MFMA + ds_read_b128", "", "", "int VALU waves alone", "MFMA + int VALU"};
    for (int mode : {1, 2, 3, 4, 5, 8, 9}) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, cyc, nm, nv, mode, 1.0001f, 0.5f);
        hipDeviceSynchronize();
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double m = 0, v = 0;
        for (int b = 0; b < 256; ++b) { for (int w = 0; w < 4; ++w) m += h[b * 8 + w]; for (int w = 4; w < 8; ++w) v += h[b * 8 + w]; }
        printf("%-28s: MFMA waves %8.0f cycles (%d MFMAs x 64 = %d), second waves %8.0f cycles (%d instructions)\n", names[mode], m / 1024, nm, nm * 64, v / 1024, nv);
    }
    return 0;
}
