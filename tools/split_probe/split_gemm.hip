// Feasibility probe (tools only, NOT part of libhumaniflow_amd.so): an fp32 GEMM  C[M][N] = A[M][K] . B[N][K]^T  computed on the bf16 MFMA with a
// three-way split of every fp32 operand (x = x0 + x1 + x2, each a bf16: 24 mantissa bits in total) and the six products x_i y_j with i + j <= 2
// (fp32 accumulate): the dropped products are <= 2^-24 relative, i.e. the result is as accurate as an fp32-MFMA GEMM.  Six 32x32x16 bf16 MFMAs
// (32 cycles each) replace eight 32x32x2 fp32 MFMAs (64 cycles each) per 16 k: 2.67x fewer MFMA cycles.  The probe measures what is left of that
// after the operand split (VALU at the LDS-store), the 1.5x LDS bytes and the rest of a plain 64x64-tile kernel.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC split_gemm.hip -o libsplit_gemm.so
#include <hip/hip_runtime.h>
#include <cstdint>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned short bf16_rn(float x) {          // round to nearest even (inputs are finite)
    const unsigned u = __float_as_uint(x);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float bf16_f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

__device__ __forceinline__ void split3(float x, unsigned short& h0, unsigned short& h1, unsigned short& h2) {
    h0 = bf16_rn(x);
    const float r1 = x - bf16_f(h0);
    h1 = bf16_rn(r1);
    h2 = bf16_rn(r1 - bf16_f(h1));
}

// B pre-split once (weights): Bs[3][N][K] bf16
__global__ void k_split_rows(const float* __restrict__ w, unsigned short* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned short a, b, c;
    split3(w[i], a, b, c);
    out[i] = a; out[n + i] = b; out[2 * n + i] = c;
}

constexpr int KC = 32, S = KC + 8;           // k-chunk, LDS row stride in bf16 (80 B: 16-byte aligned rows, conflict-free 16-byte reads)

// 64 x 64 tile, 4 waves (2 x 2), fp32 A split on the fly at the LDS store, B read pre-split.  One-chunk-ahead register prefetch.
__global__ __launch_bounds__(256) void k_split_gemm(const float* __restrict__ A, const unsigned short* __restrict__ Bs, float* __restrict__ C, int M, int N, int K) {
    __shared__ unsigned short sA[3][64 * S];
    __shared__ unsigned short sB[3][64 * S];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, li = lane & 31, lq = lane >> 5;
    const int NT = N / 64;
    const int m0 = (blockIdx.x / NT) * 64, n0 = (blockIdx.x % NT) * 64;
    const long long NK = (long long)N * K;
    // A: 64 rows x 32 floats = 512 float4 -> 2 per thread; B: 3 x 64 rows x 32 bf16 = 3 x 256 x 16 B -> 3 per thread
    const int ar = tid >> 3, ac = (tid & 7) * 4;
    const int br = tid >> 2, bc = (tid & 3) * 8;
    f32x4 pa[2];
    u16x8 pb[3];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) pa[i] = *(const f32x4*)(A + (long long)(m0 + ar + 32 * i) * K + k0 + ac);
#pragma unroll
        for (int p = 0; p < 3; ++p) pb[p] = *(const u16x8*)(Bs + p * NK + (long long)(n0 + br) * K + k0 + bc);
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            u16x4 h[3];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned short a, b, c;
                split3(pa[i][e], a, b, c);
                h[0][e] = a; h[1][e] = b; h[2][e] = c;
            }
#pragma unroll
            for (int p = 0; p < 3; ++p) *(u16x4*)&sA[p][(ar + 32 * i) * S + ac] = h[p];
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) *(u16x8*)&sB[p][br * S + bc] = pb[p];
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int rowA = (32 * wm + li) * S + 8 * lq, rowB = (32 * wn + li) * S + 8 * lq;
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += KC) {
        __syncthreads();
        stash();
        __syncthreads();
        if (k0 + KC < K) fetch(k0 + KC);
#pragma unroll
        for (int ks = 0; ks < KC; ks += 16) {
            bf16x8 a[3], b[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                a[p] = __builtin_bit_cast(bf16x8, *(const u16x8*)&sA[p][rowA + ks]);
                b[p] = __builtin_bit_cast(bf16x8, *(const u16x8*)&sB[p][rowB + ks]);
            }
            // smallest terms first
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
        }
    }
    const int n = n0 + 32 * wn + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * lq;
        C[(long long)m * N + n] = acc[r];
    }
}

extern "C" int split_rows(const float* w, unsigned short* out, long long n, void* stream) {
    hipLaunchKernelGGL(k_split_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, out, n);
    return (int)hipGetLastError();
}

extern "C" int split_gemm(const float* A, const unsigned short* Bs, float* C, int M, int N, int K, void* stream) {
    if (M % 64 || N % 64 || K % 32) return -1;
    hipLaunchKernelGGL(k_split_gemm, dim3((unsigned)((M / 64) * (N / 64))), dim3(256), 0, (hipStream_t)stream, A, Bs, C, M, N, K);
    return (int)hipGetLastError();
}
