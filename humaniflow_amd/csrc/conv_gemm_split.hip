// OPT-IN variant of the encoder's 1x1 GEMM (csrc/conv_gemm.hip) on the bf16 matrix pipe with SPLIT fp32 operands -- not the default path.
//
//   Y[m][n] = relu?( sum_k X[m][k] W[n][k]  (+ sum_k X2[m'][k] W[n][K1 + k])  + bias[n] (+ R[m][n]) )          same contract as hmf_conv1x1_gemm
//
// Every fp32 operand value is written as a sum of bf16 pieces, x = p0 + p1 (+ p2), p0 = rne_bf16(x), p1 = rne_bf16(x - p0), p2 = rne_bf16(x - p0 - p1),
// and a product x * w becomes the piece products whose magnitude is above the target:
//   terms = 3:  p0 q0 + p0 q1 + p1 q0                                   (relative error of a product ~2^-16: the "3 x bf16" split)
//   terms = 6:  p0 q0 + p0 q1 + p1 q0 + p0 q2 + p2 q0 + p1 q1           (dropped products <= 2^-24 of x w: fp32-level products, fp32 accumulate)
// each one v_mfma_f32_32x32x16_bf16 (32 cycles for 16 k) against v_mfma_f32_32x32x2_f32 (64 cycles for 2 k): 3 (6) bf16 MFMAs replace 8 fp32 MFMAs
// at 1/2 the issue time each -- 5.3x (2.7x) less matrix-pipe time.  Accumulation stays fp32 in the MFMA accumulators, bias / residual / ReLU in fp32.
//
//   * the filter is split ONCE at plan time (hmf_conv1x1_split_pack): [N/32 n-tile][K/32 chunk][piece][k-step][64 lanes][8 bf16], i.e. every
//     1 KiB block is one MFMA B operand in lane order -- it goes global -> LDS by one LDS-DMA instruction and is read back lane-linear (no conflicts);
//   * the activations stay fp32 in HBM (no second copy of any tensor): the fp32 chunk is DMA'd into the same swizzled LDS image as the fp32 kernel,
//     each wave reads its fragment (two ds_read_b128 per k-step) and splits it in registers -- v_cvt_pk_bf16_f32, shift / and, v_pk_add_f32:
//     2.5 (4.5) VALU instructions per value.  VALU and MFMA time of a SIMD add on gfx950 (profiles/r03_micro_mfma_valu_overlap.txt), so the
//     wave tile is 32 x 64 where the shape allows: one split feeds two n-tiles;
//   * LDS-DMA ring, counted vmcnt waits, raw s_barrier and the LDS-transposing 16-byte epilogue as in k_conv_gemm.
#include "hmf_common.h"
#include <type_traits>

namespace hmf {

struct GemmSplitArgs {
    const float* x;               // [M][K1]
    const float* x2;              // [B][H2][W2][K2] or null
    const unsigned short* wp;     // packed split filter
    const float* bias;            // [N] or null
    const float* res;             // [M][N] or null
    float* y;                     // [M][N]
    int M, N, K1, K2, OH, OW, H2, W2, stride2, relu;
};

typedef __attribute__((address_space(3))) void* lds_ptr_s;
typedef __bf16 bf16x8s __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2s __attribute__((ext_vector_type(2)));
typedef float f32x2s __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));

#define HMF_SVMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

static __device__ __forceinline__ void sdma16(__amdgpu_buffer_rsrc_t rsrc, lds_ptr_s dst, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voff, soff, 0, 0);
}
static __device__ __forceinline__ unsigned pack_bf16(f32x2s v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2s)); }
static __device__ __forceinline__ f32x2s unpack_bf16(unsigned p) {
    return f32x2s{__builtin_bit_cast(float, p << 16), __builtin_bit_cast(float, p & 0xffff0000u)};
}
// 8 fp32 values -> P bf16 pieces of 8
template <int P>
static __device__ __forceinline__ void split8(const f32x4 lo4, const f32x4 hi4, u32x4s (&piece)[P]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f32x2s r = i < 2 ? f32x2s{lo4[2 * i], lo4[2 * i + 1]} : f32x2s{hi4[2 * i - 4], hi4[2 * i - 3]};
#pragma unroll
        for (int p = 0; p < P; ++p) {
            piece[p][i] = pack_bf16(r);
            if (p + 1 < P) r = r - unpack_bf16(piece[p][i]);
        }
    }
}
template <int N>
static __device__ __forceinline__ void wait_vm() {          // vmcnt is a 6-bit immediate
    static_assert(N >= 0 && N < 64, "vmcnt range");
    HMF_SVMCNT(N);
}

// 2 x 2 waves, wave tile (32 TM) x (BN / 2), K chunks of 32, S-stage ring.  P = pieces per operand (2: terms = 3, 3: terms = 6).
// LDS reads per MFMA fall with the wave tile (4 TM + 2 P TN reads for 2 NPROD TM TN MFMAs per chunk): at one read per MFMA the LDS pipe of the CU
// is as busy as its four matrix pipes (a 1 KiB ds_read_b128 occupies it for 8 cycles, a bf16 MFMA one SIMD for 32).
template <int TM, int BN, int S, int P, bool DUAL, int OCC, int ABL = 0>          // ABL: timing ablations of tools/gemm_split_probe.py (wrong results)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC))) void k_conv_gemm_split(const GemmSplitArgs a) {
    constexpr int BM = 64 * TM, WM = 32 * TM, TN = BN / 64, WN = BN / 2;
    constexpr int NA = BM / 32;                               // A: BM rows x 128 B = BM / 8 requests per chunk, over 4 waves
    constexpr int NBLK = (BN / 32) * P * 2;                   // B: 1 KiB blocks per chunk (n-tile, piece, k-step)
    constexpr int NB = NBLK / 4;
    constexpr int L = NA + NB;
    constexpr int STAGE = BM * 32 + NBLK * 256;               // floats per stage
    static_assert(NBLK % 4 == 0, "blocks per wave");
    static_assert(S * STAGE >= BM * BN, "the epilogue transposes the accumulator tile through the ring's LDS");
    static_assert((S - 1) * L < 64, "vmcnt range");
    __shared__ __attribute__((aligned(16))) float smem[S * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lq = lane >> 5;

    const int NT = a.N / BN, MT = (a.M + BM - 1) / BM, T = MT * NT;
    int t = blockIdx.x;
    if ((T & 7) == 0) t = (blockIdx.x & 7) * (T >> 3) + (blockIdx.x >> 3);
    const int m0 = (t / NT) * BM, n0 = (t % NT) * BN;
    const int K = a.K1 + a.K2;
    const int nch = K / 32, nch1 = a.K1 / 32;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.M * a.K1 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, a.N * K * 2 * P, 0x00020000);
    __amdgpu_buffer_rsrc_t rx2 = rx;
    int voA[NA], voA2[NA], voB[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int row = (wave * NA + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        const int m = min(m0 + row, a.M - 1);
        voA[i] = (m * a.K1 + 4 * c) * 4;
        voA2[i] = 0;
        if (DUAL) {
            const int b = m / (a.OH * a.OW), r = m - b * a.OH * a.OW, oh = r / a.OW, ow = r - oh * a.OW;
            voA2[i] = (((b * a.H2 + oh * a.stride2) * a.W2 + ow * a.stride2) * a.K2 + 4 * c) * 4;
        }
    }
    if (DUAL) rx2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.x2, 0, (a.M / (a.OH * a.OW)) * a.H2 * a.W2 * a.K2 * 4, 0x00020000);
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int bl = wave * NB + i;                          // block of the stage: (n-tile j, piece p, k-step t) = (bl / (2 P), ...)
        const int j = bl / (2 * P), pt = bl - j * 2 * P;
        voB[i] = (((n0 / 32 + j) * nch) * 2 * P + pt) * 1024 + lane * 16;
    }
    auto issue = [&](int c) {
        float* st = smem + (c % S) * STAGE;
        const bool second = DUAL && c >= nch1;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            lds_ptr_s dst = (lds_ptr_s)(st + (wave * NA + i) * 256);
            if (second) sdma16(rx2, dst, voA2[i], (c - nch1) * 128);
            else sdma16(rx, dst, voA[i], c * 128);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) sdma16(rw, (lds_ptr_s)(st + BM * 32 + (wave * NB + i) * 256), voB[i], c * 2 * P * 1024);
    };

    const int sw = (li >> 1) & 7;
    const float* pa = smem + (wm * WM + li) * 32;
    const float* pb = smem + BM * 32 + (wn * TN * 2 * P) * 256 + lane * 4;
    int go[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) go[g] = ((2 * g + lq) ^ sw) * 4;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
    for (int c = 0; c < S - 1; ++c)
        if (c < nch) issue(c);

    // piece products, smallest first
    constexpr int NPROD = P == 2 ? 3 : 6;
    constexpr int PB3[3] = {0, 1, 0};
    constexpr int QA6[6] = {1, 2, 0, 1, 0, 0}, QB6[6] = {1, 0, 2, 0, 1, 0};

    auto compute = [&](int stage) {
        const float* sa = pa + stage * STAGE;
        const float* sb = pb + stage * STAGE;
        f32x4 ra[TM][4];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) ra[i][g] = *(const f32x4*)(sa + i * 32 * 32 + go[g]);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4s bq[TN][P];
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int p = 0; p < P; ++p) bq[j][p] = *(const u32x4s*)(sb + ((j * P + p) * 2 + ks) * 256);
            u32x4s ap[TM][P];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (ABL & 1) {
#pragma unroll
                    for (int p = 0; p < P; ++p) ap[i][p] = __builtin_bit_cast(u32x4s, ra[i][2 * ks + (p & 1)]);
                } else {
                    split8<P>(ra[i][2 * ks], ra[i][2 * ks + 1], ap[i]);
                }
            }
#pragma unroll
            for (int q = 0; q < NPROD; ++q) {
                const int ia = P == 2 ? (q == 0 ? 1 : 0) : QA6[q];
                const int ib = P == 2 ? PB3[q] : QB6[q];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        if ((ABL & 2) && q) acc[i][j][q] += __builtin_bit_cast(float, ap[i][ia][0] ^ bq[j][ib][0]);
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8s, ap[i][ia]), __builtin_bit_cast(bf16x8s, bq[j][ib]),
                                                                            acc[i][j], 0, 0, 0);
            }
        }
    };

    // steady loop: chunk c + S - 1 is requested in iteration c; the tail requests nothing and waits for everything
    int c = 0;
    for (; c + S - 1 < nch; ++c) {
        if (ABL & 4) wait_vm<0>();
        else wait_vm<(S - 2) * L>();                          // all but the newest S - 2 chunks have landed: chunk c is there (own requests)
        __builtin_amdgcn_s_barrier();                         // ... everybody's; and every wave is past chunk c - 1, whose stage is requested next
        if (!(ABL & 4)) issue(c + S - 1);
        compute(c % S);
    }
    for (; c < nch; ++c) {
        const int r = nch - 1 - c;                            // chunks requested behind chunk c (block-uniform)
        if (S >= 5 && r >= 3) wait_vm<(S >= 5 ? 3 : 0) * L>();
        else if (S >= 4 && r == 2) wait_vm<(S >= 4 ? 2 : 0) * L>();
        else if (S >= 3 && r == 1) wait_vm<(S >= 3 ? 1 : 0) * L>();
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        compute(c % S);
    }

    // ---- epilogue (as k_conv_gemm): wave-local transpose through LDS, 16-byte rows, + bias (+ residual), ReLU ----
    __builtin_amdgcn_s_barrier();
    float* tw = smem + wave * (WM * WN);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) tw[(i * 32 + 4 * lq + (r & 3) + 8 * (r >> 2)) * WN + j * 32 + li] = acc[i][j][r];
    constexpr int LPR = WN / 4, RPI = 64 / LPR, NS = WM / RPI;
    const int col = (lane % LPR) * 4, row0 = lane / LPR;
    const int nb = n0 + wn * WN + col, mb = m0 + wm * WM + row0;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) bv = *(const f32x4*)(a.bias + nb);
    constexpr int kOOB = (int)0x80000000;
    const int ybytes = a.M * a.N * 4;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.res ? a.res : a.y), 0, ybytes, 0x00020000);
    const int vo = (mb * a.N + nb) * 4;
    const int rstep = RPI * a.N * 4;
    const float lo = a.relu ? 0.f : -INFINITY;
    constexpr int HALF = NS > 8 ? NS / 2 : NS;
#pragma unroll
    for (int h = 0; h < NS; h += HALF) {
        f32x4 rv[HALF];
        if (a.res) {
#pragma unroll
            for (int s = 0; s < HALF; ++s)
                rv[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, (mb + (h + s) * RPI < a.M) ? vo + (h + s) * rstep : kOOB, 0, 0));
        } else {
#pragma unroll
            for (int s = 0; s < HALF; ++s) rv[s] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int s = 0; s < HALF; ++s) {
            f32x4 v = *(const f32x4*)(tw + (row0 + (h + s) * RPI) * WN + col);
            v = v + bv + rv[s];
            v[0] = fmaxf(v[0], lo); v[1] = fmaxf(v[1], lo); v[2] = fmaxf(v[2], lo); v[3] = fmaxf(v[3], lo);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, v), ry, (mb + (h + s) * RPI < a.M) ? vo + (h + s) * rstep : kOOB, 0, 0);
        }
    }
}

// filter split: one thread per (n-tile, chunk, k-step, lane): 8 values of row n = 32 nt + (lane & 31), k = 32 c + 16 ks + 8 (e >> 2) + 4 (lane >> 5) + (e & 3)
__global__ __launch_bounds__(256) void k_gemm_split_pack(const float* __restrict__ w, u32x4s* __restrict__ out, int N, int K, int P, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int lane = (int)(i & 63), ks = (int)((i >> 6) & 1);
    const long long bc = i >> 7;                               // nt * nch + c
    const int nch = K / 32, c = (int)(bc % nch), nt = (int)(bc / nch);
    const int n = nt * 32 + (lane & 31), q = lane >> 5;
    const float* src = w + (long long)n * K + c * 32 + 16 * ks + 4 * q;
    const f32x4 v0 = *(const f32x4*)src, v1 = *(const f32x4*)(src + 8);
    u32x4s* dst = out + ((bc * P) * 2 + ks) * 64 + lane;      // + piece * 128
    if (P == 2) {
        u32x4s pc[2];
        split8<2>(v0, v1, pc);
        dst[0] = pc[0]; dst[128] = pc[1];
    } else {
        u32x4s pc[3];
        split8<3>(v0, v1, pc);
        dst[0] = pc[0]; dst[128] = pc[1]; dst[256] = pc[2];
    }
}

template <int TM, int BN, int S, int P, int OCC, int ABL = 0>
static void launch_split(const GemmSplitArgs& a, hipStream_t st) {
    const int T = ((a.M + 64 * TM - 1) / (64 * TM)) * (a.N / BN);
    ProfScope ps(8, a.M, st);
    if (a.x2) hipLaunchKernelGGL((k_conv_gemm_split<TM, BN, S, P, true, OCC, ABL>), dim3(T), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_conv_gemm_split<TM, BN, S, P, false, OCC, ABL>), dim3(T), dim3(256), 0, st, a);
}

}  // namespace hmf

using namespace hmf;

extern "C" long long hmf_conv1x1_split_packed_bytes(int N, int K, int terms) {
    if (N <= 0 || K <= 0 || N % 32 || K % 32 || (terms != 3 && terms != 6)) return -1;
    return (long long)N * K * 2 * (terms == 3 ? 2 : 3);
}

// w_dev: the folded filter (N, K) row-major fp32 on the device (the operand of hmf_conv1x1_gemm); packed_dev: hmf_conv1x1_split_packed_bytes bytes.
extern "C" int hmf_conv1x1_split_pack(const float* w_dev, void* packed_dev, int N, int K, int terms, void* stream) {
    if (!w_dev || !packed_dev || hmf_conv1x1_split_packed_bytes(N, K, terms) < 0)
        return fail(HMF_ERR_INVALID, "hmf_conv1x1_split_pack: N=%d and K=%d must be positive multiples of 32, terms=%d one of 3, 6", N, K, terms);
    if (((uintptr_t)w_dev & 15) || ((uintptr_t)packed_dev & 15)) return fail(HMF_ERR_INVALID, "hmf_conv1x1_split_pack: pointers must be 16-byte aligned");
    const long long total = (long long)(N / 32) * (K / 32) * 128;
    hipLaunchKernelGGL(k_gemm_split_pack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_dev, (u32x4s*)packed_dev, N, K,
                       terms == 3 ? 2 : 3, total);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

// The split-operand form of hmf_conv1x1_gemm: same arguments, w_packed from hmf_conv1x1_split_pack with the same `terms`.
// tile: 0 = chosen from the shape, else ring stages * 1000000 + BM * 1000 + BN of the instantiated set.
extern "C" int hmf_conv1x1_gemm_split(const float* x, const float* x2, const void* w_packed, const float* bias, const float* residual, float* y, int B, int OH,
                                      int OW, int K1, int H2, int W2, int K2, int stride2, int N, int relu, int terms, int tile, void* stream) {
    if (!x || !w_packed || !y || B <= 0 || OH <= 0 || OW <= 0) return fail(HMF_ERR_INVALID, "hmf_conv1x1_gemm_split: bad args");
    if (terms != 3 && terms != 6) return fail(HMF_ERR_INVALID, "hmf_conv1x1_gemm_split: terms=%d (3 = bf16 hi/lo products, 6 = three-piece products)", terms);
    if (!x2) K2 = 0;
    if (K1 <= 0 || K1 % 32 || K2 % 32 || N % 64) return fail(HMF_ERR_UNSUPPORTED, "hmf_conv1x1_gemm_split: K1=%d, K2=%d must be multiples of 32 and N=%d of 64", K1, K2, N);
    if (x2 && (stride2 < 1 || (H2 - 1) / stride2 + 1 != OH || (W2 - 1) / stride2 + 1 != OW))
        return fail(HMF_ERR_INVALID, "hmf_conv1x1_gemm_split: x2 %dx%d / %d does not give %dx%d", H2, W2, stride2, OH, OW);
    if (((uintptr_t)x & 15) || ((uintptr_t)x2 & 15) || ((uintptr_t)w_packed & 15) || ((uintptr_t)y & 15) || ((uintptr_t)residual & 15) || ((uintptr_t)bias & 15))
        return fail(HMF_ERR_INVALID, "hmf_conv1x1_gemm_split: pointers must be 16-byte aligned");
    const long long M = (long long)B * OH * OW;
    if (M * N >= (1LL << 29) || M * K1 >= (1LL << 29) || (x2 && (long long)B * H2 * W2 * K2 >= (1LL << 29)) || (long long)N * (K1 + K2) >= (1LL << 28))
        return fail(HMF_ERR_UNSUPPORTED, "hmf_conv1x1_gemm_split: tensor exceeds the 32-bit byte offsets of the buffer loads");
    GemmSplitArgs a;
    a.x = x; a.x2 = x2; a.wp = (const unsigned short*)w_packed; a.bias = bias; a.res = residual; a.y = y;
    a.M = (int)M; a.N = N; a.K1 = K1; a.K2 = K2; a.OH = OH; a.OW = OW; a.H2 = H2; a.W2 = W2; a.stride2 = stride2; a.relu = relu;
    if (tile == 0) {
        // tools/gemm_split_probe.py, MI355X, B = 32: the kernel is bound by the LDS pipe (about 1 KiB of LDS traffic per bf16 MFMA: as busy as the four
        // matrix pipes together), so the largest wave tile that still leaves >= 2 work-groups per CU wins: 128 x 128 (wave 64 x 64) where that gives
        // >= 512 work-groups, 64 x 128 for >= 256, else 64 x 64 with a 4-stage ring (the deep, few-tile layer4 conv1).
        const long long m64 = (M + 63) / 64, m128 = (M + 127) / 128;
        if (N % 128 == 0 && m128 * (N / 128) >= 512) tile = 2128128;
        else if (N % 128 == 0 && m64 * (N / 128) >= 256) tile = 2064128;
        else if (m64 * (N / 64) >= 768) tile = 3064064;
        else tile = 4064064;
    }
    if ((tile % 1000) == 128 && N % 128) return fail(HMF_ERR_INVALID, "hmf_conv1x1_gemm_split: a 128-channel tile needs N %% 128 == 0");
    hipStream_t st = (hipStream_t)stream;
    const int p = terms == 3 ? 2 : 3;
#define HMF_SPLIT_CASE(code, TM, BN, S, OCC2, OCC3)                   \
    case code:                                                        \
        if (p == 2) launch_split<TM, BN, S, 2, OCC2>(a, st);          \
        else launch_split<TM, BN, S, 3, OCC3>(a, st);                 \
        break;
    switch (tile) {                                             // ring stages * 1000000 + BM * 1000 + BN; LDS bytes for terms = 3 / 6
        HMF_SPLIT_CASE(2128128, 2, 128, 2, 2, 1)                // 64 / 80 KB
        HMF_SPLIT_CASE(2064128, 1, 128, 2, 3, 2)                // 48 / 64 KB
        HMF_SPLIT_CASE(3064064, 1, 64, 3, 3, 2)                 // 48 / 60 KB
        HMF_SPLIT_CASE(4064064, 1, 64, 4, 2, 2)                 // 64 / 80 KB
#ifdef HMF_EXPERIMENTS                                          // tools/gemm_split_probe.py: tilings measured and not chosen; timing ablations
        HMF_SPLIT_CASE(3064128, 1, 128, 3, 2, 1)
        HMF_SPLIT_CASE(4064128, 1, 128, 4, 1, 1)
        HMF_SPLIT_CASE(3128128, 2, 128, 3, 1, 1)
        HMF_SPLIT_CASE(2128064, 2, 64, 2, 3, 2)
        HMF_SPLIT_CASE(3128064, 2, 64, 3, 2, 1)
        case 9000001: launch_split<1, 128, 3, 2, 2, 1>(a, st); break;      // ablations of 3064128, terms = 3: no operand split
        case 9000002: launch_split<1, 128, 3, 2, 2, 2>(a, st); break;      // one MFMA of three
        case 9000003: launch_split<1, 128, 3, 2, 2, 3>(a, st); break;      // neither
        case 9000004: launch_split<1, 128, 3, 2, 2, 4>(a, st); break;      // no DMA after the prologue
        case 9000007: launch_split<1, 128, 3, 2, 2, 7>(a, st); break;      // LDS reads + barriers only
#endif
        default: return fail(HMF_ERR_INVALID, "hmf_conv1x1_gemm_split: unknown tile %d", tile);
    }
#undef HMF_SPLIT_CASE
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}
