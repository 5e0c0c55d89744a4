// 1x1 convolutions of the ResNet encoder (row a1) as an LDS-DMA pipelined fp32-MFMA GEMM:
//
//   Y[m][n] = relu?( sum_k X[m][k] W[n][k]  (+ sum_k X2[m'][k] W[n][K1 + k])  + bias[n] (+ R[m][n]) )        m = pixel (b, oh, ow), n = output channel
//
// X (M, K1) is the NHWC activation tensor itself, W (N, K1 + K2) the folded filter in channels_last order, X2 the block input of a bottleneck whose
// downsample branch rides in the same GEMM (models/resnet.py:111-127: out = relu(bn3(conv3(h)) + bn_d(conv_d(x))); pixel m' = (b, oh * s, ow * s)).
//
// Why a second 1x1 kernel next to k_conv_igemm: the ResNet 1x1 layers are 4.3 - 12.9 GFLOP launches (27 - 82 us at the fp32 MFMA peak) and
// k_conv_igemm ran them at 0.55 - 0.75 of peak: one 32 x 32 accumulator per wave (a dependent MFMA chain that any interleaved instruction
// stalls), operands staged global -> VGPR -> LDS by the same waves that issue the MFMAs, 64 scalar stores per lane in the epilogue.  Here:
//   * wave tile (BM/2) x (BN/2) = up to 2 x 2 MFMA tiles: 4 independent accumulator chains, half the LDS fragment reads per MFMA;
//   * operands go global -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`, 1 KiB per wave instruction): no staging registers, no ds_write, the
//     MFMA waves only issue one scalar-offset instruction per KiB.  The DMA writes LDS lane-linear, so the bank-conflict-free image is produced
//     by permuting the SOURCE address: 16-byte slot c of row r (a 128-byte row = 32 k) is stored at slot c ^ ((r >> 1) & 7); every row is still
//     one full 128-byte line of global memory;
//   * S-stage LDS ring (3-4 stages of (BM + BN) x 128 B), loads issued S - 1 chunks ahead with counted `s_waitcnt vmcnt(N)` and a raw s_barrier,
//     and the chunk made visible ONE iteration before it is consumed, so the first fragments of chunk c + 1 are read under the last MFMAs of
//     chunk c: the MFMA pipe does not drain at the barrier;
//   * epilogue through LDS: every wave transposes its accumulator tile in its own LDS region and stores 16 bytes per lane (residual read the same
//     way): 4x fewer memory instructions than channel-per-lane scalar stores.
// Numerics: the same k-ordered fp32 fma chain per output as k_conv_igemm (k ascending in groups of 8; within a group the order 4q + u is a fixed
// permutation), so results agree with it to rounding of the accumulation order only.
#include "hmf_common.h"
#include <type_traits>

namespace hmf {

struct GemmArgs {
    const float* x;      // [M][K1]
    const float* x2;     // [B][H2][W2][K2] or null
    const float* w;      // [N][K1 + K2]
    const float* bias;   // [N] or null
    const float* res;    // [M][N] or null
    float* y;            // [M][N] (POOL: or null)
    float* pool;         // POOL: [M / 64][N] mean over each run of 64 rows (= the 8 x 8 pixels of one image)
    int M, N, K1, K2, OH, OW, H2, W2, stride2, relu;
    int xb;              // XCD split along n (hmf_common.h xcd_n_split)
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;

#define HMF_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

#ifdef HMF_EXPERIMENTS           // tools/gemm1x1_probe.py --timeline: shader-clock stamps of every work-group (start, first chunk visible, main loop done, end)
__device__ long long g_gemm_stamps[6 * 4096];          // per work-group: 4 shader-clock stamps + the constant 100 MHz counter at start and end
#define HMF_STAMP(k)                                                                                                   \
    do {                                                                                                               \
        if (tid == 0 && blockIdx.x < 4096) {                                                                           \
            g_gemm_stamps[blockIdx.x * 6 + (k)] = (long long)__builtin_readcyclecounter();                             \
            if ((k) == 0) g_gemm_stamps[blockIdx.x * 6 + 4] = (long long)wall_clock64();                               \
            if ((k) == 3) g_gemm_stamps[blockIdx.x * 6 + 5] = (long long)wall_clock64();                               \
        }                                                                                                              \
    } while (0)
#else
#define HMF_STAMP(k) \
    do {             \
    } while (0)
#endif

// One LDS-DMA request: 64 lanes x 16 bytes, lane l's bytes from (rsrc base + voff + soff) land at dst + 16 l.  (A non-template helper: called
// directly inside a kernel TEMPLATE the builtin silently suppresses the host-side launch stub of every instantiation -- hipcc, ROCm 7.2.)
static __device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, lds_ptr_t dst, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voff, soff, 0, 0);
}

// KS = 2 (wave tiles of ONE MFMA tile only): even / odd MFMA steps accumulate into two register sets that are added at the end, so the wave's
// MFMAs form two independent chains instead of one dependent chain.
// POOL: the global average pool of models/resnet.py:214-215 rides in the epilogue of the encoder's last convolution: a 64-row tile is exactly one
// image's 8 x 8 map, so the column means of the finished tile ARE that image's pooled features for the tile's channels -- no cross-work-group
// reduction, a fixed summation order, and the 16.8 MB map is neither written nor re-read when the caller passes y = NULL.
// ABL (experiments build, tools/whatif_1x1.py; timing only, WRONG results): & 1 = the epilogue stores accumulator registers straight from the MFMA layout
// (no transposition through LDS: the ceiling of any cheaper epilogue), & 2 = the activation panel is requested for the prologue chunks only (the
// ceiling of a panel-resident form that fetches A once per work-group instead of once per n-tile).
template <int BM, int BN, int S, bool DUAL, int OCC, int KS, bool POOL = false, int ABL = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC))) void k_conv_gemm(const GemmArgs a) {
    constexpr int TM = BM / 64, TN = BN / 64;            // 32 x 32 MFMA tiles per wave (2 x 2 waves)
    constexpr int NA = BM / 32, NB = BN / 32;            // DMA instructions per wave per chunk (8 rows of 128 B each)
    constexpr int L = NA + NB;
    constexpr int STAGE = (BM + BN) * 32;                // floats per stage
    constexpr int WM = BM / 2, WN = BN / 2;
    static_assert(S * STAGE >= BM * BN, "the epilogue transposes the accumulator tile through the ring's LDS");
    __shared__ __attribute__((aligned(16))) float smem[S * STAGE];        // ONE LDS object (a second one makes hipcc drain vmcnt before every ds_read)

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lq = lane >> 5;
    HMF_STAMP(0);

    // ---- XCD-aware tile assignment: the ids of one XCD cover a contiguous run of tiles, n fastest (the N / BN tiles sharing an activation panel share an L2)
    const int NT = a.N / BN, MT = (a.M + BM - 1) / BM, T = MT * NT;
    const int t = xcd_tile(blockIdx.x, MT, NT, a.xb);
    (void)T;
    const int m0 = (t / NT) * BM, n0 = (t % NT) * BN;
    const int K = a.K1 + a.K2;

    // ---- DMA source offsets (bytes, per lane, chunk-invariant): row R0 + (lane >> 3) of the panel, 16-byte slot (lane & 7) ^ swizzle(row) ----
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.M * a.K1 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.N * K * 4, 0x00020000);
    __amdgpu_buffer_rsrc_t rx2 = rx;
    int voA[NA], voA2[NA], voB[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int row = (wave * NA + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        const int m = min(m0 + row, a.M - 1);                    // rows beyond M read a valid row; their outputs are never stored
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
        const int row = (wave * NB + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        voB[i] = ((n0 + row) * K + 4 * c) * 4;
    }
    const int nch = K / 32, nch1 = a.K1 / 32;
    // requests [lo, hi) of the L = NA + NB that bring chunk c into stage c % S (A rows first, then B rows)
    auto issue_range = [&](int c, int lo, int hi) {
        float* st = smem + (c % S) * STAGE;
        const bool second = DUAL && c >= nch1;                  // block-uniform
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (i < lo || i >= hi) continue;
            if ((ABL & 2) && c >= S - 1) continue;
            lds_ptr_t dst = (lds_ptr_t)(st + (wave * NA + i) * 256);
            if (second) dma16(rx2, dst, voA2[i], (c - nch1) * 128);
            else dma16(rx, dst, voA[i], c * 128);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (NA + i < lo || NA + i >= hi) continue;
            dma16(rw, (lds_ptr_t)(st + BM * 32 + (wave * NB + i) * 256), voB[i], c * 128);
        }
    };
    auto issue = [&](int c) { issue_range(c, 0, L); };

    // ---- fragment addresses: lane (li, lq) of MFMA step u of k-group g takes k = 8 g + 4 lq + u of row li: one ds_read_b128 per (tile, group) ----
    const int sw = (li >> 1) & 7;
    const float* pa = smem + (wm * WM + li) * 32;
    const float* pb = smem + BM * 32 + (wn * WN + li) * 32;
    int go[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) go[g] = ((2 * g + lq) ^ sw) * 4;
    f32x4 fa[2][TM], fb[2][TN];
    auto frag = [&](int stage, int g, int slot) {
        const float* sa = pa + stage * STAGE + go[g];
        const float* sb = pb + stage * STAGE + go[g];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[slot][i] = *(const f32x4*)(sa + i * 32 * 32);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[slot][j] = *(const f32x4*)(sb + j * 32 * 32);
    };

    static_assert(KS == 1 || (TM == 1 && TN == 1), "split accumulators are for single-tile waves");
    f32x16 acc[TM][TN], acc2;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;

    // ---- prologue: chunks 0 .. S-2 in flight ----
#pragma unroll
    for (int c = 0; c < S - 1; ++c)
        if (c < nch) issue(c);

    // MORE (compile time): chunk c + S - 1 exists and is requested during iteration c (the steady loop); the last S - 1 iterations request nothing
    auto iteration = [&](int c, auto MORE) {
        constexpr bool more = decltype(MORE)::value;
        // Steady state (c >= 1): chunk c + 1 (and with it chunk c) has landed for this wave's own requests; after the barrier for everybody's, and
        // every wave is past its reads of chunk c - 1, whose stage the next request overwrites.  The first iteration starts as soon as chunk 0
        // alone is there (the launch's first requests all hit HBM at once: waiting for a second chunk costs its transfer time on every CU).
        if (c == 0) {
            if (nch >= S - 1) HMF_VMCNT((S - 2) * L);
            else HMF_VMCNT(0);
        } else if (S > 2 && c + S - 2 <= nch - 1) {
            HMF_VMCNT((S > 2 ? S - 3 : 0) * L);
        } else {
            HMF_VMCNT(0);
        }
        __builtin_amdgcn_s_barrier();
        if (c == 0) HMF_STAMP(1);
        const int stage = c % S;
        if (S == 2 || c <= 1) frag(stage, 0, 0);                // S = 2: the next chunk is still in flight during this one, no early fragments
        auto group = [&](auto G) {
            constexpr int g = decltype(G)::value;
            if (g < 3) frag(stage, g + 1, (g + 1) & 1);
            else if (S > 2 && c >= 1 && c + 1 < nch) frag((c + 1) % S, 0, 0);       // first fragments of the next chunk under this chunk's last MFMAs
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (KS == 2 && (u & 1)) {
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][0][u], fb[g & 1][0][u], acc2, 0, 0, 0);
                } else {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][i][u], fb[g & 1][j][u], acc[i][j], 0, 0, 0);
                }
                // The chunk's L DMA requests are spread over the four k-groups, each issued right behind an MFMA: issuing one costs the wave
                // ~60-90 cycles (measured: a wave alone on its SIMD lost L x ~90 cycles per chunk with all requests at the top of the
                // iteration), which is the time the MFMA just issued keeps the matrix pipe busy anyway.
                if (u == 0 && more) issue_range(c + S - 1, (g * L) / 4, ((g + 1) * L) / 4);
            }
            // pin the emitted order of this k-group: fragment reads of the next group, first MFMA step, the DMA request(s), the other steps
            __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, (KS == 2) ? 1 : TM * TN, 0);
            __builtin_amdgcn_sched_group_barrier(0x010, ((g + 1) * L) / 4 - (g * L) / 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * ((KS == 2) ? 1 : TM * TN) - ((KS == 2) ? 1 : TM * TN), 0);
        };
        group(std::integral_constant<int, 0>{});
        group(std::integral_constant<int, 1>{});
        group(std::integral_constant<int, 2>{});
        group(std::integral_constant<int, 3>{});
    };
    int c = 0;
    for (; c + S - 1 < nch; ++c) iteration(c, std::true_type{});
    for (; c < nch; ++c) iteration(c, std::false_type{});
    if (KS == 2) acc[0][0] = acc[0][0] + acc2;
    HMF_STAMP(2);

    // ---- epilogue: wave-local transpose through LDS, then 16-byte row accesses: + bias (+ residual), ReLU ----
    __builtin_amdgcn_s_barrier();                               // every wave is done reading operand stages
    float* tw = smem + wave * (WM * WN);
    if (!(ABL & 1))
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) tw[(i * 32 + 4 * lq + (r & 3) + 8 * (r >> 2)) * WN + j * 32 + li] = acc[i][j][r];
    constexpr int LPR = WN / 4;                                 // lanes per row of the wave tile
    constexpr int RPI = 64 / LPR;                               // rows per instruction
    constexpr int NS = WM / RPI;
    const int col = (lane % LPR) * 4, row0 = lane / LPR;
    const int nb = n0 + wn * WN + col, mb = m0 + wm * WM + row0;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) bv = *(const f32x4*)(a.bias + nb);
    // Row accesses are raw buffer accesses: a row beyond M gets an out-of-range offset (loads return 0, stores are dropped) -- no per-row branch,
    // so all residual loads of a pass are in flight together (a branch per row makes hipcc wait for every load separately).
    constexpr int kOOB = (int)0x80000000;
    const int ybytes = a.M * a.N * 4;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.res ? a.res : a.y), 0, ybytes, 0x00020000);
    const int vo = (mb * a.N + nb) * 4;                         // + s * RPI * N * 4
    const int rstep = RPI * a.N * 4;
    constexpr int HALF = NS > 8 ? NS / 2 : NS;                  // residual rows in flight at once (register budget)
    const float lo = a.relu ? 0.f : -INFINITY;
    f32x4 psum = {0.f, 0.f, 0.f, 0.f};
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
            f32x4 v;
            if (ABL & 1) {
                const int q = ((h + s) * 4) % (TM * TN * 16);          // any four accumulator registers (fully unrolled: constant indices): the stores keep their count and addresses
                const f32x16& A_ = acc[(q / 16) / TN][(q / 16) % TN];
                v = f32x4{A_[q % 16], A_[q % 16 + 1], A_[q % 16 + 2], A_[q % 16 + 3]};
            } else {
                v = *(const f32x4*)(tw + (row0 + (h + s) * RPI) * WN + col);
            }
            v = v + bv + rv[s];
            v[0] = fmaxf(v[0], lo); v[1] = fmaxf(v[1], lo); v[2] = fmaxf(v[2], lo); v[3] = fmaxf(v[3], lo);
            if (POOL) psum += v;
            if (!POOL || a.y)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int, v), ry,
                                                       (mb + (h + s) * RPI < a.M) ? vo + (h + s) * rstep : kOOB, 0, 0);
        }
    }
    if (POOL) {
        static_assert(!POOL || (BM == 64 && S * STAGE >= BM * BN + 2 * BN), "pool partials live behind the transposed tile");
        // lanes l, l + LPR, ... hold the same 4 columns of different rows: butterfly over the row lanes, then the two row-halves (wm) through LDS
#pragma unroll
        for (int sft = LPR; sft < 64; sft <<= 1) {
            psum[0] += __shfl_xor(psum[0], sft); psum[1] += __shfl_xor(psum[1], sft);
            psum[2] += __shfl_xor(psum[2], sft); psum[3] += __shfl_xor(psum[3], sft);
        }
        float* sp = smem + BM * BN;
        if (lane < LPR) *(f32x4*)(sp + wm * BN + wn * WN + col) = psum;
        __syncthreads();
        if (tid < BN) a.pool[(size_t)(m0 / 64) * a.N + n0 + tid] = (sp[tid] + sp[BN + tid]) * (1.0f / 64.0f);
    }
#ifdef HMF_EXPERIMENTS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    HMF_STAMP(3);
#endif
}

#ifdef HMF_EXPERIMENTS
static int g_gemm_whatif = 0;
#endif

template <int BM, int BN, int S, int OCC, int KS = 1>
static void launch_gemm(GemmArgs a, hipStream_t st) {
    const int MT = (a.M + BM - 1) / BM, NT = a.N / BN, T = MT * NT;
    // activation bytes: the x panel (+ the strided second source) once per m-tile; filter bytes once per n-tile (residual / output go to one XCD either way)
    a.xb = xcd_n_split(4LL * a.M * (a.K1 + a.K2), 4LL * a.N * (a.K1 + a.K2), MT, NT);
    ProfScope ps(8, a.M, st);
    if (a.pool) {
        if constexpr (KS == 1 && BM == 64) hipLaunchKernelGGL((k_conv_gemm<BM, BN, S, false, OCC, KS, true>), dim3(T), dim3(256), 0, st, a);
        return;
    }
#ifdef HMF_EXPERIMENTS
    if constexpr (KS == 1 && BM == 64) {
        const int wi = g_gemm_whatif;
        if ((wi & 1) || ((wi & 2) && a.K1 + a.K2 <= 256 && a.N / BN > 1)) {
            if ((wi & 3) == 1 || !((wi & 2) && a.K1 + a.K2 <= 256 && a.N / BN > 1)) {
                if (a.x2) hipLaunchKernelGGL((k_conv_gemm<BM, BN, S, true, OCC, KS, false, 1>), dim3(T), dim3(256), 0, st, a);
                else hipLaunchKernelGGL((k_conv_gemm<BM, BN, S, false, OCC, KS, false, 1>), dim3(T), dim3(256), 0, st, a);
            } else if ((wi & 3) == 2) {
                if (a.x2) hipLaunchKernelGGL((k_conv_gemm<BM, BN, S, true, OCC, KS, false, 2>), dim3(T), dim3(256), 0, st, a);
                else hipLaunchKernelGGL((k_conv_gemm<BM, BN, S, false, OCC, KS, false, 2>), dim3(T), dim3(256), 0, st, a);
            } else {
                if (a.x2) hipLaunchKernelGGL((k_conv_gemm<BM, BN, S, true, OCC, KS, false, 3>), dim3(T), dim3(256), 0, st, a);
                else hipLaunchKernelGGL((k_conv_gemm<BM, BN, S, false, OCC, KS, false, 3>), dim3(T), dim3(256), 0, st, a);
            }
            return;
        }
    }
#endif
    if (a.x2) hipLaunchKernelGGL((k_conv_gemm<BM, BN, S, true, OCC, KS>), dim3(T), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_conv_gemm<BM, BN, S, false, OCC, KS>), dim3(T), dim3(256), 0, st, a);
}

}  // namespace hmf

using namespace hmf;

#ifdef HMF_EXPERIMENTS
extern "C" int hmf_debug_set_gemm_whatif(int mask) { g_gemm_whatif = mask; return HMF_OK; }
extern "C" int hmf_debug_gemm_stamps(long long* out_host, int blocks) {
    if (!out_host || blocks < 1 || blocks > 4096) return fail(HMF_ERR_INVALID, "hmf_debug_gemm_stamps: bad args");
    HMF_HIP(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_gemm_stamps), (size_t)blocks * 6 * sizeof(long long)));
    return HMF_OK;
}
#endif

// y = relu?(x . W[:, :K1]^T (+ x2[:, ::s, ::s, :] . W[:, K1:]^T) + bias (+ residual)).  x (B*OH*OW, K1), w (N, K1 + K2) row-major, y (B*OH*OW, N);
// x2 (B, H2, W2, K2) or NULL (then K2, H2, W2, stride2 are ignored).  tile: 0 = chosen from the shape, else ring stages * 1000000 + BM * 1000 + BN of
// {64064 (4 stages), 3064064, 3064128, 2064064, 2064128}.
static int conv1x1_gemm_impl(const float* x, const float* x2, const float* w, const float* bias, const float* residual, float* y, float* pool, int B, int OH, int OW,
                             int K1, int H2, int W2, int K2, int stride2, int N, int relu, int tile, void* stream) {
    if (!x || !w || (!y && !pool) || B <= 0 || OH <= 0 || OW <= 0) return fail(HMF_ERR_INVALID, "hmf_conv1x1_gemm: bad args");
    if (pool && (x2 || OH * OW != 64 || ((uintptr_t)pool & 15)))
        return fail(HMF_ERR_UNSUPPORTED, "hmf_conv1x1_gemm_avgpool: the fused pool is built for 64-pixel maps (one image per 64-row tile) without a second K range");
    if (!x2) K2 = 0;
    if (K1 <= 0 || K1 % 32 || K2 % 32 || N % 64) return fail(HMF_ERR_UNSUPPORTED, "hmf_conv1x1_gemm: K1=%d, K2=%d must be multiples of 32 and N=%d of 64", K1, K2, N);
    if (x2 && (stride2 < 1 || (H2 - 1) / stride2 + 1 != OH || (W2 - 1) / stride2 + 1 != OW))
        return fail(HMF_ERR_INVALID, "hmf_conv1x1_gemm: x2 %dx%d / %d does not give %dx%d", H2, W2, stride2, OH, OW);
    if (((uintptr_t)x & 15) || ((uintptr_t)x2 & 15) || ((uintptr_t)w & 15) || ((uintptr_t)y & 15) || ((uintptr_t)residual & 15) || ((uintptr_t)bias & 15))
        return fail(HMF_ERR_INVALID, "hmf_conv1x1_gemm: pointers must be 16-byte aligned");
    const long long M = (long long)B * OH * OW;
    if (M * N >= (1LL << 29) || M * K1 >= (1LL << 29) || (x2 && (long long)B * H2 * W2 * K2 >= (1LL << 29)) || (long long)N * (K1 + K2) >= (1LL << 29))
        return fail(HMF_ERR_UNSUPPORTED, "hmf_conv1x1_gemm: tensor exceeds 2^29 elements (32-bit byte offsets of the buffer loads)");
    GemmArgs a;
    a.x = x; a.x2 = x2; a.w = w; a.bias = bias; a.res = residual; a.y = y; a.pool = pool;
    a.M = (int)M; a.N = N; a.K1 = K1; a.K2 = K2; a.OH = OH; a.OW = OW; a.H2 = H2; a.W2 = W2; a.stride2 = stride2; a.relu = relu;
    if (tile == 0) {
        // Measured on MI355X at B = 32 (tools/gemm1x1_probe.py; every ResNet-50 1x1 shape x every tile).  What decides is how many work-groups a
        // CU holds at once, because co-resident work-groups in different phases are what hides a work-group's first fetch and its epilogue:
        //   64 x 128 tile, 3-stage ring (72 KB, 2 per CU): deep reductions with >= 2 work-groups per CU -- 12.9 GFLOP downsample GEMMs 97-103 us
        //     (125-133 TFLOP/s; k_conv_igemm 108-122 us), layer4.0.conv1 65 us, layer4 conv3 36-37 us;
        //   64 x 64 tile, 3-stage ring (48 KB, 3 per CU): short reductions / many tiles -- layer3 conv3 (K = 256, + identity) 38.9 us vs 47.8;
        //   64 x 64 tile, 4-stage ring (64 KB, 2 per CU): the few-tile deep layers -- layer3 conv1 (512 tiles) 34.2 us vs 39.5, layer4 conv1 (256
        //     tiles: ONE wave per SIMD) 36.8 us vs 45.2.
        // 128 x 128 / 128 x 64 tiles (one work-group per CU) lose on every shape: one round of work-groups all fetching, then all storing.
        // Round 3, later: TWO stages (32 / 48 KB of LDS: five / three work-groups per CU) beat the three-stage rings on every shape with enough
        // tiles -- more co-resident work-groups in different phases hide more of each other's first fetch and epilogue than a deeper ring hides
        // of a work-group's own fetch latency: layer2 conv3 45.6 -> 41.8 us, the 12.9 GFLOP downsample GEMMs 97-103 -> 93.5-98 us, layer1 conv1
        // 39.2 -> 36.8 us.  The 128-wide tile where its work-group count fills the 768 slots evenly (<= 768 or >= 1536 work-groups; at 1,024 --
        // layer3 conv3 / downsample -- it took 43.5 / 115 us against 36.4 / 98.0 us for the 64-wide one) and the reduction is not short.
        const long long w128 = ((M + 63) / 64) * (N / 128), w64 = ((M + 63) / 64) * (N / 64);
        const int K = K1 + K2;
        if (w64 <= 512) tile = 64064;
        else if (K <= 128 || (N % 128) || (w128 > 768 && w128 < 1536)) tile = 2064064;
        else tile = 2064128;
    }
    hipStream_t st = (hipStream_t)stream;
    if ((tile % 1000) == 128 && N % 128) return fail(HMF_ERR_INVALID, "hmf_conv1x1_gemm: a 128-channel tile needs N %% 128 == 0");
    if (pool && ((tile / 1000) % 1000 != 64 || tile / 10000000 != 0))      // launch_gemm has the pooled epilogue for 64-row, unsplit-K tiles only
        return fail(HMF_ERR_UNSUPPORTED, "hmf_conv1x1_gemm_avgpool: tile %d has no pooled epilogue (64-row tiles without a K split only)", tile);
    switch (tile) {
        case 64064: launch_gemm<64, 64, 4, 2>(a, st); break;
        case 3064064: launch_gemm<64, 64, 3, 3>(a, st); break;
        case 3064128: launch_gemm<64, 128, 3, 2>(a, st); break;
        case 2064064: launch_gemm<64, 64, 2, 5>(a, st); break;          // two stages (32 KB): five work-groups per CU
        case 2064128: launch_gemm<64, 128, 2, 3>(a, st); break;          // 48 KB: three per CU
#ifdef HMF_EXPERIMENTS           // tools/gemm1x1_probe.py: the tilings measured and not chosen, split accumulator chains
        case 128128: launch_gemm<128, 128, 4, 1>(a, st); break;
        case 64128: launch_gemm<64, 128, 4, 1>(a, st); break;
        case 128064: launch_gemm<128, 64, 4, 1>(a, st); break;
        case 3128128: launch_gemm<128, 128, 3, 1>(a, st); break;
        case 3128064: launch_gemm<128, 64, 3, 2>(a, st); break;
        case 23064064: launch_gemm<64, 64, 3, 3, 2>(a, st); break;
        case 24064064: launch_gemm<64, 64, 4, 2, 2>(a, st); break;
#endif
        default: return fail(HMF_ERR_INVALID, "hmf_conv1x1_gemm: unknown tile %d (0 = auto, 64064, 3064064, 3064128, 2064064, 2064128)", tile);
    }
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

extern "C" int hmf_conv1x1_gemm(const float* x, const float* x2, const float* w, const float* bias, const float* residual, float* y, int B, int OH, int OW,
                                int K1, int H2, int W2, int K2, int stride2, int N, int relu, int tile, void* stream) {
    if (!y) return fail(HMF_ERR_INVALID, "hmf_conv1x1_gemm: null y");
    return conv1x1_gemm_impl(x, x2, w, bias, residual, y, nullptr, B, OH, OW, K1, H2, W2, K2, stride2, N, relu, tile, stream);
}

// The encoder's last 1x1 convolution with the global average pool of models/resnet.py:214-215 in its epilogue: pooled (B, N) = mean over the 8 x 8
// pixels of relu?(x . W^T + bias (+ residual)).  y (B*64, N) receives the map itself when non-NULL (tests); the inference plan passes NULL.
extern "C" int hmf_conv1x1_gemm_avgpool(const float* x, const float* w, const float* bias, const float* residual, float* y, float* pooled, int B, int OH, int OW,
                                        int K1, int N, int relu, int tile, void* stream) {
    if (!pooled) return fail(HMF_ERR_INVALID, "hmf_conv1x1_gemm_avgpool: null pooled");
    return conv1x1_gemm_impl(x, nullptr, w, bias, residual, y, pooled, B, OH, OW, K1, 0, 0, 0, 1, N, relu, tile, stream);
}
