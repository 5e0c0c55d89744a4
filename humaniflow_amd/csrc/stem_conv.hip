// Stem of the ResNet encoder (models/resnet.py:141-143,203-205: conv 7x7 / stride 2 / pad 3, 18 -> 64 channels, BatchNorm, ReLU) as ONE
// kernel that reads the proxy representation in the layout the reference hands over (NCHW fp32) and writes the NHWC activation
// map the residual stages consume: no NCHW->NHWC conversion pass, no separate bias/ReLU pass.
//
// GEMM view: M = B*OH*OW output pixels, N = 64 channels, K = 18*7*7 = 882.  A work-group (4 waves) owns an 8 x 16 tile of output
// pixels.  It first copies the (2*8+5) x (2*16+5) x 18 input patch the tile needs into LDS (56 KB; zero where the patch leaves the
// image, so the padding costs nothing later) and then runs 441 MFMA k-steps with NO further barrier: k-step s = (cp, kh, kw) covers the
// channel pair (2cp, 2cp+1), so the A operand of lane (pixel i, q) is the LDS word  base_i + q*plane + const(s)  -- one ds_read_b32
// with an immediate offset per k-step -- and the B operand (folded weights, host-packed as [s/4][n-tile][lane][s%4]) streams from
// L2 as 16-byte loads issued a group ahead.  Each wave holds 32 pixels x 64 channels (2 MFMA tiles): lane l ends with channel
// (l & 31) + 32t of 16 pixels, adds the folded-BN bias, applies the ReLU and stores 128-byte channel runs.  Two work-groups fit a CU
// (2 x 56 KB LDS), so one loads its patch while the other computes.
#include "hmf_common.h"

namespace hmf {

#ifdef HMF_EXPERIMENTS           // tools/stem_probe.py: per work-group stamps (shader clock: start, patch in LDS, main loop done, end; 100 MHz counter at start / end)
__device__ long long g_stem_stamps[6 * 4096];
#define HMF_SSTAMP(k, v)                                                              \
    do {                                                                              \
        if (threadIdx.x == 0 && blockIdx.x < 4096) g_stem_stamps[blockIdx.x * 6 + (k)] = (long long)(v); \
    } while (0)
#else
#define HMF_SSTAMP(k, v) \
    do {                 \
    } while (0)
#endif

constexpr int kStemC = 18, kStemK = 7, kStemN = 64;
constexpr int kStemTH = 8, kStemTW = 16;                                  // output tile
constexpr int kStemPH = 2 * kStemTH + 5;                                  // 21 x 37 input patch (37 = 2 * kStemTW + 5 columns: 19 even + 18 odd slots)
// LDS image of the patch: columns de-interleaved by parity (even columns in slots 0..18, odd ones in 20..37) with a row pitch of 40 floats.
// A k-step reads column 2 px + kw for the 16 px of a row: one parity, consecutive slots -> consecutive banks; the lanes' second pixel row
// is two patch rows = 80 floats further = the other 16 banks.  (Round 1 stored rows linearly with pitch 37: stride-2 columns put the 32
// lanes of a half-wave on 16 banks, 47 % of the LDS cycles were bank conflicts.)
// Round 3: a channel plane is padded from 21 to 24 rows (960 floats), so that the patch is 54 blocks of 8 rows x 40 slots = 320 floats = five
// 64-lane LDS-DMA requests each, none of which straddles a channel: the patch is fetched by `buffer_load_dword ... lds` (no staging registers,
// no ds_write, ~5 VALU per 64 elements instead of ~40 per element).  On gfx950 the fp32 MFMAs of one wave and the VALU work of the other wave of
// a SIMD do not overlap (tools/micro/mfma_valu_overlap.hip: times ADD), so every VALU instruction of the fetch and of the epilogue is paid in
// full beside the partner work-group's MFMA loop: round 2's fetch + pooling ran 42k + 37k cycles per work-group beside a 64k-cycle main loop.
constexpr int kStemPitch = 40, kStemOdd = 20;
constexpr int kStemPR = 24;                                               // rows per channel plane in LDS (21 used)
constexpr int kStemPlane = kStemPR * kStemPitch;                          // 960 floats per channel
__device__ __forceinline__ constexpr int stem_slot(int col) { return (col & 1) * kStemOdd + (col >> 1); }
constexpr int kStemSteps = (kStemC / 2) * kStemK * kStemK;                // 441 k-steps
constexpr int kStemGroups = (kStemSteps + 3) / 4;                         // 111 groups of 4

// POOL: the 3x3 / stride-2 max-pool that follows the stem (models/resnet.py:144,206) is applied to the work-group's 8 x 16 tile before it leaves
// the CU: y is then the POOLED map (B, PH, PW, 64), zeroed by the caller.  Pool windows that lie inside the tile (3 x 7 of the 5 x 9 the tile
// touches) are stored; windows shared with a neighbouring tile are combined with an integer atomic max (post-ReLU values are >= +0, so the
// float order is the unsigned order).  The 134 MB conv map (B = 32) is never written or re-read and the pool launch disappears.
typedef __attribute__((address_space(3))) void* lds_dma_ptr_t;
// one LDS-DMA request of 64 lanes x 4 bytes (non-template helper: see conv_gemm.hip dma16)
static __device__ __forceinline__ void stem_dma4(__amdgpu_buffer_rsrc_t rsrc, lds_dma_ptr_t dst, int voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 4, voff, 0, 0, 0);
}

template <bool POOL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void k_stem_conv(
    const float* __restrict__ x, const f32x4* __restrict__ wp, const float* __restrict__ bias, float* __restrict__ y, int B, int H, int W,
    int OH, int OW, int tiles_w, int tiles_hw, int relu) {
    __shared__ __attribute__((aligned(16))) float patch[kStemC * kStemPlane];     // 67.5 KB: two work-groups per CU
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lq = lane >> 5;
    const int b = blockIdx.x / tiles_hw, trem = blockIdx.x - b * tiles_hw;
    const int oh0 = (trem / tiles_w) * kStemTH, ow0 = (trem % tiles_w) * kStemTW;
    const int ih0 = 2 * oh0 - 3, iw0 = 2 * ow0 - 3;
    HMF_SSTAMP(0, __builtin_readcyclecounter());
    HMF_SSTAMP(4, wall_clock64());
    // Two work-groups share a CU: while one runs its MFMA loop the other fetches its patch or pools its tile.  Measured per work-group
    // (tools/stem_probe.py, B = 32): patch fetch 42k cycles, main loop 64k, pooling epilogue 37k -- the two VALU-dense phases (index arithmetic
    // of 55 + 55 patch elements, ~12 pooled windows per thread) crawl at ~28 cycles per instruction because they lose issue arbitration to the
    // partner's older MFMA waves, although an MFMA wave needs the issue port once per 64 cycles.  (Raising the wave priority of those phases
    // changed nothing: the two kinds of work share the vector datapath.)  Hence the DMA fetch and the float4 pooling below.

    // ---- input patch -> LDS by LDS-DMA.  Request idx (0..269) fills LDS floats [64 idx, 64 idx + 64): block k = idx / 5 (channel k / 3, rows
    // 8 (k % 3) ..+7), fifth j = idx % 5.  Lane l of fifth j is LDS float 64 j + l of the block: row rb = (64 j + l) / 40, slot (64 j + l) % 40,
    // i.e. patch column 2 slot (slots 0..18) or 2 (slot - 20) + 1 (slots 20..37); slots 19, 38, 39, rows >= 21 and pixels outside the image get
    // an out-of-range offset: the buffer load returns 0 for them and the DMA writes that 0 (the zero padding of the convolution). ----
    // The requests carry NO per-request VALU work: while the partner work-group streams MFMAs this wave gets a VALU issue slot only now and then
    // (round 3 measured 37k cycles for a fetch with 5 VALU instructions per request).  The 15 lane offsets (3 row blocks x 5 fifths, tile-relative)
    // are computed once; a request is then one SALU-built descriptor per channel plane (base = the plane, range = H * W floats, so rows above or
    // below the image fall out of range by themselves: a negative row offset wraps to a huge unsigned one) plus the precomputed VGPR.
    constexpr int kOOBs = 0x40000000;
    int vo[3][5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int f = 64 * j + lane, rb = f / kStemPitch, slot = f - rb * kStemPitch;
        const int col = slot < 19 ? 2 * slot : ((slot >= kStemOdd && slot < kStemOdd + 18) ? 2 * (slot - kStemOdd) + 1 : -1000);
        const bool cok = (col >= 0) & ((unsigned)(iw0 + col) < (unsigned)W);
#pragma unroll
        for (int b3 = 0; b3 < 3; ++b3) {
            const int r = 8 * b3 + rb;                            // patch row; rows >= 21 are LDS padding that no k-step reads: masked for determinism
            vo[b3][j] = (cok && r < kStemPH && (unsigned)(ih0 + r) < (unsigned)H) ? ((ih0 + r) * W + iw0 + col) * 4 : kOOBs;
        }
    }
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const float* ximg = x + (long long)b * kStemC * H * W;
    for (int ch = 0; ch < kStemC; ++ch) {
        const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)(ximg + (long long)ch * H * W), 0, H * W * 4, 0x00020000);
#pragma unroll
        for (int b3 = 0; b3 < 3; ++b3) {
            const int k = ch * 3 + b3;
            if ((k & 3) != wv) continue;                          // wave-uniform: wave w fetches blocks w, w + 4, ...
#pragma unroll
            for (int j = 0; j < 5; ++j)
                stem_dma4(rc, (lds_dma_ptr_t)((__attribute__((address_space(3))) float*)patch + k * 320 + 64 * j), vo[b3][j]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    HMF_SSTAMP(1, __builtin_readcyclecounter());

    // ---- 441 k-steps ----
    const int py = 2 * wave + (li >> 4), px = li & 15;
    const float* pa = patch + (2 * py) * kStemPitch + px + lq * kStemPlane;
    const f32x4* pw = wp + lane;
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // software pipeline: weights kPF groups (= 8*kPF MFMAs) ahead in a register ring, LDS operands one group ahead; the
    // sched_barriers pin the issue points (left alone, the compiler sinks every load next to its use and exposes the latency)
    constexpr int kPF = 4, kRing = kPF + 1;
    f32x4 bw[kRing][2];
    float av[2][4];
    auto ldsA = [&](int g, int slot) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int s = 4 * g + u < kStemSteps ? 4 * g + u : kStemSteps - 1;
            const int cp = s / 49, kk = s - cp * 49, kh = kk / 7, kw = kk - kh * 7;
            av[slot][u] = pa[cp * 2 * kStemPlane + kh * kStemPitch + stem_slot(kw)];
        }
    };
#pragma unroll
    for (int g = 0; g < kPF; ++g) {
        bw[g][0] = pw[g * 128];
        bw[g][1] = pw[g * 128 + 64];
    }
    ldsA(0, 0);
#pragma clang loop unroll(full)
    for (int g = 0; g < kStemGroups; ++g) {
        if (g + kPF < kStemGroups) {
            bw[(g + kPF) % kRing][0] = pw[(g + kPF) * 128];
            bw[(g + kPF) % kRing][1] = pw[(g + kPF) * 128 + 64];
        }
        if (g + 1 < kStemGroups) ldsA(g + 1, (g + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g & 1][u], bw[g % kRing][0][u], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g & 1][u], bw[g % kRing][1][u], acc[1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    HMF_SSTAMP(2, __builtin_readcyclecounter());
    if (POOL) {
        // ---- epilogue with the max-pool: tile [8][16][64] -> LDS (the patch is dead), pooled windows from LDS ----
        __syncthreads();
        float* tl = patch;                                       // [row 8][col 16][ch 64]
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int n = 32 * t + li;
            const float bv = bias ? bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * lq;
                const int row = 2 * wave + (i >> 4), col = i & 15;
                const bool in = (oh0 + row < OH) & (ow0 + col < OW);
                tl[(row * 16 + col) * 64 + n] = in ? fmaxf(acc[t][r] + bv, 0.f) + 0.f : 0.f;       // + 0.f: never -0 (its bit pattern would win the max)
            }
        }
        __syncthreads();
        // thread = (pooled position, 4 channels): nine 16-byte LDS reads per window, one 16-byte store (windows inside the tile) or four integer
        // atomic maxima (windows shared with a neighbour tile); 45 positions x 16 channel quads = 720 items over 256 threads
        const int PH = (OH - 1) / 2 + 1, PW = (OW - 1) / 2 + 1;
        const int pr0 = oh0 >> 1, pc0 = ow0 >> 1;
        for (int it = tid; it < 45 * 16; it += 256) {
            const int p = it >> 4, cq = (it & 15) * 4;
            const int a = p / 9, c = p - a * 9;                   // pooled position (pr0 + a, pc0 + c): conv rows 2a-1 .. 2a+1 of the tile
            if (pr0 + a >= PH || pc0 + c >= PW) continue;
            f32x4 m = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dr = -1; dr <= 1; ++dr) {
                const int rr = 2 * a + dr;
                if ((unsigned)rr >= 8u) continue;
#pragma unroll
                for (int dc = -1; dc <= 1; ++dc) {
                    const int cc = 2 * c + dc;
                    if ((unsigned)cc < 16u) {
                        const f32x4 v = *(const f32x4*)&tl[(rr * 16 + cc) * 64 + cq];
                        m[0] = fmaxf(m[0], v[0]); m[1] = fmaxf(m[1], v[1]); m[2] = fmaxf(m[2], v[2]); m[3] = fmaxf(m[3], v[3]);
                    }
                }
            }
            float* dst = y + (((long long)b * PH + pr0 + a) * PW + pc0 + c) * kStemN + cq;
            const bool whole = (a >= 1) & (a <= 3) & (c >= 1) & (c <= 7);
            if (whole) {
                *(f32x4*)dst = m;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) atomicMax((unsigned int*)dst + e, __float_as_uint(m[e]));
            }
        }
#ifdef HMF_EXPERIMENTS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        HMF_SSTAMP(3, __builtin_readcyclecounter());
        HMF_SSTAMP(5, wall_clock64());
#endif
        return;
    }
    // ---- epilogue: + bias, ReLU, NHWC stores (lane = channel, 128-byte runs) ----
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = 32 * t + li;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * lq;
            const int oh = oh0 + 2 * wave + (i >> 4), ow = ow0 + (i & 15);
            if (oh < OH && ow < OW) y[(((long long)b * OH + oh) * OW + ow) * kStemN + n] = relu ? fmaxf(acc[t][r] + bv, 0.f) : acc[t][r] + bv;
        }
    }
}

// max-pool 3x3 / stride 2 / pad 1 over an NHWC map (models/resnet.py:144,206): one float4 of channels per thread
__global__ __launch_bounds__(256) void k_maxpool_nhwc(const f32x4* __restrict__ x, f32x4* __restrict__ y, int B, int H, int W, int C4, int OH,
                                                      int OW) {
    const long long n = (long long)B * OH * OW * C4;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const int c = (int)(e % C4);
        long long p = e / C4;
        const int ow = (int)(p % OW);
        p /= OW;
        const int oh = (int)(p % OH), b = (int)(p / OH);
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int dh = -1; dh <= 1; ++dh) {
            const int ih = 2 * oh + dh;
            if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
            for (int dw = -1; dw <= 1; ++dw) {
                const int iw = 2 * ow + dw;
                if ((unsigned)iw >= (unsigned)W) continue;
                const f32x4 v = x[(((long long)b * H + ih) * W + iw) * C4 + c];
                m[0] = fmaxf(m[0], v[0]); m[1] = fmaxf(m[1], v[1]); m[2] = fmaxf(m[2], v[2]); m[3] = fmaxf(m[3], v[3]);
            }
        }
        y[e] = m;
    }
}

}  // namespace hmf

using namespace hmf;

#ifdef HMF_EXPERIMENTS
extern "C" int hmf_debug_stem_stamps(long long* out_host, int blocks) {
    if (!out_host || blocks < 1 || blocks > 4096) return fail(HMF_ERR_INVALID, "hmf_debug_stem_stamps: bad args");
    HMF_HIP(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_stem_stamps), (size_t)blocks * 6 * sizeof(long long)));
    return HMF_OK;
}
#endif

extern "C" long long hmf_stem_packed_floats(void) { return (long long)kStemGroups * 2 * 64 * 4; }

extern "C" int hmf_stem_pack_host(const float* w /*64x18x7x7*/, float* packed) {
    if (!w || !packed) return fail(HMF_ERR_INVALID, "hmf_stem_pack_host: null pointer");
    for (int g = 0; g < kStemGroups; ++g)
        for (int t = 0; t < 2; ++t)
            for (int l = 0; l < 64; ++l)
                for (int u = 0; u < 4; ++u) {
                    const int s = 4 * g + u, n = 32 * t + (l & 31), q = l >> 5;
                    float v = 0.f;
                    if (s < kStemSteps) {
                        const int cp = s / 49, kk = s - cp * 49;
                        v = w[(n * kStemC + 2 * cp + q) * 49 + kk];
                    }
                    packed[((g * 2 + t) * 64 + l) * 4 + u] = v;
                }
    return HMF_OK;
}

extern "C" int hmf_stem_conv(const float* x, const float* w_packed, const float* bias, float* y, int B, int C, int H, int W, int relu, void* stream) {
    if (!x || !w_packed || !y || B <= 0 || H <= 0 || W <= 0) return fail(HMF_ERR_INVALID, "hmf_stem_conv: bad args");
    if (C != kStemC) return fail(HMF_ERR_UNSUPPORTED, "hmf_stem_conv: %d input channels (the 18-channel proxy representation is built)", C);
    if ((uintptr_t)w_packed & 15) return fail(HMF_ERR_INVALID, "hmf_stem_conv: packed weights must be 16-byte aligned");
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    const int th = (OH + kStemTH - 1) / kStemTH, tw = (OW + kStemTW - 1) / kStemTW;
    const long long blocks = (long long)B * th * tw;
    if (blocks >= (1LL << 31)) return fail(HMF_ERR_UNSUPPORTED, "hmf_stem_conv: too many tiles");
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(9, B, st);
    hipLaunchKernelGGL(k_stem_conv<false>, dim3((unsigned)blocks), dim3(256), 0, st, x, (const f32x4*)w_packed, bias, y, B, H, W, OH, OW, tw, th * tw, relu);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

// conv 7x7/2 + bias + ReLU + max-pool 3x3/2 in one launch: y_pooled (B, PH, PW, 64), PH = ((H-1)/2)/2 + 1 (the inference plan's stem)
extern "C" int hmf_stem_conv_pool(const float* x, const float* w_packed, const float* bias, float* y_pooled, int B, int C, int H, int W, void* stream) {
    if (!x || !w_packed || !y_pooled || B <= 0 || H <= 0 || W <= 0) return fail(HMF_ERR_INVALID, "hmf_stem_conv_pool: bad args");
    if (C != kStemC) return fail(HMF_ERR_UNSUPPORTED, "hmf_stem_conv_pool: %d input channels (the 18-channel proxy representation is built)", C);
    if ((uintptr_t)w_packed & 15) return fail(HMF_ERR_INVALID, "hmf_stem_conv_pool: packed weights must be 16-byte aligned");
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1, PH = (OH - 1) / 2 + 1, PW = (OW - 1) / 2 + 1;
    const int th = (OH + kStemTH - 1) / kStemTH, tw = (OW + kStemTW - 1) / kStemTW;
    const long long blocks = (long long)B * th * tw;
    if (blocks >= (1LL << 31)) return fail(HMF_ERR_UNSUPPORTED, "hmf_stem_conv_pool: too many tiles");
    hipStream_t st = (hipStream_t)stream;
    if (int zr = zero_async(y_pooled, (size_t)B * PH * PW * kStemN * sizeof(float), st)) return zr;       // the identity of the max over post-ReLU values
    ProfScope ps(9, B, st);
    hipLaunchKernelGGL(k_stem_conv<true>, dim3((unsigned)blocks), dim3(256), 0, st, x, (const f32x4*)w_packed, bias, y_pooled, B, H, W, OH, OW, tw, th * tw, 1);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

extern "C" int hmf_maxpool3x3s2_nhwc(const float* x, float* y, int B, int H, int W, int C, void* stream) {
    if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0) return fail(HMF_ERR_INVALID, "hmf_maxpool3x3s2_nhwc: bad args");
    if (C % 4 != 0 || ((uintptr_t)x & 15) || ((uintptr_t)y & 15)) return fail(HMF_ERR_UNSUPPORTED, "hmf_maxpool3x3s2_nhwc: C %% 4 != 0 or unaligned");
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    const long long n = (long long)B * OH * OW * (C / 4);
    long long blocks = (n + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(k_maxpool_nhwc, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const f32x4*)x, (f32x4*)y, B, H, W, C / 4, OH, OW);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}
