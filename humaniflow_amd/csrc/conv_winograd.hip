// Winograd F(2x2, 3x3) NHWC fp32 convolution for the encoder's stride-1 3x3 layers (row a1): 2.25x fewer MFMA k-steps than the direct
// form for the same result (Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A, Lavin & Gray 2016), fused with the folded-BN epilogue.
//
// Work decomposition (one work-group = 4 waves):
//   * an 8 x 16 tile of output pixels of one image = 32 Winograd tiles (4 x 8 of 2x2 pixels) x 32 output channels;
//   * K is walked in slabs of 64 input channels; per slab the 10 x 18 x 64 input patch is copied once into LDS (raw buffer loads, zeros in the padding);
//   * wave w owns the four transform positions (xi = w, nu = 0..3).  The input transform is done in registers on the fly: lane (tile i, q)
//     reads the 2 x 4 patch pixels its row xi needs (8 ds_read_b128 at immediate offsets), forms t[b] = +-d[a1][b] +- d[a2][b] and
//     V[nu] = t0-t2, t1+t2, t2-t1, t1-t3 -- 4 MFMA A operands (4 k-steps each) for 32 VALU instructions, no transformed copy in LDS;
//   * the transformed weights U = G g G^T are computed at plan time and packed in MFMA B-operand lane order per (n-tile, wave, slab, k-group, nu):
//     they stream L2 -> VGPR one k-group ahead and never touch LDS;
//   * 16 accumulators (positions) x 32 tiles x 32 channels live in the 4 waves (4 x 16 registers each).  After the last slab they are
//     exchanged through LDS (two halves of 16 tiles, reusing the patch buffer) and every thread applies the output transform to
//     (tile, channel) pairs: + bias, + residual, ReLU, NHWC stores with lane = channel (128-byte runs).
#include "hmf_common.h"

namespace hmf {

struct WinoArgs {
    const float* x;       // [B][H][W][Cin]
    const f32x4* wp;      // packed transformed weights
    const float* bias;
    const float* res;
    float* y;             // [B][H][W][Cout]
    int B, H, W, Cin, Cout, relu, tiles_w, tiles_hw, n_tiles;
    int xb;               // XCD split along the channel tiles (hmf_common.h xcd_n_split)
    int ksplit;           // > 1: blockIdx.y walks Cin / 64 / ksplit slabs and stores its raw 2 x 2 outputs to part[blockIdx.y] (k_wino_splitk_finish adds them up)
    float* part;          // [ksplit][B][H][W][Cout]
};

#ifdef HMF_EXPERIMENTS           // tools/wino_probe.py: per work-group stamps (shader clock: start, first patch in LDS, main loop done, end; the cycles
// spent between a slab's top and its second barrier summed over slabs; the 100 MHz counter at start and end)
__device__ long long g_wino_stamps[8 * 4096];
#define HMF_WSTAMP(k, v)                                                              \
    do {                                                                              \
        if (threadIdx.x == 0 && blockIdx.x < 4096) g_wino_stamps[blockIdx.x * 8 + (k)] = (long long)(v); \
    } while (0)
#define HMF_WCLK() ((long long)__builtin_readcyclecounter())
#else
#define HMF_WSTAMP(k, v) \
    do {                 \
    } while (0)
#define HMF_WCLK() 0LL
#endif

constexpr int kWTH = 8, kWTW = 16, kWSC = 64;
constexpr int kWPH = kWTH + 2;
constexpr int kWCS = kWSC + 4;                       // LDS pixel stride (floats)
constexpr int kWNG = kWSC / 8;                       // k-groups per slab
typedef __attribute__((address_space(3))) void* wino_lds_ptr_t;
// one LDS-DMA request: 64 lanes x 16 bytes, lane l's bytes from (rsrc base + voff + soff) land at dst + 16 l (non-template helper: see conv_gemm.hip dma16)
static __device__ __forceinline__ void wino_dma16(__amdgpu_buffer_rsrc_t rsrc, wino_lds_ptr_t dst, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voff, soff, 0, 0);
}

// NIMG = 1: the 8 x 16 tile lies in one image (patch 10 x 18).  NIMG = 2: maps at most 8 pixels wide (layer4 at 256x256 input) -- the tile
// is 8 x 8 pixels of TWO consecutive images side by side (patch 10 x (10 + 10)), so the MFMA still sees 32 Winograd tiles.

// MODE (tools/winograd_ablate.py, results wrong when != 0): & 1 = no patch fetch after the first slab, & 2 = no output transform / stores.
// PIPE: 0 = transform of group g, then its MFMAs, fenced (round 1); 1 = the transform of group g + 1 is issued with the MFMAs of group g;
// 2 = the same with an explicit 1 MFMA : 1 ds_read : 3 VALU interleave.
// DMA: the slab patch goes HBM -> LDS by LDS-DMA (the same LDS image, padding slots written as zeros from out-of-range lanes): no staging registers
// (48 + 24 VGPRs), no ds_write, which is what lets the TN = 2 form (64 output channels per work-group: half the input transforms per MFMA) fit.
template <int TN, int NIMG, int MODE = 0, int PIPE = 1, int OCC = 2, bool DMA = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC))) void k_conv_winograd(const WinoArgs a) {
    // LDS image of the patch.  NIMG = 1: columns de-interleaved by parity (pixel (r, c) at slot (c & 1) * 9 + (c >> 1) of row r): the four patch
    // columns 2 tj + {0,1,2,3} of tile tj are then slots {tj, 9 + tj, tj + 1, 9 + tj + 1}, i.e. consecutive lanes read consecutive pixels (17 x 16 B
    // apart), and with a row pitch of 4 (mod 8) slots the four 16-lane groups of ds_read_b128 each cover 16 distinct bank slots: conflict-free.
    // NIMG = 2: the same de-interleave inside each image's 10 columns (slot img * 10 + (c & 1) * 5 + (c >> 1)), and MFMA row i of the wave is tile
    // (ti = i >> 3, tj = 4 h + (i & 3)) with h = bit 2 of i flipped for i in 8..23: the 16 lanes that ds_read_b128 serves together are then the 4 x 4
    // tiles of ONE image, and with a row pitch of 10 (mod 16) slots their 16 reads hit 16 distinct bank slots (round 2 measured 40.8 % of the LDS
    // cycles of this variant as bank conflicts).
    constexpr int kWPW = NIMG == 1 ? kWTW + 2 : 20, kWRP = NIMG == 1 ? (kWPW * kWCS + 8) : (kWPW * kWCS + 24);
    static_assert(NIMG != 1 || (kWRP / 4) % 8 == 4, "row pitch must be 4 (mod 8) 16-byte slots");
    static_assert(NIMG != 2 || (kWRP / 4) % 16 == 10, "row pitch must be 10 (mod 16) 16-byte slots");
    constexpr int NV = kWPH * kWPW * (kWSC / 4), NIT = (NV + 255) / 256;
    constexpr int kSlotsRow = kWRP / 4, kSlots = kWPH * kSlotsRow, NQ = (kSlots + 63) / 64, NQW = (NQ + 3) / 4;      // DMA: 16-byte slots, requests, requests per wave
    __shared__ __attribute__((aligned(16))) float lds[DMA ? NQ * 256 : kWPH * kWRP];     // patch (49 / 54 KB); reused for the accumulator exchange (2 x 32 KB halves)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lq = lane >> 5;
    HMF_WSTAMP(0, HMF_WCLK());
#ifdef HMF_EXPERIMENTS
    HMF_WSTAMP(5, wall_clock64());
    long long fetch_cyc = 0;
#endif

    const int L = xcd_tile(blockIdx.x, gridDim.x / a.n_tiles, a.n_tiles, a.xb);
    const int nt = L % a.n_tiles, pt = L / a.n_tiles;
    const int b = (pt / a.tiles_hw) * NIMG, trem = pt - (pt / a.tiles_hw) * a.tiles_hw;
    const int oh0 = (trem / a.tiles_w) * kWTH, ow0 = NIMG == 1 ? (trem % a.tiles_w) * kWTW : 0;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.B * a.H * a.W * a.Cin * 4, 0x00020000);
    constexpr int kOOB = (int)0x80000000;
    int goff[DMA ? NQW : NIT], loff[DMA ? 1 : NIT];
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    if (DMA) {
        // request q = 4 k + wave fills LDS slots [64 q, 64 q + 64): slot -> (patch row, pixel slot, channel quad); padding slots, slots past the image and
        // pixels outside the map read out of range (the DMA writes the 0 the buffer load returns)
#pragma unroll
        for (int k = 0; k < NQW; ++k) {
            const int sl = 64 * (4 * k + wv) + lane;
            const int r = sl / kSlotsRow, rs = sl - r * kSlotsRow, ps = rs / (kWCS / 4), c4 = rs - ps * (kWCS / 4);
            const int img = NIMG == 1 ? 0 : ps / 10, q10 = ps - img * 10;
            const int col = NIMG == 1 ? (ps < 9 ? 2 * ps : 2 * (ps - 9) + 1) : (q10 < 5 ? 2 * q10 : 2 * (q10 - 5) + 1);
            const int ih = oh0 + r - 1, iw = NIMG == 1 ? ow0 + col - 1 : col - 1;
            const bool ok = (r < kWPH) & (ps < kWPW) & (c4 < kWSC / 4) & ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W) & (b + img < a.B);
            goff[k] = ok ? ((((b + img) * a.H + ih) * a.W + iw) * a.Cin + 4 * c4) * 4 : kOOB;
        }
        loff[0] = 0;
    } else {
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int e = tid + 256 * k;
            const int p = e / (kWSC / 4), c4 = e - p * (kWSC / 4);
            const int r = p / kWPW, col = p - r * kWPW;
            const int img = NIMG == 1 ? 0 : col / 10;
            const int ih = oh0 + r - 1, iw = NIMG == 1 ? ow0 + col - 1 : col - img * 10 - 1;
            const bool ok = (e < NV) & ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W) & (b + img < a.B);
            goff[k] = ok ? ((((b + img) * a.H + ih) * a.W + iw) * a.Cin + 4 * c4) * 4 : kOOB;
            const int cl = col - img * 10;
            loff[k] = r * kWRP + (NIMG == 1 ? (col & 1) * 9 + (col >> 1) : img * 10 + (cl & 1) * 5 + (cl >> 1)) * kWCS + 4 * c4;
        }
    }
    const int nslab = a.Cin / kWSC;
    const int s_per = nslab / a.ksplit, s_begin = (int)blockIdx.y * s_per, s_end = s_begin + s_per;       // ksplit = 1: all slabs
    // weight stream of this wave: [n-tile][wave xi][slab][g][nu][lane][4]
    const f32x4* pw[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t)
        pw[t] = a.wp + ((MODE & 8) ? 0LL : ((long long)((nt * TN + t) * 4 + wave) * nslab + s_begin) * kWNG * 4) * 64 + lane;   // MODE & 8: every wave streams the SAME weights (L1 hits)

    // row transform of this wave: t[b] = d[a1][b] + sg * d[a2][b]   (B^T rows: d0-d2, d1+d2, d2-d1, d1-d3): one fma per element, the sign is wave-uniform
    const int a1 = wave == 0 ? 0 : (wave == 2 ? 2 : 1), a2 = wave == 3 ? 3 : (wave == 2 ? 1 : 2);
    const float sg = wave == 1 ? 1.f : -1.f;
    // MFMA row i of this wave <-> tile (ti, tj) of the 4 x 8 tiles of the work-group
    auto row_tile = [](int i) {
        return NIMG == 1 ? i : (i & ~4) | ((((i >> 2) ^ (((i >> 3) + 1) >> 1)) & 1) << 2);
    };
    const int tl = row_tile(li);
    const int ti = tl >> 3, tj = tl & 7;
    const float* pbase = lds + (2 * ti) * kWRP + (NIMG == 1 ? tj : (tj >> 2) * 10 + (tj & 3)) * kWCS + 4 * lq;
    const float* p1 = pbase + a1 * kWRP;
    const float* p2 = pbase + a2 * kWRP;

    f32x16 acc[TN][4];
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][n][r] = 0.f;

    f32x4 bw[2][TN][4];
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int n = 0; n < 4; ++n) bw[0][t][n] = pw[t][n * 64];

    for (int s = s_begin; s < s_end; ++s) {
#ifdef HMF_EXPERIMENTS
        const long long tA = HMF_WCLK();
#endif
        if (DMA) {
            if (s != s_begin) __syncthreads();                           // every wave is past its reads of the previous slab
#pragma unroll
            for (int k = 0; k < NQW; ++k)
                if (4 * k + wv < NQ)
                    wino_dma16(rx, (wino_lds_ptr_t)((__attribute__((address_space(3))) float*)lds + (4 * k + wv) * 256), goff[k], s * kWSC * 4);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        } else {
        f32x4 pv[NIT];
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int off = goff[k] == kOOB ? kOOB : goff[k] + s * kWSC * 4;
            if (MODE & 1) pv[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            else pv[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s != s_begin) __syncthreads();
#pragma unroll
        for (int k = 0; k < NIT; ++k)
            if (NV % 256 == 0 || k < NIT - 1 || tid + 256 * k < NV) *(f32x4*)&lds[loff[k]] = pv[k];
        __syncthreads();
        }
#ifdef HMF_EXPERIMENTS
        fetch_cyc += HMF_WCLK() - tA;
        if (s == s_begin) HMF_WSTAMP(1, HMF_WCLK());
#endif

        const bool last = (s + 1 == s_end);
        auto xform = [&](int g, f32x4 (&v)[4]) {
            f32x4 t[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                // column c of the tile: slot offset (c & 1) * (9 | 5) + (c >> 1) in the de-interleaved image
                const int co = ((c & 1) * (NIMG == 1 ? 9 : 5) + (c >> 1)) * kWCS;
                const f32x4 d1 = *(const f32x4*)(p1 + co + 8 * g);
                const f32x4 d2 = *(const f32x4*)(p2 + co + 8 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) t[c][e] = __builtin_fmaf(sg, d2[e], d1[e]);
            }
            v[0] = t[0] - t[2];
            v[1] = t[1] + t[2];
            v[2] = t[2] - t[1];
            v[3] = t[1] - t[3];
        };
        f32x4 v[2][4];
        if (PIPE) xform(0, v[0]);
#pragma clang loop unroll(full)
        for (int g = 0; g < kWNG; ++g) {
            if (!(last && g + 1 == kWNG)) {
#pragma unroll
                for (int t = 0; t < TN; ++t)
#pragma unroll
                    for (int n = 0; n < 4; ++n) bw[(g + 1) & 1][t][n] = pw[t][((g + 1) * 4 + n) * 64];
            }
            if (PIPE) {
                if (g + 1 < kWNG) xform(g + 1, v[(g + 1) & 1]);
            } else {
                xform(g, v[g & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int n = 0; n < 4; ++n)
#pragma unroll
                    for (int t = 0; t < TN; ++t) acc[t][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[g & 1][n][u], bw[g & 1][t][n][u], acc[t][n], 0, 0, 0);
            if (PIPE == 2) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (i < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                }
            }
            if (!PIPE) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t = 0; t < TN; ++t) pw[t] += kWNG * 4 * 64;
    }

    // ---- output transform Y = A^T M A.  The nu direction is applied in registers (a wave owns all four nu of its row xi): per (tile, channel) the
    // wave keeps P[xi][dx] = (m0 + m1 + m2, m1 - m2 - m3); the 4 x 2 values are exchanged through LDS in one pass (32 KB, the patch buffer) and every
    // thread finishes (tile, channel) pairs along xi.  (Round 1 exchanged all 16 positions in two half passes: twice the LDS traffic and barriers;
    // the output transform was 5-20 us of a 50-65 us launch, tools/winograd_ablate.py.)
    // accumulator register r of lane (j = li, q) holds tile (r & 3) + 8 (r >> 2) + 4 q, channel j
#ifdef HMF_EXPERIMENTS
    HMF_WSTAMP(2, HMF_WCLK());
    HMF_WSTAMP(4, fetch_cyc);
#endif
    if (MODE & 2) {
        if (acc[0][0][0] + acc[0][1][1] + acc[0][2][2] + acc[0][3][3] == 12345.678f) a.y[tid] = 1.f;      // keep the accumulators alive
        return;
    }
    // Second half of the exchange: thread = (tile, 4 consecutive channels): eight 16-byte LDS reads, the xi direction of the transform on float4s, and
    // one 16-byte store per output pixel (8 lanes = the 128 contiguous bytes of a pixel's 32 channels).  Round 2 finished (tile, channel) pairs with
    // 32 scalar LDS reads and 16 scalar stores per thread: the output phase was 10-13k cycles of a 40-100k cycle work-group (tools/wino_probe.py).
    const int otile = tid >> 3, oq = tid & 7;
    const int otl = row_tile(otile);
    const int oh = oh0 + 2 * (otl >> 3), ow = NIMG == 1 ? ow0 + 2 * (otl & 7) : 2 * (otl & 3);
    const int bi = b + (NIMG == 1 ? 0 : (otl >> 2) & 1);
    constexpr int kOOBo = (int)0x80000000;
    const int ybytes = a.B * a.H * a.W * a.Cout * 4;
    // split reduction: raw sums of this slab range into part[blockIdx.y] -- no bias, no residual, no ReLU (k_wino_splitk_finish applies them to the total)
    const bool partial = a.ksplit > 1;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(partial ? (void*)(a.part + (size_t)blockIdx.y * (ybytes / 4)) : (void*)a.y, 0, ybytes, 0x00020000);
    const float* resp = partial ? nullptr : a.res;
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)(resp ? resp : a.y), 0, ybytes, 0x00020000);
    const float lo = (a.relu && !partial) ? 0.f : -INFINITY;
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int n_out = (nt * TN + t) * 32 + 4 * oq;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (a.bias && !partial) bv = *(const f32x4*)(a.bias + n_out);
        int off[2][2];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx)
                off[dy][dx] = (oh + dy < a.H && ow + dx < a.W && bi < a.B) ? ((((bi * a.H + oh + dy) * a.W + ow + dx) * a.Cout) + n_out) * 4 : kOOBo;
        f32x4 rv[2][2];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx)
                rv[dy][dx] = resp ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, off[dy][dx], 0, 0)) : f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int tile = (r & 3) + 8 * (r >> 2) + 4 * lq;
            const float m0 = acc[t][0][r], m1 = acc[t][1][r], m2 = acc[t][2][r], m3 = acc[t][3][r];
            lds[((wave * 2 + 0) * 32 + tile) * 32 + li] = m0 + m1 + m2;
            lds[((wave * 2 + 1) * 32 + tile) * 32 + li] = m1 - m2 - m3;
        }
        __syncthreads();
        f32x4 pm[4][2];
#pragma unroll
        for (int xi = 0; xi < 4; ++xi)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) pm[xi][dx] = *(const f32x4*)&lds[((xi * 2 + dx) * 32 + otile) * 32 + 4 * oq];
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            f32x4 y0 = pm[0][dx] + pm[1][dx] + pm[2][dx] + bv + rv[0][dx];
            f32x4 y1 = pm[1][dx] - pm[2][dx] - pm[3][dx] + bv + rv[1][dx];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y0[e] = fmaxf(y0[e], lo);
                y1[e] = fmaxf(y1[e], lo);
            }
            typedef __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int u32x4;
            if (!(MODE & 4) || y0[0] == 12345.678f) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y0), ry, off[0][dx], 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y1), ry, off[1][dx], 0, 0);
            }
        }
    }
#ifdef HMF_EXPERIMENTS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    HMF_WSTAMP(3, HMF_WCLK());
    HMF_WSTAMP(6, wall_clock64());
#endif
}


// Small batches (the predict shape: one image): a launch has 16-64 work-groups, each walking all Cin / 64 slabs -- 256 CUs wait for 16 long loops.
// With ksplit > 1 blockIdx.y takes a slab range and k_conv_winograd stores raw sums; this kernel adds the ranges in a FIXED order (deterministic) and
// applies bias, residual and ReLU: y = relu?(sum_s part[s] + bias + residual).  One thread per 4 channels.
__global__ __launch_bounds__(256) void k_wino_splitk_finish(const f32x4* __restrict__ part, int ksplit, long long n4, const float* __restrict__ bias,
                                                            const f32x4* __restrict__ res, f32x4* __restrict__ y, int cout4, int relu) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n4) return;
    f32x4 acc = part[e];
    for (int s = 1; s < ksplit; ++s) acc += part[(long long)s * n4 + e];
    if (bias) acc += *(const f32x4*)(bias + 4 * (int)(e % cout4));
    if (res) acc += res[e];
    if (relu) { acc[0] = fmaxf(acc[0], 0.f); acc[1] = fmaxf(acc[1], 0.f); acc[2] = fmaxf(acc[2], 0.f); acc[3] = fmaxf(acc[3], 0.f); }
    y[e] = acc;
}

int splitk_finish(const float* part, int ksplit, long long n, const float* bias, const float* res, float* y, int Cout, int relu, hipStream_t st) {
    if (!part || !y || ksplit < 1 || n <= 0 || (n & 3) || (Cout & 3) || (((uintptr_t)part | (uintptr_t)y | (uintptr_t)bias | (uintptr_t)res) & 15))
        return fail(HMF_ERR_INVALID, "splitk_finish: bad args");
    const long long n4 = n / 4;
    if ((n4 + 255) / 256 >= (1LL << 31)) return fail(HMF_ERR_UNSUPPORTED, "splitk_finish: too many elements");
    hipLaunchKernelGGL(k_wino_splitk_finish, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, (const f32x4*)part, ksplit, n4, bias, (const f32x4*)res, (f32x4*)y,
                       Cout / 4, relu);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

#ifdef HMF_EXPERIMENTS
// ---------------------------------------------------------------------------------------------------------------------------------------------
// Round 4 EXPERIMENT (measured, not faster: see the dispatch comment in conv_winograd_impl): the same convolution with the patch fetch taken off the critical path.  Stamps (tools/wino_probe.py) show a work-group of
// k_conv_winograd as  first patch 6-11 k cycles | per slab: global fetch + two barriers, then 8 k-groups of MFMAs | output 2.5-6 k:  every
// slab's global -> register -> LDS copy is exposed (only co-resident work-groups hide it), and on the one-slab layers (layer1 / layer2) the
// first fetch is a third of the work-group's life.  Here a work-group walks a list of STAGES = (pixel tile, slab), TPW tiles per work-group,
// and the patch of stage i + 1 is requested WHILE stage i's MFMAs issue: one or two 16-byte loads per k-group, placed after that group's weight
// loads so that the counted vmcnt wait for the weights does not cover them (vmcnt retires in order: a patch load issued BEFORE a weight load
// would stall the next k-group on HBM latency).  The data waits in registers until the MFMA loop is done, then goes to LDS between two barriers.
// Same arithmetic, same k order, same output transform as k_conv_winograd: bitwise the same results.
// ---------------------------------------------------------------------------------------------------------------------------------------------
template <int NIMG, int PIPE, int OCC, int TPW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC))) void k_conv_winograd_pf(const WinoArgs a, const int ptiles) {
    constexpr int kWPW = NIMG == 1 ? kWTW + 2 : 20, kWRP = NIMG == 1 ? (kWPW * kWCS + 8) : (kWPW * kWCS + 24);
    constexpr int NV = kWPH * kWPW * (kWSC / 4), NIT = (NV + 255) / 256;
    __shared__ float lds[kWPH * kWRP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lq = lane >> 5;

    const int T = gridDim.x;
    int L = blockIdx.x;
    if ((T & 7) == 0) L = (blockIdx.x & 7) * (T >> 3) + (blockIdx.x >> 3);
    const int nt = L % a.n_tiles, pt0 = (L / a.n_tiles) * TPW;
    const int ntl = min(TPW, ptiles - pt0);                     // pixel tiles of this work-group
    const int nslab = a.Cin / kWSC, nstage = ntl * nslab;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.B * a.H * a.W * a.Cin * 4, 0x00020000);
    constexpr int kOOB = (int)0x80000000;
    // patch element k of this thread: LDS offset (tile-invariant) and global offset (per tile)
    int loff[NIT], goff[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int e = tid + 256 * k;
        const int p = e / (kWSC / 4), c4 = e - p * (kWSC / 4);
        const int r = p / kWPW, col = p - r * kWPW;
        const int img = NIMG == 1 ? 0 : col / 10;
        const int cl = col - img * 10;
        loff[k] = r * kWRP + (NIMG == 1 ? (col & 1) * 9 + (col >> 1) : img * 10 + (cl & 1) * 5 + (cl >> 1)) * kWCS + 4 * c4;
    }
    auto tile_geom = [&](int pt, int& b, int& oh0, int& ow0) {
        b = (pt / a.tiles_hw) * NIMG;
        const int trem = pt - (pt / a.tiles_hw) * a.tiles_hw;
        oh0 = (trem / a.tiles_w) * kWTH;
        ow0 = NIMG == 1 ? (trem % a.tiles_w) * kWTW : 0;
    };
    auto tile_goff = [&](int pt) {
        int b, oh0, ow0;
        tile_geom(pt, b, oh0, ow0);
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int e = tid + 256 * k;
            const int p = e / (kWSC / 4), c4 = e - p * (kWSC / 4);
            const int r = p / kWPW, col = p - r * kWPW;
            const int img = NIMG == 1 ? 0 : col / 10;
            const int ih = oh0 + r - 1, iw = NIMG == 1 ? ow0 + col - 1 : col - img * 10 - 1;
            const bool ok = (e < NV) & ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W) & (b + img < a.B);
            goff[k] = ok ? ((((b + img) * a.H + ih) * a.W + iw) * a.Cin + 4 * c4) * 4 : kOOB;
        }
    };
    f32x4 pv[NIT];
    auto fetch = [&](int k, int slab) {
        pv[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, goff[k] == kOOB ? kOOB : goff[k] + slab * kWSC * 4, 0, 0));
    };

    // weight stream of this wave: [n-tile][wave xi][slab][g][nu][lane][4]
    const f32x4* wbase = a.wp + (long long)(nt * 4 + wave) * nslab * kWNG * 4 * 64 + lane;
    const int a1 = wave == 0 ? 0 : (wave == 2 ? 2 : 1), a2 = wave == 3 ? 3 : (wave == 2 ? 1 : 2);
    const float sg = wave == 1 ? 1.f : -1.f;
    auto row_tile = [](int i) {
        return NIMG == 1 ? i : (i & ~4) | ((((i >> 2) ^ (((i >> 3) + 1) >> 1)) & 1) << 2);
    };
    const int tl = row_tile(li);
    const int ti = tl >> 3, tj = tl & 7;
    const float* pbase = lds + (2 * ti) * kWRP + (NIMG == 1 ? tj : (tj >> 2) * 10 + (tj & 3)) * kWCS + 4 * lq;
    const float* p1 = pbase + a1 * kWRP;
    const float* p2 = pbase + a2 * kWRP;

    f32x16 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    f32x4 bw[2][4];
#pragma unroll
    for (int n = 0; n < 4; ++n) bw[0][n] = wbase[n * 64];

    // first stage's patch
    tile_goff(pt0);
#pragma unroll
    for (int k = 0; k < NIT; ++k) fetch(k, 0);

    // output-phase constants
    const int otile = tid >> 3, oq = tid & 7;
    const int otl = row_tile(otile);
    constexpr int kOOBo = (int)0x80000000;
    const int ybytes = a.B * a.H * a.W * a.Cout * 4;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.res ? a.res : a.y), 0, ybytes, 0x00020000);
    const float lo = a.relu ? 0.f : -INFINITY;
    const int n_out = nt * 32 + 4 * oq;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) bv = *(const f32x4*)(a.bias + n_out);

    int stage = 0;
    for (int k_tile = 0; k_tile < ntl; ++k_tile) {
        for (int s = 0; s < nslab; ++s, ++stage) {
            if (stage) __syncthreads();                  // the previous stage's readers (MFMA operands or the output exchange) are done with the LDS buffer
#pragma unroll
            for (int k = 0; k < NIT; ++k)
                if (NV % 256 == 0 || k < NIT - 1 || tid + 256 * k < NV) *(f32x4*)&lds[loff[k]] = pv[k];
            __syncthreads();
            const bool has_next = stage + 1 < nstage;
            const bool next_new_tile = s + 1 == nslab;
            const int ns = next_new_tile ? 0 : s + 1;
            if (has_next && next_new_tile) tile_goff(pt0 + k_tile + 1);

            auto xform = [&](int g, f32x4 (&v)[4]) {
                f32x4 t[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int co = ((c & 1) * (NIMG == 1 ? 9 : 5) + (c >> 1)) * kWCS;
                    const f32x4 d1 = *(const f32x4*)(p1 + co + 8 * g);
                    const f32x4 d2 = *(const f32x4*)(p2 + co + 8 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[c][e] = __builtin_fmaf(sg, d2[e], d1[e]);
                }
                v[0] = t[0] - t[2];
                v[1] = t[1] + t[2];
                v[2] = t[2] - t[1];
                v[3] = t[1] - t[3];
            };
            f32x4 v[2][4];
            if (PIPE) xform(0, v[0]);
            const f32x4* ws = wbase + (long long)s * kWNG * 4 * 64;
            const f32x4* wn = wbase + (long long)ns * kWNG * 4 * 64;
#pragma clang loop unroll(full)
            for (int g = 0; g < kWNG; ++g) {
                if (g + 1 < kWNG) {
#pragma unroll
                    for (int n = 0; n < 4; ++n) bw[(g + 1) & 1][n] = ws[((g + 1) * 4 + n) * 64];
                } else if (has_next) {
#pragma unroll
                    for (int n = 0; n < 4; ++n) bw[(g + 1) & 1][n] = wn[n * 64];
                }
                // next stage's patch: NIT requests spread over the first k-groups, AFTER this group's weight requests
                if (has_next) {
                    constexpr int PER = (NIT + 5) / 6;           // 2 per group for NIT = 12 / 13: done by group 5 or 6
#pragma unroll
                    for (int k = g * PER; k < (g + 1) * PER && k < NIT; ++k) fetch(k, ns);
                }
                if (PIPE) {
                    if (g + 1 < kWNG) xform(g + 1, v[(g + 1) & 1]);
                } else {
                    xform(g, v[g & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[g & 1][n][u], bw[g & 1][n][u], acc[n], 0, 0, 0);
                if (!PIPE) __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- output transform of this pixel tile (as k_conv_winograd) ----
        int b, oh0, ow0;
        tile_geom(pt0 + k_tile, b, oh0, ow0);
        const int oh = oh0 + 2 * (otl >> 3), ow = NIMG == 1 ? ow0 + 2 * (otl & 7) : 2 * (otl & 3);
        const int bi = b + (NIMG == 1 ? 0 : (otl >> 2) & 1);
        int off[2][2];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx)
                off[dy][dx] = (oh + dy < a.H && ow + dx < a.W && bi < a.B) ? ((((bi * a.H + oh + dy) * a.W + ow + dx) * a.Cout) + n_out) * 4 : kOOBo;
        f32x4 rv[2][2];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx)
                rv[dy][dx] = a.res ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, off[dy][dx], 0, 0)) : f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int tile = (r & 3) + 8 * (r >> 2) + 4 * lq;
            const float m0 = acc[0][r], m1 = acc[1][r], m2 = acc[2][r], m3 = acc[3][r];
            lds[((wave * 2 + 0) * 32 + tile) * 32 + li] = m0 + m1 + m2;
            lds[((wave * 2 + 1) * 32 + tile) * 32 + li] = m1 - m2 - m3;
        }
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
        __syncthreads();
        f32x4 pm[4][2];
#pragma unroll
        for (int xi = 0; xi < 4; ++xi)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) pm[xi][dx] = *(const f32x4*)&lds[((xi * 2 + dx) * 32 + otile) * 32 + 4 * oq];
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            f32x4 y0 = pm[0][dx] + pm[1][dx] + pm[2][dx] + bv + rv[0][dx];
            f32x4 y1 = pm[1][dx] - pm[2][dx] - pm[3][dx] + bv + rv[1][dx];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y0[e] = fmaxf(y0[e], lo);
                y1[e] = fmaxf(y1[e], lo);
            }
            typedef __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int u32x4;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y0), ry, off[0][dx], 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y1), ry, off[1][dx], 0, 0);
        }
    }
}
#endif  // HMF_EXPERIMENTS

}  // namespace hmf

using namespace hmf;

// w_packed: [Cout/32][4 xi][Cin/64][8 g][4 nu][64 lanes = q*32 + j][4 u] = U[32 t + j][slab*64 + 8 g + 4 q + u][xi][nu],  U = G g G^T
extern "C" int hmf_conv_winograd_pack_host(const float* w /*[Cout][3][3][Cin]*/, float* packed, int Cin, int Cout) {
    if (!w || !packed) return fail(HMF_ERR_INVALID, "hmf_conv_winograd_pack_host: null pointer");
    if (Cin % kWSC || Cout % 32) return fail(HMF_ERR_UNSUPPORTED, "hmf_conv_winograd_pack_host: Cin=%d %% 64 or Cout=%d %% 32", Cin, Cout);
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const int nslab = Cin / kWSC;
    long long o = 0;
    for (int t = 0; t < Cout / 32; ++t)
        for (int xi = 0; xi < 4; ++xi)
            for (int s = 0; s < nslab; ++s)
                for (int g = 0; g < kWNG; ++g)
                    for (int nu = 0; nu < 4; ++nu)
                        for (int l = 0; l < 64; ++l)
                            for (int u = 0; u < 4; ++u) {
                                const int n = 32 * t + (l & 31), c = s * kWSC + 8 * g + 4 * (l >> 5) + u;
                                double acc = 0.0;
                                for (int p = 0; p < 3; ++p)
                                    for (int q = 0; q < 3; ++q) acc += G[xi][p] * G[nu][q] * (double)w[(((long long)n * 3 + p) * 3 + q) * Cin + c];
                                packed[o++] = (float)acc;
                            }
    return HMF_OK;
}

// The same packing on the device, from a parameter's NATIVE (out, in, 3, 3) storage (training: the filters change every optimiser step).
// dgrad = 0: the convolution's own filter f[n][p][q][c] = w[n][c][p][q]  (Cout = out, Cin = in);
// dgrad = 1: the input-gradient filter (flipped, in/out swapped) f[n][p][q][c] = w[c][n][2-p][2-q]  (Cout = in, Cin = out of the parameter).
namespace hmf {
constexpr int kWPackJobs = 48;
struct WinoPackJobs {
    const float* w[kWPackJobs];
    float* packed[kWPackJobs];
    int cin[kWPackJobs], cout[kWPackJobs], dgrad[kWPackJobs];
    int first[kWPackJobs + 1];                  // first wave of job j; first[n] = total waves
    int n;
};

// One wave per (job, n-tile t, slab s, k-group g): lane l = q * 32 + j transforms the four channels c0 .. c0 + 3 = slab * 64 + 8 g + 4 q + u of filter
// n = 32 t + j -- 36 floats read once (contiguous in the parameter for dgrad = 0), U = G g G^T in registers, sixteen 16-byte stores, each a 1 KiB run
// per wave.  (Round 2 used one thread per packed float: nine scattered reads each, 16x the read traffic; 46-72 us per layer4 filter.)
__global__ __launch_bounds__(256) void k_wino_pack_device(const WinoPackJobs jobs) {
    const int wv = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6), l = threadIdx.x & 63;
    if (wv >= jobs.first[jobs.n]) return;
    int j = 0;
    while (wv >= jobs.first[j + 1]) ++j;
    const int Cin = jobs.cin[j], Cout = jobs.cout[j], dgrad = jobs.dgrad[j];
    const float* __restrict__ w = jobs.w[j];
    const int nslab = Cin / kWSC;
    int r = wv - jobs.first[j];
    const int g = r & 7; r >>= 3;
    const int s = r % nslab, t = r / nslab;
    const int n = 32 * t + (l & 31), c0 = s * kWSC + 8 * g + 4 * (l >> 5);
    float f[4][3][3];                           // [u][p][q]
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = 0; q < 3; ++q)
                f[u][p][q] = dgrad ? w[(((long long)(c0 + u) * Cout + n) * 3 + (2 - p)) * 3 + (2 - q)] : w[(((long long)n * Cin + c0 + u) * 3 + p) * 3 + q];
    f32x4* out = (f32x4*)jobs.packed[j] + ((((long long)(t * 4) * nslab + s) * 8 + g) * 4) * 64 + l;      // + ((xi * nslab * 8) * 4 + nu) * 64
    const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
#pragma unroll
    for (int xi = 0; xi < 4; ++xi)
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
            f32x4 v;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float acc = 0.f;                // the same product order as the round-2 packer (and the host packer up to fp64 -> fp32)
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int q = 0; q < 3; ++q) acc = fmaf(G[xi][p] * G[nu][q], f[u][p][q], acc);
                v[u] = acc;
            }
            out[((long long)xi * nslab * 8 * 4 + nu) * 64] = v;
        }
}
}  // namespace hmf

// n filters in ONE launch (the training graph packs every Winograd layer's forward and input-gradient filter at the top of the step).
extern "C" int hmf_conv_winograd_pack_device_batch(const float* const* w_native, float* const* packed, const int* cin, const int* cout, const int* dgrad, int n,
                                                   void* stream) {
    if (!w_native || !packed || !cin || !cout || !dgrad || n < 1) return fail(HMF_ERR_INVALID, "hmf_conv_winograd_pack_device_batch: bad args");
    for (int base = 0; base < n; base += kWPackJobs) {
        WinoPackJobs jobs;
        jobs.n = (n - base) < kWPackJobs ? (n - base) : kWPackJobs;
        int waves = 0;
        for (int j = 0; j < jobs.n; ++j) {
            const int k = base + j;
            if (!w_native[k] || !packed[k]) return fail(HMF_ERR_INVALID, "hmf_conv_winograd_pack_device_batch: null pointer in job %d", k);
            if (cin[k] % kWSC || cout[k] % 32 || cin[k] <= 0 || cout[k] <= 0)
                return fail(HMF_ERR_UNSUPPORTED, "hmf_conv_winograd_pack_device_batch: job %d: Cin=%d %% 64 or Cout=%d %% 32", k, cin[k], cout[k]);
            jobs.w[j] = w_native[k]; jobs.packed[j] = packed[k]; jobs.cin[j] = cin[k]; jobs.cout[j] = cout[k]; jobs.dgrad[j] = dgrad[k] ? 1 : 0;
            jobs.first[j] = waves;
            waves += (cout[k] / 32) * (cin[k] / kWSC) * 8;
        }
        jobs.first[jobs.n] = waves;
        hipLaunchKernelGGL(k_wino_pack_device, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, jobs);
        HMF_LAUNCH_CHECK();
    }
    return HMF_OK;
}

extern "C" int hmf_conv_winograd_pack_device(const float* w_native, float* packed, int Cin, int Cout, int dgrad, void* stream) {
    if (!w_native || !packed) return fail(HMF_ERR_INVALID, "hmf_conv_winograd_pack_device: null pointer");
    return hmf_conv_winograd_pack_device_batch(&w_native, &packed, &Cin, &Cout, &dgrad, 1, stream);
}

extern "C" long long hmf_conv_winograd_packed_floats(int Cin, int Cout) { return (long long)Cout * 16 * Cin; }

static int wino_ksplit(long long blocks, int Cin) { return small_launch_ksplit(blocks, Cin / kWSC); }

static long long wino_blocks(int B, int H, int W, int Cout) {
    const bool pair = W <= 8;
    const int tiles_w = pair ? 1 : (W + kWTW - 1) / kWTW;
    return (long long)(pair ? (B + 1) / 2 : B) * tiles_w * ((H + kWTH - 1) / kWTH) * (Cout / 32);
}

static int conv_winograd_impl(const float* x, const float* w_packed, const float* bias, const float* residual, float* y, int B, int H, int W,
                              int Cin, int Cout, int relu, int variant, void* stream, void* ws = nullptr, long long ws_bytes = 0) {
    if (!x || !w_packed || !y || B <= 0 || H <= 0 || W <= 0) return fail(HMF_ERR_INVALID, "hmf_conv_winograd: bad args");
    if (Cin % kWSC || Cout % 32) return fail(HMF_ERR_UNSUPPORTED, "hmf_conv_winograd: Cin=%d must be a multiple of 64 and Cout=%d of 32", Cin, Cout);
    if (((uintptr_t)x & 15) || ((uintptr_t)w_packed & 15) || ((uintptr_t)y & 15) || ((uintptr_t)residual & 15) || ((uintptr_t)bias & 15))
        return fail(HMF_ERR_INVALID, "hmf_conv_winograd: x, w_packed, y, residual and bias must be 16-byte aligned (16-byte buffer accesses in the output phase)");
    if ((long long)B * H * W * Cout >= (1LL << 29) || (long long)B * H * W * Cin >= (1LL << 29))
        return fail(HMF_ERR_UNSUPPORTED, "hmf_conv_winograd: tensor exceeds 2^29 elements (32-bit byte offsets of the buffer loads)");
    WinoArgs a;
    a.x = x; a.wp = (const f32x4*)w_packed; a.bias = bias; a.res = residual; a.y = y;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.relu = relu;
    const bool pair = W <= 8;                        // narrow maps: two images per work-group tile
    a.tiles_w = pair ? 1 : (W + kWTW - 1) / kWTW;
    a.tiles_hw = a.tiles_w * ((H + kWTH - 1) / kWTH);
    // (a 64-channel work-group tile -- two n-tiles per wave -- was measured too: it needs all 256 VGPRs, spills, and loses on every layer but the
    // 128-channel ones (57 vs 62 us); removed)
    a.n_tiles = Cout / 32;
    const long long blocks = (long long)(pair ? (B + 1) / 2 : B) * a.tiles_hw * a.n_tiles;
    if (blocks >= (1LL << 31)) return fail(HMF_ERR_UNSUPPORTED, "hmf_conv_winograd: too many tiles");
    a.xb = xcd_n_split(4LL * B * H * W * Cin, 64LL * Cin * Cout, (int)(blocks / a.n_tiles), a.n_tiles);
    a.ksplit = 1;
    a.part = nullptr;
    const long long out_bytes = 4LL * B * H * W * Cout;
    if (variant == 0 && ws) {
        const int ks = wino_ksplit(blocks, Cin);
        if (ks > 1) {
            if (ws_bytes < ks * out_bytes || ((uintptr_t)ws & 15)) return fail(HMF_ERR_INVALID, "hmf_conv_winograd_ws: workspace smaller than hmf_conv_winograd_workspace_bytes or unaligned");
            a.ksplit = ks;
            a.part = (float*)ws;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(11, B * H * W, st);
#define HMF_WINO_LAUNCH(MODE, PIPE, OCC)                                                                                   \
    do {                                                                                                                   \
        if (pair) hipLaunchKernelGGL((k_conv_winograd<1, 2, MODE, PIPE, 2>), dim3((unsigned)blocks), dim3(256), 0, st, a); \
        else hipLaunchKernelGGL((k_conv_winograd<1, 1, MODE, PIPE, OCC>), dim3((unsigned)blocks), dim3(256), 0, st, a);    \
    } while (0)
#ifdef HMF_EXPERIMENTS
    const int ptiles = (pair ? (B + 1) / 2 : B) * a.tiles_hw;
#endif
#define HMF_WINO_PF(NIMG, PIPE, OCC, TPW)                                                                                                     \
    hipLaunchKernelGGL((k_conv_winograd_pf<NIMG, PIPE, OCC, TPW>), dim3((unsigned)(((ptiles + TPW - 1) / TPW) * a.n_tiles)), dim3(256), 0, st, a, ptiles)
    switch (variant) {
        case 0:
            // Measured round-robin, median of 7 (tools/winograd_ablate.py, B = 32): with one or two 64-channel slabs (layer1 / layer2) a work-group is
            // mostly its first patch fetch and its output phase, so a third work-group per CU and the transform of group g + 1 issued under the MFMAs of
            // group g pay (52.4 -> 47.7 us, 51.3 -> 44.6 us); with 4+ slabs they do not (layer3 43.2 vs 44.3 us) and on 8-pixel maps (one wave per
            // SIMD) the pipelined transform loses outright (49.3 vs 62.7 us).
            // Round 4, measured and rejected (tools/winograd_ablate.py --pf; bitwise the same results): the stage-pipelined k_conv_winograd_pf -- the next
            // slab's / next pixel tile's patch requested under the current MFMA loop -- is no faster where it fits the register file (layer3 45.2 vs 45.5 us,
            // layer4 52.5 vs 51.1, layer1 at two waves per SIMD 49.6 vs 47.0) and spills at three waves per SIMD (+48 prefetch registers on 163): the
            // co-resident work-groups already hide the fetch; the kernel stays in the experiments build.
            // Round 4, later (tools/winograd_ablate.py --dma; bitwise the same results): the slab patch by LDS-DMA into the same LDS image (template flag
            // DMA: no staging registers, no ds_write) is no faster by itself (layer1-4: 48.5 / 45.5 / 44.2 / 51.0 vs 47.3 / 44.7 / 44.0 / 51.1 us) but frees
            // the 72 VGPRs that made the 64-channel work-group spill; that form (TN = 2: half the input transforms per MFMA, 252 VGPRs, no spill) wins
            // only on layer2 alone (42.4 vs 44.7 us) and leaves the three-stream step where it was (2.90-2.93 ms for every mix of layers against
            // 2.91): the transform VALU is not what the step waits for.  Both stay in the experiments build (variants 500 / 540 / 541).
            if (!pair && Cin <= 128) hipLaunchKernelGGL((k_conv_winograd<1, 1, 0, 1, 3>), dim3((unsigned)blocks, a.ksplit), dim3(256), 0, st, a);
            else if (pair) hipLaunchKernelGGL((k_conv_winograd<1, 2, 0, 0, 2>), dim3((unsigned)blocks, a.ksplit), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((k_conv_winograd<1, 1, 0, 0, 2>), dim3((unsigned)blocks, a.ksplit), dim3(256), 0, st, a);
            if (a.ksplit > 1) {
                HMF_LAUNCH_CHECK();
                if (int rc = splitk_finish(a.part, a.ksplit, out_bytes / 4, bias, residual, y, Cout, relu, st)) return rc;
            }
            break;
#ifdef HMF_EXPERIMENTS
        case 300:           // = the product dispatch, for A/B against the stage-pipelined kernel
            if (!pair && Cin <= 128) hipLaunchKernelGGL((k_conv_winograd<1, 1, 0, 1, 3>), dim3((unsigned)blocks), dim3(256), 0, st, a);
            else HMF_WINO_LAUNCH(0, 0, 2);
            break;
        case 301: if (pair) HMF_WINO_PF(2, 0, 2, 1); else if (Cin <= 128) HMF_WINO_PF(1, 1, 3, 2); else HMF_WINO_PF(1, 0, 2, 1); break;
        case 311: if (pair) HMF_WINO_PF(2, 0, 2, 1); else HMF_WINO_PF(1, 1, 3, 1); break;      // stage-pipelined variants: PIPE / OCC / tiles per work-group
        case 312: if (pair) HMF_WINO_PF(2, 0, 2, 2); else HMF_WINO_PF(1, 1, 3, 2); break;
        case 314: if (pair) HMF_WINO_PF(2, 0, 2, 2); else HMF_WINO_PF(1, 1, 3, 4); break;
        case 321: if (pair) HMF_WINO_PF(2, 0, 2, 1); else HMF_WINO_PF(1, 0, 2, 1); break;
        case 322: if (pair) HMF_WINO_PF(2, 0, 2, 2); else HMF_WINO_PF(1, 0, 2, 2); break;
        case 332: if (pair) HMF_WINO_PF(2, 1, 2, 2); else HMF_WINO_PF(1, 1, 2, 2); break;
        case 100: HMF_WINO_LAUNCH(0, 0, 2); break;          // tools: the round-2 configuration on every shape
        case 1: HMF_WINO_LAUNCH(1, 0, 2); break;       // 1..3, 10.., 20..: tools/winograd_ablate.py only
        case 2: HMF_WINO_LAUNCH(2, 0, 2); break;
        case 3: HMF_WINO_LAUNCH(3, 0, 2); break;
        case 10: HMF_WINO_LAUNCH(0, 0, 3); break;
        case 13: HMF_WINO_LAUNCH(3, 0, 3); break;
        case 20: HMF_WINO_LAUNCH(0, 1, 2); break;
        case 30: HMF_WINO_LAUNCH(0, 1, 3); break;
        case 40:            // two n-tiles per wave (64 output channels per work-group): half the patch fetches and input transforms per MFMA
        case 41:
            if (pair || Cout % 64) return fail(HMF_ERR_INVALID, "hmf_conv_winograd_variant: variant 40/41 needs Cout %% 64 == 0 and maps wider than 8");
            a.n_tiles = Cout / 64;
            a.xb = 1;
            if (variant == 40) hipLaunchKernelGGL((k_conv_winograd<2, 1, 0, 0, 2>), dim3((unsigned)(blocks / 2)), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((k_conv_winograd<2, 1, 0, 1, 2>), dim3((unsigned)(blocks / 2)), dim3(256), 0, st, a);
            break;
        case 500:           // LDS-DMA patch fetch, otherwise the product configuration
            if (!pair && Cin <= 128) hipLaunchKernelGGL((k_conv_winograd<1, 1, 0, 1, 3, true>), dim3((unsigned)blocks), dim3(256), 0, st, a);
            else if (pair) hipLaunchKernelGGL((k_conv_winograd<1, 2, 0, 0, 2, true>), dim3((unsigned)blocks), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((k_conv_winograd<1, 1, 0, 0, 2, true>), dim3((unsigned)blocks), dim3(256), 0, st, a);
            break;
        case 540:           // LDS-DMA + two n-tiles per wave (64 output channels per work-group), fenced / pipelined transform
        case 541:
            if (pair || Cout % 64) return fail(HMF_ERR_INVALID, "hmf_conv_winograd_variant: variant 540/541 needs Cout %% 64 == 0 and maps wider than 8");
            a.n_tiles = Cout / 64;
            a.xb = 1;
            if (variant == 540) hipLaunchKernelGGL((k_conv_winograd<2, 1, 0, 0, 2, true>), dim3((unsigned)(blocks / 2)), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((k_conv_winograd<2, 1, 0, 1, 2, true>), dim3((unsigned)(blocks / 2)), dim3(256), 0, st, a);
            break;
        case 22: HMF_WINO_LAUNCH(0, 2, 2); break;
        case 4: HMF_WINO_LAUNCH(4, 0, 2); break;
        case 8: HMF_WINO_LAUNCH(8, 0, 2); break;
        case 11: HMF_WINO_LAUNCH(11, 0, 2); break;
#endif
        default: return fail(HMF_ERR_INVALID, "hmf_conv_winograd_variant: unknown variant %d", variant);
    }
#undef HMF_WINO_LAUNCH
#undef HMF_WINO_PF
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

extern "C" long long hmf_conv_winograd_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % kWSC || Cout % 32) return 0;
    const int ks = wino_ksplit(wino_blocks(B, H, W, Cout), Cin);
    return ks > 1 ? ks * 4LL * B * H * W * Cout : 0;
}

extern "C" int hmf_conv_winograd_ws(const float* x, const float* w_packed, const float* bias, const float* residual, float* y, int B, int H, int W,
                                    int Cin, int Cout, int relu, void* ws, long long ws_bytes, void* stream) {
    return conv_winograd_impl(x, w_packed, bias, residual, y, B, H, W, Cin, Cout, relu, 0, stream, ws, ws_bytes);
}

extern "C" int hmf_conv_winograd(const float* x, const float* w_packed, const float* bias, const float* residual, float* y, int B, int H, int W,
                                 int Cin, int Cout, int relu, void* stream) {
    return conv_winograd_impl(x, w_packed, bias, residual, y, B, H, W, Cin, Cout, relu, 0, stream);
}

#ifdef HMF_EXPERIMENTS
extern "C" int hmf_debug_wino_stamps(long long* out_host, int blocks) {
    if (!out_host || blocks < 1 || blocks > 4096) return fail(HMF_ERR_INVALID, "hmf_debug_wino_stamps: bad args");
    HMF_HIP(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_wino_stamps), (size_t)blocks * 8 * sizeof(long long)));
    return HMF_OK;
}

// timing experiments (tools/winograd_ablate.py); a variant other than 0 / 10 / 20 computes wrong results
extern "C" int hmf_conv_winograd_variant(const float* x, const float* w_packed, const float* bias, const float* residual, float* y, int B, int H, int W,
                                         int Cin, int Cout, int relu, int variant, void* stream) {
    return conv_winograd_impl(x, w_packed, bias, residual, y, B, H, W, Cin, Cout, relu, variant, stream);
}
#endif
