// Stem of the ResNet encoder (models/resnet.py:141-144,203-206: conv 7x7 / stride 2 / pad 3, 18 -> 64 channels, folded BatchNorm, ReLU, max-pool)
// as a POLYPHASE fast convolution: 100 products per 2 x 2 output tile and channel instead of the 196 of the direct form, with transform
// constants 0, +-1 and 1/2 only (the arithmetic of Winograd F(2,3): no large-point conditioning loss).
//
// One dimension.  Outputs o = 2t, 2t + 1 read the strip s[j] = x[4t - 3 + j], j = 0..8:  y0 = sum_k s[k] w[k],  y1 = sum_k s[k + 2] w[k].
// A stride-2 filter never mixes sample parities, so the 7 taps split into a 4-tap stride-1 filter on the even samples and a 3-tap one on the odd
// samples.  The 3-tap one is F(2,3) (4 products); the 4-tap one is F(2,3) on its first three taps plus its last tap directly (4 + 2 products):
//     m0 = (s0 - s4) w0                m1 = (s2 + s4) (w0 + w2 + w4)/2       m2 = (s4 - s2) (w0 - w2 + w4)/2       m3 = (s2 - s6) w4
//     m4 = s6 w6       m5 = s8 w6      m6 = (s1 - s5) w1                     m7 = (s3 + s5) (w1 + w3 + w5)/2       m8 = (s5 - s3) (w1 - w3 + w5)/2
//     m9 = (s3 - s7) w5                y0 = m0 + m4 + m6 + (m1 + m7) + (m2 + m8)            y1 = (m5 - m3 - m9) + (m1 + m7) - (m2 + m8)
// 10 products for 2 outputs instead of 14.  The products fall into four classes by where they go: A0 = {0,4,6} (y0 only), A1 = {5,-3,-9} (y1 only,
// the sign moved into the sample difference), P = {1,7} (both, + +), Q = {2,8} (both, + -):  y0 = A0 + P + Q,  y1 = A1 + P - Q.
// Two dimensions: the tensor product, 100 products in 4 x 4 classes.  Because the MFMA accumulates, every product of a class is chained into ONE
// accumulator tile: there is NO per-product output transform -- the 16 class sums are combined once per work-group.
//
// GEMM view.  A work-group (4 waves) owns 8 x 16 output pixels = 32 two-by-two tiles (M = 32), N = 64 channels, and per product a K = 18
// contraction over the input channels (9 MFMA k-steps of `v_mfma_f32_32x32x2_f32`): 100 x 9 x 2 = 1,800 MFMAs instead of 3,528.  Wave (nh, vh)
// owns output channels 32 nh .. +31 and the vertical product classes {A0, P} (vh = 0) or {A1, Q} (vh = 1): 50 products, 8 class accumulators
// (128 VGPRs).  The A operand (transformed input) is formed on the fly from the input patch in LDS: per vertical product two patch rows of 9
// samples are read and combined (9 add/sub), then the 10 horizontal combinations (8 add/sub) -- 1.6 LDS reads and 1.5 VALU per MFMA, no transformed
// tile ever stored.  The B operand (transformed filters, host-packed per wave role in issue order) streams from L2 as 16-byte loads.
// Patch in LDS: columns de-interleaved mod 4 (slot = (col & 3) * 10 + col / 4) with a row pitch of 42 floats: the 32 tiles of a half-wave read
// addresses 168 tr + tc + const = banks 8 tr + tc: conflict-free.  A channel plane is 1,024 floats (21 x 42 used): 16 LDS-DMA requests.
#include "hmf_common.h"

namespace hmf {

#ifdef HMF_EXPERIMENTS           // tools/stem_probe.py --polyphase: per work-group stamps (shader clock: start, patch in LDS, main loop done, tile in LDS, end; 100 MHz counter at start / end)
__device__ long long g_sw_stamps[12 * 4096];
#define HMF_SWSTAMP(k, v)                                                                \
    do {                                                                                 \
        if (threadIdx.x == 0 && blockIdx.x < 4096) g_sw_stamps[blockIdx.x * 12 + (k)] = (long long)(v); \
    } while (0)
#else
#define HMF_SWSTAMP(k, v) \
    do {                  \
    } while (0)
#endif

constexpr int kSwC = 18, kSwN = 64;
constexpr int kSwTH = 8, kSwTW = 16;
constexpr int kSwRows = 21;
constexpr int kSwP = 42, kSwQ = 10, kSwPlane = 1024;
constexpr int kSwPairs = kSwC / 2;                 // 9 channel pairs = MFMA k-steps per product
constexpr int kSwRing = 4;                         // channel-pair slots of the patch ring in LDS
constexpr int kSwRim = 24;                         // pooled positions on the rim of a tile's 5 x 9 (shared with neighbour tiles)
// rim slot of tile-relative pooled position (a, c): top row, bottom row, left column, right column
__host__ __device__ constexpr int sw_rim_slot(int a, int c) { return a == 0 ? c : a == 4 ? 9 + c : c == 0 ? 17 + a : 20 + a; }
constexpr int kSwGroups = 13;                      // 16-byte weight groups per channel pair and wave role (50 products + 2 zero slots)
constexpr int kSwTotalGroups = kSwPairs * kSwGroups;   // 117

// 1-D product table: value = s[ja] (mode 0), s[ja] + s[jb] (mode 1) or s[ja] - s[jb] (mode -1); the y1-only products 3 and 9 carry their minus sign here
struct SwProd { int ja, jb, mode, cls; };          // cls: 0 = A0, 1 = A1, 2 = P, 3 = Q
__host__ __device__ constexpr SwProd sw_prod(int v) {
    return v == 0 ? SwProd{0, 4, -1, 0} : v == 1 ? SwProd{2, 4, 1, 2} : v == 2 ? SwProd{4, 2, -1, 3} : v == 3 ? SwProd{6, 2, -1, 1}
         : v == 4 ? SwProd{6, 6, 0, 0}  : v == 5 ? SwProd{8, 8, 0, 1} : v == 6 ? SwProd{1, 5, -1, 0} : v == 7 ? SwProd{3, 5, 1, 2}
         : v == 8 ? SwProd{5, 3, -1, 3} : SwProd{7, 3, -1, 1};
}
// vertical products of a wave role, class-major: vh = 0 -> A0 = {0,4,6}, P = {1,7};  vh = 1 -> A1 = {3,5,9}, Q = {2,8}
__host__ __device__ constexpr int sw_vlist(int vh, int i) {
    return vh == 0 ? (i == 0 ? 0 : i == 1 ? 4 : i == 2 ? 6 : i == 3 ? 1 : 7) : (i == 0 ? 3 : i == 1 ? 5 : i == 2 ? 9 : i == 3 ? 2 : 8);
}
// filter-side transform: row v of the 10 x 7 matrix (taps 0..6)
static inline double sw_gcoef(int v, int k) {
    switch (v) {
        case 0: return k == 0 ? 1.0 : 0.0;
        case 1: return (k == 0 || k == 2 || k == 4) ? 0.5 : 0.0;
        case 2: return (k == 0 || k == 4) ? 0.5 : (k == 2 ? -0.5 : 0.0);
        case 3: return k == 4 ? 1.0 : 0.0;
        case 4: return k == 6 ? 1.0 : 0.0;
        case 5: return k == 6 ? 1.0 : 0.0;
        case 6: return k == 1 ? 1.0 : 0.0;
        case 7: return (k == 1 || k == 3 || k == 5) ? 0.5 : 0.0;
        case 8: return (k == 1 || k == 5) ? 0.5 : (k == 3 ? -0.5 : 0.0);
        default: return k == 5 ? 1.0 : 0.0;
    }
}

typedef __attribute__((address_space(3))) void* sw_lds_ptr_t;
static __device__ __forceinline__ void sw_dma4(__amdgpu_buffer_rsrc_t rsrc, sw_lds_ptr_t dst, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 4, voff, soff, 0, 0);
}

__device__ __forceinline__ constexpr int sw_off(int j, int b) { return j * kSwP + (b & 3) * kSwQ + (b >> 2); }

// the 10 horizontal combinations of one vertically combined strip t[0..8]
#define HMF_SW_HOPS(op, t)                                                                                            \
    do {                                                                                                              \
        op[0] = t[0] - t[4]; op[1] = t[2] + t[4]; op[2] = t[4] - t[2]; op[3] = t[6] - t[2]; op[4] = t[6];             \
        op[5] = t[8];        op[6] = t[1] - t[5]; op[7] = t[3] + t[5]; op[8] = t[5] - t[3]; op[9] = t[7] - t[3];      \
    } while (0)

// raw patch rows of vertical product vi of a channel pair (9 or 18 LDS reads), and their combination into the 10 MFMA A operands.
// The row base goes through an opaque register: the compiler then addresses the row's samples as base + small constant (ds_read2_b32 pairs), one
// VALU add per row -- left to itself it rebuilt `lane base + large constant` for every pair (9 adds per product row instead of 2).
typedef const __attribute__((address_space(3))) float* sw_ldsf_t;
__device__ __forceinline__ sw_ldsf_t sw_opaque(sw_ldsf_t p) {
    asm volatile("" : "+v"(p));
    return p;
}
template <int VH>
__device__ __forceinline__ void sw_load(sw_ldsf_t pa, int slot, int vi, float (&raw)[18]) {
    const SwProd pr = sw_prod(sw_vlist(VH, vi));
    const sw_ldsf_t pa_ = sw_opaque(pa + slot * 2 * kSwPlane + pr.ja * kSwP);
#pragma unroll
    for (int b = 0; b < 9; ++b) raw[b] = pa_[sw_off(0, b)];
    if (pr.mode != 0) {
        const sw_ldsf_t pb_ = sw_opaque(pa + slot * 2 * kSwPlane + pr.jb * kSwP);
#pragma unroll
        for (int b = 0; b < 9; ++b) raw[9 + b] = pb_[sw_off(0, b)];
    }
}
template <int VH>
__device__ __forceinline__ void sw_combine(int vi, const float (&raw)[18], float (&op)[10]) {
    const SwProd pr = sw_prod(sw_vlist(VH, vi));
    float t[9];
#pragma unroll
    for (int b = 0; b < 9; ++b) t[b] = pr.mode == 0 ? raw[b] : (pr.mode > 0 ? raw[b] + raw[9 + b] : raw[b] - raw[9 + b]);
    HMF_SW_HOPS(op, t);
}

#define HMF_SW_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

// The input patch streams through a ring of kSwRing channel-pair slots (2 planes of 1,024 floats each): pair s + 3 is requested (LDS-DMA, this wave's
// quarter of the 32 requests) while pair s is consumed, so no phase of the work-group waits for the whole patch.
struct SwFetch {
    __amdgpu_buffer_rsrc_t rx;       // the image's 18 planes
    int plane_bytes;                 // H * W * 4
    int vo[4];                       // lane offsets of this wave's requests j = 4 jj + wave of a plane (out of range where the slot is padding / outside the image)
    int wv;                          // wave index (scalar)
    float* ring;                     // LDS
};
static __device__ __forceinline__ void sw_fetch_pair(const SwFetch& f, int s) {
    const int slot = s % kSwRing;
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
            sw_dma4(f.rx, (sw_lds_ptr_t)((__attribute__((address_space(3))) float*)f.ring + (slot * 2 + k) * kSwPlane + 64 * (4 * jj + f.wv)), f.vo[jj],
                    (2 * s + k) * f.plane_bytes);
}

// main loop of one wave role: 9 channel pairs x 5 vertical x 10 horizontal products, each chained into its class accumulator
// acc[vertical class][horizontal class].  Software pipeline inside a pair: the patch rows of the next vertical product are requested before the 10
// MFMAs of the current one and combined after them; filter groups arrive kPF groups (4 kPF MFMAs) ahead in a register ring, as buffer loads whose
// group offset is a scalar (no address arithmetic on the vector pipe).  vmcnt is in-order: by the time a wave reaches the barrier of pair s it has
// consumed filter groups requested after its share of pair s, so that share has landed; the counted wait only makes this explicit.
template <int VH>
__device__ __forceinline__ void sw_main(sw_ldsf_t pa, __amdgpu_buffer_rsrc_t rw, int wvo, const SwFetch& f, f32x16 (&acc)[2][4]) {
    constexpr int kPF = 6, kRing = kPF + 1;
    f32x4 bw[kRing];
#pragma unroll
    for (int g = 0; g < kPF; ++g) bw[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, wvo, g * 1024, 0));
#pragma unroll
    for (int s = 0; s < kSwRing - 1; ++s) sw_fetch_pair(f, s);
    float raw[18], op[2][10];
#pragma clang loop unroll(full)
    for (int s = 0; s < kSwPairs; ++s) {
        // own requests of pair s complete (everything but the requests of the up to two later pairs and the filter groups in flight) ...
        if (s + 2 < kSwPairs) HMF_SW_VMCNT(16 + (s == 0 ? 0 : kPF));
        else if (s + 1 < kSwPairs) HMF_SW_VMCNT(8 + kPF);
        else HMF_SW_VMCNT(kPF);
        __builtin_amdgcn_s_barrier();                                // ... everybody's, and every wave is past its reads of pair s - 1
        if (s + kSwRing - 1 < kSwPairs) sw_fetch_pair(f, s + kSwRing - 1);
        const int slot = s % kSwRing;
        sw_load<VH>(pa, slot, 0, raw);
        sw_combine<VH>(0, raw, op[0]);
#pragma unroll
        for (int vi = 0; vi < 5; ++vi) {
            if (vi + 1 < 5) sw_load<VH>(pa, slot, vi + 1, raw);
            __builtin_amdgcn_sched_barrier(0);
            const int vc = vi < 3 ? 0 : 1;
#pragma unroll
            for (int h = 0; h < 10; ++h) {
                const int p = vi * 10 + h, g = s * kSwGroups + (p >> 2), u = p & 3;
                if (u == 0 && g + kPF < kSwTotalGroups)
                    bw[(g + kPF) % kRing] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, wvo, (g + kPF) * 1024, 0));
                acc[vc][sw_prod(h).cls] = __builtin_amdgcn_mfma_f32_32x32x2f32(op[vi & 1][h], bw[g % kRing][u], acc[vc][sw_prod(h).cls], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (vi + 1 < 5) sw_combine<VH>(vi + 1, raw, op[(vi + 1) & 1]);
        }
    }
}

template <bool POOL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void k_stem_wino(
    const float* __restrict__ x, const f32x4* __restrict__ wp, const float* __restrict__ bias, float* __restrict__ y, float* __restrict__ halo, int B, int H,
    int W, int OH, int OW, int tiles_w, int tiles_hw, int relu) {
    // ring of 4 channel-pair slots (32 KB; after the main loop: the exchange buffer of the wave pairs) + the finished 8 x 16 x 64 tile (32 KB): two work-groups per CU
    __shared__ __attribute__((aligned(16))) float lds[kSwRing * 2 * kSwPlane + kSwTH * kSwTW * kSwN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lq = lane >> 5;
    // Work-groups are dealt to the 8 XCDs round-robin: give each XCD a contiguous run of tiles, so that the horizontally adjacent tiles that share
    // 128-byte lines of the input rows (and the vertically adjacent ones that share 5 of their 21 rows) meet in ONE L2 at the same time.
    int L = blockIdx.x;
    if ((gridDim.x & 7) == 0) L = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int b = L / tiles_hw, trem = L - b * tiles_hw;
    const int oh0 = (trem / tiles_w) * kSwTH, ow0 = (trem % tiles_w) * kSwTW;
    const int ih0 = 2 * oh0 - 3, iw0 = 2 * ow0 - 3;
    HMF_SWSTAMP(0, __builtin_readcyclecounter());
    HMF_SWSTAMP(5, wall_clock64());
    HMF_SWSTAMP(8, __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)));                    // HW_ID
    HMF_SWSTAMP(9, __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)));                   // XCC_ID

    // ---- patch requests: request j of a plane fills plane floats [64 j, 64 j + 64): row = f / 42, slot = f % 42 -> column 4 (slot % 10) + slot / 10.
    // Unused slots, rows >= 21 and pixels outside the image get an out-of-range offset: the DMA writes the 0 the buffer load returns (= the zero
    // padding of the convolution).  Wave w issues requests w, w + 4, w + 8, w + 12 of every plane: four lane offsets, computed once. ----
    constexpr int kOOBs = 0x40000000;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    SwFetch f;
    f.rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (long long)b * kSwC * H * W), 0, kSwC * H * W * 4, 0x00020000);
    f.plane_bytes = H * W * 4;
    f.wv = wv;
    f.ring = lds;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int fl = 64 * (4 * jj + wv) + lane, row = fl / kSwP, slot = fl - row * kSwP;
        const int grp = slot / kSwQ, idx = slot - grp * kSwQ, col = 4 * idx + grp;
        const bool ok = (row < kSwRows) & (slot < 4 * kSwQ) & (col <= 2 * kSwTW + 4) & ((unsigned)(ih0 + row) < (unsigned)H) & ((unsigned)(iw0 + col) < (unsigned)W);
        f.vo[jj] = ok ? ((ih0 + row) * W + iw0 + col) * 4 : kOOBs;
    }

    // ---- 450 products per wave ----
    const int nh = wv >> 1, vh = wv & 1;
    const int tr = li >> 3, tc = li & 7;                              // the lane's 2 x 2 tile: output rows 2 tr, 2 tr + 1, columns 2 tc, 2 tc + 1
    const sw_ldsf_t pa = (sw_ldsf_t)lds + lq * kSwPlane + (4 * tr) * kSwP + tc;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(wp + (long long)(nh * 2 + vh) * kSwTotalGroups * 64), 0, kSwTotalGroups * 1024, 0x00020000);
    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
    if (vh == 0) sw_main<0>(pa, rw, lane * 16, f, acc); else sw_main<1>(pa, rw, lane * 16, f, acc);
    HMF_SWSTAMP(2, __builtin_readcyclecounter());
#ifdef HMF_EXPERIMENTS
    if (threadIdx.x == 64 && blockIdx.x < 4096) g_sw_stamps[blockIdx.x * 12 + 7] = (long long)__builtin_readcyclecounter();      // wave 1 (vh = 1)
#endif

    // ---- class sums -> the wave's share of the four outputs of every tile.  Horizontal: C0 = A0 + P + Q, C1 = A1 + P - Q per vertical class;
    // vertical: wave vh = 0 holds E = A0v and P = Pv: output rows (E + P, P); wave vh = 1 holds F = A1v and Q = Qv: rows (Q, F - Q).
    // Wave vh finishes output row vh of its tiles: it hands its share of the other row to its partner through LDS (8 x 16-byte writes per lane). ----
    f32x16 mine[2], give[2];                                         // [output column]
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const f32x16 c0 = c == 0 ? acc[0][0] + acc[0][2] + acc[0][3] : acc[0][1] + acc[0][2] - acc[0][3];     // vertical class 0 of this wave
        const f32x16 c1 = c == 0 ? acc[1][0] + acc[1][2] + acc[1][3] : acc[1][1] + acc[1][2] - acc[1][3];     // vertical class 1
        if (vh == 0) { mine[c] = c0 + c1; give[c] = c1; }
        else         { mine[c] = c0 - c1; give[c] = c1; }
    }
    __builtin_amdgcn_s_barrier();                                    // every wave is done with the ring
    f32x4* xb = (f32x4*)lds;                                         // [wave][column][quad][lane]
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) xb[((wv * 2 + c) * 4 + q) * 64 + lane] = f32x4{give[c][4 * q], give[c][4 * q + 1], give[c][4 * q + 2], give[c][4 * q + 3]};
    __syncthreads();
    float* tl = lds + kSwRing * 2 * kSwPlane;                        // [row 8][col 16][ch 64]
    const int n = 32 * nh + li;
    {
        const float bv = bias ? bias[n] : 0.f;
        f32x4 got[2][4];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) got[c][q] = xb[(((wv ^ 1) * 2 + c) * 4 + q) * 64 + lane];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * lq;
                const int row = 2 * (i >> 3) + vh, col = 2 * (i & 7) + c;
                // row 0: (E + P) + Q, row 1: P + (F - Q) -- the order of the two-launch form's sums is kept (vh = 0's share first)
                const float v = (vh == 0 ? mine[c][r] + got[c][r >> 2][r & 3] : got[c][r >> 2][r & 3] + mine[c][r]) + bv;
                const bool in = (oh0 + row < OH) & (ow0 + col < OW);
                if (POOL) {
                    tl[(row * 16 + col) * 64 + n] = in ? fmaxf(v, 0.f) + 0.f : 0.f;              // + 0.f: never -0 (its bit pattern would win the max)
                } else if (in) {
                    y[(((long long)b * OH + oh0 + row) * OW + ow0 + col) * kSwN + n] = relu ? fmaxf(v, 0.f) : v;
                }
            }
    }
    if (!POOL) return;
    __syncthreads();
    HMF_SWSTAMP(3, __builtin_readcyclecounter());
    // ---- max-pool 3x3 / stride 2 / pad 1 of the tile.  The tile touches 5 x 9 pooled positions: the 3 x 7 whose windows lie inside it are final and
    // stored; the 24 on its rim also take rows / columns of a neighbour tile: their PARTIAL maxima go to the tile's halo record (plain stores) and
    // k_stem_pool_rim combines the two to four records of a rim position afterwards.  No atomics, no zero-filled map: every pooled value is written
    // exactly once.  (k_stem_conv<true> combines the rim with integer atomic maxima on a pre-zeroed map: 28 us of a 315-us launch here, mostly the
    // acknowledgement latency at the end of every work-group.) ----
    const int PH = (OH - 1) / 2 + 1, PW = (OW - 1) / 2 + 1;
    const int pr0 = oh0 >> 1, pc0 = ow0 >> 1;
    float* hrec = halo + (long long)L * (kSwRim * kSwN);
    for (int it = tid; it < 45 * 16; it += 256) {                    // item = (pooled position, 4 channels): nine 16-byte LDS reads, one 16-byte store
        const int p = it >> 4, cq = (it & 15) * 4;
        const int a = p / 9, c = p - a * 9;
        if (pr0 + a >= PH || pc0 + c >= PW) continue;
        f32x4 m = {0.f, 0.f, 0.f, 0.f};                              // window rows / columns clamped into the tile: a duplicate does not change a maximum,
#pragma unroll                                                       // and nine unconditional reads can be in flight together
        for (int dr = -1; dr <= 1; ++dr) {
            const int rr = min(max(2 * a + dr, 0), 7);
#pragma unroll
            for (int dc = -1; dc <= 1; ++dc) {
                const int cc = min(max(2 * c + dc, 0), 15);
                const f32x4 v = *(const f32x4*)&tl[(rr * 16 + cc) * 64 + cq];
                m[0] = fmaxf(m[0], v[0]); m[1] = fmaxf(m[1], v[1]); m[2] = fmaxf(m[2], v[2]); m[3] = fmaxf(m[3], v[3]);
            }
        }
        const bool whole = (a >= 1) & (a <= 3) & (c >= 1) & (c <= 7);
        float* dst = whole ? y + (((long long)b * PH + pr0 + a) * PW + pc0 + c) * kSwN + cq : hrec + sw_rim_slot(a, c) * kSwN + cq;
        *(f32x4*)dst = m;
    }
#ifdef HMF_EXPERIMENTS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    HMF_SWSTAMP(4, __builtin_readcyclecounter());
    HMF_SWSTAMP(6, wall_clock64());
#endif
}

// Rim positions of the pooled map (pooled row % 4 == 0 or pooled column % 8 == 0): the maximum over the halo records of the up to four tiles
// whose 5 x 9 ranges contain the position.  One thread per (position, 4 channels).
__global__ __launch_bounds__(256) void k_stem_pool_rim(const float* __restrict__ halo, float* __restrict__ y, int B, int PH, int PW, int tiles_h, int tiles_w,
                                                       int rows_rim, int cols_plain) {
    // enumeration per image: first the rim rows (every column), then the rim columns of the other rows
    const int per_img = rows_rim * PW + (PH - rows_rim) * cols_plain;
    const unsigned e = blockIdx.x * 256u + threadIdx.x;
    const int cq = (int)(e & 15) * 4;
    const unsigned q = e >> 4;
    if (q >= (unsigned)per_img * (unsigned)B) return;
    const int b = (int)(q / (unsigned)per_img);
    int r = (int)(q - (unsigned)b * (unsigned)per_img);
    int pr, pc;
    if (r < rows_rim * PW) {
        pr = (r / PW) * 4;
        pc = r % PW;
    } else {
        r -= rows_rim * PW;
        const int rowi = r / cols_plain;                              // index among the rows with pr % 4 != 0
        pr = rowi + rowi / 3 + 1;
        pc = (r % cols_plain) * 8;
    }
    f32x4 m = {0.f, 0.f, 0.f, 0.f};                                    // post-ReLU values are >= 0
    const int ty1 = pr >> 2, tx1 = pc >> 3;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
        const int ty = ty1 - dy, a = pr - 4 * ty;
        if (ty < 0 || ty >= tiles_h || a > 4) continue;
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int tx = tx1 - dx, c = pc - 8 * tx;
            if (tx < 0 || tx >= tiles_w || c > 8) continue;
            const f32x4 v = *(const f32x4*)&halo[(((long long)b * tiles_h + ty) * tiles_w + tx) * (kSwRim * kSwN) + sw_rim_slot(a, c) * kSwN + cq];
            m[0] = fmaxf(m[0], v[0]); m[1] = fmaxf(m[1], v[1]); m[2] = fmaxf(m[2], v[2]); m[3] = fmaxf(m[3], v[3]);
        }
    }
    *(f32x4*)&y[(((long long)b * PH + pr) * PW + pc) * kSwN + cq] = m;
}

// transformed filters of wave role (nh, vh), group g = pair * 13 + q, lane, slot u:  product p = 4 q + u -> (vertical vi = p / 10, horizontal h = p % 10)
__host__ __device__ static inline void sw_slot(long long e, int& n, int& ch, int& v, int& h, bool& live) {
    const int u = (int)(e & 3), l = (int)((e >> 2) & 63);
    const long long gg = e >> 8;
    const int g = (int)(gg % kSwTotalGroups), role = (int)(gg / kSwTotalGroups);
    const int s = g / kSwGroups, q = g - s * kSwGroups, p = 4 * q + u;
    live = p < 50;
    const int vi = live ? p / 10 : 0;
    h = live ? p - 10 * vi : 0;
    v = sw_vlist(role & 1, vi);
    n = 32 * (role >> 1) + (l & 31);
    ch = 2 * s + (l >> 5);
}

__device__ __forceinline__ float sw_g(int v, int k) {
    switch (v) {
        case 0: return k == 0 ? 1.f : 0.f;
        case 1: return (k == 0 || k == 2 || k == 4) ? 0.5f : 0.f;
        case 2: return (k == 0 || k == 4) ? 0.5f : (k == 2 ? -0.5f : 0.f);
        case 3: return k == 4 ? 1.f : 0.f;
        case 4: return k == 6 ? 1.f : 0.f;
        case 5: return k == 6 ? 1.f : 0.f;
        case 6: return k == 1 ? 1.f : 0.f;
        case 7: return (k == 1 || k == 3 || k == 5) ? 0.5f : 0.f;
        case 8: return (k == 1 || k == 5) ? 0.5f : (k == 3 ? -0.5f : 0.f);
        default: return k == 5 ? 1.f : 0.f;
    }
}

// device-side packer (training re-packs the live filter every step): one thread per packed float, <= 9 taps each, accumulated in double like the host packer
__global__ __launch_bounds__(256) void k_stem_wino_pack(const float* __restrict__ w, float* __restrict__ packed, long long total) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    int n, ch, v, h;
    bool live;
    sw_slot(e, n, ch, v, h, live);
    double a = 0.0;
    if (live) {
        const float* wf = w + ((long long)n * kSwC + ch) * 49;
        for (int kh = 0; kh < 7; ++kh) {
            const float gv = sw_g(v, kh);
            if (gv == 0.f) continue;
            for (int kw = 0; kw < 7; ++kw) {
                const float gh = sw_g(h, kw);
                if (gh != 0.f) a += (double)gv * (double)gh * (double)wf[kh * 7 + kw];
            }
        }
    }
    packed[e] = (float)a;
}

}  // namespace hmf

using namespace hmf;

#ifdef HMF_EXPERIMENTS
extern "C" int hmf_debug_stem_wino_stamps(long long* out_host, int blocks) {
    if (!out_host || blocks < 1 || blocks > 4096) return fail(HMF_ERR_INVALID, "hmf_debug_stem_wino_stamps: bad args");
    HMF_HIP(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_sw_stamps), (size_t)blocks * 12 * sizeof(long long)));
    return HMF_OK;
}
#endif

extern "C" long long hmf_stem_wino_packed_floats(void) { return 4LL * kSwTotalGroups * 64 * 4; }

extern "C" long long hmf_stem_wino_pool_workspace_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    const long long th = (OH + kSwTH - 1) / kSwTH, tw = (OW + kSwTW - 1) / kSwTW;
    return (long long)B * th * tw * kSwRim * kSwN * (long long)sizeof(float);
}

extern "C" int hmf_stem_wino_pack_host(const float* w /*64x18x7x7*/, float* packed) {
    if (!w || !packed) return fail(HMF_ERR_INVALID, "hmf_stem_wino_pack_host: null pointer");
    const long long total = hmf_stem_wino_packed_floats();
    for (long long e = 0; e < total; ++e) {
        int n, ch, v, h;
        bool live;
        sw_slot(e, n, ch, v, h, live);
        double a = 0.0;
        if (live) {
            const float* wf = w + ((long long)n * kSwC + ch) * 49;
            for (int kh = 0; kh < 7; ++kh) {
                const double gv = sw_gcoef(v, kh);
                if (gv == 0.0) continue;
                for (int kw = 0; kw < 7; ++kw) {
                    const double gh = sw_gcoef(h, kw);
                    if (gh != 0.0) a += gv * gh * (double)wf[kh * 7 + kw];
                }
            }
        }
        packed[e] = (float)a;
    }
    return HMF_OK;
}

extern "C" int hmf_stem_wino_pack_device(const float* w, float* packed, void* stream) {
    if (!w || !packed) return fail(HMF_ERR_INVALID, "hmf_stem_wino_pack_device: null pointer");
    const long long total = hmf_stem_wino_packed_floats();
    hipLaunchKernelGGL(k_stem_wino_pack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, packed, total);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

static int stem_wino_launch(const float* x, const float* w_packed, const float* bias, float* y, int B, int C, int H, int W, int relu, bool pool, void* ws,
                            long long ws_bytes, void* stream, const char* who) {
    if (!x || !w_packed || !y || B <= 0 || H <= 0 || W <= 0) return fail(HMF_ERR_INVALID, "%s: bad args", who);
    if (C != kSwC) return fail(HMF_ERR_UNSUPPORTED, "%s: %d input channels (the 18-channel proxy representation is built)", who, C);
    if ((uintptr_t)w_packed & 15) return fail(HMF_ERR_INVALID, "%s: packed weights must be 16-byte aligned", who);
    if ((long long)H * W * 4 * kSwC >= (1LL << 30)) return fail(HMF_ERR_UNSUPPORTED, "%s: image too large for a buffer descriptor", who);
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1, PH = (OH - 1) / 2 + 1, PW = (OW - 1) / 2 + 1;
    const int th = (OH + kSwTH - 1) / kSwTH, tw = (OW + kSwTW - 1) / kSwTW;
    const long long blocks = (long long)B * th * tw;
    if (blocks >= (1LL << 31)) return fail(HMF_ERR_UNSUPPORTED, "%s: too many tiles", who);
    if (pool && (!ws || ws_bytes < hmf_stem_wino_pool_workspace_bytes(B, H, W) || ((uintptr_t)ws & 15)))
        return fail(HMF_ERR_INVALID, "%s: workspace missing, unaligned or smaller than hmf_stem_wino_pool_workspace_bytes", who);
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(9, B, st);
    if (pool) {
        hipLaunchKernelGGL(k_stem_wino<true>, dim3((unsigned)blocks), dim3(256), 0, st, x, (const f32x4*)w_packed, bias, y, (float*)ws, B, H, W, OH, OW, tw, th * tw, 1);
        HMF_LAUNCH_CHECK();
        const int rows_rim = (PH + 3) / 4, cols_plain = (PW + 7) / 8;                    // pooled rows with pr % 4 == 0; rim columns (pc % 8 == 0) of the other rows
        const long long n = ((long long)rows_rim * PW + (long long)(PH - rows_rim) * cols_plain) * B * 16;
        if (n >= (1LL << 31)) return fail(HMF_ERR_UNSUPPORTED, "%s: pooled map too large", who);
        hipLaunchKernelGGL(k_stem_pool_rim, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)ws, y, B, PH, PW, th, tw, rows_rim, cols_plain);
    } else {
        hipLaunchKernelGGL(k_stem_wino<false>, dim3((unsigned)blocks), dim3(256), 0, st, x, (const f32x4*)w_packed, bias, y, (float*)nullptr, B, H, W, OH, OW, tw, th * tw, relu);
    }
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

extern "C" int hmf_stem_wino(const float* x, const float* w_packed, const float* bias, float* y, int B, int C, int H, int W, int relu, void* stream) {
    return stem_wino_launch(x, w_packed, bias, y, B, C, H, W, relu, false, nullptr, 0, stream, "hmf_stem_wino");
}

extern "C" int hmf_stem_wino_pool(const float* x, const float* w_packed, const float* bias, float* y_pooled, int B, int C, int H, int W, void* ws, long long ws_bytes,
                                  void* stream) {
    return stem_wino_launch(x, w_packed, bias, y_pooled, B, C, H, W, 1, true, ws, ws_bytes, stream, "hmf_stem_wino_pool");
}
