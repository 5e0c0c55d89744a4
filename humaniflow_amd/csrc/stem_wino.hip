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

constexpr int kSwC = 18, kSwN = 64;
constexpr int kSwTH = 8, kSwTW = 16;
constexpr int kSwRows = 21;
constexpr int kSwP = 42, kSwQ = 10, kSwPlane = 1024;
constexpr int kSwPairs = kSwC / 2;                 // 9 channel pairs = MFMA k-steps per product
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
static __device__ __forceinline__ void sw_dma4(__amdgpu_buffer_rsrc_t rsrc, sw_lds_ptr_t dst, int voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 4, voff, 0, 0, 0);
}

__device__ __forceinline__ constexpr int sw_off(int j, int b) { return j * kSwP + (b & 3) * kSwQ + (b >> 2); }

// the 10 horizontal combinations of one vertically combined strip t[0..8]
#define HMF_SW_HOPS(op, t)                                                                                            \
    do {                                                                                                              \
        op[0] = t[0] - t[4]; op[1] = t[2] + t[4]; op[2] = t[4] - t[2]; op[3] = t[6] - t[2]; op[4] = t[6];             \
        op[5] = t[8];        op[6] = t[1] - t[5]; op[7] = t[3] + t[5]; op[8] = t[5] - t[3]; op[9] = t[7] - t[3];      \
    } while (0)

// raw patch rows of vertical product vi of channel pair s (9 or 18 LDS reads), and their combination into the 10 MFMA A operands
template <int VH>
__device__ __forceinline__ void sw_load(const float* __restrict__ pa, int s, int vi, float (&raw)[18]) {
    const SwProd pr = sw_prod(sw_vlist(VH, vi));
    const float* p = pa + s * 2 * kSwPlane;
#pragma unroll
    for (int b = 0; b < 9; ++b) {
        raw[b] = p[sw_off(pr.ja, b)];
        if (pr.mode != 0) raw[9 + b] = p[sw_off(pr.jb, b)];
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

// main loop of one wave role: 9 channel pairs x 5 vertical x 10 horizontal products, each chained into its class accumulator
// acc[vertical class][horizontal class].  Software pipeline: the patch rows of the next vertical product are requested before the 10 MFMAs of the
// current one and combined after them; filter groups arrive kPF groups (16 MFMAs) ahead in a register ring.
template <int VH>
__device__ __forceinline__ void sw_main(const float* __restrict__ pa, const f32x4* __restrict__ pw, f32x16 (&acc)[2][4]) {
    constexpr int kPF = 4, kRing = kPF + 1, kIters = kSwPairs * 5;
    f32x4 bw[kRing];
#pragma unroll
    for (int g = 0; g < kPF; ++g) bw[g] = pw[g * 64];
    float raw[18], op[2][10];
    sw_load<VH>(pa, 0, 0, raw);
    sw_combine<VH>(0, raw, op[0]);
#pragma clang loop unroll(full)
    for (int it = 0; it < kIters; ++it) {
        const int s = it / 5, vi = it - 5 * s;
        if (it + 1 < kIters) sw_load<VH>(pa, (it + 1) / 5, (it + 1) % 5, raw);
        __builtin_amdgcn_sched_barrier(0);
        const int vc = vi < 3 ? 0 : 1;
#pragma unroll
        for (int h = 0; h < 10; ++h) {
            const int p = vi * 10 + h, g = s * kSwGroups + (p >> 2), u = p & 3;
            if (u == 0 && g + kPF < kSwTotalGroups) bw[(g + kPF) % kRing] = pw[(g + kPF) * 64];
            acc[vc][sw_prod(h).cls] = __builtin_amdgcn_mfma_f32_32x32x2f32(op[it & 1][h], bw[g % kRing][u], acc[vc][sw_prod(h).cls], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (it + 1 < kIters) sw_combine<VH>((it + 1) % 5, raw, op[(it + 1) & 1]);
    }
}

template <bool POOL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void k_stem_wino(
    const float* __restrict__ x, const f32x4* __restrict__ wp, const float* __restrict__ bias, float* __restrict__ y, int B, int H, int W,
    int OH, int OW, int tiles_w, int tiles_hw, int relu) {
    __shared__ __attribute__((aligned(16))) float patch[kSwC * kSwPlane];           // 72 KB: two work-groups per CU
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lq = lane >> 5;
    const int b = blockIdx.x / tiles_hw, trem = blockIdx.x - b * tiles_hw;
    const int oh0 = (trem / tiles_w) * kSwTH, ow0 = (trem % tiles_w) * kSwTW;
    const int ih0 = 2 * oh0 - 3, iw0 = 2 * ow0 - 3;

    // ---- input patch -> LDS by LDS-DMA: request (channel, j) fills plane floats [64 j, 64 j + 64); lane offsets computed once (16 VGPRs), the
    // channel enters through the buffer descriptor's base.  Unused slots, rows >= 21 and pixels outside the image read out of range: the DMA
    // writes the 0 the buffer load returns (= the zero padding of the convolution). ----
    constexpr int kOOBs = 0x40000000;
    int vo[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int f = 64 * j + lane, row = f / kSwP, slot = f - row * kSwP;
        const int grp = slot / kSwQ, idx = slot - grp * kSwQ, col = 4 * idx + grp;
        const bool ok = (row < kSwRows) & (slot < 4 * kSwQ) & (col <= 2 * kSwTW + 4) & ((unsigned)(ih0 + row) < (unsigned)H) & ((unsigned)(iw0 + col) < (unsigned)W);
        vo[j] = ok ? ((ih0 + row) * W + iw0 + col) * 4 : kOOBs;
    }
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const float* ximg = x + (long long)b * kSwC * H * W;
    for (int ch = 0; ch < kSwC; ++ch) {
        const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)(ximg + (long long)ch * H * W), 0, H * W * 4, 0x00020000);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if ((j & 3) != wv) continue;                              // wave-uniform: wave w issues requests w, w + 4, ... of every plane
            sw_dma4(rc, (sw_lds_ptr_t)((__attribute__((address_space(3))) float*)patch + ch * kSwPlane + 64 * j), vo[j]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- 450 products per wave ----
    const int nh = wv >> 1, vh = wv & 1;
    const int tr = li >> 3, tc = li & 7;                              // the lane's 2 x 2 tile: output rows 2 tr, 2 tr + 1, columns 2 tc, 2 tc + 1
    const float* pa = patch + lq * kSwPlane + (4 * tr) * kSwP + tc;
    const f32x4* pw = wp + (long long)(nh * 2 + vh) * kSwTotalGroups * 64 + lane;
    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
    if (vh == 0) sw_main<0>(pa, pw, acc); else sw_main<1>(pa, pw, acc);

    // ---- class sums -> the wave's share of the four outputs of every tile.  Horizontal: C0 = A0 + P + Q, C1 = A1 + P - Q per vertical class;
    // vertical: wave vh = 0 holds E = A0v and P = Pv: rows (E + P, P); wave vh = 1 holds F = A1v and Q = Qv: rows (Q, F - Q). ----
    f32x16 part[2][2];                                               // [output row][output column]
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const f32x16 c0 = c == 0 ? acc[0][0] + acc[0][2] + acc[0][3] : acc[0][1] + acc[0][2] - acc[0][3];     // vertical class 0 of this wave
        const f32x16 c1 = c == 0 ? acc[1][0] + acc[1][2] + acc[1][3] : acc[1][1] + acc[1][2] - acc[1][3];     // vertical class 1
        if (vh == 0) { part[0][c] = c0 + c1; part[1][c] = c1; }
        else         { part[0][c] = c1;      part[1][c] = c0 - c1; }
    }
    __syncthreads();                                                 // every wave is done with the patch
    float* tl = patch;                                               // [row 8][col 16][ch 64]
    const int n = 32 * nh + li;
    if (vh == 0) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = (r & 3) + 8 * (r >> 2) + 4 * lq;
                    tl[((2 * (i >> 3) + rr) * 16 + 2 * (i & 7) + c) * 64 + n] = part[rr][c][r];
                }
    }
    __syncthreads();
    if (vh == 1) {
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = (r & 3) + 8 * (r >> 2) + 4 * lq;
                    const int row = 2 * (i >> 3) + rr, col = 2 * (i & 7) + c;
                    const float v = tl[(row * 16 + col) * 64 + n] + part[rr][c][r] + bv;
                    const bool in = (oh0 + row < OH) & (ow0 + col < OW);
                    if (POOL) {
                        tl[(row * 16 + col) * 64 + n] = in ? fmaxf(v, 0.f) + 0.f : 0.f;          // + 0.f: never -0 (its bit pattern would win the max)
                    } else if (in) {
                        y[(((long long)b * OH + oh0 + row) * OW + ow0 + col) * kSwN + n] = relu ? fmaxf(v, 0.f) : v;
                    }
                }
    }
    if (!POOL) return;
    __syncthreads();
    // ---- max-pool 3x3 / stride 2 / pad 1 of the tile (as k_stem_conv<true>): windows inside the tile are stored, windows shared with a neighbour tile
    // are combined with an integer atomic max (post-ReLU values are >= +0); y is the POOLED map, zeroed by the caller ----
    const int PH = (OH - 1) / 2 + 1, PW = (OW - 1) / 2 + 1;
    const int pr0 = oh0 >> 1, pc0 = ow0 >> 1;
    for (int it = tid; it < 45 * 16; it += 256) {
        const int p = it >> 4, cq = (it & 15) * 4;
        const int a = p / 9, c = p - a * 9;
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
        float* dst = y + (((long long)b * PH + pr0 + a) * PW + pc0 + c) * kSwN + cq;
        const bool whole = (a >= 1) & (a <= 3) & (c >= 1) & (c <= 7);
        if (whole) {
            *(f32x4*)dst = m;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicMax((unsigned int*)dst + e, __float_as_uint(m[e]));
        }
    }
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

extern "C" long long hmf_stem_wino_packed_floats(void) { return 4LL * kSwTotalGroups * 64 * 4; }

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

static int stem_wino_launch(const float* x, const float* w_packed, const float* bias, float* y, int B, int C, int H, int W, int relu, bool pool, void* stream,
                            const char* who) {
    if (!x || !w_packed || !y || B <= 0 || H <= 0 || W <= 0) return fail(HMF_ERR_INVALID, "%s: bad args", who);
    if (C != kSwC) return fail(HMF_ERR_UNSUPPORTED, "%s: %d input channels (the 18-channel proxy representation is built)", who, C);
    if ((uintptr_t)w_packed & 15) return fail(HMF_ERR_INVALID, "%s: packed weights must be 16-byte aligned", who);
    if ((long long)H * W * 4 >= (1LL << 31)) return fail(HMF_ERR_UNSUPPORTED, "%s: image plane too large for a buffer descriptor", who);
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1, PH = (OH - 1) / 2 + 1, PW = (OW - 1) / 2 + 1;
    const int th = (OH + kSwTH - 1) / kSwTH, tw = (OW + kSwTW - 1) / kSwTW;
    const long long blocks = (long long)B * th * tw;
    if (blocks >= (1LL << 31)) return fail(HMF_ERR_UNSUPPORTED, "%s: too many tiles", who);
    hipStream_t st = (hipStream_t)stream;
    if (pool) {
        if (int zr = zero_async(y, (size_t)B * PH * PW * kSwN * sizeof(float), st)) return zr;       // the identity of the max over post-ReLU values
    }
    ProfScope ps(9, B, st);
    if (pool) hipLaunchKernelGGL(k_stem_wino<true>, dim3((unsigned)blocks), dim3(256), 0, st, x, (const f32x4*)w_packed, bias, y, B, H, W, OH, OW, tw, th * tw, 1);
    else hipLaunchKernelGGL(k_stem_wino<false>, dim3((unsigned)blocks), dim3(256), 0, st, x, (const f32x4*)w_packed, bias, y, B, H, W, OH, OW, tw, th * tw, relu);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

extern "C" int hmf_stem_wino(const float* x, const float* w_packed, const float* bias, float* y, int B, int C, int H, int W, int relu, void* stream) {
    return stem_wino_launch(x, w_packed, bias, y, B, C, H, W, relu, false, stream, "hmf_stem_wino");
}

extern "C" int hmf_stem_wino_pool(const float* x, const float* w_packed, const float* bias, float* y_pooled, int B, int C, int H, int W, void* stream) {
    return stem_wino_launch(x, w_packed, bias, y_pooled, B, C, H, W, 1, true, stream, "hmf_stem_wino_pool");
}
