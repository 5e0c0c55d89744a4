// Weight gradient of the stride-1 3x3 convolutions of the encoder's training graph (train/train_humaniflow.py:351 reaches it through
// loss.backward(); BASELINE configs[4]) as Winograd F(3x3, 2x2):
//
//   dW[co][ci][kh][kw] = sum_{b, oh, ow} gy[b][oh][ow][co] * x[b][oh + kh - 1][ow + kw - 1][ci]
//
// Per 2 x 2 tile of the output gradient (and the 4 x 4 input tile around it) the 3 x 3 correlation is
//   dW_tile = A^T [ (G g G^T) .* (B^T d B) ] A,     B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 -1 0 1],  G = [1 0; 1/2 1/2; 1/2 -1/2; 0 1],
//                                                     A^T = [1 1 1 0; 0 1 -1 0; 0 1 1 1]      (Toom-Cook points 0, 1, -1, inf; checked numerically)
// i.e. 16 multiplies per tile instead of 36: the sum over tiles is 16 independent GEMMs  M[xi][nu][co][ci] = sum_tiles V[xi][nu][tile][co] U[xi][nu][tile][ci]
// with 2.25x fewer MFMA k-steps than the direct form, and the inverse transform A^T M A runs ONCE over the accumulated 16 x Cout x Cin array
// (k_wgw_finish, which also writes the parameter's own (Cout, Cin, 3, 3) layout: no permute pass).  The 1/2 factors of G are applied there too.
//
// k_wgw: a work-group = 32 output channels x 64 input channels x a slice of the tiles; wave w owns transform row xi = w and its four
// columns nu, for both 32-channel halves of ci: 8 accumulators.  MFMA operands are CHANNEL-major (lane = channel, the k index of an MFMA step is
// the tile: lanes 0-31 take tile 2s, lanes 32-63 tile 2s + 1), so both operands are read from LDS as conflict-free ds_read_b32 with
// compile-time offsets and transformed in registers: per k-step 6 VALU (g) + 16 VALU (d, two halves) + 20 LDS reads for 8 MFMAs.  On gfx950 that
// VALU work adds to the MFMA time, so the expected MFMA utilisation is ~0.75 -- of 4/9 of the direct work.
// Tiles are staged 16 at a time (TR x TC tile rows x columns; two LDS stages): the gy patch (2 TR x 2 TC pixels x 32 channels) and the input patch
// ((2 TR + 2) x (2 TC + 2) pixels x 64 channels, zero outside the image) arrive by LDS-DMA with lane offsets computed once per work-group
// (an out-of-range offset makes the DMA write 0: the convolution's zero padding); pixel-major LDS rows are what lane = channel reads want, so no
// swizzle is needed.  Lanes 0-31 / 32-63 walk two disjoint runs of ADJACENT tiles, so two of the four row-transformed input columns of a k-step are
// the previous k-step's (kept in registers): per k-step 6 + 2 x 6 VALU and 4 + 2 x 4 LDS reads for 8 MFMAs.
// Split over the tiles: every work-group writes its 8 x 16 accumulator registers as a partial sum (plain 256-byte stores); k_wgw_reduce adds
// the slices into M and k_wgw_finish applies the inverse transform.  No atomics: the result is reproducible run to run.  Measured at B = 72
// (ResNet-18 layers, tools/wgw_probe.py): 109-118 us per layer vs 240-250 us for the direct kernel (134-144 us with an atomic accumulation pass).
#include "hmf_common.h"
#include <type_traits>

namespace hmf {

struct WgwArgs {
    const float* gy;     // [B][H][W][Cout]
    const float* x;      // [B][H][W][Cin]
    float* m;            // partial sums [slices][tiles][32768] (k_wgw) / [16][Cout][Cin] behind them (k_wgw_reduce)
    int B, H, W, Cin, Cout, chunks_per_img, chunks_per_row, nchunks, chunks_per_slice, ci_blocks;
    int ablate;          // experiments only (tools/wgw_probe.py): 1 = no accumulation pass, 2 = float atomics instead of partial stores (wrong results)
};

typedef __attribute__((address_space(3))) void* lds_ptr_w;
static __device__ __forceinline__ void wgw_dma16(__amdgpu_buffer_rsrc_t rsrc, lds_ptr_w dst, int voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voff, 0, 0, 0);
}

// TR x TC = 16 or 32 tiles per chunk (TC = tiles per chunk row); k-step s: lanes 0-31 take tile s, lanes 32-63 tile KS + s of the chunk.
// DB: two LDS stages -- the DMA requests of chunk c + 1 are issued right after the barrier that opens chunk c and land under its MFMAs (one barrier
// per chunk).  Single-staged (round 3 first version) every chunk began with its own fetch, exposed: 1-2 us of every ~4.5 us.
template <int TR, int TC, bool DB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void k_wgw(const WgwArgs a) {
    constexpr int GH = 2 * TR, GW = 2 * TC;                 // gy patch (pixels)
    constexpr int XH = GH + 2, XW = GW + 2;                 // input patch
    constexpr int NGI = GH * GW / 8;                        // DMA requests for the gy patch: 8 pixels x 128 B each
    constexpr int NXI = (XH * XW + 3) / 4;                  // ... for the input patch: 4 pixels x 256 B each
    constexpr int XPIX = NXI * 4;
    constexpr int STAGE = GH * GW * 32 + XPIX * 64;
    __shared__ __attribute__((aligned(16))) float lds[(DB ? 2 : 1) * STAGE];
    float* sG = lds;
    float* sX = lds + GH * GW * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lq = lane >> 5;
    const int blk = blockIdx.x, cb = blk % a.ci_blocks, ob = blk / a.ci_blocks;
    const int co0 = ob * 32, ci0 = cb * 64;
    const int slice = blockIdx.y;
    const int c_begin = slice * a.chunks_per_slice, c_end = min(a.nchunks, c_begin + a.chunks_per_slice);

    // ---- lane offsets of this wave's DMA requests, relative to the chunk's first gy pixel (oh0, ow0) of image b; columns outside the image are
    // marked per request (bit 0: the lane's pixel is in patch column 0, bit 1: in the last patch column); rows outside are ranged out by the
    // per-image descriptor: a negative offset wraps to a huge unsigned one ----
    constexpr int kOOBw = 0x40000000;
    constexpr int NGW = (NGI + 3) / 4, NXW = (NXI + 3) / 4;   // requests per wave
    int gofs[NGW], xofs[NXW], xedge[NXW];
#pragma unroll
    for (int k = 0; k < NGW; ++k) {
        const int i = wave + 4 * k;                          // request index
        const int p = 8 * i + (lane >> 3);                   // patch pixel
        const int r = p / GW, c = p - r * GW;
        gofs[k] = (i < NGI) ? ((r * a.W + c) * a.Cout + co0 + 4 * (lane & 7)) * 4 : kOOBw;
    }
#pragma unroll
    for (int k = 0; k < NXW; ++k) {
        const int i = wave + 4 * k;
        const int p = 4 * i + (lane >> 4);
        const int r = p / XW, c = p - r * XW;
        const bool ok = (i < NXI) && p < XH * XW;
        xofs[k] = ok ? (((r - 1) * a.W + c - 1) * a.Cin + ci0 + 4 * (lane & 15)) * 4 : kOOBw;
        xedge[k] = ok ? ((c == 0 ? 1 : 0) | (c == XW - 1 ? 2 : 0)) : 0;
    }

    // transform constants of this wave (xi = wave): g row r[b] = c1 g[0][b] + c2 g[1][b] (G without its 1/2: applied by k_wgw_finish);
    // d row t[c] = d[a1][c] + sd d[a2][c]  (B^T rows: d0 - d2, d1 + d2, d2 - d1, d3 - d1)
    const float c1 = wave == 3 ? 0.f : 1.f, c2 = wave == 0 ? 0.f : (wave == 2 ? -1.f : 1.f);
    const int a1 = wave == 0 ? 0 : (wave == 1 ? 1 : (wave == 2 ? 2 : 3)), a2 = wave == 0 ? 2 : (wave == 1 ? 2 : 1);
    const float sd = wave == 1 ? 1.f : -1.f;
    constexpr int KS = TR * TC / 2;                                            // k-steps per chunk
    // the second lane half starts KS tiles further: KS / TC tile rows (KS >= TC) or KS tile columns (KS < TC) into the patch
    constexpr int QG = KS >= TC ? (2 * (KS / TC)) * GW : 2 * KS, QX = KS >= TC ? (2 * (KS / TC)) * XW : 2 * KS;     // pixels
    const float* pg0 = sG + lq * QG * 32 + li;
    const float* px10 = sX + (a1 * XW) * 64 + lq * QX * 64 + li;
    const float* px20 = sX + (a2 * XW) * 64 + lq * QX * 64 + li;

    f32x16 acc[4][2];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][h][r] = 0.f;

    // DMA requests of chunk ch into stage `buf`
    auto issue = [&](int ch, int buf) {
        const int b = ch / a.chunks_per_img, rem = ch - b * a.chunks_per_img;
        const int crow = rem / a.chunks_per_row, ccol = rem - crow * a.chunks_per_row;
        const int oh0 = crow * GH, ow0 = ccol * GW;
        const bool edgeL = ccol == 0, edgeR = ccol == a.chunks_per_row - 1;
        const int emask = (edgeL ? 1 : 0) | (edgeR ? 2 : 0);
        // per-image descriptors: rows above / below the image fall out of range
        const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)(a.gy + (long long)b * a.H * a.W * a.Cout), 0, a.H * a.W * a.Cout * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (long long)b * a.H * a.W * a.Cin), 0, a.H * a.W * a.Cin * 4, 0x00020000);
        const int gbase = (oh0 * a.W + ow0) * a.Cout * 4, xbase = (oh0 * a.W + ow0) * a.Cin * 4;
        float* dG = sG + buf * STAGE;
        float* dX = sX + buf * STAGE;
#pragma unroll
        for (int k = 0; k < NGW; ++k) {
            const int i = wave + 4 * k;
            if (i < NGI) wgw_dma16(rg, (lds_ptr_w)(dG + i * 256), gofs[k] + gbase);
        }
#pragma unroll
        for (int k = 0; k < NXW; ++k) {
            const int i = wave + 4 * k;
            if (i < NXI) wgw_dma16(rx, (lds_ptr_w)(dX + i * 256), (xedge[k] & emask) ? kOOBw : xofs[k] + xbase);
        }
    };
    if (DB && c_begin < c_end) issue(c_begin, 0);
    for (int ch = c_begin; ch < c_end; ++ch) {
        const int buf = DB ? ((ch - c_begin) & 1) : 0;
        if (DB) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's requests of chunk ch have landed
            __builtin_amdgcn_s_barrier();                             // ... everybody's; and every wave is done reading the other stage
            if (ch + 1 < c_end) issue(ch + 1, buf ^ 1);
        } else {
            if (ch != c_begin) __builtin_amdgcn_s_barrier();          // every wave is done reading the previous chunk
            issue(ch, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        const float* pg = pg0 + buf * STAGE;
        const float* px1 = px10 + buf * STAGE;
        const float* px2 = px20 + buf * STAGE;

        float tk[2][2];                                            // row-transformed columns 2, 3 of the previous tile = columns 0, 1 of this one
        // Raw operands of k-step s (tile s of this lane half: (tr, tc) = (s / TC, s % TC) relative to the half's first tile; compile time after
        // unrolling): 4 gy values, and per channel half the two rows x (4 or, behind a left neighbour, 2) columns of the input tile.  They are read
        // ONE STEP AHEAD of their use: the LDS round trip runs under the previous step's MFMAs instead of in front of its own (round 3 first version:
        // reads, wait, transform, MFMAs in turn -- 52 % MFMA utilisation with two waves per SIMD).
        struct Raw {
            float g[4];
            float d1[2][4], d2[2][4];
        };
        auto load = [&](Raw& R, auto S) {
            constexpr int s_ = decltype(S)::value, tr = s_ / TC, tc = s_ % TC;
            const float* g0 = pg + ((2 * tr) * GW + 2 * tc) * 32;
            R.g[0] = g0[0]; R.g[1] = g0[32]; R.g[2] = g0[GW * 32]; R.g[3] = g0[GW * 32 + 32];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float* d1 = px1 + ((2 * tr) * XW + 2 * tc) * 64 + 32 * h;
                const float* d2 = px2 + ((2 * tr) * XW + 2 * tc) * 64 + 32 * h;
#pragma unroll
                for (int c = (tc == 0 ? 0 : 2); c < 4; ++c) { R.d1[h][c] = d1[c * 64]; R.d2[h][c] = d2[c * 64]; }
            }
        };
        auto step = [&](const Raw& R, auto S) {
            constexpr int s_ = decltype(S)::value, tc = s_ % TC;
            // ---- V = G g G^T, row xi ----
            const float r0 = c1 * R.g[0] + c2 * R.g[2], r1 = c1 * R.g[1] + c2 * R.g[3];
            const float v[4] = {r0, r0 + r1, r0 - r1, r1};
            // ---- U = B^T d B, row xi, both 32-channel halves ----
            float u[2][4];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float t[4];
                if (tc == 0) {
                    t[0] = __builtin_fmaf(sd, R.d2[h][0], R.d1[h][0]);
                    t[1] = __builtin_fmaf(sd, R.d2[h][1], R.d1[h][1]);
                } else {
                    t[0] = tk[h][0];
                    t[1] = tk[h][1];
                }
                t[2] = __builtin_fmaf(sd, R.d2[h][2], R.d1[h][2]);
                t[3] = __builtin_fmaf(sd, R.d2[h][3], R.d1[h][3]);
                tk[h][0] = t[2];
                tk[h][1] = t[3];
                u[h][0] = t[0] - t[2];
                u[h][1] = t[1] + t[2];
                u[h][2] = t[2] - t[1];
                u[h][3] = t[3] - t[1];
            }
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int h = 0; h < 2; ++h) acc[n][h] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[n], u[h][n], acc[n][h], 0, 0, 0);
        };
        Raw R0, R1;
        load(R0, std::integral_constant<int, 0>{});
        auto pair = [&](auto S) {                                   // k-steps S and S + 1
            constexpr int s_ = decltype(S)::value;
            if (s_ + 1 < KS) load(R1, std::integral_constant<int, (s_ + 1 < KS ? s_ + 1 : 0)>{});
            __builtin_amdgcn_sched_barrier(0);
            step(R0, S);
            __builtin_amdgcn_sched_barrier(0);
            if (s_ + 1 < KS) {
                if (s_ + 2 < KS) load(R0, std::integral_constant<int, (s_ + 2 < KS ? s_ + 2 : 0)>{});
                __builtin_amdgcn_sched_barrier(0);
                step(R1, std::integral_constant<int, (s_ + 1 < KS ? s_ + 1 : 0)>{});
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        static_assert(KS == 8 || KS == 16, "k-steps per chunk");
        pair(std::integral_constant<int, 0>{});
        pair(std::integral_constant<int, 2>{});
        pair(std::integral_constant<int, 4>{});
        pair(std::integral_constant<int, 6>{});
        if constexpr (KS == 16) {
            pair(std::integral_constant<int, 8>{});
            pair(std::integral_constant<int, 10>{});
            pair(std::integral_constant<int, 12>{});
            pair(std::integral_constant<int, 14>{});
        }
    }

    // ---- partial sums of this work-group, plain 256-byte stores in register order: P[slice][tile][xi = wave][nu][half][r][lane] (k_wgw_reduce adds the
    // slices).  Round 3's first version added them to M with float atomics: 67 MB of atomic traffic per layer at the ~1.5 TB/s the L2 atomic units
    // sustain = 40-47 us of a 134-143 us launch (tools/wgw_probe.py, atomic pass ablated: 94-100 us). ----
    if (c_begin >= c_end) return;
#ifdef HMF_EXPERIMENTS
    if (a.ablate == 1 && acc[0][0][0] != 12345.678f) return;        // tools/wgw_probe.py: no accumulation pass at all (timing only)
    if (a.ablate == 2) {                                             // ... the round-3 first version: float atomics into one array (timing only: it is not zeroed)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float* mp = a.m + ((size_t)(wave * 4 + n) * a.Cout + co0) * a.Cin + ci0 + 32 * h + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) atomicAdd(mp + (size_t)((r & 3) + 8 * (r >> 2) + 4 * lq) * a.Cin, acc[n][h][r]);
            }
        return;
    }
#endif
    float* pp = a.m + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * (4 * 2 * 16 * 64) + lane;
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 16; ++r) pp[((n * 2 + h) * 16 + r) * 64] = acc[n][h][r];
}

// M[xi][nu][co][ci] = sum over slices of P[slice][tile][xi][nu][half][r][lane]; register r of lane (li, q) is row co0 + (r & 3) + 8 (r >> 2) + 4 q,
// column ci0 + 32 half + li of the tile's 32 x 64 block
__global__ __launch_bounds__(256) void k_wgw_reduce(const float* __restrict__ P, float* __restrict__ M, int tiles, int slices, int Cout, int Cin, int ci_blocks) {
    const int gidx = blockIdx.x * 256 + threadIdx.x;             // tile * 32768 + idx
    if (gidx >= tiles * 32768) return;
    const int t = gidx >> 15, idx = gidx & 32767;
    const size_t stride = (size_t)tiles * 32768;
    const float* p = P + (size_t)t * 32768 + idx;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 8 <= slices; k += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(k + u) * stride];
        s0 += v[0] + v[4]; s1 += v[1] + v[5]; s2 += v[2] + v[6]; s3 += v[3] + v[7];
    }
    for (; k < slices; ++k) s0 += p[(size_t)k * stride];
    const int lane = idx & 63, r = (idx >> 6) & 15, h = (idx >> 10) & 1, nu = (idx >> 11) & 3, xi = idx >> 13;
    const int co = (t / ci_blocks) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), ci = (t % ci_blocks) * 64 + 32 * h + (lane & 31);
    M[((size_t)(xi * 4 + nu) * Cout + co) * Cin + ci] = (s0 + s1) + (s2 + s3);
}

// dW[co][ci][kh][kw] = sum_{xi, nu} A^T[kh][xi] s[xi] A^T[kw][nu] s[nu] M[xi][nu][co][ci],  s = (1, 1/2, 1/2, 1): the 1/2 of G's middle rows
__global__ __launch_bounds__(256) void k_wgw_finish(const float* __restrict__ m, float* __restrict__ dw, int n) {
    const int e = blockIdx.x * 256 + threadIdx.x;                 // (co, ci) pair
    if (e >= n) return;
    float M[4][4];
#pragma unroll
    for (int xi = 0; xi < 4; ++xi)
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
            const float sc = ((xi == 1 || xi == 2) ? 0.5f : 1.f) * ((nu == 1 || nu == 2) ? 0.5f : 1.f);
            M[xi][nu] = sc * m[(size_t)(xi * 4 + nu) * n + e];
        }
    float T[3][4];                                                // A^T M: rows kh
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
        T[0][nu] = M[0][nu] + M[1][nu] + M[2][nu];
        T[1][nu] = M[1][nu] - M[2][nu];
        T[2][nu] = M[1][nu] + M[2][nu] + M[3][nu];
    }
    float* o = dw + (size_t)e * 9;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        o[kh * 3 + 0] = T[kh][0] + T[kh][1] + T[kh][2];
        o[kh * 3 + 1] = T[kh][1] - T[kh][2];
        o[kh * 3 + 2] = T[kh][1] + T[kh][2] + T[kh][3];
    }
}

}  // namespace hmf

using namespace hmf;

static int g_wgw_target_wgs = 512;
static int g_wgw_ablate = 0;
static int g_wgw_single_stage = 0;       // 1: the single-staged 32-tile-chunk form (round 3 first version), for tools/wgw_probe.py
#ifdef HMF_EXPERIMENTS
extern "C" int hmf_debug_set_wgw_target_wgs(int n) { g_wgw_target_wgs = n > 0 ? n : 512; return HMF_OK; }      // tools/wgw_probe.py
extern "C" int hmf_debug_set_wgw_single_stage(int on) { g_wgw_single_stage = on & 1; g_wgw_ablate = on >> 1; return HMF_OK; }      // bits 1..: ablate
#endif

// workspace: the work-groups' partial sums ((target work-groups + tiles) x 32 x 64 x 16 floats at most) + the 16 * Cout * Cin Winograd-domain
// accumulator behind them; dw_native = (Cout, Cin, 3, 3) row-major, the parameter's own layout
#ifdef HMF_EXPERIMENTS
constexpr int kWgwMaxWgs = 1024;            // tools/wgw_probe.py sweeps the work-group target
#else
constexpr int kWgwMaxWgs = 512;
#endif
static size_t wgw_partial_floats(int Cin, int Cout) { return (size_t)(kWgwMaxWgs + (Cout / 32) * (Cin / 64)) * 32768; }
extern "C" size_t hmf_conv_wgrad_winograd_workspace_floats(int Cin, int Cout) { return wgw_partial_floats(Cin, Cout) + (size_t)16 * Cin * Cout; }

// chunk geometry TR * 100 + TC for a map, or 0.  Double-staged (the product path): 16-tile chunks, two stages of ~36 KB = two work-groups per CU;
// single-staged: the largest chunk (32 tiles) whose rows tile the map.
static int wgw_geometry(int H, int W, int single = 0) {
    if (H < 2 || W < 8 || (H & 1)) return 0;
    if (single) {
        if (W % 32 == 0) return H % 4 == 0 ? 216 : 116;
        if (W == 16) return H % 8 == 0 ? 408 : (H % 4 == 0 ? 208 : 0);
        if (W == 8) return H % 8 == 0 ? 404 : 0;
        return 0;
    }
    if (W % 16 == 0 && H % 4 == 0) return 208;
    if (W % 32 == 0) return 116;
    if (W == 8) return H % 8 == 0 ? 404 : 0;
    return 0;
}

extern "C" int hmf_conv_wgrad_winograd_supported(int H, int W, int Cin, int Cout) {
    return (Cin % 64 == 0 && Cout % 32 == 0 && wgw_geometry(H, W) != 0) ? 1 : 0;
}

extern "C" int hmf_conv_wgrad_winograd(const float* gy, const float* x, float* dw_native, float* workspace, int B, int H, int W, int Cin, int Cout,
                                       void* stream) {
    if (!gy || !x || !dw_native || !workspace || B <= 0) return fail(HMF_ERR_INVALID, "hmf_conv_wgrad_winograd: bad args");
    if (!hmf_conv_wgrad_winograd_supported(H, W, Cin, Cout))
        return fail(HMF_ERR_UNSUPPORTED, "hmf_conv_wgrad_winograd: H=%d W=%d Cin=%d Cout=%d (W a multiple of 16 with H %% 4 == 0, or W a multiple of 32 with H even, or W = 8 with H %% 8 == 0; Cin %% 64, Cout %% 32)", H, W, Cin, Cout);
    if (((uintptr_t)gy & 15) || ((uintptr_t)x & 15)) return fail(HMF_ERR_INVALID, "hmf_conv_wgrad_winograd: gy and x must be 16-byte aligned");
    if ((long long)H * W * std::max(Cin, Cout) * 4 >= (1LL << 30)) return fail(HMF_ERR_UNSUPPORTED, "hmf_conv_wgrad_winograd: image exceeds the 32-bit offset range");
    WgwArgs a;
    a.gy = gy; a.x = x; a.m = workspace; a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.ablate = g_wgw_ablate;
    const int single = (g_wgw_single_stage && wgw_geometry(H, W, 1)) ? 1 : 0;
    const int geo = wgw_geometry(H, W, single), TR = geo / 100, TC = geo % 100;
    a.chunks_per_row = (W / 2) / TC;
    a.chunks_per_img = ((H / 2) / TR) * a.chunks_per_row;
    a.nchunks = B * a.chunks_per_img;
    a.ci_blocks = Cin / 64;
    const int tiles = (Cout / 32) * a.ci_blocks;
    int slices = std::max(1, std::min(a.nchunks, (g_wgw_target_wgs + tiles - 1) / tiles));   // ~2 work-groups per CU in one round
    a.chunks_per_slice = (a.nchunks + slices - 1) / slices;
    slices = (a.nchunks + a.chunks_per_slice - 1) / a.chunks_per_slice;
    hipStream_t st = (hipStream_t)stream;
    if ((size_t)tiles * slices * 32768 > wgw_partial_floats(Cin, Cout)) return fail(HMF_ERR_WORKSPACE, "hmf_conv_wgrad_winograd: %d x %d work-groups exceed the workspace", tiles, slices);
    float* mred = workspace + wgw_partial_floats(Cin, Cout);
    {
        ProfScope ps(13, B * H * W, st);                             // weight-gradient family (bench.py --train)
#ifdef HMF_EXPERIMENTS
        if (single) {
            switch (geo) {
                case 216: hipLaunchKernelGGL((k_wgw<2, 16, false>), dim3(tiles, slices), dim3(256), 0, st, a); break;
                case 116: hipLaunchKernelGGL((k_wgw<1, 16, false>), dim3(tiles, slices), dim3(256), 0, st, a); break;
                case 408: hipLaunchKernelGGL((k_wgw<4, 8, false>), dim3(tiles, slices), dim3(256), 0, st, a); break;
                case 208: hipLaunchKernelGGL((k_wgw<2, 8, false>), dim3(tiles, slices), dim3(256), 0, st, a); break;
                default: hipLaunchKernelGGL((k_wgw<4, 4, false>), dim3(tiles, slices), dim3(256), 0, st, a); break;
            }
        } else
#endif
        {
            switch (geo) {
                case 208: hipLaunchKernelGGL((k_wgw<2, 8, true>), dim3(tiles, slices), dim3(256), 0, st, a); break;
                case 116: hipLaunchKernelGGL((k_wgw<1, 16, true>), dim3(tiles, slices), dim3(256), 0, st, a); break;
                default: hipLaunchKernelGGL((k_wgw<4, 4, true>), dim3(tiles, slices), dim3(256), 0, st, a); break;
            }
        }
        HMF_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_wgw_reduce, dim3((tiles * 32768 + 255) / 256), dim3(256), 0, st, workspace, mred, tiles, slices, Cout, Cin, a.ci_blocks);
        HMF_LAUNCH_CHECK();
        const int n = Cin * Cout;
        hipLaunchKernelGGL(k_wgw_finish, dim3((n + 255) / 256), dim3(256), 0, st, mred, dw_native, n);
        HMF_LAUNCH_CHECK();
    }
    return HMF_OK;
}
