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
// Tiles are staged 16 or 32 at a time (TR x TC tile rows x columns): the gy patch (2 TR x 2 TC pixels x 32 channels) and the input patch
// ((2 TR + 2) x (2 TC + 2) pixels x 64 channels, zero outside the image) arrive by LDS-DMA with lane offsets computed once per work-group
// (an out-of-range offset makes the DMA write 0: the convolution's zero padding); pixel-major LDS rows are what lane = channel reads want, so no
// swizzle is needed.  Lanes 0-31 / 32-63 walk two disjoint runs of ADJACENT tiles, so two of the four row-transformed input columns of a k-step are
// the previous k-step's (kept in registers): per k-step 6 + 2 x 6 VALU and 4 + 2 x 4 LDS reads for 8 MFMAs.  Measured at B = 72 (ResNet-18
// layers, rocprof): 141-154 us per layer with 16-tile chunks and no reuse vs 246-252 us for the direct kernel.  Split over the tiles: every work-group adds its 8 x 16 accumulator registers to M with float atomics (M zeroed first).
#include "hmf_common.h"

namespace hmf {

struct WgwArgs {
    const float* gy;     // [B][H][W][Cout]
    const float* x;      // [B][H][W][Cin]
    float* m;            // [16][Cout][Cin], zeroed
    int B, H, W, Cin, Cout, chunks_per_img, chunks_per_row, nchunks, chunks_per_slice, ci_blocks;
};

typedef __attribute__((address_space(3))) void* lds_ptr_w;
static __device__ __forceinline__ void wgw_dma16(__amdgpu_buffer_rsrc_t rsrc, lds_ptr_w dst, int voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voff, 0, 0, 0);
}

// TR x TC = 16 or 32 tiles per chunk (TC = tiles per chunk row); k-step s: lanes 0-31 take tile s, lanes 32-63 tile KS + s of the chunk
template <int TR, int TC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void k_wgw(const WgwArgs a) {
    constexpr int GH = 2 * TR, GW = 2 * TC;                 // gy patch (pixels)
    constexpr int XH = GH + 2, XW = GW + 2;                 // input patch
    constexpr int NGI = GH * GW / 8;                        // DMA requests for the gy patch: 8 pixels x 128 B each
    constexpr int NXI = (XH * XW + 3) / 4;                  // ... for the input patch: 4 pixels x 256 B each
    constexpr int XPIX = NXI * 4;
    __shared__ __attribute__((aligned(16))) float lds[GH * GW * 32 + XPIX * 64];
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
    const float* pg = sG + lq * QG * 32 + li;
    const float* px1 = sX + (a1 * XW) * 64 + lq * QX * 64 + li;
    const float* px2 = sX + (a2 * XW) * 64 + lq * QX * 64 + li;

    f32x16 acc[4][2];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][h][r] = 0.f;

    for (int ch = c_begin; ch < c_end; ++ch) {
        const int b = ch / a.chunks_per_img, rem = ch - b * a.chunks_per_img;
        const int crow = rem / a.chunks_per_row, ccol = rem - crow * a.chunks_per_row;
        const int oh0 = crow * GH, ow0 = ccol * GW;
        const bool edgeL = ccol == 0, edgeR = ccol == a.chunks_per_row - 1;
        const int emask = (edgeL ? 1 : 0) | (edgeR ? 2 : 0);
        // per-image descriptors: rows above / below the image fall out of range
        const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)(a.gy + (long long)b * a.H * a.W * a.Cout), 0, a.H * a.W * a.Cout * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (long long)b * a.H * a.W * a.Cin), 0, a.H * a.W * a.Cin * 4, 0x00020000);
        const int gbase = (oh0 * a.W + ow0) * a.Cout * 4, xbase = (oh0 * a.W + ow0) * a.Cin * 4;
        if (ch != c_begin) __builtin_amdgcn_s_barrier();          // every wave is done reading the previous chunk
#pragma unroll
        for (int k = 0; k < NGW; ++k) {
            const int i = wave + 4 * k;
            if (i < NGI) wgw_dma16(rg, (lds_ptr_w)(sG + i * 256), gofs[k] + gbase);
        }
#pragma unroll
        for (int k = 0; k < NXW; ++k) {
            const int i = wave + 4 * k;
            if (i < NXI) wgw_dma16(rx, (lds_ptr_w)(sX + i * 256), (xedge[k] & emask) ? kOOBw : xofs[k] + xbase);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();

        float tk[2][2];                                            // row-transformed columns 2, 3 of the previous tile = columns 0, 1 of this one
#pragma unroll
        for (int s = 0; s < KS; ++s) {                             // tile s of this lane half: (tr, tc) = (s / TC, s % TC) relative to the half's first tile
            const int tr = s / TC, tc = s % TC;                    // compile time after unrolling (KS < TC: one row, tr = 0)
            // ---- V = G g G^T, row xi ----
            const float* g0 = pg + ((2 * tr) * GW + 2 * tc) * 32;
            const float g00 = g0[0], g01 = g0[32], g10 = g0[GW * 32], g11 = g0[GW * 32 + 32];
            const float r0 = c1 * g00 + c2 * g10, r1 = c1 * g01 + c2 * g11;
            const float v[4] = {r0, r0 + r1, r0 - r1, r1};
            // ---- U = B^T d B, row xi, both 32-channel halves ----
            float u[2][4];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float* d1 = px1 + ((2 * tr) * XW + 2 * tc) * 64 + 32 * h;
                const float* d2 = px2 + ((2 * tr) * XW + 2 * tc) * 64 + 32 * h;
                float t[4];
                if (tc == 0) {
                    t[0] = __builtin_fmaf(sd, d2[0], d1[0]);
                    t[1] = __builtin_fmaf(sd, d2[64], d1[64]);
                } else {
                    t[0] = tk[h][0];
                    t[1] = tk[h][1];
                }
                t[2] = __builtin_fmaf(sd, d2[2 * 64], d1[2 * 64]);
                t[3] = __builtin_fmaf(sd, d2[3 * 64], d1[3 * 64]);
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
        }
    }

    // ---- one atomic pass: M[xi][nu][co0 + row][ci0 + 32 h + li] (128-byte runs; register r of lane (li, q) is row (r & 3) + 8 (r >> 2) + 4 q) ----
    if (c_begin >= c_end) return;
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float* mp = a.m + ((size_t)(wave * 4 + n) * a.Cout + co0) * a.Cin + ci0 + 32 * h + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) atomicAdd(mp + (size_t)((r & 3) + 8 * (r >> 2) + 4 * lq) * a.Cin, acc[n][h][r]);
        }
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
#ifdef HMF_EXPERIMENTS
extern "C" int hmf_debug_set_wgw_target_wgs(int n) { g_wgw_target_wgs = n > 0 ? n : 512; return HMF_OK; }      // tools/wgw_probe.py
#endif

// workspace = 16 * Cout * Cin floats (the Winograd-domain accumulator); dw_native = (Cout, Cin, 3, 3) row-major, the parameter's own layout
extern "C" size_t hmf_conv_wgrad_winograd_workspace_floats(int Cin, int Cout) { return (size_t)16 * Cin * Cout; }

// chunk geometry TR * 100 + TC for a map, or 0: the largest chunk (32 tiles: fewer barriers and first-fetch waits per MFMA) whose rows tile the map
static int wgw_geometry(int H, int W) {
    if (H < 2 || W < 8 || (H & 1)) return 0;
    if (W % 32 == 0) return H % 4 == 0 ? 216 : 116;
    if (W == 16) return H % 8 == 0 ? 408 : (H % 4 == 0 ? 208 : 0);
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
        return fail(HMF_ERR_UNSUPPORTED, "hmf_conv_wgrad_winograd: H=%d W=%d Cin=%d Cout=%d (H even and W a multiple of 32, or W = 16 with H %% 4 == 0, or W = 8 with H %% 8 == 0; Cin %% 64, Cout %% 32)", H, W, Cin, Cout);
    if (((uintptr_t)gy & 15) || ((uintptr_t)x & 15)) return fail(HMF_ERR_INVALID, "hmf_conv_wgrad_winograd: gy and x must be 16-byte aligned");
    if ((long long)H * W * std::max(Cin, Cout) * 4 >= (1LL << 30)) return fail(HMF_ERR_UNSUPPORTED, "hmf_conv_wgrad_winograd: image exceeds the 32-bit offset range");
    WgwArgs a;
    a.gy = gy; a.x = x; a.m = workspace; a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    const int geo = wgw_geometry(H, W), TR = geo / 100, TC = geo % 100;
    a.chunks_per_row = (W / 2) / TC;
    a.chunks_per_img = ((H / 2) / TR) * a.chunks_per_row;
    a.nchunks = B * a.chunks_per_img;
    a.ci_blocks = Cin / 64;
    const int tiles = (Cout / 32) * a.ci_blocks;
    int slices = std::max(1, std::min(a.nchunks, (g_wgw_target_wgs + tiles - 1) / tiles));   // ~2 work-groups per CU in one round
    a.chunks_per_slice = (a.nchunks + slices - 1) / slices;
    slices = (a.nchunks + a.chunks_per_slice - 1) / a.chunks_per_slice;
    hipStream_t st = (hipStream_t)stream;
    const size_t mfloats = (size_t)16 * Cin * Cout;
    if (int zr = zero_async(workspace, mfloats * sizeof(float), st)) return zr;
    {
        ProfScope ps(13, B * H * W, st);                             // weight-gradient family (bench.py --train)
        switch (geo) {
            case 216: hipLaunchKernelGGL((k_wgw<2, 16>), dim3(tiles, slices), dim3(256), 0, st, a); break;
            case 116: hipLaunchKernelGGL((k_wgw<1, 16>), dim3(tiles, slices), dim3(256), 0, st, a); break;
            case 408: hipLaunchKernelGGL((k_wgw<4, 8>), dim3(tiles, slices), dim3(256), 0, st, a); break;
            case 208: hipLaunchKernelGGL((k_wgw<2, 8>), dim3(tiles, slices), dim3(256), 0, st, a); break;
            default: hipLaunchKernelGGL((k_wgw<4, 4>), dim3(tiles, slices), dim3(256), 0, st, a); break;
        }
        HMF_LAUNCH_CHECK();
        const int n = Cin * Cout;
        hipLaunchKernelGGL(k_wgw_finish, dim3((n + 255) / 256), dim3(256), 0, st, workspace, dw_native, n);
        HMF_LAUNCH_CHECK();
    }
    return HMF_OK;
}
