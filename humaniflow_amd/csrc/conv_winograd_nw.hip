// EXPERIMENTS BUILD ONLY (tools/wino_nw_probe.py) -- built, bitwise equal to k_conv_winograd, measured SLOWER, not shipped; kept as the evidence.
// Winograd F(2x2, 3x3) NHWC fp32 convolution, second form (round 5): the work-group tile of csrc/conv_winograd.hip turned by 90 degrees.
//
// RESULT (MI355X, B = 32, profiles/r05_wino_nw_probe.txt): layer2 (32x32, 128 -> 128) 56.5 us vs 48.2 us for k_conv_winograd, layer3 (16x16, 256 -> 256;
// 128 work-groups on 256 CUs) 94 vs 47 us, 32x32 256 -> 256 188 vs 165 us.  With the operand stream, the patch DMA and the window reads ALL ablated
// (mode 7) the loop reaches 107 TFLOP/s of executed MFMA work (68 %) as fenced VALU bursts and 54-63 % interleaved: ONE wave per SIMD cannot issue
// its dependent VALU chains (t -> v, 128 per 64 MFMAs) faster than their ~8-cycle dependency latency, and an fp32 MFMA does not overlap them.  The
// 256-register accumulator set that removes the exchange is exactly what forbids the second wave that would fill those slots.  What this says
// about k_conv_winograd: its 47-53 % is not the patch fetch or the exchange alone -- co-resident waves are what covers the transform latency.
//
// k_conv_winograd gives a work-group 32 Winograd tiles x 32 OUTPUT CHANNELS and splits the sixteen transform positions over its four waves (wave =
// row xi).  Every 32 output channels of a layer therefore pay their own patch fetch, their own input transform and an accumulator exchange
// through LDS (two barriers) before the output transform -- on the 64- and 128-channel layers (one or two slabs of K) those phases are as long
// as the MFMA loop itself (tools/wino_probe.py: 8.2 k cycles of MFMA issue in a 20 k-cycle work-group; MfmaUtil 47-53 %, round-4 review).
//
// Here a work-group owns the same 8 x 16 output pixels but NT * 32 output channels, and a WAVE owns one n-tile with ALL sixteen positions:
//   * 16 accumulators x 16 registers = 256 accumulator registers per lane, one wave per SIMD (the 512-entry unified file), no second work-group
//     on the CU needed: the phases that used to be hidden by co-resident work-groups are overlapped inside the wave instead --
//   * the slab patch arrives by LDS-DMA into a DOUBLE buffer (2 x 49 KB): slab s + 1 lands under the MFMAs of slab s, one barrier per slab, no
//     staging registers;
//   * per k-group a lane reads its tile's 4 x 4 window once (16 ds_read_b128) and walks xi = 0..3: one fma per element for the row transform, the
//     column transform, then 16 MFMAs against the pre-transformed filters of (n-tile, xi), which stream L2 -> VGPR one xi-step (1024 MFMA cycles)
//     ahead through raw buffer loads with scalar offsets -- the same packed operand as k_conv_winograd ([n-tile][xi][slab][g][nu][lane][4]), read
//     with another index order;
//   * the output transform A^T M A is IN REGISTERS (a lane holds all sixteen positions of its (tile, channel) pairs): no accumulator exchange, no
//     barrier; lane = channel, so a half-wave stores the 128 contiguous bytes of a pixel's 32 channels.
// One patch fetch and (per wave) one window read serve NT n-tiles; per MFMA: 0.25 LDS reads, 2 VALU, 0.25 operand loads, no LDS writes.
// Same arithmetic in the same order as k_conv_winograd (row transform fma(sg, d2, d1), MFMA k order, output sums): BITWISE the same result
// (tests/test_gpu_encoder.py).  Built for maps at least 16 pixels wide (NIMG = 1 layout), Cin % 64 == 0, Cout % 128 == 0 (NT = 4).
#include "hmf_common.h"
#ifdef HMF_EXPERIMENTS

namespace hmf {

struct WinoNwArgs {
    const float* x;       // [B][H][W][Cin]
    const f32x4* wp;      // packed transformed weights (hmf_conv_winograd_pack_*)
    const float* bias;
    const float* res;
    float* y;             // [B][H][W][Cout]
    int B, H, W, Cin, Cout, relu, tiles_w, tiles_hw, n_groups;
    int xb;               // XCD split along the channel groups (hmf_common.h xcd_n_split)
    int mode;             // experiments build only (tools/wino_nw_probe.py; wrong results when != 0): & 1 no next-slab patch DMA, & 2 no operand stream, & 4 no window reads
};

constexpr int kNwTH = 8, kNwTW = 16, kNwSC = 64;
constexpr int kNwPH = kNwTH + 2, kNwPW = kNwTW + 2;
constexpr int kNwCS = kNwSC + 4;                     // LDS pixel stride (floats): the layout of k_conv_winograd<.., NIMG = 1>
constexpr int kNwNG = kNwSC / 8;                     // k-groups per slab
constexpr int kNwRP = kNwPW * kNwCS + 8;             // row pitch: 4 (mod 8) 16-byte slots -> conflict-free ds_read_b128 of the window columns
static_assert((kNwRP / 4) % 8 == 4, "row pitch must be 4 (mod 8) 16-byte slots");
constexpr int kNwSlotsRow = kNwRP / 4, kNwSlots = kNwPH * kNwSlotsRow, kNwNQ = (kNwSlots + 63) / 64;
typedef __attribute__((address_space(3))) void* nw_lds_ptr_t;

constexpr int kNwNT = 4;                             // n-tiles (= waves) per work-group
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_conv_wino_nw(const WinoNwArgs a) {
    constexpr int NT = kNwNT;
    __shared__ __attribute__((aligned(16))) float lds[2 * kNwNQ * 256];           // two patch buffers of 49 KB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lq = lane >> 5;

    const int L = xcd_tile(blockIdx.x, gridDim.x / a.n_groups, a.n_groups, a.xb);
    const int ng = L % a.n_groups, pt = L / a.n_groups;
    const int b = pt / a.tiles_hw, trem = pt - b * a.tiles_hw;
    const int oh0 = (trem / a.tiles_w) * kNwTH, ow0 = (trem % a.tiles_w) * kNwTW;
    const int ntw = ng * NT + wv;                                                  // this wave's n-tile

    // ---- patch by LDS-DMA: request q = NT k + wave fills LDS slots [64 q, 64 q + 64) of a buffer; padding slots, slots past the image and pixels
    // outside the map read out of range (the DMA writes the 0 the buffer load returns: the convolution's zero padding)
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.B * a.H * a.W * a.Cin * 4, 0x00020000);
    constexpr int kOOB = (int)0x80000000;
    constexpr int NQW = (kNwNQ + NT - 1) / NT;
    int goff[NQW];
#pragma unroll
    for (int k = 0; k < NQW; ++k) {
        const int sl = 64 * (NT * k + wv) + lane;
        const int r = sl / kNwSlotsRow, rs = sl - r * kNwSlotsRow, ps = rs / (kNwCS / 4), c4 = rs - ps * (kNwCS / 4);
        const int col = ps < 9 ? 2 * ps : 2 * (ps - 9) + 1;                        // columns de-interleaved by parity
        const int ih = oh0 + r - 1, iw = ow0 + col - 1;
        const bool ok = (r < kNwPH) & (ps < kNwPW) & (c4 < kNwSC / 4) & ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W);
        goff[k] = ok ? (((b * a.H + ih) * a.W + iw) * a.Cin + 4 * c4) * 4 : kOOB;
    }
    auto issue_patch = [&](int buf, int slab) {
#pragma unroll
        for (int k = 0; k < NQW; ++k)
            if (NT * k + wv < kNwNQ)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (nw_lds_ptr_t)((__attribute__((address_space(3))) float*)lds + (buf * kNwNQ + NT * k + wv) * 256), 16,
                                                         goff[k], slab * kNwSC * 4, 0, 0);
    };

    // ---- pre-transformed filters of (n-tile ntw, xi, slab s, k-group g, nu): 1 KiB blocks, lane l reads bytes [16 l, 16 l + 16)
    const int nslab = a.Cin / kNwSC;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, a.Cout * 16 * a.Cin * 4, 0x00020000);
    const int wlane = lane * 16;
    auto wblock = [&](int xi, int s, int g) { return (((ntw * 4 + xi) * nslab + s) * kNwNG + g) * 4 * 1024; };      // byte offset of nu = 0
    auto load_w = [&](f32x4 (&dst)[4], int soff) {
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) dst[nu] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, wlane, soff + nu * 1024, 0));
    };

    // MFMA row li of this wave = tile (ti, tj) of the 4 x 8 tiles; its 4 x 4 window starts at patch row 2 ti, column 2 tj
    const int ti = li >> 3, tj = li & 7;
    const int pwin = (2 * ti) * kNwRP + tj * kNwCS + 4 * lq;

    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    f32x4 bw[2][4];
    issue_patch(0, 0);
    load_w(bw[0], wblock(0, 0, 0));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    auto read_window = [&](f32x4 (&d)[4][4], const float* pb, int g) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) d[r][c] = *(const f32x4*)(pb + r * kNwRP + ((c & 1) * 9 + (c >> 1)) * kNwCS + 8 * g);
    };
    f32x4 d[4][4];                                                                 // the window of a k-group: [row][column]
    for (int s = 0; s < nslab; ++s) {
        const bool more = s + 1 < nslab;
        const int nbuf = (s + 1) & 1;
        const float* pb = lds + (s & 1) * (kNwNQ * 256) + pwin;
        read_window(d, pb, 0);
        for (int go = 0; go < kNwNG / 4; ++go) {
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) {
                const int g = go * 4 + gi;
                // B^T d B of the whole k-group as ONE VALU burst (128 instructions), then 64 MFMAs with no VALU between them: an fp32 MFMA and a
                // VALU instruction do not overlap on a SIMD, and with one wave per SIMD every VALU -> MFMA operand hand-off inside the MFMA stream
                // showed as pipe bubbles (interleaved: 54-63 % MFMA duty with the operand stream, the patch DMA and the window reads all ablated)
                f32x4 v[4][4];
#pragma unroll
                for (int xi = 0; xi < 4; ++xi) {
                    constexpr int A1[4] = {0, 1, 2, 1}, A2[4] = {2, 2, 1, 3};     // rows of B^T: d0 - d2, d1 + d2, d2 - d1, d1 - d3
                    const float sg = xi == 1 ? 1.f : -1.f;
                    f32x4 t[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int e = 0; e < 4; ++e) t[c][e] = __builtin_fmaf(sg, d[A2[xi]][c][e], d[A1[xi]][c][e]);
                    v[xi][0] = t[0] - t[2];
                    v[xi][1] = t[1] + t[2];
                    v[xi][2] = t[2] - t[1];
                    v[xi][3] = t[1] - t[3];
                }
                __builtin_amdgcn_sched_barrier(0);
#ifdef HMF_EXPERIMENTS
                if (!(a.mode & 4))
#endif
                if (g + 1 < kNwNG) read_window(d, pb, g + 1);                      // the next k-group's window: its LDS latency hides under this group's MFMAs
#pragma unroll
                for (int xi = 0; xi < 4; ++xi) {
                    // The next slab's patch: ONE LDS-DMA request per step over the first steps of the slab (it lands under this slab's MFMAs).  All
                    // at the slab's top they would sit in front of this step's operand loads in the in-order vmcnt queue: the first steps would
                    // wait for the whole patch.
                    const int q = gi * 4 + xi;
#ifdef HMF_EXPERIMENTS
                    if (!(a.mode & 1))
#endif
                    if (q < NQW && go == 0 && more && NT * q + wv < kNwNQ)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (nw_lds_ptr_t)((__attribute__((address_space(3))) float*)lds + (nbuf * kNwNQ + NT * q + wv) * 256),
                                                                 16, goff[q < NQW ? q : 0], (s + 1) * kNwSC * 4, 0, 0);
                    // operands of the NEXT step (xi + 1, or xi = 0 of the next k-group / slab) requested now; past the last slab the offset runs out
                    // of range for the last n-tile and the buffer load returns zeros that nothing consumes
                    const int nxi = (xi + 1) & 3, ngi = g + (xi == 3 ? 1 : 0);
#ifdef HMF_EXPERIMENTS
                    if (!(a.mode & 2))
#endif
                    load_w(bw[(xi + 1) & 1], wblock(nxi, s + (ngi >> 3), ngi & 7));
                    __builtin_amdgcn_sched_barrier(0);          // keep the requests AHEAD of this step's 16 MFMAs (left alone the scheduler sinks them behind
                                                                // the MFMAs and the next step opens with an L2 round trip)
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int nu = 0; nu < 4; ++nu)
                            acc[xi * 4 + nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[xi][nu][u], bw[xi & 1][nu][u], acc[xi * 4 + nu], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (more) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // this wave's share of the next patch has landed
            __syncthreads();                                                       // ... everyone's has, and everyone is done reading this one
        }
    }

    // ---- output transform Y = A^T M A in registers: accumulator register r of lane (li, lq) holds tile (r & 3) + 8 (r >> 2) + 4 lq, channel 32 ntw + li.
    // Same summation order as k_conv_winograd's two-step form.
    const int n = ntw * 32 + li;
    const float bv = a.bias ? a.bias[n] : 0.f;
    const float lo = a.relu ? 0.f : -INFINITY;
    const int ybytes = a.B * a.H * a.W * a.Cout * 4;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.res ? a.res : a.y), 0, ybytes, 0x00020000);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int tile = (r & 3) + 8 * (r >> 2) + 4 * lq;
        const int oh = oh0 + 2 * (tile >> 3), ow = ow0 + 2 * (tile & 7);
        int off[2][2];
        float rv[2][2];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                off[dy][dx] = (oh + dy < a.H && ow + dx < a.W) ? ((((b * a.H + oh + dy) * a.W + ow + dx) * a.Cout) + n) * 4 : kOOB;
                rv[dy][dx] = a.res ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, off[dy][dx], 0, 0)) : 0.f;
            }
        float pm[4][2];
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
            const float m0 = acc[xi * 4][r], m1 = acc[xi * 4 + 1][r], m2 = acc[xi * 4 + 2][r], m3 = acc[xi * 4 + 3][r];
            pm[xi][0] = m0 + m1 + m2;
            pm[xi][1] = m1 - m2 - m3;
        }
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const float y0 = fmaxf(pm[0][dx] + pm[1][dx] + pm[2][dx] + bv + rv[0][dx], lo);
            const float y1 = fmaxf(pm[1][dx] - pm[2][dx] - pm[3][dx] + bv + rv[1][dx], lo);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, y0), ry, off[0][dx], 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, y1), ry, off[1][dx], 0, 0);
        }
    }
}

}  // namespace hmf

using namespace hmf;

static int g_nw_mode = 0;
extern "C" int hmf_debug_set_wino_nw_mode(int mode) { g_nw_mode = mode; return HMF_OK; }

// Same contract as hmf_conv_winograd (csrc/conv_winograd.hip) and the same w_packed; returns HMF_ERR_UNSUPPORTED for shapes this form is not built
// for (maps narrower than 16 pixels, Cout not a multiple of 128): the caller keeps hmf_conv_winograd for those.
extern "C" int hmf_conv_winograd_nw(const float* x, const float* w_packed, const float* bias, const float* residual, float* y, int B, int H, int W,
                                    int Cin, int Cout, int relu, void* stream) {
    if (!x || !w_packed || !y || B <= 0 || H <= 0 || W <= 0) return fail(HMF_ERR_INVALID, "hmf_conv_winograd_nw: bad args");
    if (Cin % kNwSC || Cout % 128 || W < 16)
        return fail(HMF_ERR_UNSUPPORTED, "hmf_conv_winograd_nw: Cin=%d must be a multiple of 64, Cout=%d of 128, W=%d >= 16", Cin, Cout, W);
    if (((uintptr_t)x & 15) || ((uintptr_t)w_packed & 15) || ((uintptr_t)y & 3) || ((uintptr_t)residual & 3) || ((uintptr_t)bias & 3))
        return fail(HMF_ERR_INVALID, "hmf_conv_winograd_nw: x and w_packed must be 16-byte aligned");
    if ((long long)B * H * W * Cout >= (1LL << 29) || (long long)B * H * W * Cin >= (1LL << 29) || (long long)Cout * 16 * Cin >= (1LL << 29))
        return fail(HMF_ERR_UNSUPPORTED, "hmf_conv_winograd_nw: tensor exceeds 2^29 elements (32-bit byte offsets of the buffer accesses)");
    WinoNwArgs a;
    a.x = x; a.wp = (const f32x4*)w_packed; a.bias = bias; a.res = residual; a.y = y;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.relu = relu;
    a.tiles_w = (W + kNwTW - 1) / kNwTW;
    a.tiles_hw = a.tiles_w * ((H + kNwTH - 1) / kNwTH);
    const int nt = 4;
    a.n_groups = Cout / (32 * nt);
    const long long ptiles = (long long)B * a.tiles_hw, blocks = ptiles * a.n_groups;
    if (blocks >= (1LL << 31)) return fail(HMF_ERR_UNSUPPORTED, "hmf_conv_winograd_nw: too many tiles");
    a.xb = xcd_n_split(4LL * B * H * W * Cin, 64LL * Cin * Cout, (int)ptiles, a.n_groups);
    a.mode = 0;
#ifdef HMF_EXPERIMENTS
    a.mode = g_nw_mode;
#endif
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(11, B * H * W, st);
    hipLaunchKernelGGL(k_conv_wino_nw, dim3((unsigned)blocks), dim3(256), 0, st, a);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}
#endif  // HMF_EXPERIMENTS
