// Patch-resident NHWC fp32 convolution for the encoder's stride-1 3x3 and its 1x1 layers (row a1), fused with the folded-BN epilogue.
//
// Same GEMM as conv_igemm.hip (M = B*OH*OW pixels, N = Cout, K = KS*KS*Cin) but organised like the stem kernel so that the MFMA loop
// runs barrier-free for hundreds of k-steps:
//   * a work-group (4 waves) owns a TH x TW tile of output pixels of ONE image and BN output channels;
//   * K is walked in slabs of SC input channels.  Per slab the (TH+KS-1) x (TW+KS-1) x SC input patch is copied ONCE into LDS (raw
//     buffer loads: taps that fall into the zero padding / outside the image come back as zeros), so a 3x3 convolution reads its
//     input from L2 once instead of nine times and all nine taps are served by the same LDS-resident patch;
//   * inside a slab the A operand of lane (pixel i, q) for tap (kh, kw) and channel group g is ONE ds_read_b128 at
//     base_i + const(kh, kw, g)  (pixel stride SC+4 floats: conflict-free), i.e. an immediate offset, no address arithmetic;
//   * the B operand (weights) never touches LDS: it is pre-packed at plan time into the exact MFMA lane order
//     [n-tile][slab][tap][g][lane][4] and streamed from L2 with 16-byte loads through a register ring kPF groups ahead;
//   * KS*KS*SC/8 groups x 4 MFMAs (x TN n-tiles) run between two barriers (576 k-steps for a 3x3 slab of 64 channels).
// Epilogue as in conv_igemm.hip: lane = output channel, + bias, + residual, ReLU, 128-byte channel runs in NHWC.
//
// Wave -> work map: PG pixel groups x (4/PG) channel groups; a pixel group is 32 pixels = 32/TW rows of the tile; every wave owns
// TN n-tiles of 32 channels: BN = (4/PG) * TN * 32.
#include "hmf_common.h"

namespace hmf {

struct PatchArgs {
    const float* x;       // [B][H][W][Cin]
    const f32x4* wp;      // packed weights (hmf_conv_pack_patch)
    const float* bias;
    const float* res;
    float* y;             // [B][OH][OW][Cout]
    int B, H, W, Cin, OH, OW, Cout, stride, relu, tiles_w, tiles_hw, n_tiles;
    int xb;               // XCD split along the channel tiles (hmf_common.h xcd_n_split)
    int ksplit;           // > 1: blockIdx.y walks Cin / SC / ksplit slabs and stores raw sums to part[blockIdx.y] (hmf_common.h small_launch_ksplit)
    float* part;          // [ksplit][B][OH][OW][Cout]
};

template <int KS, int TW, int PG, int TN, int SC, int ST>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void k_conv_patch(const PatchArgs a) {
    constexpr int CG = 4 / PG;                      // channel groups
    constexpr int RPG = 32 / TW;                    // tile rows per pixel group
    constexpr int TH = PG * RPG;
    constexpr int PH = (TH - 1) * ST + KS, PW = (TW - 1) * ST + KS;   // ST = stride of a 3x3 layer (1x1: the loader strides, patch is compact)
    constexpr int CS = SC + 4;                      // LDS pixel stride (floats)
    constexpr int NG = SC / 8;                      // k-groups per tap per slab
    constexpr int NV = PH * PW * (SC / 4);          // float4 per slab patch
    constexpr int NIT = (NV + 255) / 256;
    // LDS row pitch (floats): ds_read_b128 is served in four 16-lane groups ({0-3,12-15,20-27}, ...) over 64 banks, so the 16-byte slots
    // of a group must be distinct mod 16.  Pixel stride CS = 17 slots; a 16-wide tile needs the row pitch = 0 (mod 16 slots), an 8-wide
    // tile = 8 (mod 16 slots); the stride-2 variant stores the columns split by parity (below).
    constexpr int kDense = PW * CS;
    // Stride 2 (round 2): patch columns de-interleaved by parity (column c at slot (c & 1) * kHalf + (c >> 1)), so the lanes' columns 2 px + kw are
    // consecutive slots; two tile rows of a 16-lane group are 2 patch rows apart: row pitch = 4 (mod 8) slots (as k_conv_winograd's patch image).
    constexpr int kHalf = (PW + 1) / 2;
    constexpr int RPF = ST == 2 ? ((kDense / 4 + 3) / 8 * 8 + 4) * 4 : (TW == 16 ? (kDense + 63) / 64 * 64 : ((kDense - 32 + 63) / 64 * 64 + 32));
    static_assert(ST != 2 || (RPF >= kDense && (RPF / 4) % 8 == 4), "stride-2 row pitch");
    // weight ring: slot = group % kRing must stay in phase over slabs -> kRing divides GPS (8, 16 or 72 groups per slab)
    constexpr int kRing = 8, kPF = kRing - 2;
    __shared__ float patch[PH * RPF];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lq = lane >> 5;
    const int pg = wave % PG, cg = wave / PG;

    // block -> (image, pixel tile, channel tile); channel tile fastest so the blocks sharing an input patch are neighbours
    // and XCD-aware (work-group ids round-robin over the 8 XCDs: give each XCD a contiguous run of logical tiles -> one L2 per patch)
    const int L = xcd_tile(blockIdx.x, gridDim.x / a.n_tiles, a.n_tiles, a.xb);
    const int nt = L % a.n_tiles, pt = L / a.n_tiles;
    const int b = pt / a.tiles_hw, trem = pt - b * a.tiles_hw;
    const int oh0 = (trem / a.tiles_w) * TH, ow0 = (trem % a.tiles_w) * TW;

    // ---- patch fetch coordinates (fixed over slabs) ----
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.B * a.H * a.W * a.Cin * 4, 0x00020000);
    constexpr int kOOB = (int)0x80000000;
    int goff[NIT], loff[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int e = tid + 256 * k;
        const int p = e / (SC / 4), c4 = e - p * (SC / 4);
        const int r = p / PW, col = p - r * PW;
        const int ih = KS == 1 ? (oh0 + r) * a.stride : oh0 * ST + r - 1, iw = KS == 1 ? (ow0 + col) * a.stride : ow0 * ST + col - 1;
        const bool ok = (e < NV) & ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W);
        goff[k] = ok ? (((b * a.H + ih) * a.W + iw) * a.Cin + 4 * c4) * 4 : kOOB;
        loff[k] = r * RPF + (ST == 2 ? (col & 1) * kHalf + (col >> 1) : col) * CS + 4 * c4;
    }
    const int nslab = a.Cin / SC;
    constexpr int GPS = KS * KS * NG;               // groups per slab
    // this wave's weight stream: [n-tile][slab][tap][g][lane][4]; n-tile index = (nt * CG + cg) * TN + t
    const f32x4* pw[TN];
    const int s_per = nslab / a.ksplit, s_begin = (int)blockIdx.y * s_per, s_end = s_begin + s_per;      // ksplit = 1: all slabs
#pragma unroll
    for (int t = 0; t < TN; ++t) pw[t] = a.wp + (((long long)((nt * CG + cg) * TN + t) * nslab + s_begin) * GPS) * 64 + lane;

    const int prow = pg * RPG + li / TW, pcol = li % TW;
    const float* pa = patch + prow * ST * RPF + pcol * (ST == 2 ? 1 : ST) * CS + 4 * lq;

    f32x16 acc[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    f32x4 bw[kRing][TN];
#pragma unroll
    for (int g = 0; g < kPF; ++g)
#pragma unroll
        for (int t = 0; t < TN; ++t) bw[g][t] = pw[t][g * 64];

    for (int s = s_begin; s < s_end; ++s) {
        // ---- slab patch: global -> registers -> LDS ----
        f32x4 pv[NIT];
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int off = goff[k] == kOOB ? kOOB : goff[k] + s * SC * 4;
            pv[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s != s_begin) __syncthreads();           // everyone is done reading the previous slab
#pragma unroll
        for (int k = 0; k < NIT; ++k)
            if (NV % 256 == 0 || k < NIT - 1 || tid + 256 * k < NV) *(f32x4*)&patch[loff[k]] = pv[k];
        __syncthreads();

        // ---- barrier-free MFMA run over the slab ----
        const bool last = (s + 1 == s_end);
        f32x4 fa[2];
        fa[0] = *(const f32x4*)pa;
#pragma clang loop unroll(full)
        for (int gi = 0; gi < GPS; ++gi) {
            // weights kPF groups ahead (runs into the next slab's groups: the stream is contiguous); the very last groups of the
            // kernel would read past the stream, so they are skipped
            if (!(last && gi + kPF >= GPS)) {
#pragma unroll
                for (int t = 0; t < TN; ++t) bw[(gi + kPF) % kRing][t] = pw[t][(gi + kPF) * 64];
            }
            if (gi + 1 < GPS) {
                const int tap = (gi + 1) / NG, g = (gi + 1) - tap * NG;
                const int kw = tap % KS, cslot = ST == 2 ? (kw & 1) * kHalf + (kw >> 1) : kw;
                fa[(gi + 1) & 1] = *(const f32x4*)(pa + (tap / KS) * RPF + cslot * CS + 8 * g);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int t = 0; t < TN; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[gi & 1][u], bw[gi % kRing][t][u], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t = 0; t < TN; ++t) pw[t] += GPS * 64;
        // ring slots are indexed by gi % kRing; keep the phase consistent across slabs by rotating when GPS % kRing != 0
        if (GPS % kRing != 0) {
            f32x4 tmp[kRing][TN];
#pragma unroll
            for (int i = 0; i < kRing; ++i)
#pragma unroll
                for (int t = 0; t < TN; ++t) tmp[i][t] = bw[(i + GPS) % kRing][t];
#pragma unroll
            for (int i = 0; i < kRing; ++i)
#pragma unroll
                for (int t = 0; t < TN; ++t) bw[i][t] = tmp[i][t];
        }
    }

    // ---- epilogue (split reduction: raw sums of this slab range, no bias / residual / ReLU: splitk_finish applies them to the total) ----
    const bool partial = a.ksplit > 1;
    float* const yout = partial ? a.part + (size_t)blockIdx.y * ((size_t)a.B * a.OH * a.OW * a.Cout) : a.y;
    const float* const resp = partial ? nullptr : a.res;
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int n = ((nt * CG + cg) * TN + t) * 32 + li;
        const float bv = (a.bias && !partial) ? a.bias[n] : 0.f;
        long long yoff[16];
        bool okp[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * lq;
            const int oh = oh0 + pg * RPG + i / TW, ow = ow0 + i % TW;
            okp[r] = (oh < a.OH) & (ow < a.OW);
            yoff[r] = (((long long)b * a.OH + min(oh, a.OH - 1)) * a.OW + min(ow, a.OW - 1)) * a.Cout + n;
        }
        float rv[16];
        if (resp) {
#pragma unroll
            for (int r = 0; r < 16; ++r) rv[r] = resp[yoff[r]];
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) rv[r] = 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[t][r] + bv + rv[r];
            if (a.relu && !partial) v = fmaxf(v, 0.f);
            if (okp[r]) yout[yoff[r]] = v;
        }
    }
}

template <int KS, int TW, int PG, int TN, int SC, int ST>
static int launch_patch(PatchArgs a, hipStream_t st) {
    constexpr int TH = PG * (32 / TW), BN = (4 / PG) * TN * 32;
    a.tiles_w = (a.OW + TW - 1) / TW;
    a.tiles_hw = a.tiles_w * ((a.OH + TH - 1) / TH);
    a.n_tiles = a.Cout / BN;
    const long long blocks = (long long)a.B * a.tiles_hw * a.n_tiles;
    a.xb = xcd_n_split(4LL * a.B * a.H * a.W * a.Cin, 4LL * KS * KS * a.Cin * a.Cout, a.B * a.tiles_hw, a.n_tiles);
    a.ksplit = a.part ? small_launch_ksplit(blocks, a.Cin / SC) : 1;
    ProfScope ps(10, a.B * a.OH * a.OW, st);
    hipLaunchKernelGGL((k_conv_patch<KS, TW, PG, TN, SC, ST>), dim3((unsigned)blocks, a.ksplit), dim3(256), 0, st, a);
    HMF_LAUNCH_CHECK();
    if (a.ksplit > 1) return splitk_finish(a.part, a.ksplit, (long long)a.B * a.OH * a.OW * a.Cout, a.bias, a.res, a.y, a.Cout, a.relu, st);
    return HMF_OK;
}

// work-groups of a launch before any split (for the workspace size)
static long long patch_blocks(int B, int OH, int OW, int Cout, int TW, int TH, int BN) {
    return (long long)B * ((OW + TW - 1) / TW) * ((OH + TH - 1) / TH) * (Cout / BN);
}

}  // namespace hmf

using namespace hmf;

// cfg = TW*1000 + PG*100 + TN*10 (+ 1: 32-channel slabs instead of 64): 16420 (8x16 pixels x 64 channels), 16410 (8x16 x 32), 8210 (8x8 x 64),
// 8211 (8x8 x 64 with 32-channel slabs: half the LDS per work-group, stride-2 layers only)
static int patch_cfg_ok(int cfg) { return cfg == 16420 || cfg == 16410 || cfg == 8210 || cfg == 8211; }
static int patch_cfg_bn(int cfg) { const int pg = (cfg / 100) % 10, tn = (cfg / 10) % 10; return (4 / pg) * tn * 32; }
static int patch_cfg_sc(int cfg) { return (cfg % 10) == 1 ? 32 : 64; }

extern "C" long long hmf_conv_patch_packed_floats(int Cin, int Cout, int ks) { return (long long)Cout * ks * ks * Cin; }

// w: [Cout][ks][ks][Cin] (channels_last weight); packed: [Cout/32][Cin/SC][ks*ks][SC/8][lane = q*32 + j][4]
extern "C" int hmf_conv_patch_pack_host(const float* w, float* packed, int Cin, int Cout, int ks, int cfg) {
    if (!w || !packed) return fail(HMF_ERR_INVALID, "hmf_conv_patch_pack_host: null pointer");
    if (!patch_cfg_ok(cfg)) return fail(HMF_ERR_INVALID, "hmf_conv_patch_pack_host: unknown cfg %d", cfg);
    const int SC = patch_cfg_sc(cfg);
    if (Cin % SC || Cout % 32 || (ks != 1 && ks != 3)) return fail(HMF_ERR_UNSUPPORTED, "hmf_conv_patch_pack_host: Cin=%d Cout=%d ks=%d", Cin, Cout, ks);
    const int taps = ks * ks, NG = SC / 8, nslab = Cin / SC;
    long long o = 0;
    for (int t = 0; t < Cout / 32; ++t)
        for (int s = 0; s < nslab; ++s)
            for (int tap = 0; tap < taps; ++tap)
                for (int g = 0; g < NG; ++g)
                    for (int l = 0; l < 64; ++l)
                        for (int u = 0; u < 4; ++u) {
                            const int n = 32 * t + (l & 31), c = s * SC + 8 * g + 4 * (l >> 5) + u;
                            packed[o++] = w[((long long)n * taps + tap) * Cin + c];
                        }
    return HMF_OK;
}

// tile geometry of a cfg: pixel tile TW x TH, channel tile BN (the 3x3 / stride 2 layers always run cfg 8210's 8 x 8 x 64)
static void patch_cfg_tile(int cfg, int stride, int& TW, int& TH, int& BN) {
    if (stride == 2 && cfg != 8211) cfg = 8210;
    TW = cfg / 1000;
    const int pg = (cfg / 100) % 10;
    TH = pg * (32 / TW);
    BN = patch_cfg_bn(cfg);
}

static int conv_patch_impl(const float* x, const float* w_packed, const float* bias, const float* residual, float* y, int B, int H, int W, int Cin, int Cout, int ks,
                           int stride, int relu, int cfg, void* ws, long long ws_bytes, void* stream);

extern "C" long long hmf_conv_patch_workspace_bytes(int B, int H, int W, int Cin, int Cout, int ks, int stride, int cfg) {
    if (B <= 0 || H <= 0 || W <= 0 || ks != 3 || (stride != 1 && stride != 2) || !patch_cfg_ok(cfg) || Cin % 64 || (cfg == 8211 && stride != 2)) return 0;
    int TW, TH, BN;
    patch_cfg_tile(cfg, stride, TW, TH, BN);
    if (Cout % BN) return 0;
    const int OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
    const int ks_ = small_launch_ksplit(patch_blocks(B, OH, OW, Cout, TW, TH, BN), Cin / patch_cfg_sc(cfg));
    return ks_ > 1 ? ks_ * 4LL * B * OH * OW * Cout : 0;
}

extern "C" int hmf_conv_patch_ws(const float* x, const float* w_packed, const float* bias, const float* residual, float* y, int B, int H, int W, int Cin, int Cout,
                                 int ks, int stride, int relu, int cfg, void* ws, long long ws_bytes, void* stream) {
    return conv_patch_impl(x, w_packed, bias, residual, y, B, H, W, Cin, Cout, ks, stride, relu, cfg, ws, ws_bytes, stream);
}

extern "C" int hmf_conv_patch(const float* x, const float* w_packed, const float* bias, const float* residual, float* y, int B, int H, int W,
                              int Cin, int Cout, int ks, int stride, int relu, int cfg, void* stream) {
    return conv_patch_impl(x, w_packed, bias, residual, y, B, H, W, Cin, Cout, ks, stride, relu, cfg, nullptr, 0, stream);
}

static int conv_patch_impl(const float* x, const float* w_packed, const float* bias, const float* residual, float* y, int B, int H, int W, int Cin, int Cout, int ks,
                           int stride, int relu, int cfg, void* ws, long long ws_bytes, void* stream) {
    if (!x || !w_packed || !y || B <= 0 || H <= 0 || W <= 0) return fail(HMF_ERR_INVALID, "hmf_conv_patch: bad args");
    if (!patch_cfg_ok(cfg)) return fail(HMF_ERR_INVALID, "hmf_conv_patch: unknown cfg %d", cfg);
    if (ks != 3 || (stride != 1 && stride != 2)) return fail(HMF_ERR_UNSUPPORTED, "hmf_conv_patch: ks=%d stride=%d (3x3, stride 1 or 2)", ks, stride);
    if (ks == 3 && stride == 2 && cfg != 8210 && cfg != 8211) return fail(HMF_ERR_UNSUPPORTED, "hmf_conv_patch: 3x3 / stride 2 is built for cfg 8210 / 8211 only");
    if (cfg == 8211 && stride != 2) return fail(HMF_ERR_UNSUPPORTED, "hmf_conv_patch: cfg 8211 (32-channel slabs) is built for the stride-2 layers only");
    const int SC = patch_cfg_sc(cfg), BN = patch_cfg_bn(cfg);
    if (Cin % SC || Cout % BN) return fail(HMF_ERR_UNSUPPORTED, "hmf_conv_patch: Cin=%d %% %d or Cout=%d %% %d", Cin, SC, Cout, BN);
    if (((uintptr_t)x & 15) || ((uintptr_t)w_packed & 15)) return fail(HMF_ERR_INVALID, "hmf_conv_patch: x and w_packed must be 16-byte aligned");
    PatchArgs a;
    a.x = x; a.wp = (const f32x4*)w_packed; a.bias = bias; a.res = residual; a.y = y;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.stride = stride; a.relu = relu;
    a.OH = (H - 1) / stride + 1;      // pad = ks/2: (H + 2*(ks/2) - ks) / stride + 1
    a.OW = (W - 1) / stride + 1;
    if ((long long)B * a.OH * a.OW * Cout >= (1LL << 29) || (long long)B * H * W * Cin >= (1LL << 29))
        return fail(HMF_ERR_UNSUPPORTED, "hmf_conv_patch: tensor exceeds 2^29 elements (32-bit byte offsets of the buffer loads)");
    hipStream_t st = (hipStream_t)stream;
    a.part = nullptr;
    a.ksplit = 1;
    if (ws) {                                                     // the small-launch form: raw partial sums in ws, finished by splitk_finish
        const long long need = hmf_conv_patch_workspace_bytes(B, H, W, Cin, Cout, ks, stride, cfg);
        if (need) {
            if (ws_bytes < need || ((uintptr_t)ws & 15) || ((uintptr_t)y & 15) || ((uintptr_t)bias & 15) || ((uintptr_t)residual & 15))
                return fail(HMF_ERR_INVALID, "hmf_conv_patch_ws: workspace smaller than hmf_conv_patch_workspace_bytes, or ws / y / bias / residual not 16-byte aligned");
            a.part = (float*)ws;
        }
    }
#define PCASE(CFG, KS_, TW_, PG_, TN_, SC_) \
    if (cfg == CFG && ks == KS_) return launch_patch<KS_, TW_, PG_, TN_, SC_, 1>(a, st);
    if (ks == 3 && stride == 2 && cfg == 8211) return launch_patch<3, 8, 2, 1, 32, 2>(a, st);
    if (ks == 3 && stride == 2) return launch_patch<3, 8, 2, 1, 64, 2>(a, st);
    PCASE(16420, 3, 16, 4, 2, 64) PCASE(16410, 3, 16, 4, 1, 64) PCASE(8210, 3, 8, 2, 1, 64)
    // 1x1 layers were measured on this kernel as well (64- and 128-channel slabs, single- and double-buffered patches): equal to or slower than
    // k_conv_igemm on every ResNet shape, so those instantiations are not built
#undef PCASE
    return fail(HMF_ERR_UNSUPPORTED, "hmf_conv_patch: cfg %d is not built for ks=%d", cfg, ks);
}
