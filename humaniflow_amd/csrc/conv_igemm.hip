// Implicit-GEMM NHWC fp32 convolution with the folded-BatchNorm epilogue fused in, for the ResNet encoder (row a1).
//
//   Y[m][n] = relu?( sum_{kh,kw,c} X[b][oh*s+kh-p][ow*s+kw-p][c] * Wt[n][kh][kw][c]  + bias[n] (+ R[m][n]) ),   m = (b, oh, ow)
//
// i.e. the GEMM  (M = B*OH*OW) x (K = KS*KS*Cin) x (N = Cout)  whose A operand is gathered from the NHWC activation tensor on the
// fly (zero rows where the tap falls into the padding) and whose B operand is the weight tensor in channels_last memory order
// [Cout][KS][KS][Cin] (what torch's `.contiguous(memory_format=channels_last)` of a (Cout,Cin,KS,KS) weight already is).
//
// Tiling for CDNA4: a work-group of 4 waves (2x2) owns a BM x BN output tile; every wave owns (BM/2) x (BN/2) = TM x TN MFMA tiles of
// v_mfma_f32_32x32x2_f32 (64-lane waves: pixels are the A rows, output channels the B columns, so lane l ends up holding channel
// n0 + (l & 31) of 16 pixels per tile -> 128-byte coalesced row stores in NHWC).  K is walked in chunks of 32 channels of one filter
// tap: the chunk's A (BM x 32) and B (BN x 32) panels are fetched global -> registers one chunk ahead (16 B per lane, 128 B runs),
// written to a double-buffered LDS panel with a 36-float row stride (conflict-free for the 16-byte row-major writes AND for the
// operand reads), and read back as ONE ds_read_b128 per lane per MFMA tile per 8 k (the k order inside a chunk is free as long as
// A and B agree: lane (i, q) takes k = 8g + 4q + u for the u-th of four MFMAs).  One barrier per chunk.  The epilogue adds the
// per-channel bias (the folded BN shift), the residual branch and the ReLU in registers: the activation tensor is written exactly
// once and never re-read by an elementwise pass.
//
// Block -> tile map is XCD-aware: consecutive work-group ids round-robin over the 8 XCDs, so the ids of one XCD are remapped to a
// CONTIGUOUS run of (m-tile, n-tile) pairs with n fastest: the N/BN blocks that share an activation panel run on the same L2.
#include "hmf_common.h"

namespace hmf {

struct ConvArgs {
    const float* x;      // [B][H][W][Cin]
    const float* w;      // [Cout][KS][KS][Cin]
    const float* bias;   // [Cout] or null
    const float* res;    // [M][Cout] or null
    float* y;            // [M][Cout]
    int B, H, W, Cin, OH, OW, Cout, stride, pad, relu, M;
    // DUAL (1x1 only): a second input concatenated along K -- y = relu(x . W[:, :Cin] + x2(strided) . W[:, Cin:] + bias): a bottleneck's last
    // 1x1 convolution and its downsample branch (models/resnet.py:111-127) as ONE GEMM, so the identity tensor is never written or re-read
    const float* x2;     // [B][H2][W2][Cin2]
    int H2, W2, Cin2, stride2;
};

constexpr int kConvKC = 32;             // channels per k-chunk
constexpr int kConvS = kConvKC + 4;     // LDS row stride in floats

// ABL != 0: ablation builds for tools/igemm_ablate.py (timing only, results are wrong): 1 = no global fetch after the first chunk,
// 2 = no per-chunk barrier, 4 = no epilogue loads/stores.
template <int BM, int BN, int KS, bool DUAL, int ABL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void k_conv_igemm(const ConvArgs a) {
    constexpr int KC = kConvKC, S = kConvS;
    constexpr int TM = BM / 64, TN = BN / 64;      // MFMA tiles per wave
    constexpr int RA = BM / 32, RB = BN / 32;      // panel rows fetched per thread
    __shared__ float sA[2][BM * S];
    __shared__ float sB[2][BN * S];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lq = lane >> 5;

    // ---- XCD-aware tile assignment ----
    const int NT = a.Cout / BN, MT = (a.M + BM - 1) / BM, T = MT * NT;
    int L = blockIdx.x;
    if ((T & 7) == 0) L = (blockIdx.x & 7) * (T >> 3) + (blockIdx.x >> 3);
    const int m0 = (L / NT) * BM, n0 = (L % NT) * BN;

    // ---- per-thread fetch coordinates ----
    // Activation rows are fetched with raw buffer loads: a tap that falls into the zero padding (or a row beyond M) gets an
    // out-of-range byte offset and the hardware returns zeros -- no branch, no select, loads of one chunk all in flight together.
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.B * a.H * a.W * a.Cin * 4, 0x00020000);
    constexpr int kOOB = (int)0x80000000;
    const int col4 = (tid & 7) * 4, rbase = tid >> 3;
    int offA[RA], ih0[RA], iw0[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int m = m0 + rbase + 32 * i;
        if (m < a.M) {
            const int b = m / (a.OH * a.OW), r = m - b * a.OH * a.OW;
            const int oh = r / a.OW, ow = r - oh * a.OW;
            ih0[i] = oh * a.stride - a.pad;
            iw0[i] = ow * a.stride - a.pad;
            offA[i] = (((b * a.H + ih0[i]) * a.W + iw0[i]) * a.Cin + col4) * 4;
        } else {
            ih0[i] = -(1 << 20);
            iw0[i] = -(1 << 20);
            offA[i] = kOOB;
        }
    }
    const int Ktot = KS * KS * a.Cin + (DUAL ? a.Cin2 : 0);
    const float* pB[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) pB[i] = a.w + (long long)(n0 + rbase + 32 * i) * Ktot + col4;
    __amdgpu_buffer_rsrc_t rx2 = rx;
    int offA2[RA];
    if (DUAL) {
        rx2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.x2, 0, a.B * a.H2 * a.W2 * a.Cin2 * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int m = m0 + rbase + 32 * i;
            offA2[i] = kOOB;
            if (m < a.M) {
                const int b = m / (a.OH * a.OW), r = m - b * a.OH * a.OW;
                const int oh = r / a.OW, ow = r - oh * a.OW;
                offA2[i] = (((b * a.H2 + oh * a.stride2) * a.W2 + ow * a.stride2) * a.Cin2 + col4) * 4;
            }
        }
    }

    f32x4 pa[RA], pb[RB], qa[RA], qb[RB];                // two register stages: the fetch runs up to two chunks ahead (ABL & 8: one)
    int kh = 0, kw = 0, c0 = 0;
    auto fetch_into = [&](f32x4 (&pa)[RA], f32x4 (&pb)[RB]) {
        const int tapoff = ((kh * a.W + kw) * a.Cin + c0) * 4;
        const bool second = DUAL && c0 >= a.Cin;               // block-uniform: the K range of the second input
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            int off = offA[i] + tapoff;
            if (KS != 1) off = (((unsigned)(ih0[i] + kh) < (unsigned)a.H) & ((unsigned)(iw0[i] + kw) < (unsigned)a.W)) ? off : kOOB;
            if (DUAL && second) pa[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx2, offA2[i] + (c0 - a.Cin) * 4, 0, 0));
            else pa[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
        }
        const int woff = (kh * KS + kw) * a.Cin + c0;
#pragma unroll
        for (int i = 0; i < RB; ++i) pb[i] = *(const f32x4*)(pB[i] + woff);
        c0 += KC;
        if (!DUAL && c0 == a.Cin) {
            c0 = 0;
            if (++kw == KS) { kw = 0; ++kh; }
        }
    };
    auto stash_from = [&](int buf, const f32x4 (&pa)[RA], const f32x4 (&pb)[RB]) {
#pragma unroll
        for (int i = 0; i < RA; ++i) *(f32x4*)&sA[buf][(rbase + 32 * i) * S + col4] = pa[i];
#pragma unroll
        for (int i = 0; i < RB; ++i) *(f32x4*)&sB[buf][(rbase + 32 * i) * S + col4] = pb[i];
    };
    auto fetch = [&]() { fetch_into(pa, pb); };
    auto stash = [&](int buf) { stash_from(buf, pa, pb); };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nchunks = KS * KS * (a.Cin / KC) + (DUAL ? a.Cin2 / KC : 0);
    fetch();
    stash(0);
    __syncthreads();
    const int rowA = (wm * (BM / 2) + li) * S + lq * 4, rowB = (wn * (BN / 2) + li) * S + lq * 4;
    constexpr int G = KC / 8;
    f32x4 fa[2][TM], fb[2][TN];
    auto frag = [&](int buf, int g, int slot) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[slot][i] = *(const f32x4*)&sA[buf][rowA + i * 32 * S + g * 8];
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[slot][j] = *(const f32x4*)&sB[buf][rowB + j * 32 * S + g * 8];
    };
    auto compute = [&](int buf) {
        frag(buf, 0, 0);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (g + 1 < G) frag(buf, g + 1, (g + 1) & 1);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][i][u], fb[g & 1][j][u], acc[i][j], 0, 0, 0);
        }
    };
    if (ABL & 8) {                                  // one chunk ahead (the round-1 loop)
        for (int c = 0; c < nchunks; ++c) {
            const int buf = c & 1;
            if (c + 1 < nchunks && !(ABL & 1)) fetch();
            compute(buf);
            if (c + 1 < nchunks && !(ABL & 1)) stash(buf ^ 1);
            if (!(ABL & 2)) __syncthreads();
        }
    } else {
        // Two chunks ahead: while chunk c is multiplied out of LDS[c & 1], chunk c+1 sits in one register stage (requested one iteration ago,
        // written to LDS at the end of this one) and chunk c+2 is requested into the other.  Measured (tools/igemm_ablate.py): the fetch
        // wait that the one-ahead loop exposes is 4-10 us of a 40-46 us layer when only 1-2 work-groups fit a CU (layer3/layer4).
        if (nchunks > 1 && !(ABL & 1)) fetch_into(qa, qb);                       // chunk 1
        for (int c = 0; c < nchunks; c += 2) {
            if (c + 2 < nchunks && !(ABL & 1)) fetch_into(pa, pb);               // chunk c+2
            compute(0);
            if (c + 1 < nchunks && !(ABL & 1)) stash_from(1, qa, qb);            // chunk c+1
            if (!(ABL & 2)) __syncthreads();
            if (c + 1 >= nchunks) break;
            if (c + 3 < nchunks && !(ABL & 1)) fetch_into(qa, qb);               // chunk c+3
            compute(1);
            if (c + 2 < nchunks && !(ABL & 1)) stash_from(0, pa, pb);            // chunk c+2
            if (!(ABL & 2)) __syncthreads();
        }
    }
    if (ABL & 4) {                                  // keep the accumulators alive without the epilogue traffic
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) t += acc[i][j][r];
        if (a.M < 0) a.y[tid] = t;
        return;
    }

    // ---- epilogue: + bias, + residual, ReLU, 128-byte row stores.  All residual loads of a tile column are issued before use. ----
    const bool full = (m0 + BM <= a.M);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / 2) + j * 32 + li;
        const float bv = a.bias ? a.bias[n] : 0.f;
        float rv[TM][16];
        if (a.res) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int mb = m0 + wm * (BM / 2) + i * 32 + 4 * lq;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = min(mb + (r & 3) + 8 * (r >> 2), a.M - 1);
                    rv[i][r] = a.res[(long long)m * a.Cout + n];
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[i][r] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + wm * (BM / 2) + i * 32 + 4 * lq;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb + (r & 3) + 8 * (r >> 2);
                float v = acc[i][j][r] + bv + rv[i][r];
                if (a.relu) v = fmaxf(v, 0.f);
                if (full || m < a.M) a.y[(long long)m * a.Cout + n] = v;     // (non-temporal stores measured: no gain here, and the training graph re-reads y at once)
            }
        }
    }
}

template <int BM, int BN, int KS, bool DUAL = false>
static int launch_conv(const ConvArgs& a, hipStream_t st) {
    const int T = ((a.M + BM - 1) / BM) * (a.Cout / BN);
    ProfScope ps(8, a.M, st);
    hipLaunchKernelGGL((k_conv_igemm<BM, BN, KS, DUAL, 0>), dim3(T), dim3(256), 0, st, a);
    return 0;
}

#ifdef HMF_EXPERIMENTS        // tools/ only (humaniflow_amd.build --experiments): never part of libhumaniflow_amd.so
template <int ABL>
static int launch_conv_abl(const ConvArgs& a, hipStream_t st) {
    const int T = ((a.M + 63) / 64) * (a.Cout / 64);
    hipLaunchKernelGGL((k_conv_igemm<64, 64, 1, false, ABL>), dim3(T), dim3(256), 0, st, a);
    return 0;
}
#endif

}  // namespace hmf

using namespace hmf;

// tile: 0 = auto, else BM*1000 + BN (64128, 64064; 128-row tiles were measured and lose on every ResNet shape: 2 work-groups per CU only)
extern "C" int hmf_conv2d_nhwc(const float* x, const float* w, const float* bias, const float* residual, float* y, int B, int H, int W,
                               int Cin, int Cout, int ks, int stride, int relu, int tile, void* stream) {
    if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0) return fail(HMF_ERR_INVALID, "hmf_conv2d_nhwc: bad args");
    if (ks != 1 && ks != 3) return fail(HMF_ERR_UNSUPPORTED, "hmf_conv2d_nhwc: kernel size %d (1 and 3 are built)", ks);
    if (stride != 1 && stride != 2) return fail(HMF_ERR_UNSUPPORTED, "hmf_conv2d_nhwc: stride %d", stride);
    if (Cin % kConvKC != 0 || Cout % 64 != 0) return fail(HMF_ERR_UNSUPPORTED, "hmf_conv2d_nhwc: Cin=%d must be a multiple of 32 and Cout=%d of 64", Cin, Cout);
    if (((uintptr_t)x & 15) || ((uintptr_t)w & 15)) return fail(HMF_ERR_INVALID, "hmf_conv2d_nhwc: x and w must be 16-byte aligned");
    ConvArgs a;
    a.x2 = nullptr; a.H2 = a.W2 = a.Cin2 = a.stride2 = 0;
    a.x = x; a.w = w; a.bias = bias; a.res = residual; a.y = y;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.stride = stride; a.pad = ks / 2; a.relu = relu;
    a.OH = (H + 2 * a.pad - ks) / stride + 1;
    a.OW = (W + 2 * a.pad - ks) / stride + 1;
    const long long M = (long long)B * a.OH * a.OW;
    if (M * (long long)Cout >= (1LL << 29) || (long long)B * H * W * Cin >= (1LL << 29))
        return fail(HMF_ERR_UNSUPPORTED, "hmf_conv2d_nhwc: tensor exceeds 2^29 elements (32-bit byte offsets of the buffer loads)");
    a.M = (int)M;
    if (tile == 0) {
        // measured on MI355X (tools/conv_probe.py, B=32): the 64x64 tile (36 KB LDS -> 4 work-groups = 16 waves per CU) wins or ties on
        // every ResNet layer; 64x128 is ahead by 2-3 % on the 3x3 layers with Cout >= 128 and many rows.
        tile = (ks == 3 && Cout % 128 == 0 && M >= 32768) ? 64128 : 64064;
    }
    hipStream_t st = (hipStream_t)stream;
#define CONV_CASE(BM, BN)                                                    \
    if (ks == 1) launch_conv<BM, BN, 1>(a, st); else launch_conv<BM, BN, 3>(a, st); \
    break;
    switch (tile) {
        case 64128: if (Cout % 128) return fail(HMF_ERR_INVALID, "hmf_conv2d_nhwc: tile 64x128 needs Cout %% 128 == 0"); CONV_CASE(64, 128)
        case 64064: CONV_CASE(64, 64)
#ifdef HMF_EXPERIMENTS
        case 1064064: launch_conv_abl<1>(a, st); break;
        case 2064064: launch_conv_abl<2>(a, st); break;
        case 3064064: launch_conv_abl<3>(a, st); break;
        case 4064064: launch_conv_abl<4>(a, st); break;
        case 7064064: launch_conv_abl<7>(a, st); break;
        case 8064064: launch_conv_abl<8>(a, st); break;
#endif
        default: return fail(HMF_ERR_INVALID, "hmf_conv2d_nhwc: unknown tile %d (0 = auto, 64064, 64128)", tile);
    }
#undef CONV_CASE
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

// y = relu?(x . W[:, :Cin] + x2[:, ::stride2, ::stride2, :] . W[:, Cin:] + bias): two 1x1 convolutions that share their output, as one GEMM over the
// concatenated K.  x (B,OH,OW,Cin); x2 (B,H2,W2,Cin2) with OH = (H2-1)/stride2+1; w (Cout, Cin+Cin2) row-major.
extern "C" int hmf_conv1x1_dual_nhwc(const float* x, const float* x2, const float* w, const float* bias, float* y, int B, int OH, int OW, int Cin,
                                     int H2, int W2, int Cin2, int stride2, int Cout, int relu, void* stream) {
    if (!x || !x2 || !w || !y || B <= 0 || OH <= 0 || OW <= 0) return fail(HMF_ERR_INVALID, "hmf_conv1x1_dual_nhwc: bad args");
    if (stride2 != 1 && stride2 != 2) return fail(HMF_ERR_UNSUPPORTED, "hmf_conv1x1_dual_nhwc: stride %d", stride2);
    if ((H2 - 1) / stride2 + 1 != OH || (W2 - 1) / stride2 + 1 != OW) return fail(HMF_ERR_INVALID, "hmf_conv1x1_dual_nhwc: x2 %dx%d / %d does not give %dx%d", H2, W2, stride2, OH, OW);
    if (Cin % kConvKC || Cin2 % kConvKC || Cout % 64) return fail(HMF_ERR_UNSUPPORTED, "hmf_conv1x1_dual_nhwc: Cin=%d, Cin2=%d must be multiples of 32 and Cout=%d of 64", Cin, Cin2, Cout);
    if (((uintptr_t)x & 15) || ((uintptr_t)x2 & 15) || ((uintptr_t)w & 15)) return fail(HMF_ERR_INVALID, "hmf_conv1x1_dual_nhwc: x, x2 and w must be 16-byte aligned");
    const long long M = (long long)B * OH * OW;
    if (M * Cout >= (1LL << 29) || M * Cin >= (1LL << 29) || (long long)B * H2 * W2 * Cin2 >= (1LL << 29))
        return fail(HMF_ERR_UNSUPPORTED, "hmf_conv1x1_dual_nhwc: tensor exceeds 2^29 elements");
    ConvArgs a;
    a.x = x; a.w = w; a.bias = bias; a.res = nullptr; a.y = y;
    a.B = B; a.H = OH; a.W = OW; a.Cin = Cin; a.Cout = Cout; a.stride = 1; a.pad = 0; a.relu = relu; a.OH = OH; a.OW = OW; a.M = (int)M;
    a.x2 = x2; a.H2 = H2; a.W2 = W2; a.Cin2 = Cin2; a.stride2 = stride2;
    launch_conv<64, 64, 1, true>(a, (hipStream_t)stream);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}
