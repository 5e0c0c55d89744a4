// Input gradient of a stride-2 3x3 convolution (pad 1) of the encoder's training graph (row a1 / configs[4]; models/resnet.py:53-67 BasicBlock
// conv1 of layer2-4.0, :95-100 Bottleneck conv2), optionally + the input gradient of the block's stride-2 1x1 downsample branch (:42-50) and
// + a residual gradient, as ONE launch of parity-class GEMMs on the LDS-DMA pipeline of csrc/conv_gemm.hip.
//
//   gx[b][ih][iw][ci] = sum_{kh,kw,co} gy[b][oh][ow][co] w[co][ci][kh][kw],   oh = (ih + 1 - kh) / 2,  ow = (iw + 1 - kw) / 2  (integral, in range)
//
// Only taps with kh = ih + 1 (mod 2) reach input row ih: pixels (2a + pa, 2b + pb) of one parity class (pa, pb) see (1 + pa)(1 + pb) of the 9 taps, always
// from the same relative positions gy[a + dh][b + dw], dh = (pa and kh == 0), dw = (pb and kw == 0).  So each class is a GEMM
//   M = B x ceil((H - pa) / 2) x ceil((W - pb) / 2) pixels,  N = Cin,  K = taps x Cout,   A = [gy shifted by the class's taps],  B = the taps' filters,
// 9 tap-GEMMs for 4 output pixels instead of the 36 of the zero-inserted map (round 2: torch.zeros + strided copy + a full-resolution
// Winograd convolution = 270-435 us per layer at B = 72; the MFMA work alone was 4x what the stride needs).  The downsample branch's gradient lands
// on class (0, 0) only (its pixels are the sampled ones) as one more K range from the second gradient tensor.  Every pixel of gx belongs to exactly
// one class, so gx needs no zero fill.
//
// Work-group = 64 pixels of one class x 64 input channels, 4 waves (2 x 2, 32 x 32 MFMA tile each), K chunks of 32 through a 3-stage LDS-DMA ring
// (source-swizzled 128-byte rows, counted vmcnt, raw barriers: conv_gemm.hip).  Pixels whose source lies outside gy read an out-of-range buffer
// offset: the DMA writes zeros.
#include "hmf_common.h"

namespace hmf {

struct DgradS2Args {
    const float* gy;      // [B][OH][OW][Co]
    const float* wt;      // [Ci][9][Co]   = w.permute(1, 2, 3, 0)
    const float* gy2;     // [B][OH][OW][Co2] or null
    const float* wt2;     // [Ci][Co2]     = w_downsample^T
    const float* res;     // [B][H][W][Ci] or null
    float* gx;            // [B][H][W][Ci]
    int B, OH, OW, H, W, Co, Co2, Ci;
    int tiles[4];         // work-groups of class c start at tiles[c] (classes in the order 3, 1, 2, 0: most taps first)
};

typedef __attribute__((address_space(3))) void* lds_ptr_d;
#define HMF_DVMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
static __device__ __forceinline__ void ddma16(__amdgpu_buffer_rsrc_t rsrc, lds_ptr_d dst, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voff, soff, 0, 0);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void k_conv_dgrad_s2(const DgradS2Args a) {
    constexpr int S = 3, NA = 2, NB = 2, L = NA + NB, STAGE = 128 * 32;
    constexpr int kOOB = (int)0x80000000;
    __shared__ __attribute__((aligned(16))) float smem[S * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lq = lane >> 5;

    // class of this work-group (block-uniform)
    const int order[4] = {3, 1, 2, 0};
    int slot = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if ((int)blockIdx.x >= a.tiles[k]) slot = k;
    const int cls = order[slot], pa = cls >> 1, pb = cls & 1;
    const int Ac = (a.H - pa + 1) / 2, Bc = (a.W - pb + 1) / 2, Mc = a.B * Ac * Bc;
    const int NT = a.Ci / 64;
    const int t = (int)blockIdx.x - a.tiles[slot];
    const int m0 = (t / NT) * 64, n0 = (t % NT) * 64;
    const int nth = 1 + pa, ntw = 1 + pb, ntap = nth * ntw;
    const int cps = a.Co / 32, cps2 = (a.gy2 && cls == 0) ? a.Co2 / 32 : 0;
    const int nch = ntap * cps + cps2;

    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)a.gy, 0, a.B * a.OH * a.OW * a.Co * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg2 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.gy2 ? a.gy2 : a.gy), 0, a.B * a.OH * a.OW * (a.gy2 ? a.Co2 : a.Co) * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.wt, 0, a.Ci * 9 * a.Co * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw2 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.wt2 ? a.wt2 : a.wt), 0, a.Ci * (a.wt2 ? a.Co2 : 9 * a.Co) * 4, 0x00020000);

    // A rows: per tap (th, tw) the shifted source pixel of each of this lane's rows; tap (th, tw): kh = pa ? 2 th : 1, dh = pa && th == 0 (kw, dw alike)
    int voA[4][NA], voA2[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int row = (wave * NA + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        const int m = m0 + row;
        const int img = m / (Ac * Bc), r = m - img * (Ac * Bc), ia = r / Bc, ib = r - ia * Bc;
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
            const int th = tp >> 1, tw = tp & 1;
            const int oh = ia + ((pa && th == 0) ? 1 : 0), ow = ib + ((pb && tw == 0) ? 1 : 0);
            const bool ok = m < Mc && th < nth && tw < ntw && oh < a.OH && ow < a.OW;
            voA[tp][i] = ok ? (((img * a.OH + oh) * a.OW + ow) * a.Co + 4 * c) * 4 : kOOB;
        }
        voA2[i] = (m < Mc && ia < a.OH && ib < a.OW) ? (((img * a.OH + ia) * a.OW + ib) * a.Co2 + 4 * c) * 4 : kOOB;
    }
    int voB[NB], voB2[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int row = (wave * NB + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        voB[i] = ((n0 + row) * 9 * a.Co + 4 * c) * 4;
        voB2[i] = ((n0 + row) * a.Co2 + 4 * c) * 4;
    }
    // chunk c: source tap (c / cps) in the order th-major over the class's taps, then the downsample range
    auto issue = [&](int c) {
        float* st = smem + (c % S) * STAGE;
        const int src = c / cps, kc = c - src * cps;                  // block-uniform
        if (src >= ntap) {                                            // second gradient tensor (class 0 only)
            const int k2 = c - ntap * cps;
#pragma unroll
            for (int i = 0; i < NA; ++i) ddma16(rg2, (lds_ptr_d)(st + (wave * NA + i) * 256), voA2[i], k2 * 128);
#pragma unroll
            for (int i = 0; i < NB; ++i) ddma16(rw2, (lds_ptr_d)(st + 64 * 32 + (wave * NB + i) * 256), voB2[i], k2 * 128);
            return;
        }
        const int th = src / ntw, tw = src - th * ntw;
        const int kh = pa ? 2 * th : 1, kw = pb ? 2 * tw : 1;
        const int tp = th * 2 + tw;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            lds_ptr_d dst = (lds_ptr_d)(st + (wave * NA + i) * 256);
            const int vo = tp == 0 ? voA[0][i] : (tp == 1 ? voA[1][i] : (tp == 2 ? voA[2][i] : voA[3][i]));
            ddma16(rg, dst, vo, kc * 128);
        }
        const int so = ((kh * 3 + kw) * a.Co + kc * 32) * 4;
#pragma unroll
        for (int i = 0; i < NB; ++i) ddma16(rw, (lds_ptr_d)(st + 64 * 32 + (wave * NB + i) * 256), voB[i], so);
    };

    const int sw = (li >> 1) & 7;
    const float* pa_ = smem + (wm * 32 + li) * 32;
    const float* pb_ = smem + 64 * 32 + (wn * 32 + li) * 32;
    int go[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) go[g] = ((2 * g + lq) ^ sw) * 4;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

#pragma unroll
    for (int c = 0; c < S - 1; ++c)
        if (c < nch) issue(c);
    auto compute = [&](int stage) {
        f32x4 fa[4], fb[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            fa[g] = *(const f32x4*)(pa_ + stage * STAGE + go[g]);
            fb[g] = *(const f32x4*)(pb_ + stage * STAGE + go[g]);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g][u], fb[g][u], acc, 0, 0, 0);
    };
    int c = 0;
    for (; c + S - 1 < nch; ++c) {
        HMF_DVMCNT((S - 2) * L);
        __builtin_amdgcn_s_barrier();
        issue(c + S - 1);
        compute(c % S);
    }
    for (; c < nch; ++c) {
        if (nch - 1 - c >= 1) HMF_DVMCNT(L);
        else HMF_DVMCNT(0);
        __builtin_amdgcn_s_barrier();
        compute(c % S);
    }

    // ---- epilogue: wave-local transpose through LDS, 16-byte rows scattered to the class's pixels (+ residual) ----
    __builtin_amdgcn_s_barrier();
    float* tw_ = smem + wave * (32 * 32);
#pragma unroll
    for (int r = 0; r < 16; ++r) tw_[(4 * lq + (r & 3) + 8 * (r >> 2)) * 32 + li] = acc[r];
    const int col = (lane & 7) * 4, row0 = lane >> 3;                  // 8 lanes per row, 8 rows per instruction, 4 passes
    const int nb = n0 + wn * 32 + col;
    const int gbytes = a.B * a.H * a.W * a.Ci * 4;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)a.gx, 0, gbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.res ? a.res : a.gx), 0, gbytes, 0x00020000);
    int off[4];
    f32x4 rv[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int m = m0 + wm * 32 + row0 + 8 * s;
        const int img = m / (Ac * Bc), r = m - img * (Ac * Bc), ia = r / Bc, ib = r - ia * Bc;
        off[s] = m < Mc ? (((img * a.H + 2 * ia + pa) * a.W + 2 * ib + pb) * a.Ci + nb) * 4 : kOOB;
        rv[s] = a.res ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, off[s], 0, 0)) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        f32x4 v = *(const f32x4*)(tw_ + (row0 + 8 * s) * 32 + col);
        v = v + rv[s];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int, v), ry, off[s], 0, 0);
    }
}

}  // namespace hmf

using namespace hmf;

// gx = dgrad of a 3x3 / stride 2 / pad 1 convolution (+ dgrad of a 1x1 / stride 2 convolution of gy2) (+ residual).  gy (B,OH,OW,Co) NHWC with
// OH = (H - 1) / 2 + 1; wt (Ci, 3, 3, Co) = the parameter (Co, Ci, 3, 3) permuted (1, 2, 3, 0); gy2 (B,OH,OW,Co2) and wt2 (Ci, Co2) = the 1x1 parameter
// (Co2, Ci) transposed, or both NULL; residual / gx (B,H,W,Ci).  Co, Co2 % 32 == 0, Ci % 64 == 0, 16-byte aligned pointers.  Profiler tag 10.
extern "C" int hmf_conv_dgrad_s2(const float* gy, const float* wt, const float* gy2, const float* wt2, const float* residual, float* gx, int B, int H, int W,
                                 int Co, int Co2, int Ci, void* stream) {
    if (!gy || !wt || !gx || B <= 0 || H <= 1 || W <= 1 || (!gy2) != (!wt2)) return fail(HMF_ERR_INVALID, "hmf_conv_dgrad_s2: bad args");
    if (!gy2) Co2 = 0;
    if (Co <= 0 || Co % 32 || Co2 % 32 || Ci <= 0 || Ci % 64) return fail(HMF_ERR_UNSUPPORTED, "hmf_conv_dgrad_s2: Co=%d, Co2=%d must be multiples of 32 and Ci=%d of 64", Co, Co2, Ci);
    if (((uintptr_t)gy & 15) || ((uintptr_t)wt & 15) || ((uintptr_t)gy2 & 15) || ((uintptr_t)wt2 & 15) || ((uintptr_t)residual & 15) || ((uintptr_t)gx & 15))
        return fail(HMF_ERR_INVALID, "hmf_conv_dgrad_s2: pointers must be 16-byte aligned");
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    if ((long long)B * H * W * Ci >= (1LL << 29) || (long long)B * OH * OW * (Co > Co2 ? Co : Co2) >= (1LL << 29) || (long long)Ci * 9 * Co >= (1LL << 29))
        return fail(HMF_ERR_UNSUPPORTED, "hmf_conv_dgrad_s2: tensor exceeds the 32-bit byte offsets of the buffer loads");
    DgradS2Args a;
    a.gy = gy; a.wt = wt; a.gy2 = gy2; a.wt2 = wt2; a.res = residual; a.gx = gx;
    a.B = B; a.OH = OH; a.OW = OW; a.H = H; a.W = W; a.Co = Co; a.Co2 = Co2; a.Ci = Ci;
    const int order[4] = {3, 1, 2, 0};
    int total = 0;
    for (int k = 0; k < 4; ++k) {
        const int pa = order[k] >> 1, pb = order[k] & 1;
        const long long Mc = (long long)B * ((H - pa + 1) / 2) * ((W - pb + 1) / 2);
        a.tiles[k] = total;
        total += (int)((Mc + 63) / 64) * (Ci / 64);
    }
    ProfScope ps(10, B * H * W, (hipStream_t)stream);          // booked with the strided 3x3 family
    hipLaunchKernelGGL(k_conv_dgrad_s2, dim3(total), dim3(256), 0, (hipStream_t)stream, a);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}
