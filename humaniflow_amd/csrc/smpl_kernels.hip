// SMPL shape/pose blend-shapes + linear-blend skinning + 90 joints for gfx950 (MI355X).
//
// Replaces smplx.lbs.lbs / vertices2joints / VertexJointSelector as driven by the reference's
// models/smpl.py:27-41 (call sites evaluate/evaluate_humaniflow.py:124-127,168-172,
// predict/predict_humaniflow.py:128-131,160-163).  Launch sequence of hmf_smpl_lbs:
//   1. k_rigid_chain   rest joints (J_template + Jdirs.beta), 24-joint kinematic chain, per-row skinning
//                      transforms A (M x 24 x 3x4) -> workspace; posed joints -> joints[:, :24]
//   2. k_lbs_mfma      v_posed = [R-I | beta | 1] (M x 224) . Bpack (224 x 3V) on fp32 MFMA 32x32x2,
//                      skinning transform blended from LDS and applied in-register, vertices written once
//   3. k_joints_compact  21 vertex picks + 45 sparse regressed joints from the compact (M, <= 276, 3) copy of the vertices they touch, which the
//                      skinning epilogue of step 2 writes next to the vertices (round 4; k_joints gathers from the vertices themselves: T-pose bodies)
// Data layout and roofline: DESIGN.md "hmf_smpl_lbs".
#include <vector>
#include <utility>
#include <cmath>
#include "hmf_common.h"
#include <type_traits>

namespace hmf {

constexpr int kVT = 216;             // vertex tiles of 32 (6912 >= 6890)
constexpr int kVTperX = 27;          // vertex tiles per XCD-aligned vertex range (216 / 8)
constexpr int kKG = 28;              // k-groups of 4 MFMA k-steps (2 k each): K_pad = 224
constexpr int kKpad = kKG * 8;       // 224 >= 207 + num_betas + 1
constexpr int kPoseFeat = 207;
constexpr int kRowsPerWG = 64;       // two 32-row MFMA tiles per wave
constexpr int kAtr = kJ * 12;        // 288 floats of skinning transforms per row
constexpr int kMaxBetas = 16;

struct SmplDev {
    int num_betas;
    int skin_slots;                  // ELL width: max non-zeros per row of lbs_weights
    int depth_max;
    int parents[kJ];
    int depth[kJ];
    float* blend;                    // packed [kVT][kKG][3][64][4]
    int* skin_off;                   // [kVT][skin_slots][32] byte offset (joint*48) into a row's A block
    float* skin_w;                   // [kVT][skin_slots][32]
    float* j_template;               // 24 x 3
    float* j_dirs;                   // 24 x 3 x nb
    int* jr_ptr;                     // 67 CSR row pointers: 21 picks + 45 regressed joints
    int* jr_idx;
    float* jr_w;
    float* tp_dirs;                  // [nb][3V]
    float* tp_tmpl;                  // [3V]
    unsigned short* blend16;         // bf16 operand of the configs[2] variant: [kVT][kKG16][3][64 lanes][8]
    unsigned short* blend16_lo;      // bf16(x - bf16(x)) of the same operand (impl = 3: hi + lo split, three MFMAs per product)
    // round 6, experiments build only (MEPI: measured not faster): the DENSE skinning weights W[joint][vertex] as the B operand of a matrix-pipe transform blend, bf16 hi | lo:
    // [kVT][piece 2][ 64 lanes x 8 (joints 8 (lane >> 5) + i)  |  64 lanes x 4 (joints 16 + 4 (lane >> 5) + i) ]
    unsigned short* skin_w16;
    // vertex-major transpose of the gathered-joint CSR (backward of joints 24..89): unique vertices and, per vertex, (output joint, weight)
    int n_ju;
    int* ju_vid;                     // [n_ju]
    int* ju_ptr;                     // [n_ju + 1]
    int* ju_out;                     // output joint index 0..65 (joint 24 + i)
    float* ju_w;
    // joints 24..89 without re-reading the vertices: the <= kMaxJU vertices they touch leave the skinning epilogue a second time, into a compact
    // (M, n_ju, 3) buffer, and k_joints_compact finishes the 66 joints from it
    int* ju_slot;                    // [kVT * 32]: compact slot of vertex v, or -1
    int* jr_slot;                    // CSR column of jr_ptr in compact slots (= slot of jr_idx[e])
    // T-pose bodies without materialising them (k_tpose_errors): per vertex the rows [template | shape basis] of its three coordinates,
    // [V][3][tp_stride] (tp_stride = nb + 1 rounded up to 4), and the first / second moments of those rows over the vertices in fp64:
    // tp_mom = mean[3][nb + 1] | gram[(nb + 1)^2]  (T(beta) = T0 + S beta is affine, so mean T and sum |T|^2 are closed forms of beta)
    float* tp_pack;
    double* tp_mom;
    int tp_stride;
    // k_smpl_joints_bwd: the pose / shape / template rows of the <= kMaxJU joint vertices as a dense matrix [kPoseFeat + nb + 1][ju_bstride] (entry
    // u * 3 + c of row k = blend[k][ju_vid[u]][c]): both of its passes stream it instead of gathering from the packed MFMA operand
    float* ju_blend;
    int ju_bstride;
};
constexpr int kMaxJU = 320;          // >= 21 picks + 255 regressor non-zeros of the reference's tables

}  // namespace hmf

struct hmf_smpl {
    hmf::SmplDev d;
    void* arena;
};

namespace hmf {

__host__ __device__ inline size_t blend_index(int k, int v, int c) {
    int vt = v >> 5, l31 = v & 31;
    int ks = k >> 1, half = k & 1;
    int kg = ks >> 2, u = ks & 3;
    int idx = u * 3 + c;
    int lane = half * 32 + l31;
    return ((((size_t)vt * kKG + kg) * 3 + (idx >> 2)) * 64 + lane) * 4 + (idx & 3);
}

// ---- bf16 variant (BASELINE configs[2]: "bf16 MFMA blendshape GEMMs"): v_mfma_f32_32x32x16_bf16, fp32 accumulate ----
// K order as the fp32 operand (207 pose-feature rows | num_betas shape rows), no template row: v_template is added in fp32 after the GEMM
// (a bf16 template would cost 2^-9 of ~1 m).  Lane (j = l & 31, q = l >> 5) of k-group kg holds k = 16 kg + 8 q + e, e = 0..7, for A and B alike.
constexpr int kKG16 = kKpad / 16;                 // 14 MFMA k-groups
constexpr int kW16PerPiece = 64 * 8 + 64 * 4;          // bf16 values of one piece of a vertex tile's dense skinning-weight operand
constexpr int kAtr16Plane = kRowsPerWG * kJ;          // bf16 values of one (piece, element) plane of a row group's transforms: [row 64][joint 24]
constexpr int kAtr16Group = 2 * 12 * kAtr16Plane;      // bf16 values per row group: [piece hi | lo][element 12][row][joint] = 73,728 B, as the fp32 block
constexpr int kP16Stride = kKpad + 8;             // bf16 row stride of the row-major pose-feature panel (464 B = 29 16-byte slots: conflict-free b128 reads)
__host__ __device__ inline unsigned short f32_to_bf16(float x) {
    unsigned int u;
    memcpy(&u, &x, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);              // round to nearest even
    return (unsigned short)(u >> 16);
}
__host__ __device__ inline float bf16_to_f32(unsigned short h) {
    const unsigned int u = (unsigned int)h << 16;
    float x;
    memcpy(&x, &u, 4);
    return x;
}
__host__ __device__ inline size_t blend16_index(int k, int v, int c) {
    return ((((size_t)(v >> 5) * kKG16 + (k >> 4)) * 3 + c) * 64 + (((k >> 3) & 1) * 32 + (v & 31))) * 8 + (k & 7);
}

// k-th entry of the extended pose feature of one row: [vec(R_1..R_23) - I | beta | 1 | 0...]
__device__ __forceinline__ float pose_feature(int k, int nb, const float* __restrict__ body_row,
                                              const float* __restrict__ beta_row) {
    if (k < kPoseFeat) {
        int e = k % 9;
        return body_row[k] - ((e == 0 || e == 4 || e == 8) ? 1.0f : 0.0f);
    }
    if (k < kPoseFeat + nb) return beta_row[k - kPoseFeat];
    return (k == kPoseFeat + nb) ? 1.0f : 0.0f;
}

// ---------------------------------------------------------------------------------------------------
// 1. rigid chain: block = 8 rows x 24 joints, level-synchronous over the kinematic tree depth.
//    smplx lbs.batch_rigid_transform: G_i = G_parent @ [[R_i, J_i - J_parent],[0,1]], A_i = [G_R | G_t - G_R J_i].
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(192) void k_rigid_chain(SmplDev d, const float* __restrict__ glob, int Mg,
                                                     const float* __restrict__ body, const float* __restrict__ betas,
                                                     int Mb, int M, float* __restrict__ atr, float* __restrict__ joints,
                                                     float* __restrict__ panel, unsigned short* __restrict__ panel16,
                                                     unsigned short* __restrict__ panel16_lo, int gN, float* __restrict__ full_pose,
                                                     unsigned short* __restrict__ atr16) {
    __shared__ float sJ[8][kJ][3];
    __shared__ float sG[8][kJ][12];
    const int j = threadIdx.x % kJ, r = threadIdx.x / kJ;
    const int row = blockIdx.x * 8 + r;
    const bool live = row < M;
    const int nb = d.num_betas;
    float Jc[3] = {0.f, 0.f, 0.f};
    float R[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
    if (live) {
        const float* b = betas + (size_t)(Mb == 1 ? 0 : row) * nb;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float acc = d.j_template[j * 3 + c];
            for (int k = 0; k < nb; ++k) acc = fmaf(d.j_dirs[(j * 3 + c) * nb + k], b[k], acc);
            Jc[c] = acc;
        }
        if (j == 0) {
            if (glob) {
                // gN > 0: the packed rows of one HumaniflowModel.forward -- samples b * gN + n, then the point-estimate rows -- share image b's rotation
                const int grow = Mg == 1 ? 0 : (gN > 0 ? (row < Mg * gN ? row / gN : row - Mg * gN) : row);
                const float* g = glob + (size_t)grow * 9;
#pragma unroll
                for (int e = 0; e < 9; ++e) R[e] = g[e];
            }
        } else if (body) {
            const float* p = body + ((size_t)row * 23 + (j - 1)) * 9;
#pragma unroll
            for (int e = 0; e < 9; ++e) R[e] = p[e];
        }
    }
    // extended pose feature [vec(R_1..R_23) - I | beta | 1 | 0] of this row, written in the A-operand panel layout
    // panel[row / 64][k][row % 64] that k_lbs_mfma stages into LDS with plain coalesced float4 copies (dead rows: zeros)
    if (panel) {
        float* pr = panel + (size_t)(row >> 6) * (kKpad * 64) + (row & 63);
        if (j > 0) {
#pragma unroll
            for (int e = 0; e < 9; ++e)
                pr[((j - 1) * 9 + e) * 64] = live ? R[e] - ((e == 0 || e == 4 || e == 8) ? 1.0f : 0.0f) : 0.0f;
        } else {
            const float* b = betas + (size_t)((Mb == 1 || !live) ? 0 : row) * nb;
            for (int k = kPoseFeat; k < kKpad; ++k)
                pr[k * 64] = !live ? 0.0f : (k < kPoseFeat + nb ? b[k - kPoseFeat] : (k == kPoseFeat + nb ? 1.0f : 0.0f));
        }
    }
    if (panel16) {                                  // row-major bf16 copy of the same features (no constant-1 column); impl = 3: + the bf16 remainders
        unsigned short* pr = panel16 + (size_t)row * kP16Stride;
        unsigned short* pl = panel16_lo ? panel16_lo + (size_t)row * kP16Stride : nullptr;
        auto put = [&](int k, float x) {
            const unsigned short h = f32_to_bf16(x);
            pr[k] = h;
            if (pl) pl[k] = f32_to_bf16(x - bf16_to_f32(h));
        };
        if (j > 0) {
#pragma unroll
            for (int e = 0; e < 9; ++e) put((j - 1) * 9 + e, live ? R[e] - ((e == 0 || e == 4 || e == 8) ? 1.0f : 0.0f) : 0.f);
        } else {
            const float* b = betas + (size_t)((Mb == 1 || !live) ? 0 : row) * nb;
            for (int k = kPoseFeat; k < kP16Stride; ++k) put(k, (live && k < kPoseFeat + nb) ? b[k - kPoseFeat] : 0.f);
        }
    }
    sJ[r][j][0] = Jc[0]; sJ[r][j][1] = Jc[1]; sJ[r][j][2] = Jc[2];
    __syncthreads();
    const int par = d.parents[j];
    float rel[3] = {Jc[0], Jc[1], Jc[2]};
    if (j > 0) {
        rel[0] -= sJ[r][par][0]; rel[1] -= sJ[r][par][1]; rel[2] -= sJ[r][par][2];
    }
    float G[12];
    const int my_depth = d.depth[j];
    for (int lev = 0; lev <= d.depth_max; ++lev) {
        if (my_depth == lev) {
            if (j == 0) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    G[i * 4 + 0] = R[i * 3 + 0]; G[i * 4 + 1] = R[i * 3 + 1]; G[i * 4 + 2] = R[i * 3 + 2];
                    G[i * 4 + 3] = rel[i];
                }
            } else {
                float P[12];
#pragma unroll
                for (int e = 0; e < 12; ++e) P[e] = sG[r][par][e];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        float a = P[i * 4 + 0] * R[0 * 3 + c];
                        a = fmaf(P[i * 4 + 1], R[1 * 3 + c], a);
                        a = fmaf(P[i * 4 + 2], R[2 * 3 + c], a);
                        G[i * 4 + c] = a;
                    }
                    float t = P[i * 4 + 0] * rel[0];
                    t = fmaf(P[i * 4 + 1], rel[1], t);
                    t = fmaf(P[i * 4 + 2], rel[2], t);
                    G[i * 4 + 3] = t + P[i * 4 + 3];
                }
            }
#pragma unroll
            for (int e = 0; e < 12; ++e) sG[r][j][e] = G[e];
        }
        __syncthreads();
    }
    // round 6 (impl = 3): the same 3 x 4 transforms as bf16 hi | lo planes in the A-operand order of k_lbs_mfma16's transform blend,
    // [row group][piece][element 12][row 64][joint 24]; dead rows of the last group are zeros (the grid is padded to whole groups)
    float A12[12];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float t = G[i * 4 + 0] * Jc[0];
        t = fmaf(G[i * 4 + 1], Jc[1], t);
        t = fmaf(G[i * 4 + 2], Jc[2], t);
        A12[i * 4 + 0] = G[i * 4 + 0]; A12[i * 4 + 1] = G[i * 4 + 1]; A12[i * 4 + 2] = G[i * 4 + 2];
        A12[i * 4 + 3] = G[i * 4 + 3] - t;
    }
    if (atr16) {
        unsigned short* g16 = atr16 + (size_t)(row >> 6) * kAtr16Group + (row & 63) * kJ + j;
#pragma unroll
        for (int e = 0; e < 12; ++e) {
            const float x = live ? A12[e] : 0.f;
            const unsigned short h = f32_to_bf16(x);
            g16[e * kAtr16Plane] = h;
            g16[(12 + e) * kAtr16Plane] = f32_to_bf16(x - bf16_to_f32(h));
        }
    }
    if (!live) return;
    if (full_pose) {                                // smplx SMPL.forward: full_pose = cat([global_orient, body_pose], dim=1)
        float* o = full_pose + ((size_t)row * kJ + j) * 9;
#pragma unroll
        for (int e = 0; e < 9; ++e) o[e] = R[e];
    }
    if (joints) {
        float* o = joints + ((size_t)row * kOutJ + j) * 3;
        o[0] = G[3]; o[1] = G[7]; o[2] = G[11];
    }
    float* a = atr + ((size_t)row * kJ + j) * 12;
#pragma unroll
    for (int e = 0; e < 12; ++e) a[e] = A12[e];
}

typedef __attribute__((address_space(3))) void* lbs_lds_ptr_t;
// one LDS-DMA request: 64 lanes x 16 bytes, lane l's bytes from (rsrc base + voff + soff) land at dst + 16 l (non-template helper: see conv_gemm.hip dma16)
static __device__ __forceinline__ void lbs_dma16(__amdgpu_buffer_rsrc_t rsrc, lbs_lds_ptr_t dst, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voff, soff, 0, 0);
}

// Per-tile constants of the skinning epilogue of k_lbs_mfma.
struct EpiCtx {
    float* obase;                    // verts + ((row0 + 4*half) * V + v) * 3
    bool ok_v, full_rows;
    int row_lim;                     // M - row0 - 4*half
    float* cj;                       // compact joint-vertex buffer + ((row0 + 4*half) * n_ju + slot of this lane's vertex) * 3, or null
    int cj_row;                      // n_ju * 3
    bool tile_has_ju;                // wave-uniform: some vertex of this tile feeds a gathered joint
    bool lane_has_ju;
};

template <int SLOTS>
__device__ __forceinline__ void skin_element(const SmplDev& d, const char* __restrict__ lbase, const int (&off)[SLOTS > 0 ? SLOTS : 1],
                                             const float (&w)[SLOTS > 0 ? SLOTS : 1], int nslots, int vt, int l31,
                                             const f32x16 (&acc)[2][3], int rt, int rt_row, int reg, const EpiCtx& e) {
    const int rc = rt_row * 32 + (reg & 3) + 8 * (reg >> 2);      // rt: accumulator set, rt_row: which 32 rows of the group it holds
    const char* base = lbase + rc * kAtr * 4;
    f32x4 T0 = {0.f, 0.f, 0.f, 0.f}, T1 = T0, T2 = T0;
    if (SLOTS > 0) {
        f32x4 ga[SLOTS > 0 ? SLOTS : 1][3];
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const f32x4* p = (const f32x4*)(base + off[s]);
            ga[s][0] = p[0]; ga[s][1] = p[1]; ga[s][2] = p[2];
        }
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) { T0 += w[s] * ga[s][0]; T1 += w[s] * ga[s][1]; T2 += w[s] * ga[s][2]; }
    } else {
        for (int s = 0; s < nslots; ++s) {
            const int o = d.skin_off[((size_t)vt * nslots + s) * 32 + l31];
            const float ww = d.skin_w[((size_t)vt * nslots + s) * 32 + l31];
            const f32x4* p = (const f32x4*)(base + o);
            T0 += ww * p[0]; T1 += ww * p[1]; T2 += ww * p[2];
        }
    }
    const float px = acc[rt][0][reg], py = acc[rt][1][reg], pz = acc[rt][2][reg];
    float x = fmaf(T0[2], pz, fmaf(T0[1], py, T0[0] * px)) + T0[3];
    float y = fmaf(T1[2], pz, fmaf(T1[1], py, T1[0] * px)) + T1[3];
    float z = fmaf(T2[2], pz, fmaf(T2[1], py, T2[0] * px)) + T2[3];
    asm volatile("" : "+v"(x), "+v"(y), "+v"(z));        // keep the body OUT of the exec-masked store block (otherwise the compiler sinks each body into its own exec-masked block and serialises 32 x 4 LDS round trips per tile)
    if (e.ok_v && (e.full_rows || rc < e.row_lim)) {
        float* o = e.obase + rc * (kV * 3);
        // streaming stores: the 267 MB of vertices are written once and never re-read by this kernel; keeping them out of the way of the
        // 2.3 MB operand range that each XCD's L2 is supposed to hold (r01 PMC: the operand was re-fetched ~9x)
        // ONE 12-byte store per lane: the 32 vertices of a row are then one dense 384-byte run per instruction.  (Three dword stores at a 12-byte lane
        // stride left the three partial passes over every line to the write combiner: PMC counted 318 MB written for 267 MB of vertices.)
        typedef float f32x3u __attribute__((ext_vector_type(3), aligned(4)));
        __builtin_nontemporal_store(f32x3u{x, y, z}, (f32x3u*)o);
    }
    if (e.tile_has_ju) {                                  // wave-uniform; ~150 of the 216 vertex tiles hold one of the <= 276 joint vertices
        if (e.lane_has_ju && (e.full_rows || rc < e.row_lim)) {
            float* q = e.cj + rc * e.cj_row;
            q[0] = x; q[1] = y; q[2] = z;
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// 2a. The same kernel on 32-row groups (round 5, the round-4 review's "<= 64 KB LDS" variant): pose-feature panel [224][32] (28 KB) + 32 x 24 skinning
//     transforms (36 KB) = 64 KB per work-group, so TWO work-groups (four waves per SIMD) share a CU -- or one leaves room for another batch's convolution
//     work-groups.  A wave tile is 32 rows x 32 vertices x 3 components (one accumulator set): every operand fragment streamed from L2 now serves 32 rows
//     instead of 64.  Reads the same 64-row panels (half of one) and the same packed operand as k_lbs_mfma; same arithmetic per (row, vertex): bitwise the
//     same vertices.  SMPL(lbs_rows=32) / hmf_smpl_lbs impl = 4.
// ---------------------------------------------------------------------------------------------------
template <int SLOTS>
__global__ __launch_bounds__(512, 4) void k_lbs_mfma32(SmplDev d, int M, int nrg32, int wpx, const float* __restrict__ panel,
                                                     const float* __restrict__ atr, float* __restrict__ verts, float* __restrict__ cjv) {
    constexpr int kR = 32;
    __shared__ __attribute__((aligned(16))) float smem[kKpad * kR + kR * kAtr];      // 28 + 36 KB
    float* sPF = smem;
    float* sA = smem + kKpad * kR;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int xr = blockIdx.x & 7;
    const int chunk = blockIdx.x >> 3;
    const int T = nrg32 * kVTperX;
    int t = (int)((long long)chunk * T / wpx);
    const int t1 = (int)((long long)(chunk + 1) * T / wpx);
    const int nslots = d.skin_slots;
    const int ks_valid = (kPoseFeat + d.num_betas + 2) / 2;
    while (t < t1) {
        const int rg = t / kVTperX;                       // 32-row group
        const int seg_end = min(t1, (rg + 1) * kVTperX);
        const int row0 = rg * kR;
        {
            // rows [32 (rg & 1), +32) of the 64-row panel rg >> 1: element (k, r) at k * 64 + r
            const f32x4* src = (const f32x4*)(panel + (size_t)(rg >> 1) * (kKpad * 64)) + (rg & 1) * 8;
            f32x4* dst = (f32x4*)sPF;
            for (int i = tid; i < kKpad * kR / 4; i += 512) dst[i] = src[(i >> 3) * 16 + (i & 7)];
            const f32x4* src2 = (const f32x4*)(atr + (size_t)row0 * kAtr);
            f32x4* dst2 = (f32x4*)sA;
            const int n4 = min(kR, M - row0) * (kAtr / 4);
            for (int ii = tid; ii < kR * (kAtr / 4); ii += 512) {
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
                dst2[ii] = (ii < n4) ? src2[ii] : z;
            }
        }
        __syncthreads();
        for (int tt = t + wave; tt < seg_end; tt += 8) {
            const int vt = xr * kVTperX + (tt - rg * kVTperX);
            int off[SLOTS > 0 ? SLOTS : 1];
            float w[SLOTS > 0 ? SLOTS : 1];
            if (SLOTS > 0) {
#pragma unroll
                for (int sl = 0; sl < SLOTS; ++sl) {
                    off[sl] = d.skin_off[((size_t)vt * nslots + sl) * 32 + l31];
                    w[sl] = d.skin_w[((size_t)vt * nslots + sl) * 32 + l31];
                }
            }
            const int jslot = cjv ? d.ju_slot[vt * 32 + l31] : -1;
            f32x16 acc[2][3];
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int e = 0; e < 16; ++e) { acc[0][c][e] = 0.f; acc[1][c][e] = 0.f; }
            const f32x4* bp = (const f32x4*)(d.blend + (size_t)vt * kKG * 3 * 64 * 4) + lane;
            const float* ap = sPF + half * kR + l31;
            f32x4 b0 = bp[0], b1 = bp[64], b2 = bp[128];
            float a0[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) a0[u] = ap[u * 2 * kR];
#pragma unroll 2
            for (int kg = 0; kg < kKG; ++kg) {
                f32x4 n0 = b0, n1 = b1, n2 = b2;
                float m0[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) m0[u] = a0[u];
                if (kg + 1 < kKG) {
                    n0 = bp[(kg + 1) * 192]; n1 = bp[(kg + 1) * 192 + 64]; n2 = bp[(kg + 1) * 192 + 128];
#pragma unroll
                    for (int u = 0; u < 4; ++u) m0[u] = ap[((kg + 1) * 4 + u) * 2 * kR];
                }
                const float bv[12] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3], b2[0], b2[1], b2[2], b2[3]};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int ks = kg * 4 + u;
                    if (ks >= ks_valid) continue;                                   // wave-uniform
#pragma unroll
                    for (int c = 0; c < 3; ++c) acc[0][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], bv[u * 3 + c], acc[0][c], 0, 0, 0);
                }
                b0 = n0; b1 = n1; b2 = n2;
#pragma unroll
                for (int u = 0; u < 4; ++u) a0[u] = m0[u];
            }
            EpiCtx e;
            const int v = vt * 32 + l31;
            e.ok_v = v < kV;
            e.full_rows = row0 + kR <= M;
            e.row_lim = M - row0 - 4 * half;
            e.obase = verts + ((size_t)(row0 + 4 * half) * kV + v) * 3;
            e.lane_has_ju = jslot >= 0;
            e.tile_has_ju = __builtin_amdgcn_ballot_w64(e.lane_has_ju) != 0;
            e.cj_row = d.n_ju * 3;
            e.cj = cjv + ((size_t)(row0 + 4 * half) * d.n_ju + (jslot >= 0 ? jslot : 0)) * 3;
            const char* lbase = (const char*)(sA + 4 * half * kAtr);
#pragma unroll
            for (int el = 0; el < 16; ++el) skin_element<SLOTS>(d, lbase, off, w, nslots, vt, l31, acc, 0, 0, el, e);
        }
        __syncthreads();
        t = seg_end;
    }
}

// ---------------------------------------------------------------------------------------------------
// 2. fused pose-blend GEMM + skinning.  One workgroup (8 waves, 2 per SIMD) per CU; it owns a contiguous chunk of
//    (row-group of 64 rows, vertex tile) pairs inside ONE of the 8 vertex ranges (27 tiles = 2.3 MB of packed operand,
//    L2-resident per XCD when block b lands on XCD b%8 -- speed only).  Per wave tile: D(64 rows x 32 verts x 3 comps) =
//    112 k-steps x 6 v_mfma_f32_32x32x2_f32.  LDS: pose-feature panel [224][64] (A operand, conflict-free ds_read_b32,
//    pre-laid-out by k_rigid_chain so staging is a coalesced float4 copy) + skinning transforms [64][24][12]
//    (ds_read_b128 gather by joint).  The B operand streams L2 -> VGPR as dwordx4, one k-group ahead; the next tile's
//    first k-group and skinning slots are fetched before the current tile's epilogue.
//    ABL != 0: ablation builds (tools/lbs_tune.py) -- 1 no epilogue, 2 no B stream, 4 no A reads.
//    Measured on MI355X at M = 3200: bare MFMA loop 260 us, + operands 285 us, + epilogue 328 us.
// ---------------------------------------------------------------------------------------------------
template <int SLOTS, int ABL>
__global__ __launch_bounds__(512, 2) void k_lbs_mfma(SmplDev d, int M, int nrg, int wpx, const float* __restrict__ panel,
                                                   const float* __restrict__ atr, float* __restrict__ verts,
                                                   long long* __restrict__ cycles, float* __restrict__ cjv) {
    const long long c_start = cycles ? (long long)__builtin_readcyclecounter() : 0;
    __shared__ __attribute__((aligned(16))) float smem[kKpad * kRowsPerWG + kRowsPerWG * kAtr];  // 56 + 72 KB
    float* sPF = smem;
    float* sA = smem + kKpad * kRowsPerWG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int xr = blockIdx.x & 7;                 // vertex range (XCD-affine for L2 residency; speed only)
    const int chunk = blockIdx.x >> 3;
    const int T = nrg * kVTperX;
    int t = (int)((long long)chunk * T / wpx);
    const int t1 = (int)((long long)(chunk + 1) * T / wpx);
    const int nslots = d.skin_slots;
    const int ks_valid = (kPoseFeat + d.num_betas + 2) / 2;      // k-steps that carry data; the zero padding up to 112 is skipped
    while (t < t1) {
        const int rg = t / kVTperX;
        const int seg_end = min(t1, (rg + 1) * kVTperX);
        const int row0 = rg * kRowsPerWG;
        {
            const f32x4* src = (const f32x4*)(panel + (size_t)rg * (kKpad * 64));
            f32x4* dst = (f32x4*)sPF;
#pragma unroll
            // (round 4, measured and rejected: streaming / non-temporal loads for this once-per-segment copy, meant to keep the 2.3 MB operand range
            // of the XCD in its L2: +15 us at M = 3,232 (tools/lbs_tune.py 0 8) -- the copy sits on the critical path of every segment and the
            // non-temporal path is slower to first data; ABL & 8 keeps the variant in the experiments build)
            for (int i = 0; i < kKpad * 64 / 4 / 512; ++i) dst[tid + i * 512] = (ABL & 8) ? __builtin_nontemporal_load(src + tid + i * 512) : src[tid + i * 512];
            if (!(ABL & 32)) {
                const f32x4* src2 = (const f32x4*)(atr + (size_t)row0 * kAtr);
                f32x4* dst2 = (f32x4*)sA;
                const int n4 = min(kRowsPerWG, M - row0) * (kAtr / 4);
#pragma unroll
                for (int i = 0; i < kRowsPerWG * (kAtr / 4) / 512; ++i) {
                    const int ii = tid + i * 512;
                    f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    dst2[ii] = (ii < n4) ? ((ABL & 8) ? __builtin_nontemporal_load(src2 + ii) : src2[ii]) : z;
                }
            }
        }
        __syncthreads();
        // ABL & 32 (round 6, k_lbs_mfma16's ADMA): the skinning transforms by LDS-DMA under the segment's first main loops, one extra barrier before the first epilogue
        bool atr_pending = false;
        if (ABL & 32) {
            const int live = min(kRowsPerWG, M - row0) * kAtr * 4;
            const __amdgpu_buffer_rsrc_t ratr = __builtin_amdgcn_make_buffer_rsrc((void*)(atr + (size_t)row0 * kAtr), 0, live, 0x00020000);
            constexpr int kReq = kRowsPerWG * kAtr * 4 / 1024 / 8;
#pragma unroll
            for (int i = 0; i < kReq; ++i)
                lbs_dma16(ratr, (lbs_lds_ptr_t)((char*)sA + (wave * kReq + i) * 1024), lane * 16, (wave * kReq + i) * 1024);
            atr_pending = true;
        }
        // Work items of this wave inside the segment: full rounds of 8 vertex tiles (64 rows x 32 vertices per wave), then the partial round.
        // A partial round of <= 4 tiles is split by rows: two waves share a vertex tile, 32 rows each, so the 27 tiles of a range cost
        // 3.5 rounds instead of 4 (the B operand of those tiles is fetched by both waves).
        // Round 3: the half tiles are also what de-phases the two waves of a SIMD.  Waves w and w + 4 share a SIMD (a work-group's waves are
        // placed on SIMDs 0,2,1,3,0,2,1,3); run in lock step they reach their skinning epilogues -- ~40 % of a tile's time, no MFMA -- together
        // and the matrix pipe idles (MfmaUtil 61 % in round 2).  So the <= 4 split tiles of a segment come FIRST for waves 4-7 (up to four
        // half tiles) and LAST for waves 0-3: from then on a SIMD's two waves are half a tile apart, one in its MFMA loop while the other skins.
        const int cnt = seg_end - t, rf = cnt >> 3, rem = cnt & 7;
        const bool split = rem > 0 && rem <= 4;
        const int nB = split ? min(2 * rem, 4) : 0, nA = split ? 2 * rem - nB : 0;
        const bool lead = wave >= 4 && (wave - 4) < nB;          // this wave starts with a half tile
        auto item = [&](int k, int& rh) -> int {
            rh = -1;
            if (split) {
                if (lead) {
                    if (k == 0) { rh = (wave - 4) & 1; return t + ((wave - 4) >> 1); }
                    --k;
                }
                if (k < rf) return t + rem + k * 8 + wave;
                if (k == rf && wave < 4 && wave < nA) { rh = (nB + wave) & 1; return t + ((nB + wave) >> 1); }
                return -1;
            }
            if (k < rf) return t + k * 8 + wave;
            if (k == rf && rem > 0) return wave < rem ? t + rf * 8 + wave : -1;
            return -1;
        };
        int kitem = 0, rh;
        int tt = item(0, rh);
        f32x4 b0, b1, b2;
        int off[SLOTS > 0 ? SLOTS : 1];
        float w[SLOTS > 0 ? SLOTS : 1];
        int jslot = -1;
        if (tt >= 0) {
            const int vt = xr * kVTperX + (tt - rg * kVTperX);
            const f32x4* bp = (const f32x4*)(d.blend + (size_t)vt * kKG * 3 * 64 * 4) + lane;
            b0 = bp[0]; b1 = bp[64]; b2 = bp[128];
            if (cjv) jslot = d.ju_slot[vt * 32 + l31];
            if (SLOTS > 0) {
#pragma unroll
                for (int s = 0; s < SLOTS; ++s) {
                    off[s] = d.skin_off[((size_t)vt * nslots + s) * 32 + l31];
                    w[s] = d.skin_w[((size_t)vt * nslots + s) * 32 + l31];
                }
            }
        }
        while (tt >= 0) {
            const int vt = xr * kVTperX + (tt - rg * kVTperX);
            f32x16 acc[2][3];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[a][c][e] = 0.f;
            const f32x4* bp = (const f32x4*)(d.blend + (size_t)vt * kKG * 3 * 64 * 4) + lane;
            const float* ap = sPF + half * 64 + l31 + (rh > 0 ? 32 : 0);      // half tile: this wave's 32 rows feed accumulator set 0
            // A operands (LDS) of k-group 0; inside the loop both operand streams run one k-group ahead of the MFMAs, so a
            // wave never waits on the LDS queue behind its neighbours' epilogue gathers
            float a0[4], a1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { a0[u] = ap[u * 128]; a1[u] = ap[u * 128 + 32]; }
#pragma unroll 2
            for (int kg = 0; kg < kKG; ++kg) {
                f32x4 n0 = b0, n1 = b1, n2 = b2;
                float m0[4], m1[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { m0[u] = a0[u]; m1[u] = a1[u]; }
                if (kg + 1 < kKG && !(ABL & 2)) {
                    n0 = bp[(kg + 1) * 192]; n1 = bp[(kg + 1) * 192 + 64]; n2 = bp[(kg + 1) * 192 + 128];
                }
                if (kg + 1 < kKG && !(ABL & 4)) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) { m0[u] = ap[((kg + 1) * 4 + u) * 128]; m1[u] = ap[((kg + 1) * 4 + u) * 128 + 32]; }
                }
                // round 6: the next k-group's requests stay ahead of this group's MFMAs (k_lbs_mfma16's main lever).  Here 24 fp32 MFMAs = 1,536 cycles per
                // k-group cover the L2 round trip either way: measured neutral (two boxes: 340 -> 328 us and 342 -> 345 us at M = 3,232; ABL & 16 = unpinned)
                if (!(ABL & 16)) __builtin_amdgcn_sched_barrier(0);
                const float bv[12] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3], b2[0], b2[1], b2[2], b2[3]};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int ks = kg * 4 + u;
                    if (ks >= ks_valid) continue;                                   // wave-uniform
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        acc[0][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], bv[u * 3 + c], acc[0][c], 0, 0, 0);
                        if (rh < 0) acc[1][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], bv[u * 3 + c], acc[1][c], 0, 0, 0);
                    }
                }
                b0 = n0; b1 = n1; b2 = n2;
#pragma unroll
                for (int u = 0; u < 4; ++u) { a0[u] = m0[u]; a1[u] = m1[u]; }
            }
            // next tile's first operand group + skinning slots: in flight during this tile's epilogue
            int nrh;
            const int ntt = item(kitem + 1, nrh);
            int noff[SLOTS > 0 ? SLOTS : 1];
            float nw[SLOTS > 0 ? SLOTS : 1];
            int njslot = -1;
            if (ntt >= 0) {
                const int nvt = xr * kVTperX + (ntt - rg * kVTperX);
                const f32x4* nbp = (const f32x4*)(d.blend + (size_t)nvt * kKG * 3 * 64 * 4) + lane;
                b0 = nbp[0]; b1 = nbp[64]; b2 = nbp[128];
                if (cjv) njslot = d.ju_slot[nvt * 32 + l31];
                if (SLOTS > 0) {
#pragma unroll
                    for (int s = 0; s < SLOTS; ++s) {
                        noff[s] = d.skin_off[((size_t)nvt * nslots + s) * 32 + l31];
                        nw[s] = d.skin_w[((size_t)nvt * nslots + s) * 32 + l31];
                    }
                }
            }
            if (ABL & 1) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int c = 0; c < 3; ++c) asm volatile("" ::"v"(acc[a][c]));
            } else {
                if ((ABL & 32) && atr_pending) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    atr_pending = false;
                }
                EpiCtx e;
                const int v = vt * 32 + l31;
                e.ok_v = v < kV;
                e.full_rows = row0 + kRowsPerWG <= M;
                // half tile: accumulator set 0 holds rows 32 rh .. 32 rh + 31 -> shift the row bases, run the first 16 elements of the same code
                const int rsh = rh > 0 ? 32 : 0;
                e.row_lim = M - row0 - 4 * half - rsh;
                e.obase = verts + ((size_t)(row0 + 4 * half + rsh) * kV + v) * 3;
                e.lane_has_ju = jslot >= 0;
                e.tile_has_ju = __builtin_amdgcn_ballot_w64(e.lane_has_ju) != 0;
                e.cj_row = d.n_ju * 3;
                e.cj = cjv + ((size_t)(row0 + 4 * half + rsh) * d.n_ju + (jslot >= 0 ? jslot : 0)) * 3;
                const char* lbase = (const char*)(sA + (4 * half + rsh) * kAtr);
#pragma unroll
                for (int el = 0; el < 32; ++el) {
                    if (el == 16 && rh >= 0) break;                       // wave-uniform
                    skin_element<SLOTS>(d, lbase, off, w, nslots, vt, l31, acc, el >> 4, el >> 4, el & 15, e);
                }
            }
            if (SLOTS > 0) {
#pragma unroll
                for (int s = 0; s < SLOTS; ++s) { off[s] = noff[s]; w[s] = nw[s]; }
            }
            jslot = njslot;
            tt = ntt;
            rh = nrh;
            ++kitem;
        }
        if ((ABL & 32) && atr_pending) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        __syncthreads();
        t = seg_end;
    }
    if (cycles && threadIdx.x == 0) cycles[blockIdx.x] = (long long)__builtin_readcyclecounter() - c_start;
}

// Scalar cross-check kernel (tests only): one thread per (row, vertex), same k order and the same
// skinning op order as the MFMA kernel, so both must agree bit for bit.
__global__ void k_lbs_scalar(SmplDev d, const float* __restrict__ body, const float* __restrict__ betas, int Mb,
                             int M, const float* __restrict__ atr, float* __restrict__ verts) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)M * kV) return;
    const int row = (int)(gid / kV), v = (int)(gid % kV);
    const int nb = d.num_betas;
    const float* brow = body + (size_t)row * kPoseFeat;
    const float* beta = betas + (size_t)(Mb == 1 ? 0 : row) * nb;
    float p[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < kKpad; ++k) {
        const float a = pose_feature(k, nb, brow, beta);
#pragma unroll
        for (int c = 0; c < 3; ++c) p[c] = fmaf(a, d.blend[blend_index(k, v, c)], p[c]);
    }
    const int vt = v >> 5, l31 = v & 31;
    f32x4 T0 = {0.f, 0.f, 0.f, 0.f}, T1 = T0, T2 = T0;
    for (int s = 0; s < d.skin_slots; ++s) {
        const int o = d.skin_off[((size_t)vt * d.skin_slots + s) * 32 + l31];
        const float w = d.skin_w[((size_t)vt * d.skin_slots + s) * 32 + l31];
        const f32x4* q = (const f32x4*)((const char*)(atr + (size_t)row * kAtr) + o);
        T0 += w * q[0]; T1 += w * q[1]; T2 += w * q[2];
    }
    float* o = verts + ((size_t)row * kV + v) * 3;
    o[0] = fmaf(T0[2], p[2], fmaf(T0[1], p[1], T0[0] * p[0])) + T0[3];
    o[1] = fmaf(T1[2], p[2], fmaf(T1[1], p[1], T1[0] * p[0])) + T1[3];
    o[2] = fmaf(T2[2], p[2], fmaf(T2[1], p[1], T2[0] * p[0])) + T2[3];
}

// ---------------------------------------------------------------------------------------------------
// 3. joints 24..89: vertex picks (weight 1) and the three sparse extra regressors (models/smpl.py:30-34)
// ---------------------------------------------------------------------------------------------------
__global__ void k_joints(SmplDev d, const float* __restrict__ verts, int M, float* __restrict__ joints) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int nextra = kOutJ - kJ;
    if (gid >= M * nextra) return;
    const int row = gid / nextra, i = gid % nextra;
    const float* vr = verts + (size_t)row * kV * 3;
    float x = 0.f, y = 0.f, z = 0.f;
    for (int e = d.jr_ptr[i]; e < d.jr_ptr[i + 1]; ++e) {
        const float w = d.jr_w[e];
        const float* p = vr + (size_t)d.jr_idx[e] * 3;
        x = fmaf(w, p[0], x); y = fmaf(w, p[1], y); z = fmaf(w, p[2], z);
    }
    float* o = joints + ((size_t)row * kOutJ + kJ + i) * 3;
    o[0] = x; o[1] = y; o[2] = z;
}

// The same 66 joints from the compact (M, n_ju, 3) copy of the <= 276 vertices they touch, written by the skinning epilogue: the same fma chain
// per joint as k_joints (bit-identical results), 3.3 KB per body instead of a gather over its 82.7 KB of vertices.
__global__ void k_joints_compact(SmplDev d, const float* __restrict__ cjv, int M, float* __restrict__ joints) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int nextra = kOutJ - kJ;
    if (gid >= M * nextra) return;
    const int row = gid / nextra, i = gid % nextra;
    const float* vr = cjv + (size_t)row * d.n_ju * 3;
    float x = 0.f, y = 0.f, z = 0.f;
    for (int e = d.jr_ptr[i]; e < d.jr_ptr[i + 1]; ++e) {
        const float w = d.jr_w[e];
        const float* p = vr + (size_t)d.jr_slot[e] * 3;
        x = fmaf(w, p[0], x); y = fmaf(w, p[1], y); z = fmaf(w, p[2], z);
    }
    float* o = joints + ((size_t)row * kOutJ + kJ + i) * 3;
    o[0] = x; o[1] = y; o[2] = z;
}

// ---------------------------------------------------------------------------------------------------
// T-pose bodies: v_template + shapedirs . beta.  HBM-write bound: 8 B per lane per row, the 2*nb basis
// values of a lane's two coordinates live in registers across the row loop.
// ---------------------------------------------------------------------------------------------------
constexpr int kTpRows = 16;
__global__ __launch_bounds__(256) void k_tpose(SmplDev d, const float* __restrict__ betas, int M,
                                               float* __restrict__ verts) {
    const int pair = blockIdx.x * 256 + threadIdx.x;       // coordinate pair index in [0, 3V/2)
    const int row0 = blockIdx.y * kTpRows;
    if (pair >= kV * 3 / 2) return;
    const int nb = d.num_betas;
    float2 base = ((const float2*)d.tp_tmpl)[pair];
    float2 sd[kMaxBetas];
#pragma unroll
    for (int k = 0; k < kMaxBetas; ++k)
        if (k < nb) sd[k] = ((const float2*)(d.tp_dirs + (size_t)k * kV * 3))[pair];
    const int rend = min(row0 + kTpRows, M);
    for (int row = row0; row < rend; ++row) {
        const float* b = betas + (size_t)row * nb;       // wave-uniform -> scalar loads
        float2 o = base;
#pragma unroll
        for (int k = 0; k < kMaxBetas; ++k)
            if (k < nb) { o.x = fmaf(sd[k].x, b[k], o.x); o.y = fmaf(sd[k].y, b[k], o.y); }
        ((float2*)(verts + (size_t)row * kV * 3))[pair] = o;
    }
}

// ---------------------------------------------------------------------------------------------------
// k_tpose_errors: the two T-pose metrics (PVE-T, PVE-T-SC: metrics/eval_metrics_tracker.py:200-222 via utils/eval_utils.py:107-125) of a
// (prediction betas, target betas) pair WITHOUT the T-pose bodies ever existing in HBM.  T(beta) = T0 + S beta is affine in beta, so
//   * the moments the scale + translation correction needs (mean, sum |.|^2 of both point sets) are closed forms of beta through the
//     fp64 mean / Gram matrix of the rows [T0 | S] (hmf_smpl_create) -- no pass over the vertices;
//   * the errors need one pass: D = S (beta_p - beta_t) (= P - T), e0 = |D|, e1 = |(a - 1) T + a D + (mu_t - a mu_p)|.
// Lane = pair (the `group` predictions of one image share a target), a work-group walks a chunk of vertices: the chunk's 3 x (nb + 1) rows
// and its target vertices sit in LDS and are read at wave-uniform addresses (broadcast reads: one row fetch per work-group instead of one per
// pair), no cross-lane reduction anywhere; chunk partial sums are added in a fixed order by k_tpose_errors_finish.
// Before: k_tpose wrote 267 MB of T-pose bodies per step and k_pointset_errors_reg read them back (round-4 review).
// ---------------------------------------------------------------------------------------------------
constexpr int kTeChunk = 65;           // 6890 = 106 x 65
constexpr int kTeChunks = (kV + kTeChunk - 1) / kTeChunk;
// rows: [V][3][TS] (SmplDev::tp_pack) as a direct __restrict__ argument, so that the wave-uniform row reads of the main loop are SCALAR loads
// (the rows of a vertex arrive in SGPRs and feed the FMAs as scalar operands: no LDS or vector-memory traffic per pair).
// Pair pl of image b: pl < group -> row b * group + pl; pl == group (packed rows only, n_img_rows = 1) -> the point-estimate row B * group + b.
template <int NB>
__global__ __launch_bounds__(128) void k_tpose_errors(const float* __restrict__ rows_all, const double* __restrict__ mom, const float* __restrict__ betas_p,
                                                      const float* __restrict__ betas_t, int group, int packed_extra, int npairs,
                                                      float* __restrict__ partial /* [kTeChunks][npairs][2] */) {
    constexpr int NQ = NB + 1, TS = (NB + 1 + 3) & ~3;
    __shared__ float Tl[kTeChunk * 3];
    const int img = blockIdx.x, chunk = blockIdx.y;
    const int v0 = chunk * kTeChunk, nv = min(kTeChunk, kV - v0);
    const float* bt = betas_t + (size_t)img * NB;                     // wave-uniform
    const float* rows = rows_all + (size_t)v0 * 3 * TS;
    for (int e = threadIdx.x; e < nv * 3; e += blockDim.x) {          // the chunk's target vertices, k_tpose's fma order
        const float* row = rows + e * TS;
        float t = row[0];
#pragma unroll
        for (int k = 0; k < NB; ++k) t = fmaf(row[1 + k], bt[k], t);
        Tl[e] = t;
    }
    const int pl = blockIdx.z * blockDim.x + threadIdx.x;             // pair within the image
    const bool live = pl < group + packed_extra;
    const size_t pair = !live ? 0 : (pl < group ? (size_t)img * group + pl : (size_t)gridDim.x * group + img);
    const float* bp = betas_p + pair * NB;
    float db[NB];
    double qp[NQ], qt[NQ];
    qp[0] = qt[0] = 1.0;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const float p = bp[k], t = bt[k];
        db[k] = p - t;
        qp[1 + k] = p; qt[1 + k] = t;
    }
    // moments: mean_c = m[c] . q, sum |.|^2 = q^T G q
    double mp[3], mt[3], sp = 0.0, st = 0.0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int i = 0; i < NQ; ++i) { a += mom[c * NQ + i] * qp[i]; b += mom[c * NQ + i] * qt[i]; }
        mp[c] = a; mt[c] = b;
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int j = 0; j < NQ; ++j) { const double g = mom[3 * NQ + i * NQ + j]; a += g * qp[j]; b += g * qt[j]; }
        sp += a * qp[i]; st += b * qt[i];
    }
    const double n = kV;
    const double varp = sp - n * (mp[0] * mp[0] + mp[1] * mp[1] + mp[2] * mp[2]);
    const double vart = st - n * (mt[0] * mt[0] + mt[1] * mt[1] + mt[2] * mt[2]);
    const float sa = (float)(sqrt(vart / n) / sqrt(varp / n));        // solve_alignment's sc_a (csrc/eval_kernels.hip)
    const float sa1 = sa - 1.f;
    float kc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) kc[c] = fmaf(-sa, (float)mp[c], (float)mt[c]);          // mu_t - a mu_p
    __syncthreads();
    float e0 = 0.f, e1 = 0.f;
#pragma unroll 2
    for (int v = 0; v < nv; ++v) {
        float D[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* row = rows + (v * 3 + c) * TS;              // wave-uniform address: scalar loads
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < NB; ++k) a = fmaf(row[1 + k], db[k], a);
            D[c] = a;
        }
        e0 += __builtin_amdgcn_sqrtf(D[0] * D[0] + D[1] * D[1] + D[2] * D[2]);      // v_sqrt_f32 (1 ulp): mean distances of ~1e-2 m
        const float x = fmaf(sa1, Tl[v * 3], fmaf(sa, D[0], kc[0]));
        const float y = fmaf(sa1, Tl[v * 3 + 1], fmaf(sa, D[1], kc[1]));
        const float z = fmaf(sa1, Tl[v * 3 + 2], fmaf(sa, D[2], kc[2]));
        e1 += __builtin_amdgcn_sqrtf(x * x + y * y + z * z);
    }
    if (live) {
        float* o = partial + ((size_t)chunk * npairs + pair) * 2;      // chunk-major: the finish pass reads a chunk's pairs as one run
        o[0] = e0; o[1] = e1;
    }
}
__global__ __launch_bounds__(64) void k_tpose_errors_finish(const float* __restrict__ partial, int npairs, float* __restrict__ out) {
    const int p = blockIdx.x * 64 + threadIdx.x;
    if (p >= npairs) return;
    double a = 0.0, b = 0.0;
#pragma unroll 8
    for (int c = 0; c < kTeChunks; ++c) {
        const float2 v = ((const float2*)partial)[(size_t)c * npairs + p];
        a += v.x; b += v.y;
    }
    out[3 * p] = (float)(a / kV); out[3 * p + 1] = (float)(b / kV); out[3 * p + 2] = 0.f;
}

template <int NB>
static void launch_tpose_errors(const SmplDev& d, const float* bp, const float* bt, int nimg, int group, int extra, int npairs, float* partial, hipStream_t st) {
    const int threads = group + extra > 64 ? 128 : 64;
    hipLaunchKernelGGL((k_tpose_errors<NB>), dim3(nimg, kTeChunks, (group + extra + threads - 1) / threads), dim3(threads), 0, st, (const float*)d.tp_pack,
                       (const double*)d.tp_mom, bp, bt, group, extra, npairs, partial);
}

__global__ void k_rest_joints(SmplDev d, const float* __restrict__ betas, int M, float* __restrict__ joints) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= M * kJ) return;
    const int row = gid / kJ, j = gid % kJ, nb = d.num_betas;
    const float* b = betas + (size_t)row * nb;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float acc = d.j_template[j * 3 + c];
        for (int k = 0; k < nb; ++k) acc = fmaf(d.j_dirs[(j * 3 + c) * nb + k], b[k], acc);
        joints[((size_t)row * kOutJ + j) * 3 + c] = acc;
    }
}

// compute_vertex_variance_from_samples, utils/sampling_utils.py:22-33 (two passes, like the reference)
__global__ void k_vertex_variance(const float* __restrict__ verts, int N, int V, float* __restrict__ avg_l2,
                                  float* __restrict__ dir_std) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    float m[3] = {0.f, 0.f, 0.f};
    for (int n = 0; n < N; ++n) {
        const float* p = verts + ((size_t)n * V + v) * 3;
        m[0] += p[0]; m[1] += p[1]; m[2] += p[2];
    }
    m[0] /= N; m[1] /= N; m[2] /= N;
    float s[3] = {0.f, 0.f, 0.f}, l2 = 0.f;
    for (int n = 0; n < N; ++n) {
        const float* p = verts + ((size_t)n * V + v) * 3;
        const float dx = p[0] - m[0], dy = p[1] - m[1], dz = p[2] - m[2];
        s[0] += dx * dx; s[1] += dy * dy; s[2] += dz * dz;
        l2 += sqrtf(dx * dx + dy * dy + dz * dz);
    }
    avg_l2[v] = l2 / N;
    dir_std[v * 3 + 0] = sqrtf(s[0] / N); dir_std[v * 3 + 1] = sqrtf(s[1] / N); dir_std[v * 3 + 2] = sqrtf(s[2] / N);
}

// ---------------------------------------------------------------------------------------------------
// 2b. bf16 variant of the fused pose-blend GEMM + skinning (impl = 2): same tiling, operands in bf16 on
//     v_mfma_f32_32x32x16_bf16 (14 k-groups x 6 MFMAs per wave tile instead of 112 x 6), fp32 accumulate, fp32 template add and fp32
//     skinning.  With 1/16 of the MFMA time the kernel is bound by the skinning epilogue (LDS gathers + VALU) and the 82,680 B/row store.
// ---------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// SPLIT (impl = 3): both operands as bf16 hi + bf16 lo (x = hi + lo to 2^-17 relative); every product is three bf16 MFMAs
// (hi.hi + hi.lo + lo.hi, fp32 accumulate; lo.lo ~ 2^-18 dropped), which brings the vertices from 1.5e-3 m (plain bf16 operands) to within
// 1e-4 m of the fp32 oracle -- the north-star tolerance -- at 3/16 of the fp32 MFMA time: still bound by the vertex write.
// ABL (experiments build only, tools/lbs16_tune.py; results wrong when != 0): & 1 no skinning epilogue, & 2 no B-operand stream (the first k-group's
// fragments serve every k-group), & 4 no MFMAs, & 8 skinning without the vertex stores.
// Packed fp32 VALU (v_pk_fma_f32) is switched off for the library (build.py: beside fp32 MFMAs it is an anti-lever) and back ON for this kernel alone: with
// bf16 MFMAs the matrix pipe is busy 8 k of a wave tile's ~40 k cycles and the 48 blend FMAs per (row, vertex) of the skinning epilogue pair up.  (The
// attribute is honoured by the device pass; the host pass warns that it ignores it.)  Measured: no change (the epilogue is bound by LDS gather bandwidth).
#define HMF_LBS16_TARGET __attribute__((target("packed-fp32-ops")))
#ifdef HMF_EXPERIMENTS
// tools/lbs16_tune.py: shader-clock cycles every wave spends staging + waiting at the segment barriers / between tile start and its first k-group / in the
// main loop / in the skinning epilogue, summed over all waves (lane 0 adds at the end of the kernel), + the number of wave tiles
__device__ unsigned long long g_lbs16_phase[8];
#define HMF_L16_T(var) const long long var = (ABL & 64) ? (long long)__builtin_readcyclecounter() : 0        // a stamp drains the wave's LDS / scalar queue: only in the ABL & 64 build
#define HMF_L16_ADD(k, v) ph[k] += (v)
#else
#define HMF_L16_T(var)
#define HMF_L16_ADD(k, v)
#endif
// PF: k-groups of B fragments requested ahead of their use (a ring of PF + 1 register sets).  PIN: a scheduling barrier keeps every k-group's requests where
// they are written -- left alone the scheduler SINKS them to their first use (same VGPR count for PF = 1, 2, 3), and the main loop of rounds 2-5 was 14 serial
// L2 round trips per wave tile (26.6 k cycles for 8 k cycles of MFMA issue; pinned: 10.8 k).  LATE: the skinning offsets / weights and the template
// (11 registers) are loaded after the main loop instead of being held through it (no scratch left: 254 VGPRs, was 256 + 56 B).
// MEPI (round 6; EXPERIMENTS BUILD ONLY -- correct, within 1.4e-5 m of the fp32 kernel, and not faster: 283-286 vs 268 us at M = 6,464, profiles/r06_lbs16_ablation.txt):
// the skinning epilogue on the matrix pipe.  T[row][vertex] = sum_s w_s A[row][joint_s] is a dense product
// [32 rows x 24 joints] x [24 joints x 32 vertices] per transform element once the 4-sparse weights are written densely: bf16 hi + lo operands (three MFMAs per
// product as in the main loop; k = 16 + 8), fp32 accumulate, and it lands in the accumulator layout of the posed offsets (lane = vertex, register = row), so the
// 3 x 4 apply is 4 FMAs per register.  Per wave tile: 144 MFMAs (4.6 k cycles), 96 LDS fragment reads and 384 FMAs replace 384 16-byte LDS gathers and ~1,900
// FMAs (the gathers' 393 KB through the CU's LDS return path were what bounded the kernel: profiles/r06_lbs16_ablation.txt).  The transforms arrive as bf16
// planes [piece][element][row 64][joint 24] written by k_rigid_chain (48-byte rows: the 16 lanes of a ds_read_b128 group hit 16 different 4-bank slots).
template <int SLOTS, bool SPLIT, int ABL = 0, int PF = 1, bool LATE = true, bool PIN = true, bool HALF_TILES = false, bool XT = false, bool ADMA = true, bool MEPI = false>
__global__ __launch_bounds__(512, 2) HMF_LBS16_TARGET void k_lbs_mfma16(SmplDev d, int M, int nrg, int wpx, const unsigned short* __restrict__ panel16,
                                                     const unsigned short* __restrict__ panel16_lo, const float* __restrict__ atr, float* __restrict__ verts,
                                                     float* __restrict__ cjv, const unsigned short* __restrict__ atr16) {
    static_assert(!MEPI || (ADMA && SPLIT && kAtr16Group * 2 == kRowsPerWG * kAtr * 4), "the matrix-pipe epilogue reads the bf16 hi + lo transform planes staged by LDS-DMA");
    constexpr int kPanelF = kRowsPerWG * kP16Stride / 2;                     // floats of one bf16 panel
    __shared__ __attribute__((aligned(16))) float smem[(SPLIT ? 2 : 1) * kPanelF + kRowsPerWG * kAtr];   // 29 (+ 29) + 72 KB
    const unsigned short* sP = (const unsigned short*)smem;
    float* sA = smem + (SPLIT ? 2 : 1) * kPanelF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int xr = blockIdx.x & 7;
    const int chunk = blockIdx.x >> 3;
    const int T = nrg * kVTperX;
    int t = (int)((long long)chunk * T / wpx);
    const int t1 = (int)((long long)(chunk + 1) * T / wpx);
    const int nslots = d.skin_slots;
#ifdef HMF_EXPERIMENTS
    long long ph[5] = {0, 0, 0, 0, 0};
#endif
    while (t < t1) {
        const int rg = t / kVTperX;
        const int seg_end = min(t1, (rg + 1) * kVTperX);
        const int row0 = rg * kRowsPerWG;
        HMF_L16_T(ts0);
        {
            const f32x4* src = (const f32x4*)(panel16 + (size_t)row0 * kP16Stride);
            f32x4* dst = (f32x4*)smem;
            constexpr int n16 = kRowsPerWG * kP16Stride * 2 / 16;
            for (int i = tid; i < n16; i += 512) dst[i] = src[i];
            if (SPLIT) {
                const f32x4* srcl = (const f32x4*)(panel16_lo + (size_t)row0 * kP16Stride);
                for (int i = tid; i < n16; i += 512) dst[n16 + i] = srcl[i];
            }
            if (!ADMA) {
                const f32x4* src2 = (const f32x4*)(atr + (size_t)row0 * kAtr);
                f32x4* dst2 = (f32x4*)sA;
                const int n4 = min(kRowsPerWG, M - row0) * (kAtr / 4);
#pragma unroll
                for (int i = 0; i < kRowsPerWG * (kAtr / 4) / 512; ++i) {
                    const int ii = tid + i * 512;
                    f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    dst2[ii] = (ii < n4) ? src2[ii] : z;
                }
            }
        }
        __syncthreads();
        // ADMA: the 72 KB of skinning transforms are needed by the EPILOGUE only: they go global -> LDS by LDS-DMA (no registers, no ds_write: nine 1-KiB
        // requests per wave, rows beyond M read as zeros through the descriptor's range) while the segment's first main loops run; every wave waits for its
        // own requests and meets the others at one extra barrier before its first epilogue of the segment.
        bool atr_pending = false;
        if (ADMA) {
            const int live = MEPI ? kAtr16Group * 2 : min(kRowsPerWG, M - row0) * kAtr * 4;         // the bf16 planes carry their dead rows as zeros
            const __amdgpu_buffer_rsrc_t ratr = __builtin_amdgcn_make_buffer_rsrc(
                MEPI ? (void*)(atr16 + (size_t)rg * kAtr16Group) : (void*)(atr + (size_t)row0 * kAtr), 0, live, 0x00020000);
            constexpr int kReq = kRowsPerWG * kAtr * 4 / 1024 / 8;               // 1-KiB requests per wave
            static_assert(kReq * 8 * 1024 == kRowsPerWG * kAtr * 4, "transform block = whole requests");
#pragma unroll
            for (int i = 0; i < kReq; ++i)
                lbs_dma16(ratr, (lbs_lds_ptr_t)((char*)sA + (wave * kReq + i) * 1024), lane * 16, (wave * kReq + i) * 1024);
            atr_pending = true;
        }
        HMF_L16_T(ts1);
        HMF_L16_ADD(0, ts1 - ts0);
        // Work items as in k_lbs_mfma: full rounds of 8 vertex tiles, then a partial round of <= 4 tiles split by rows (two waves share a vertex tile, 32 rows
        // each: the 27 tiles of a range cost 3.5 rounds instead of 4), the half tiles FIRST for waves 4-7 and LAST for waves 0-3 -- a SIMD's two waves are then
        // half a tile apart, one in its main loop (matrix pipe, L2 stream) while the other skins (LDS gathers: the epilogue is bound by the CU's LDS return path).
        const int cnt = seg_end - t, rf = cnt >> 3, rem = cnt & 7;
        const bool splitr = HALF_TILES && rem > 0 && rem <= 4;
        const int nB = splitr ? min(2 * rem, 4) : 0, nA = splitr ? 2 * rem - nB : 0;
        const bool lead = wave >= 4 && (wave - 4) < nB;          // this wave starts with a half tile
        auto item = [&](int k, int& rh_) -> int {
            rh_ = -1;
            if (splitr) {
                if (lead) {
                    if (k == 0) { rh_ = (wave - 4) & 1; return t + ((wave - 4) >> 1); }
                    --k;
                }
                if (k < rf) return t + rem + k * 8 + wave;
                if (k == rf && wave < 4 && wave < nA) { rh_ = (nB + wave) & 1; return t + ((nB + wave) >> 1); }
                return -1;
            }
            if (k < rf) return t + k * 8 + wave;
            if (k == rf && rem > 0) return wave < rem ? t + rf * 8 + wave : -1;
            return -1;
        };
        int rh, kitem = 0;
        // XT (experiments, measured slower: 322 vs 311 us, 28 B of scratch): the first k-group of the NEXT tile of this wave is requested before the skinning
        // epilogue of the current one and waits in registers
        f32x4 bnext[SPLIT ? 6 : 3];
        bool have_next = false;
        // the tile body exists twice (whole tile / half tile, compile time): a run-time `rh < 0` around the second accumulator set's MFMAs splits the k-group
        // into basic blocks and un-pins its requests (measured: main loop 10.2 k -> 19.1 k cycles)
        auto tile = [&](int tt, auto HALFT) {
            constexpr bool kHalf = decltype(HALFT)::value;
            HMF_L16_T(tt0);
            const int vt = xr * kVTperX + (tt - rg * kVTperX);
            int off[SLOTS > 0 ? SLOTS : 1];
            float w[SLOTS > 0 ? SLOTS : 1];
            const int v = vt * 32 + l31;
            float tm[3] = {0.f, 0.f, 0.f};
            f32x4 w0h = {0.f, 0.f, 0.f, 0.f}, w0l = w0h;               // MEPI: this vertex tile's dense skinning weights, B-operand fragments (k = 16 | 8, hi | lo)
            f32x2 w1h = {0.f, 0.f}, w1l = w1h;
            auto load_epilogue_operands = [&]() {
                if (MEPI) {
                    const unsigned short* wb = d.skin_w16 + (size_t)vt * 2 * kW16PerPiece;
                    w0h = ((const f32x4*)wb)[lane]; w1h = ((const f32x2*)(wb + 512))[lane];
                    w0l = ((const f32x4*)(wb + kW16PerPiece))[lane]; w1l = ((const f32x2*)(wb + kW16PerPiece + 512))[lane];
                } else if (SLOTS > 0) {
#pragma unroll
                    for (int s = 0; s < SLOTS; ++s) {
                        off[s] = d.skin_off[((size_t)vt * nslots + s) * 32 + l31];
                        w[s] = d.skin_w[((size_t)vt * nslots + s) * 32 + l31];
                    }
                }
                if (v < kV) { tm[0] = d.tp_tmpl[v * 3]; tm[1] = d.tp_tmpl[v * 3 + 1]; tm[2] = d.tp_tmpl[v * 3 + 2]; }
            };
            if (!LATE) load_epilogue_operands();
            f32x16 acc[2][3];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[a][c][e] = 0.f;
            const f32x4* bp = (const f32x4*)(d.blend16 + (size_t)vt * kKG16 * 3 * 64 * 8) + lane;
            const f32x4* bpl = (const f32x4*)(d.blend16_lo + (size_t)vt * kKG16 * 3 * 64 * 8) + lane;
            const unsigned short* ap = sP + (l31 + (rh > 0 ? 32 : 0)) * kP16Stride + half * 8;      // half tile: this wave's 32 rows feed accumulator set 0
            const unsigned short* apl = ap + kRowsPerWG * kP16Stride;                  // the lo panel follows the hi panel in LDS
            constexpr int NSET = PF + 1;
            f32x4 bs[NSET][SPLIT ? 6 : 3];
            auto loadB = [&](int kg, int slot) {
                bs[slot][0] = bp[kg * 192]; bs[slot][1] = bp[kg * 192 + 64]; bs[slot][2] = bp[kg * 192 + 128];
                if (SPLIT) { bs[slot][3] = bpl[kg * 192]; bs[slot][4] = bpl[kg * 192 + 64]; bs[slot][5] = bpl[kg * 192 + 128]; }
            };
            if (XT && PF == 1 && have_next) {
#pragma unroll
                for (int i = 0; i < (SPLIT ? 6 : 3); ++i) bs[0][i] = bnext[i];
            } else {
#pragma unroll
                for (int p = 0; p < PF; ++p) loadB(p, p);
            }
            f32x4 a0 = *(const f32x4*)ap, a1 = *(const f32x4*)(ap + 32 * kP16Stride);
            f32x4 al0 = a0, al1 = a1;
            if (SPLIT) {
                al0 = *(const f32x4*)apl; al1 = *(const f32x4*)(apl + 32 * kP16Stride);
            }
            HMF_L16_T(tt1);
#pragma unroll
            for (int kg = 0; kg < kKG16; ++kg) {
                f32x4 m0 = a0, m1 = a1, ml0 = al0, ml1 = al1;
                if (kg + PF < kKG16 && !(ABL & 2)) loadB(kg + PF, (kg + PF) % NSET);
                if (LATE && XT && kg == kKG16 - 2) load_epilogue_operands();       // two k-groups before the epilogue: their latency under the last MFMAs
                if (PIN) __builtin_amdgcn_sched_barrier(0);        // the requests stay HERE (see PIN above)
                if (kg + 1 < kKG16) {
                    m0 = *(const f32x4*)(ap + (kg + 1) * 16);
                    m1 = *(const f32x4*)(ap + 32 * kP16Stride + (kg + 1) * 16);
                    if (SPLIT) {
                        ml0 = *(const f32x4*)(apl + (kg + 1) * 16);
                        ml1 = *(const f32x4*)(apl + 32 * kP16Stride + (kg + 1) * 16);
                    }
                }
                const int cur = (ABL & 2) ? (kg < PF ? kg : 0) : kg % NSET;
                const f32x4 b0 = bs[cur][0], b1 = bs[cur][1], b2 = bs[cur][2];
                const f32x4 bl0 = bs[cur][SPLIT ? 3 : 0], bl1 = bs[cur][SPLIT ? 4 : 1], bl2 = bs[cur][SPLIT ? 5 : 2];
                const bf16x8 A0 = __builtin_bit_cast(bf16x8, a0), A1 = __builtin_bit_cast(bf16x8, a1);
                const bf16x8 B[3] = {__builtin_bit_cast(bf16x8, b0), __builtin_bit_cast(bf16x8, b1), __builtin_bit_cast(bf16x8, b2)};
                if (ABL & 4) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) { acc[0][c][kg] += a0[c] * b0[c] + al0[c] * bl1[c]; acc[1][c][kg] += a1[c] * b2[c] + al1[c] * bl2[c]; }
                } else {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    acc[0][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B[c], acc[0][c], 0, 0, 0);
                    if (!kHalf) acc[1][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B[c], acc[1][c], 0, 0, 0);
                }
                }
                if (SPLIT && !(ABL & 4)) {
                    const bf16x8 L0 = __builtin_bit_cast(bf16x8, al0), L1 = __builtin_bit_cast(bf16x8, al1);
                    const bf16x8 BL[3] = {__builtin_bit_cast(bf16x8, bl0), __builtin_bit_cast(bf16x8, bl1), __builtin_bit_cast(bf16x8, bl2)};
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        acc[0][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, BL[c], acc[0][c], 0, 0, 0);
                        if (!kHalf) acc[1][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, BL[c], acc[1][c], 0, 0, 0);
                        acc[0][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(L0, B[c], acc[0][c], 0, 0, 0);
                        if (!kHalf) acc[1][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(L1, B[c], acc[1][c], 0, 0, 0);
                    }
                }
                a0 = m0; a1 = m1; al0 = ml0; al1 = ml1;
            }
            HMF_L16_T(tt2);
            if (LATE && !XT) load_epilogue_operands();
            if (XT && PF == 1) {
                int nrh;
                const int ntt = item(kitem + 1, nrh);
                have_next = ntt >= 0;
                if (have_next) {
                    const int nvt = xr * kVTperX + (ntt - rg * kVTperX);
                    const f32x4* nbp = (const f32x4*)(d.blend16 + (size_t)nvt * kKG16 * 3 * 64 * 8) + lane;
                    bnext[0] = nbp[0]; bnext[1] = nbp[64]; bnext[2] = nbp[128];
                    if (SPLIT) {
                        const f32x4* nbpl = (const f32x4*)(d.blend16_lo + (size_t)nvt * kKG16 * 3 * 64 * 8) + lane;
                        bnext[3] = nbpl[0]; bnext[4] = nbpl[64]; bnext[5] = nbpl[128];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[a][c][e] += tm[c];
            if (ADMA && atr_pending) {                 // wave-uniform: first epilogue of the segment
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                atr_pending = false;
            }
            EpiCtx e;
            e.ok_v = v < kV;
            e.full_rows = row0 + kRowsPerWG <= M;
            const int rsh = rh > 0 ? 32 : 0;                  // half tile: accumulator set 0 holds rows 32 rh .. 32 rh + 31 -> shift the row bases, run the first 16 elements
            e.row_lim = M - row0 - 4 * half - rsh;
            e.obase = verts + ((size_t)(row0 + 4 * half + rsh) * kV + v) * 3;
            const int jslot = cjv ? d.ju_slot[vt * 32 + l31] : -1;
            e.lane_has_ju = jslot >= 0;
            e.tile_has_ju = __builtin_amdgcn_ballot_w64(e.lane_has_ju) != 0;
            e.cj_row = d.n_ju * 3;
            e.cj = cjv + ((size_t)(row0 + 4 * half + rsh) * d.n_ju + (jslot >= 0 ? jslot : 0)) * 3;
            const char* lbase = (const char*)(sA + (4 * half + rsh) * kAtr);
            if (ABL & 1) {
                if (e.ok_v && l31 == 0) {
                    float sacc = 0.f;
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int c = 0; c < 3; ++c)
#pragma unroll
                            for (int q = 0; q < 16; ++q) sacc += acc[a][c][q];
                    e.obase[0] = sacc;
                }
            } else if (MEPI) {
                if (ABL & 8) e.ok_v = e.ok_v && (row0 < 0);
                typedef short s16x4 __attribute__((ext_vector_type(4)));
                const unsigned short* sT = (const unsigned short*)sA;
                const bf16x8 W0h = __builtin_bit_cast(bf16x8, w0h), W0l = __builtin_bit_cast(bf16x8, w0l);
                const s16x4 W1h = __builtin_bit_cast(s16x4, w1h), W1l = __builtin_bit_cast(s16x4, w1l);
#pragma unroll
                for (int a = 0; a < (kHalf ? 1 : 2); ++a) {
                    const unsigned short* prow = sT + (32 * a + rsh + l31) * kJ;          // this lane's A-operand row in every (piece, element) plane
                    float ox[16], oy[16];
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        // the four elements q of transform row c in two pairs: two independent accumulator chains of six MFMAs each (a lone chain waits out every
                        // MFMA's latency), folded into vals as they finish (v = T0 px; v = fma(T1, py, v); v = fma(T2, pz, v); v += T3: skin_element's order)
                        float vals[16];
#pragma unroll
                        for (int qp = 0; qp < 2; ++qp) {
                            bf16x8 A0h[2], A0l[2];
                            s16x4 A1h[2], A1l[2];
                            f32x16 T[2];
#pragma unroll
                            for (int u = 0; u < 2; ++u) {
                                const unsigned short* ph = prow + (4 * c + 2 * qp + u) * kAtr16Plane;
                                const unsigned short* pl = ph + 12 * kAtr16Plane;
                                A0h[u] = __builtin_bit_cast(bf16x8, *(const f32x4*)(ph + 8 * half));
                                A1h[u] = __builtin_bit_cast(s16x4, *(const f32x2*)(ph + 16 + 4 * half));
                                A0l[u] = __builtin_bit_cast(bf16x8, *(const f32x4*)(pl + 8 * half));
                                A1l[u] = __builtin_bit_cast(s16x4, *(const f32x2*)(pl + 16 + 4 * half));
#pragma unroll
                                for (int r = 0; r < 16; ++r) T[u][r] = 0.f;
                            }
#pragma unroll
                            for (int u = 0; u < 2; ++u) T[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0h[u], W0h, T[u], 0, 0, 0);
#pragma unroll
                            for (int u = 0; u < 2; ++u) T[u] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(A1h[u], W1h, T[u], 0, 0, 0);
#pragma unroll
                            for (int u = 0; u < 2; ++u) T[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0h[u], W0l, T[u], 0, 0, 0);
#pragma unroll
                            for (int u = 0; u < 2; ++u) T[u] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(A1h[u], W1l, T[u], 0, 0, 0);
#pragma unroll
                            for (int u = 0; u < 2; ++u) T[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0l[u], W0h, T[u], 0, 0, 0);
#pragma unroll
                            for (int u = 0; u < 2; ++u) T[u] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(A1l[u], W1h, T[u], 0, 0, 0);
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                if (qp == 0) vals[r] = fmaf(T[1][r], acc[a][1][r], T[0][r] * acc[a][0][r]);
                                else vals[r] = fmaf(T[0][r], acc[a][2][r], vals[r]) + T[1][r];
                                asm volatile("" : "+v"(vals[r]));      // the fold happens HERE: otherwise the compiler sinks it into the exec-masked store blocks and keeps (spills) every T
                            }
                            if (!(ABL & 128)) __builtin_amdgcn_sched_barrier(0);          // one pair at a time: left alone the scheduler hoists the fragment reads and MFMAs of all 24 (half, row, element) blocks
                        }
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float val = vals[r];
                            if (c == 0) ox[r] = val;
                            else if (c == 1) oy[r] = val;
                            else {
                                const int rc = 32 * a + (r & 3) + 8 * (r >> 2);
                                const float x = ox[r], y = oy[r], z = val;
                                if (e.ok_v && (e.full_rows || rc < e.row_lim)) {
                                    typedef float f32x3u __attribute__((ext_vector_type(3), aligned(4)));
                                    __builtin_nontemporal_store(f32x3u{x, y, z}, (f32x3u*)(e.obase + rc * (kV * 3)));
                                }
                                if (e.tile_has_ju) {
                                    if (e.lane_has_ju && (e.full_rows || rc < e.row_lim)) {
                                        float* q = e.cj + rc * e.cj_row;
                                        q[0] = x; q[1] = y; q[2] = z;
                                    }
                                }
                            }
                        }
                    }
                }
            } else {
                if (ABL & 8) e.ok_v = e.ok_v && (row0 < 0);          // never true, not provable: the skinning is computed, nothing stored
#pragma unroll
                for (int el = 0; el < 32; ++el) {
                    if (el == 16 && kHalf) break;
                    skin_element<SLOTS>(d, lbase, off, w, nslots, vt, l31, acc, el >> 4, el >> 4, el & 15, e);
                }
            }
            HMF_L16_T(tt3);
            HMF_L16_ADD(1, tt1 - tt0); HMF_L16_ADD(2, tt2 - tt1); HMF_L16_ADD(3, tt3 - tt2); HMF_L16_ADD(4, kHalf ? 1 : 2);                 // counted in half tiles
        };
        for (int tt = item(0, rh); tt >= 0; tt = item(++kitem, rh)) {
            if (rh < 0) tile(tt, std::false_type{});
            else tile(tt, std::true_type{});
        }
        HMF_L16_T(ts2);
        if (ADMA && atr_pending) {                     // a wave without a tile in this segment still meets the others at the transform barrier
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        __syncthreads();
        HMF_L16_T(ts3);
        HMF_L16_ADD(0, ts3 - ts2);                    // the wait at the segment's closing barrier = idle time of waves with fewer tiles: booked with staging
        t = seg_end;
    }
#ifdef HMF_EXPERIMENTS
    if ((ABL & 64) && lane == 0) {
#pragma unroll
        for (int k = 0; k < 5; ++k) atomicAdd(&g_lbs16_phase[k], (unsigned long long)ph[k]);
    }
#endif
}

// ---------------------------------------------------------------------------------------------------
// The 90 output joints WITHOUT the body (round 5): what the training loss consumes (2D reprojection of joints[:, ALL_JOINTS_TO_COCO_MAP],
// train/train_humaniflow.py:283-318; the vertices of the same SMPL call feed only the reference's metric tracker).  One work-group per row: rest
// joints, the 24-joint chain level by level, then only the <= kMaxJU vertices the gathered joints touch go through the blend shapes (the dense
// joint-vertex matrix of k_smpl_joints_bwd) and the skinning; joints 0..23 are the chain's translations, 24..89 the regressor rows of
// k_joints_compact.  218 x 3 x n_ju FMAs per row instead of 218 x 3 x 6890 + an 82.7 KB vertex row: ~20 us for the training batch against
// k_rigid_chain + k_lbs_mfma + k_joints_compact (~140 us).  The same function as hmf_smpl_lbs's joints up to fp32 summation order.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_smpl_joints_fwd(SmplDev d, const float* __restrict__ glob, const float* __restrict__ body,
                                                         const float* __restrict__ betas, int M, float* __restrict__ joints) {
    __shared__ float sR[kJ][9], sJ[kJ][3], sG[kJ][12], sA[kJ][12];
    __shared__ float s_feat[kKpad], s_v[kMaxJU][3];
    const int tid = threadIdx.x, row = blockIdx.x, nb = d.num_betas;
    const float* beta = betas + (size_t)row * nb;
    for (int i = tid; i < kJ * 9; i += 256) {
        const int j = i / 9, e = i % 9;
        sR[j][e] = j == 0 ? glob[(size_t)row * 9 + e] : body[((size_t)row * 23 + (j - 1)) * 9 + e];
    }
    for (int i = tid; i < kJ * 3; i += 256) {
        const int j = i / 3, c = i % 3;
        float acc = d.j_template[j * 3 + c];
        for (int k = 0; k < nb; ++k) acc = fmaf(d.j_dirs[(j * 3 + c) * nb + k], beta[k], acc);
        sJ[j][c] = acc;
    }
    __syncthreads();
    for (int k = tid; k < kKpad; k += 256) s_feat[k] = pose_feature(k, nb, &sR[1][0], beta);
    for (int L = 0; L <= d.depth_max; ++L) {                          // smplx batch_rigid_transform, level by level (the expressions of k_smpl_joints_bwd)
        for (int idx = tid; idx < kJ * 12; idx += 256) {
            const int j = idx / 12, e = idx - j * 12, i = e >> 2, c = e & 3;
            if (d.depth[j] != L) continue;
            const int par = d.parents[j];
            if (j == 0) {
                sG[0][e] = c < 3 ? sR[0][i * 3 + c] : sJ[0][i];
            } else if (c < 3) {
                sG[j][e] = sG[par][i * 4 + 0] * sR[j][0 * 3 + c] + sG[par][i * 4 + 1] * sR[j][1 * 3 + c] + sG[par][i * 4 + 2] * sR[j][2 * 3 + c];
            } else {
                const float rel[3] = {sJ[j][0] - sJ[par][0], sJ[j][1] - sJ[par][1], sJ[j][2] - sJ[par][2]};
                sG[j][e] = sG[par][i * 4 + 0] * rel[0] + sG[par][i * 4 + 1] * rel[1] + sG[par][i * 4 + 2] * rel[2] + sG[par][i * 4 + 3];
            }
        }
        __syncthreads();
    }
    for (int idx = tid; idx < kJ * 12; idx += 256) {
        const int j = idx / 12, e = idx - j * 12, i = e >> 2, c = e & 3;
        sA[j][e] = c < 3 ? sG[j][e] : sG[j][i * 4 + 3] - (sG[j][i * 4 + 0] * sJ[j][0] + sG[j][i * 4 + 1] * sJ[j][1] + sG[j][i * 4 + 2] * sJ[j][2]);
    }
    __syncthreads();
    const int kdata = kPoseFeat + nb + 1;
    for (int u = tid; u < d.n_ju; u += 256) {
        const int v = d.ju_vid[u], vt = v >> 5, l31 = v & 31;
        float vp[3] = {0.f, 0.f, 0.f};
        const float* jb = d.ju_blend + u * 3;
#pragma unroll 8
        for (int k = 0; k < kdata; ++k) {
            const float a = s_feat[k];
            const float* r = jb + (size_t)k * d.ju_bstride;
#pragma unroll
            for (int c = 0; c < 3; ++c) vp[c] = fmaf(a, r[c], vp[c]);
        }
        float T[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) T[i] = 0.f;
        for (int s = 0; s < d.skin_slots; ++s) {
            const int jn = d.skin_off[((size_t)vt * d.skin_slots + s) * 32 + l31] / 48;
            const float w = d.skin_w[((size_t)vt * d.skin_slots + s) * 32 + l31];
            if (w != 0.f) {
#pragma unroll
                for (int i = 0; i < 12; ++i) T[i] = fmaf(w, sA[jn][i], T[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) s_v[u][i] = T[i * 4 + 0] * vp[0] + T[i * 4 + 1] * vp[1] + T[i * 4 + 2] * vp[2] + T[i * 4 + 3];
    }
    __syncthreads();
    float* o = joints + (size_t)row * kOutJ * 3;
    for (int i = tid; i < kJ * 3; i += 256) o[i] = sG[i / 3][(i % 3) * 4 + 3];
    for (int i = tid; i < kOutJ - kJ; i += 256) {
        float x = 0.f, y = 0.f, z = 0.f;
        for (int e = d.jr_ptr[i]; e < d.jr_ptr[i + 1]; ++e) {
            const float w = d.jr_w[e];
            const float* p = s_v[d.jr_slot[e]];
            x = fmaf(w, p[0], x); y = fmaf(w, p[1], y); z = fmaf(w, p[2], z);
        }
        o[(kJ + i) * 3 + 0] = x; o[(kJ + i) * 3 + 1] = y; o[(kJ + i) * 3 + 2] = z;
    }
}

// ---------------------------------------------------------------------------------------------------
// Backward of the 90 output joints w.r.t. (global_orient, body_pose, betas) -- what the 2D reprojection loss of the training step needs
// (losses/humaniflow_loss.py:44-60 on joints[:, ALL_JOINTS_TO_COCO_MAP]; train/train_humaniflow.py:283-318).  One work-group per row:
// rest joints and the 24-joint chain are recomputed, the <= 276 vertices that the gathered joints touch are pushed back through the
// skinning blend, the pose/shape blend-shapes (sparse rows of the packed operand) and the chain.  Vertex gradients are not built
// (LOSS.APPLY_POINT_EST_LOSS = False in the reference configuration).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_smpl_joints_bwd(SmplDev d, const float* __restrict__ glob, const float* __restrict__ body,
                                                         const float* __restrict__ betas, int M, const float* __restrict__ g_joints,
                                                         float* __restrict__ g_glob, float* __restrict__ g_body, float* __restrict__ g_betas) {
    __shared__ float sR[kJ][9], sJ[kJ][3], sG[kJ][12], sA[kJ][12], s_gA[kJ][12], s_gG[kJ][12], s_gJ[kJ][3], s_gR[kJ][9];
    __shared__ float s_gj[kOutJ][3], s_feat[kKpad], s_gfeat[kKpad], s_gvp[kMaxJU][3], s_gJrel[kJ][3];
    const int tid = threadIdx.x, row = blockIdx.x, nb = d.num_betas;
    const float* beta = betas + (size_t)row * nb;
    for (int i = tid; i < kJ * 9; i += 256) {
        const int j = i / 9, e = i % 9;
        sR[j][e] = j == 0 ? glob[(size_t)row * 9 + e] : body[((size_t)row * 23 + (j - 1)) * 9 + e];
    }
    for (int i = tid; i < kOutJ * 3; i += 256) s_gj[i / 3][i % 3] = g_joints[(size_t)row * kOutJ * 3 + i];
    for (int i = tid; i < kJ * 12; i += 256) { s_gA[i / 12][i % 12] = 0.f; s_gG[i / 12][i % 12] = 0.f; }
    for (int i = tid; i < kJ * 3; i += 256) {
        const int j = i / 3, c = i % 3;
        float acc = d.j_template[j * 3 + c];
        for (int k = 0; k < nb; ++k) acc = fmaf(d.j_dirs[(j * 3 + c) * nb + k], beta[k], acc);
        sJ[j][c] = acc;
        s_gJ[j][c] = 0.f;
    }
    __syncthreads();
    for (int k = tid; k < kKpad; k += 256) { s_feat[k] = pose_feature(k, nb, &sR[1][0], beta); s_gfeat[k] = 0.f; }
    __syncthreads();
    // smplx batch_rigid_transform, level by level of the kinematic tree (depth <= 8 for SMPL): thread = (joint, element of its 3 x 4 transform), the same
    // expressions as the one-thread loop this replaces (two serial 24-joint chains through LDS were ~100 us of the kernel's 170)
    for (int L = 0; L <= d.depth_max; ++L) {
        for (int idx = tid; idx < kJ * 12; idx += 256) {
            const int j = idx / 12, e = idx - j * 12, i = e >> 2, c = e & 3;
            if (d.depth[j] != L) continue;
            const int par = d.parents[j];
            if (j == 0) {
                sG[0][e] = c < 3 ? sR[0][i * 3 + c] : sJ[0][i];
            } else if (c < 3) {
                sG[j][e] = sG[par][i * 4 + 0] * sR[j][0 * 3 + c] + sG[par][i * 4 + 1] * sR[j][1 * 3 + c] + sG[par][i * 4 + 2] * sR[j][2 * 3 + c];
            } else {
                const float rel[3] = {sJ[j][0] - sJ[par][0], sJ[j][1] - sJ[par][1], sJ[j][2] - sJ[par][2]};
                sG[j][e] = sG[par][i * 4 + 0] * rel[0] + sG[par][i * 4 + 1] * rel[1] + sG[par][i * 4 + 2] * rel[2] + sG[par][i * 4 + 3];
            }
        }
        __syncthreads();
    }
    for (int idx = tid; idx < kJ * 12; idx += 256) {
        const int j = idx / 12, e = idx - j * 12, i = e >> 2, c = e & 3;
        sA[j][e] = c < 3 ? sG[j][e] : sG[j][i * 4 + 3] - (sG[j][i * 4 + 0] * sJ[j][0] + sG[j][i * 4 + 1] * sJ[j][1] + sG[j][i * 4 + 2] * sJ[j][2]);
    }
    __syncthreads();
    // ---- the vertices the gathered joints read: joint gradient -> vertex gradient -> skinning blend / posed-vertex gradients ----
    const int kdata = kPoseFeat + nb + 1;
    for (int u = tid; u < d.n_ju; u += 256) {
        float gv[3] = {0.f, 0.f, 0.f};
        for (int e = d.ju_ptr[u]; e < d.ju_ptr[u + 1]; ++e) {
            const float w = d.ju_w[e];
            const float* gj = s_gj[kJ + d.ju_out[e]];
            gv[0] = fmaf(w, gj[0], gv[0]); gv[1] = fmaf(w, gj[1], gv[1]); gv[2] = fmaf(w, gj[2], gv[2]);
        }
        const int v = d.ju_vid[u], vt = v >> 5, l31 = v & 31;
        float vp[4] = {0.f, 0.f, 0.f, 1.f};
        const float* jb = d.ju_blend + u * 3;                         // row k of the dense joint-vertex matrix: the same values as blend_index(k, v, c)
#pragma unroll 8
        for (int k = 0; k < kdata; ++k) {
            const float a = s_feat[k];
            const float* r = jb + (size_t)k * d.ju_bstride;
#pragma unroll
            for (int c = 0; c < 3; ++c) vp[c] = fmaf(a, r[c], vp[c]);
        }
        float T[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) T[i] = 0.f;
        for (int s = 0; s < d.skin_slots; ++s) {
            const int jn = d.skin_off[((size_t)vt * d.skin_slots + s) * 32 + l31] / 48;
            const float w = d.skin_w[((size_t)vt * d.skin_slots + s) * 32 + l31];
            if (w != 0.f) {
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    T[i] = fmaf(w, sA[jn][i], T[i]);
                    atomicAdd(&s_gA[jn][i], w * gv[i >> 2] * vp[i & 3]);        // d L / d A_j = w * gv (x) [v_posed; 1]
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) s_gvp[u][c] = T[0 * 4 + c] * gv[0] + T[1 * 4 + c] * gv[1] + T[2 * 4 + c] * gv[2];
    }
    __syncthreads();
    for (int k = tid; k < kPoseFeat + nb; k += 256) {     // blend-shape rows of the touched vertices: d L / d pose feature, d L / d beta
        float acc = 0.f;
        const float* r = d.ju_blend + (size_t)k * d.ju_bstride;       // contiguous row: (vertex, component) in the order of s_gvp
        const float* gq = &s_gvp[0][0];
        const int n3 = d.n_ju * 3;
        int i = 0;
        for (; i + 4 <= n3; i += 4) {
            const f32x4 q = *(const f32x4*)(r + i);
            acc = fmaf(q[0], gq[i], acc); acc = fmaf(q[1], gq[i + 1], acc); acc = fmaf(q[2], gq[i + 2], acc); acc = fmaf(q[3], gq[i + 3], acc);
        }
        for (; i < n3; ++i) acc = fmaf(r[i], gq[i], acc);
        s_gfeat[k] = acc;
    }
    __syncthreads();
    // ---- reverse of the chain, level by level from the leaves.  Same expressions and the same per-accumulator order as the one-thread loop it
    // replaces (j = 23 .. 1: a parent receives its children in DESCENDING index order, after its own descendants are complete).
    for (int idx = tid; idx < kJ * 3; idx += 256) {                   // A_R = G_R, A_t = G_t - G_R J;  posed joint j = G_t
        const int j = idx / 3, i = idx - j * 3;
        const float gat = s_gA[j][i * 4 + 3];
#pragma unroll
        for (int c = 0; c < 3; ++c) s_gG[j][i * 4 + c] = s_gA[j][i * 4 + c] - gat * sJ[j][c];
        s_gG[j][i * 4 + 3] = gat + s_gj[j][i];
    }
    __syncthreads();
    for (int idx = tid; idx < kJ * 3; idx += 256) {                   // s_gJ[j][c] -= G_R[i][c] * g_A_t[i], i = 0, 1, 2 in order
        const int j = idx / 3, c = idx - j * 3;
        float acc = s_gJ[j][c];
#pragma unroll
        for (int i = 0; i < 3; ++i) acc -= sG[j][i * 4 + c] * s_gA[j][i * 4 + 3];
        s_gJ[j][c] = acc;
    }
    __syncthreads();
    float (*s_gr)[3] = s_gJrel;                                        // g_rel of every joint (LDS)
    for (int L = d.depth_max; L >= 1; --L) {
        // step A: joints of level L (their s_gG is final): g_R_j = G_R_par^T g_G_R_j, g_rel = G_R_par^T g_G_t_j, s_gJ[j] += g_rel
        for (int idx = tid; idx < kJ * 12; idx += 256) {
            const int j = idx / 12, e = idx - j * 12;
            if (j == 0 || d.depth[j] != L) continue;
            const int par = d.parents[j];
            if (e < 9) {
                const int a = e / 3, c = e - a * 3;
                s_gR[j][a * 3 + c] = sG[par][0 * 4 + a] * s_gG[j][0 * 4 + c] + sG[par][1 * 4 + a] * s_gG[j][1 * 4 + c] + sG[par][2 * 4 + a] * s_gG[j][2 * 4 + c];
            } else {
                const int a = e - 9;
                const float gr = sG[par][0 * 4 + a] * s_gG[j][0 * 4 + 3] + sG[par][1 * 4 + a] * s_gG[j][1 * 4 + 3] + sG[par][2 * 4 + a] * s_gG[j][2 * 4 + 3];
                s_gr[j][a] = gr;
                s_gJ[j][a] += gr;
            }
        }
        __syncthreads();
        // step B: parents at level L - 1 gather their level-L children in descending index order
        for (int idx = tid; idx < kJ * 15; idx += 256) {
            const int p = idx / 15, e = idx - p * 15;
            if (d.depth[p] != L - 1) continue;
            for (int j = kJ - 1; j > p; --j) {
                if (d.parents[j] != p) continue;
                if (e < 12) {
                    const int i = e >> 2, a = e & 3;
                    if (a < 3) {
                        const float rel_a = sJ[j][a] - sJ[p][a];
                        s_gG[p][i * 4 + a] += s_gG[j][i * 4 + 0] * sR[j][a * 3 + 0] + s_gG[j][i * 4 + 1] * sR[j][a * 3 + 1] + s_gG[j][i * 4 + 2] * sR[j][a * 3 + 2]
                                              + s_gG[j][i * 4 + 3] * rel_a;
                    } else {
                        s_gG[p][i * 4 + 3] += s_gG[j][i * 4 + 3];
                    }
                } else {
                    s_gJ[p][e - 12] -= s_gr[j][e - 12];
                }
            }
        }
        __syncthreads();
    }
    if (tid < 9) s_gR[0][tid] = s_gG[0][(tid / 3) * 4 + (tid % 3)];
    if (tid >= 64 && tid < 67) s_gJ[0][tid - 64] += s_gG[0][(tid - 64) * 4 + 3];
    __syncthreads();
    for (int i = tid; i < 9; i += 256) g_glob[(size_t)row * 9 + i] = s_gR[0][i];
    for (int i = tid; i < kPoseFeat; i += 256) g_body[(size_t)row * kPoseFeat + i] = s_gR[1 + i / 9][i % 9] + s_gfeat[i];
    for (int k = tid; k < nb; k += 256) {
        float acc = s_gfeat[kPoseFeat + k];
        for (int j = 0; j < kJ; ++j)
            for (int c = 0; c < 3; ++c) acc = fmaf(s_gJ[j][c], d.j_dirs[(j * 3 + c) * nb + k], acc);
        g_betas[(size_t)row * nb + k] = acc;
    }
}

// ---------------------------------------------------------------------------------------------------
// host-side packing
// ---------------------------------------------------------------------------------------------------
static int validate(const hmf_smpl_host_t* m) {
    if (!m || !m->v_template_host || !m->shapedirs_host || !m->posedirs_host || !m->J_regressor_host ||
        !m->lbs_weights_host || !m->parents_host || !m->extra_vertex_ids_host || !m->J_regressor_more_host)
        return fail(HMF_ERR_INVALID, "hmf_smpl: null model array");
    if (m->num_betas < 1 || m->num_betas > kMaxBetas)
        return fail(HMF_ERR_UNSUPPORTED, "hmf_smpl: num_betas=%d outside [1,%d]", m->num_betas, kMaxBetas);
    if (m->parents_host[0] != -1) return fail(HMF_ERR_INVALID, "hmf_smpl: parents[0] must be -1");
    for (int j = 1; j < kJ; ++j)
        if (m->parents_host[j] < 0 || m->parents_host[j] >= j)
            return fail(HMF_ERR_INVALID, "hmf_smpl: parents[%d]=%d is not an earlier joint", j, m->parents_host[j]);
    return HMF_OK;
}

static void pack_blend(const hmf_smpl_host_t* m, float* out) {
    const int nb = m->num_betas;
    memset(out, 0, sizeof(float) * (size_t)kVT * kKG * 3 * 64 * 4);
    for (int v = 0; v < kV; ++v)
        for (int c = 0; c < 3; ++c) {
            for (int k = 0; k < kPoseFeat; ++k) out[blend_index(k, v, c)] = m->posedirs_host[(size_t)k * kV * 3 + v * 3 + c];
            for (int k = 0; k < nb; ++k) out[blend_index(kPoseFeat + k, v, c)] = m->shapedirs_host[((size_t)v * 3 + c) * nb + k];
            out[blend_index(kPoseFeat + nb, v, c)] = m->v_template_host[v * 3 + c];
        }
}

static void pack_blend16(const hmf_smpl_host_t* m, unsigned short* out, bool lo = false) {
    const int nb = m->num_betas;
    memset(out, 0, sizeof(unsigned short) * (size_t)kVT * kKG16 * 3 * 64 * 8);
    for (int v = 0; v < kV; ++v)
        for (int c = 0; c < 3; ++c) {
            auto enc = [&](float x) { const unsigned short h = f32_to_bf16(x); return lo ? f32_to_bf16(x - bf16_to_f32(h)) : h; };
            for (int k = 0; k < kPoseFeat; ++k) out[blend16_index(k, v, c)] = enc(m->posedirs_host[(size_t)k * kV * 3 + v * 3 + c]);
            for (int k = 0; k < nb; ++k) out[blend16_index(kPoseFeat + k, v, c)] = enc(m->shapedirs_host[((size_t)v * 3 + c) * nb + k]);
        }
}

}  // namespace hmf

using namespace hmf;

static long long* g_lbs_cycles = nullptr;

extern "C" {

#ifdef HMF_EXPERIMENTS
/* diagnostic (tools/lbs_tune.py): device buffer of >= 256 int64; every k_lbs_mfma block stores its shader-clock span */
int hmf_debug_set_lbs_cycles(void* dev_ptr) { g_lbs_cycles = (long long*)dev_ptr; return HMF_OK; }
/* reads and clears the phase counters of k_lbs_mfma16 (8 uint64: staging + barrier waits, tile start -> first k-group, main loop, epilogue, wave tiles) */
int hmf_debug_lbs16_phases(unsigned long long* out_host) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HMF_HIP(hipDeviceSynchronize());
    HMF_HIP(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(hmf::g_lbs16_phase), sizeof(z)));
    HMF_HIP(hipMemcpyToSymbol(HIP_SYMBOL(hmf::g_lbs16_phase), z, sizeof(z)));
    return HMF_OK;
}
#endif

size_t hmf_smpl_packed_blend_floats(int) { return (size_t)kVT * kKG * 3 * 64 * 4; }
size_t hmf_smpl_packed_blend_index(int k, int v, int c) { return blend_index(k, v, c); }

int hmf_smpl_pack_blend_host(const hmf_smpl_host_t* m, float* out) {
    int rc = validate(m);
    if (rc) return rc;
    if (!out) return fail(HMF_ERR_INVALID, "hmf_smpl_pack_blend_host: null output");
    pack_blend(m, out);
    return HMF_OK;
}

int hmf_smpl_create(hmf_smpl_t** out, const hmf_smpl_host_t* m) {
    if (!out) return fail(HMF_ERR_INVALID, "hmf_smpl_create: null out");
    *out = nullptr;
    int rc = validate(m);
    if (rc) return rc;
    if (hmf_device_count() <= 0) return fail(HMF_ERR_NO_DEVICE, "hmf_smpl_create: no HIP device visible");
    const int nb = m->num_betas;
    // --- ELL skin weights -------------------------------------------------------------------------
    int slots = 1;
    for (int v = 0; v < kV; ++v) {
        int n = 0;
        for (int j = 0; j < kJ; ++j) n += m->lbs_weights_host[(size_t)v * kJ + j] != 0.0f;
        slots = std::max(slots, n);
    }
    if (slots > 4 && slots < 8) slots = 8;
    if (slots < 4) slots = 4;
    std::vector<int> soff((size_t)kVT * slots * 32, 0);
    std::vector<float> sw((size_t)kVT * slots * 32, 0.f);
    for (int v = 0; v < kV; ++v) {
        int s = 0;
        for (int j = 0; j < kJ; ++j) {
            const float w = m->lbs_weights_host[(size_t)v * kJ + j];
            if (w != 0.0f) {
                soff[((size_t)(v >> 5) * slots + s) * 32 + (v & 31)] = j * 48;
                sw[((size_t)(v >> 5) * slots + s) * 32 + (v & 31)] = w;
                ++s;
            }
        }
    }
    // --- the same weights, dense, as the bf16 hi | lo B operand of k_lbs_mfma16's transform blend (SmplDev::skin_w16) --------------
    std::vector<unsigned short> w16((size_t)kVT * 2 * kW16PerPiece, 0);
#ifdef HMF_EXPERIMENTS
    for (int v = 0; v < kV; ++v)
        for (int j = 0; j < kJ; ++j) {
            const float w = m->lbs_weights_host[(size_t)v * kJ + j];
            const unsigned short hi = f32_to_bf16(w), lo = f32_to_bf16(w - bf16_to_f32(hi));
            // joints 0..15: k-group of 16 (lane = vertex + 32 (k >> 3), element k & 7); joints 16..23: k-group of 8 (lane = vertex + 32 ((k - 16) >> 2), element (k - 16) & 3)
            const size_t at = j < 16 ? (size_t)(((j >> 3) * 32 + (v & 31)) * 8 + (j & 7)) : (size_t)(64 * 8 + ((((j - 16) >> 2) * 32 + (v & 31)) * 4 + ((j - 16) & 3)));
            w16[((size_t)(v >> 5) * 2 + 0) * kW16PerPiece + at] = hi;
            w16[((size_t)(v >> 5) * 2 + 1) * kW16PerPiece + at] = lo;
        }
#else
    w16.resize(8);                                        // the product library has no consumer of the table (SmplDev::skin_w16 stays a valid pointer)
#endif
    // --- rest joints as an affine function of beta (fp64 accumulate, rounded once) ------------------
    std::vector<float> jt(kJ * 3), jd((size_t)kJ * 3 * nb);
    for (int j = 0; j < kJ; ++j)
        for (int c = 0; c < 3; ++c) {
            double a = 0.0;
            for (int v = 0; v < kV; ++v) a += (double)m->J_regressor_host[(size_t)j * kV + v] * m->v_template_host[v * 3 + c];
            jt[j * 3 + c] = (float)a;
            for (int k = 0; k < nb; ++k) {
                double b = 0.0;
                for (int v = 0; v < kV; ++v)
                    b += (double)m->J_regressor_host[(size_t)j * kV + v] * m->shapedirs_host[((size_t)v * 3 + c) * nb + k];
                jd[((size_t)j * 3 + c) * nb + k] = (float)b;
            }
        }
    // --- CSR of the 66 gathered joints ----------------------------------------------------------------
    std::vector<int> ptr(1, 0), idx;
    std::vector<float> wts;
    for (int i = 0; i < 21; ++i) {
        const int v = m->extra_vertex_ids_host[i];
        if (v < 0 || v >= kV) return fail(HMF_ERR_INVALID, "hmf_smpl_create: extra vertex id %d out of range", v);
        idx.push_back(v); wts.push_back(1.0f); ptr.push_back((int)idx.size());
    }
    for (int i = 0; i < 45; ++i) {
        for (int v = 0; v < kV; ++v) {
            const float w = m->J_regressor_more_host[(size_t)i * kV + v];
            if (w != 0.0f) { idx.push_back(v); wts.push_back(w); }
        }
        ptr.push_back((int)idx.size());
    }
    // --- vertex-major transpose of that CSR (joints backward) ------------------------------------------
    std::vector<int> ju_vid, ju_ptr(1, 0), ju_out;
    std::vector<float> ju_w;
    {
        std::vector<int> slot(kV, -1);
        std::vector<std::vector<std::pair<int, float>>> per;
        for (int i = 0; i < 66; ++i)
            for (int e = ptr[i]; e < ptr[i + 1]; ++e) {
                if (slot[idx[e]] < 0) { slot[idx[e]] = (int)ju_vid.size(); ju_vid.push_back(idx[e]); per.emplace_back(); }
                per[slot[idx[e]]].push_back({i, wts[e]});
            }
        for (auto& l : per) {
            for (auto& pr : l) { ju_out.push_back(pr.first); ju_w.push_back(pr.second); }
            ju_ptr.push_back((int)ju_out.size());
        }
        if ((int)ju_vid.size() > kMaxJU) return fail(HMF_ERR_UNSUPPORTED, "hmf_smpl_create: %zu vertices feed the gathered joints (> %d)", ju_vid.size(), kMaxJU);
    }
    std::vector<int> ju_slot((size_t)kVT * 32, -1), jr_slot(idx.size());
    for (size_t u = 0; u < ju_vid.size(); ++u) ju_slot[ju_vid[u]] = (int)u;
    for (size_t e = 0; e < idx.size(); ++e) jr_slot[e] = ju_slot[idx[e]];
    // --- T-pose basis [nb][3V] -----------------------------------------------------------------------
    std::vector<float> tpd((size_t)nb * kV * 3);
    for (int v = 0; v < kV; ++v)
        for (int c = 0; c < 3; ++c)
            for (int k = 0; k < nb; ++k) tpd[(size_t)k * kV * 3 + v * 3 + c] = m->shapedirs_host[((size_t)v * 3 + c) * nb + k];
    const int tps = (nb + 1 + 3) & ~3, nq = nb + 1;
    std::vector<float> tpp((size_t)kV * 3 * tps, 0.f);
    std::vector<double> tpm((size_t)3 * nq + (size_t)nq * nq, 0.0);
    for (int v = 0; v < kV; ++v)
        for (int c = 0; c < 3; ++c) {
            float* row = &tpp[((size_t)v * 3 + c) * tps];
            row[0] = m->v_template_host[v * 3 + c];
            for (int k = 0; k < nb; ++k) row[1 + k] = m->shapedirs_host[((size_t)v * 3 + c) * nb + k];
            for (int i = 0; i < nq; ++i) {
                tpm[(size_t)c * nq + i] += (double)row[i] / kV;
                for (int j = 0; j < nq; ++j) tpm[(size_t)3 * nq + (size_t)i * nq + j] += (double)row[i] * (double)row[j];
            }
        }
    const int jbs = (((int)ju_vid.size() * 3 + 3) & ~3), jbk = kPoseFeat + nb + 1;
    std::vector<float> jub((size_t)jbk * std::max(jbs, 4), 0.f);
    for (size_t u = 0; u < ju_vid.size(); ++u)
        for (int c = 0; c < 3; ++c) {
            const int v = ju_vid[u];
            for (int k = 0; k < kPoseFeat; ++k) jub[(size_t)k * jbs + u * 3 + c] = m->posedirs_host[(size_t)k * kV * 3 + v * 3 + c];
            for (int k = 0; k < nb; ++k) jub[(size_t)(kPoseFeat + k) * jbs + u * 3 + c] = m->shapedirs_host[((size_t)v * 3 + c) * nb + k];
            jub[(size_t)(kPoseFeat + nb) * jbs + u * 3 + c] = m->v_template_host[v * 3 + c];
        }
    std::vector<float> blend((size_t)kVT * kKG * 3 * 64 * 4);
    pack_blend(m, blend.data());
    std::vector<unsigned short> blend16((size_t)kVT * kKG16 * 3 * 64 * 8), blend16l(blend16.size());
    pack_blend16(m, blend16.data());
    pack_blend16(m, blend16l.data(), true);

    // --- one device arena -----------------------------------------------------------------------------
    size_t off = 0;
    auto place = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_blend = place(blend.size() * 4), o_soff = place(soff.size() * 4), o_sw = place(sw.size() * 4);
    const size_t o_jt = place(jt.size() * 4), o_jd = place(jd.size() * 4), o_ptr = place(ptr.size() * 4);
    const size_t o_idx = place(idx.size() * 4), o_w = place(wts.size() * 4), o_tpd = place(tpd.size() * 4);
    const size_t o_tpt = place((size_t)kV * 3 * 4), o_b16 = place(blend16.size() * 2), o_b16l = place(blend16l.size() * 2);
    const size_t o_jv = place(ju_vid.size() * 4 + 4), o_jp = place(ju_ptr.size() * 4), o_jo = place(ju_out.size() * 4 + 4), o_jw = place(ju_w.size() * 4 + 4);
    const size_t o_jsl = place(ju_slot.size() * 4), o_jrs = place(jr_slot.size() * 4 + 4);
    const size_t o_tpp = place(tpp.size() * 4), o_tpm = place(tpm.size() * 8), o_jub = place(jub.size() * 4), o_w16 = place(w16.size() * 2);
    void* arena = nullptr;
    HMF_HIP(hipMalloc(&arena, off));
    char* base = (char*)arena;
    auto up = [&](size_t o, const void* src, size_t bytes) { return hipMemcpy(base + o, src, bytes, hipMemcpyHostToDevice); };
    hipError_t e = up(o_blend, blend.data(), blend.size() * 4);
    if (e == hipSuccess) e = up(o_soff, soff.data(), soff.size() * 4);
    if (e == hipSuccess) e = up(o_sw, sw.data(), sw.size() * 4);
    if (e == hipSuccess) e = up(o_jt, jt.data(), jt.size() * 4);
    if (e == hipSuccess) e = up(o_jd, jd.data(), jd.size() * 4);
    if (e == hipSuccess) e = up(o_ptr, ptr.data(), ptr.size() * 4);
    if (e == hipSuccess) e = up(o_idx, idx.data(), idx.size() * 4);
    if (e == hipSuccess) e = up(o_w, wts.data(), wts.size() * 4);
    if (e == hipSuccess) e = up(o_tpd, tpd.data(), tpd.size() * 4);
    if (e == hipSuccess) e = up(o_tpt, m->v_template_host, (size_t)kV * 3 * 4);
    if (e == hipSuccess) e = up(o_b16, blend16.data(), blend16.size() * 2);
    if (e == hipSuccess) e = up(o_b16l, blend16l.data(), blend16l.size() * 2);
    if (e == hipSuccess && !ju_vid.empty()) e = up(o_jv, ju_vid.data(), ju_vid.size() * 4);
    if (e == hipSuccess) e = up(o_jp, ju_ptr.data(), ju_ptr.size() * 4);
    if (e == hipSuccess && !ju_out.empty()) e = up(o_jo, ju_out.data(), ju_out.size() * 4);
    if (e == hipSuccess && !ju_w.empty()) e = up(o_jw, ju_w.data(), ju_w.size() * 4);
    if (e == hipSuccess) e = up(o_jsl, ju_slot.data(), ju_slot.size() * 4);
    if (e == hipSuccess && !jr_slot.empty()) e = up(o_jrs, jr_slot.data(), jr_slot.size() * 4);
    if (e == hipSuccess) e = up(o_tpp, tpp.data(), tpp.size() * 4);
    if (e == hipSuccess) e = up(o_tpm, tpm.data(), tpm.size() * 8);
    if (e == hipSuccess) e = up(o_jub, jub.data(), jub.size() * 4);
    if (e == hipSuccess) e = up(o_w16, w16.data(), w16.size() * 2);
    if (e != hipSuccess) {
        (void)hipFree(arena);
        return fail(HMF_ERR_HIP, "hmf_smpl_create: upload failed: %s", hipGetErrorString(e));
    }
    hmf_smpl* h = new hmf_smpl();
    h->arena = arena;
    SmplDev& d = h->d;
    d.num_betas = nb;
    d.skin_slots = slots;
    d.depth_max = 0;
    for (int j = 0; j < kJ; ++j) {
        d.parents[j] = m->parents_host[j];
        d.depth[j] = j == 0 ? 0 : d.depth[m->parents_host[j]] + 1;
        d.depth_max = std::max(d.depth_max, d.depth[j]);
    }
    d.blend = (float*)(base + o_blend);
    d.skin_off = (int*)(base + o_soff);
    d.skin_w = (float*)(base + o_sw);
    d.j_template = (float*)(base + o_jt);
    d.j_dirs = (float*)(base + o_jd);
    d.jr_ptr = (int*)(base + o_ptr);
    d.jr_idx = (int*)(base + o_idx);
    d.jr_w = (float*)(base + o_w);
    d.tp_dirs = (float*)(base + o_tpd);
    d.tp_tmpl = (float*)(base + o_tpt);
    d.blend16 = (unsigned short*)(base + o_b16);
    d.blend16_lo = (unsigned short*)(base + o_b16l);
    d.n_ju = (int)ju_vid.size();
    d.ju_vid = (int*)(base + o_jv);
    d.ju_ptr = (int*)(base + o_jp);
    d.ju_out = (int*)(base + o_jo);
    d.ju_w = (float*)(base + o_jw);
    d.ju_slot = (int*)(base + o_jsl);
    d.jr_slot = (int*)(base + o_jrs);
    d.tp_pack = (float*)(base + o_tpp);
    d.tp_mom = (double*)(base + o_tpm);
    d.tp_stride = tps;
    d.ju_blend = (float*)(base + o_jub);
    d.skin_w16 = (unsigned short*)(base + o_w16);
    d.ju_bstride = jbs;
    *out = h;
    return HMF_OK;
}

void hmf_smpl_destroy(hmf_smpl_t* h) {
    if (!h) return;
    (void)hipFree(h->arena);
    delete h;
}

static size_t ws_atr_bytes(int M) { return align_up((size_t)std::max(M, 1) * kAtr * sizeof(float), 256); }
static size_t ws_panel_bytes(int M) { return (size_t)((std::max(M, 1) + kRowsPerWG - 1) / kRowsPerWG) * kKpad * 64 * sizeof(float); }
static size_t ws_panel16_bytes(int M) { return (size_t)((std::max(M, 1) + kRowsPerWG - 1) / kRowsPerWG) * kRowsPerWG * kP16Stride * 2; }
static size_t ws_cj_bytes(int M) { return align_up((size_t)std::max(M, 1) * kMaxJU * 3 * sizeof(float), 256); }
static size_t ws_atr16_bytes(int M) { return (size_t)((std::max(M, 1) + kRowsPerWG - 1) / kRowsPerWG) * kAtr16Group * 2; }
#ifdef HMF_EXPERIMENTS
size_t hmf_smpl_lbs_workspace_bytes(int M) { return ws_atr_bytes(M) + ws_panel_bytes(M) + 2 * ws_panel16_bytes(M) + ws_cj_bytes(M) + ws_atr16_bytes(M); }
#else
size_t hmf_smpl_lbs_workspace_bytes(int M) { return ws_atr_bytes(M) + ws_panel_bytes(M) + 2 * ws_panel16_bytes(M) + ws_cj_bytes(M); }
#endif

static int smpl_lbs_impl(const hmf_smpl_t* h, const float* global_orient, int Mg, int gN, const float* body_pose, const float* betas,
                         int Mb, int M, float* verts, float* joints, float* full_pose, void* workspace, size_t workspace_bytes, int impl,
                         void* stream) {
    if (M == 0) return HMF_OK;                                       // empty batch: nothing to do, pointers may be null
    if (!h || !global_orient || !body_pose || !betas || !verts) return fail(HMF_ERR_INVALID, "hmf_smpl_lbs: null argument");
    if (M < 0 || (Mb != 1 && Mb != M) || (gN == 0 && Mg != 1 && Mg != M))
        return fail(HMF_ERR_INVALID, "hmf_smpl_lbs: M=%d Mb=%d Mg=%d (Mb, Mg must be 1 or M)", M, Mb, Mg);
    if (!workspace || workspace_bytes < hmf_smpl_lbs_workspace_bytes(M))
        return fail(HMF_ERR_WORKSPACE, "hmf_smpl_lbs: workspace %zu < %zu bytes", workspace_bytes, hmf_smpl_lbs_workspace_bytes(M));
#ifndef HMF_EXPERIMENTS
    if (impl < 0 || impl > 4) return fail(HMF_ERR_INVALID, "hmf_smpl_lbs: impl=%d (0 = fp32 MFMA, 1 = scalar cross-check, 2 = bf16 operands, 3 = bf16 hi+lo split operands, 4 = fp32 MFMA on 32-row groups)", impl);
#endif
    hipStream_t st = (hipStream_t)stream;
    float* atr = (float*)workspace;
    float* panel = (float*)((char*)workspace + ws_atr_bytes(M));
    unsigned short* panel16 = (unsigned short*)((char*)workspace + ws_atr_bytes(M) + ws_panel_bytes(M));
    unsigned short* panel16_lo = (unsigned short*)((char*)panel16 + ws_panel16_bytes(M));
    float* cjv_all = (float*)((char*)panel16_lo + ws_panel16_bytes(M));
    float* cjv = (joints && (impl & 15) != 1) ? cjv_all : nullptr;      // the scalar cross-check kernel keeps k_joints
    unsigned short* atr16 = nullptr;
#ifdef HMF_EXPERIMENTS
    if ((impl & 15) == 3 && (impl >> 4) >= 25 && (impl >> 4) <= 27) atr16 = (unsigned short*)((char*)cjv_all + ws_cj_bytes(M));      // the matrix-pipe epilogue variants read bf16 transform planes
#endif
    const bool split = (impl & 15) == 3;
    const bool bf16 = (impl & 15) == 2 || split;
    const int nrg_all = (M + kRowsPerWG - 1) / kRowsPerWG;
    {
        ProfScope ps(2, M, st);      // grid padded to whole row groups: dead rows zero their slice of the panel
        hipLaunchKernelGGL(k_rigid_chain, dim3(nrg_all * (kRowsPerWG / 8)), dim3(192), 0, st, h->d, global_orient, Mg, body_pose, betas,
                           Mb, M, atr, joints, bf16 ? (float*)nullptr : panel, bf16 ? panel16 : (unsigned short*)nullptr,
                           split ? panel16_lo : (unsigned short*)nullptr, gN, full_pose, atr16);
    }
    HMF_LAUNCH_CHECK();
    if ((impl & 15) == 1) {
        const long long n = (long long)M * kV;
        hipLaunchKernelGGL(k_lbs_scalar, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, h->d, body_pose, betas, Mb, M,
                           atr, verts);
    } else if (bf16) {
        ProfScope ps(12, M, st);
        const int nrg = (M + kRowsPerWG - 1) / kRowsPerWG;
        const int T = nrg * kVTperX;
        const int wpx = std::max(1, std::min(32, (T + 7) / 8));
        const dim3 grid(8 * wpx), block(512);
#define HMF_LBS16(S)                                                                                                                       \
    do {                                                                                                                                   \
        if (split) hipLaunchKernelGGL((k_lbs_mfma16<S, true>), grid, block, 0, st, h->d, M, nrg, wpx, panel16, panel16_lo, atr, verts, cjv, atr16);     \
        else hipLaunchKernelGGL((k_lbs_mfma16<S, false>), grid, block, 0, st, h->d, M, nrg, wpx, panel16, panel16_lo, atr, verts, cjv, atr16);          \
    } while (0)
#ifdef HMF_EXPERIMENTS
#define HMF_LBS16A(A) hipLaunchKernelGGL((k_lbs_mfma16<4, true, A>), grid, block, 0, st, h->d, M, nrg, wpx, panel16, panel16_lo, atr, verts, cjv, atr16)
        if (h->d.skin_slots == 4 && split && (impl >> 4)) {
            switch (impl >> 4) {
                case 1: HMF_LBS16A(1); break;
                case 2: HMF_LBS16A(2); break;
                case 3: HMF_LBS16A(3); break;
                case 4: HMF_LBS16A(4); break;
                case 5: HMF_LBS16A(5); break;
                case 6: HMF_LBS16A(6); break;
                case 7: HMF_LBS16A(7); break;
                case 8: HMF_LBS16A(8); break;
                case 10: HMF_LBS16A(10); break;
                case 12: HMF_LBS16A(12); break;
#define HMF_LBS16V(...) hipLaunchKernelGGL((k_lbs_mfma16<__VA_ARGS__>), grid, block, 0, st, h->d, M, nrg, wpx, panel16, panel16_lo, atr, verts, cjv, atr16)
                case 16: HMF_LBS16V(4, true, 0, 2, true, true); break;        // round 6 (tools/lbs16_tune.py): B ring two k-groups ahead
                case 17: HMF_LBS16V(4, true, 0, 1, false, false, false); break; // the round-5 form: requests unpinned, epilogue operands held through the loop, whole tiles
                case 18: HMF_LBS16V(4, true, 0, 2, false, true); break;
                case 19: HMF_LBS16V(4, true, 0, 3, true, true); break;
                case 21: HMF_LBS16V(4, true, 2, 1, true, true); break;        // the product form without its B stream
                case 22: HMF_LBS16V(4, true, 0, 1, true, true, true); break;         // + the half-tile rounds of k_lbs_mfma (3.5 rounds, waves de-phased): measured no gain
                case 23: HMF_LBS16V(4, true, 0, 1, true, true, false, true); break;   // + the next tile's first k-group requested before the epilogue: measured slower
                case 24: HMF_LBS16V(4, true, 0, 1, true, true, false, false, false, false); break;   // gather epilogue, transforms staged through registers before the first barrier
                case 25: HMF_LBS16V(4, true, 0, 1, true, true, false, false, true, false); break;
                case 26: HMF_LBS16V(4, true, 128, 1, true, true, false, false, true, true); break;   // matrix-pipe epilogue without the scheduling barrier between element pairs
                case 27: HMF_LBS16V(4, true, 8, 1, true, true, false, false, true, true); break;     // matrix-pipe epilogue without the vertex stores    // the gather epilogue (fp32 transforms, 4 slots) instead of the matrix-pipe blend
                case 64: HMF_LBS16V(4, true, 64, 1, true, true, false); break;      // the product form with phase stamps (hmf_debug_lbs16_phases)
                case 65: HMF_LBS16V(4, true, 64, 1, true, true, true); break;      // ... with the half tiles
                case 66: HMF_LBS16V(4, true, 64, 1, false, false, false); break;   // the round-5 form with phase stamps
#undef HMF_LBS16V
                default: HMF_LBS16A(0); break;
            }
        } else
#undef HMF_LBS16A
#endif
        if (h->d.skin_slots == 4) HMF_LBS16(4);
        else if (h->d.skin_slots == 8) HMF_LBS16(8);
        else HMF_LBS16(0);
#undef HMF_LBS16
    } else if ((impl & 15) == 4) {
        ProfScope ps(1, M, st);
        const int nrg32 = (M + 31) / 32;
        const int T = nrg32 * kVTperX;
        const int wpx = std::max(1, std::min(64, (T + 7) / 8));          // two work-groups per CU: up to 64 per XCD
        const dim3 grid(8 * wpx), block(512);
        if (h->d.skin_slots == 4) hipLaunchKernelGGL((k_lbs_mfma32<4>), grid, block, 0, st, h->d, M, nrg32, wpx, panel, atr, verts, cjv);
        else if (h->d.skin_slots == 8) hipLaunchKernelGGL((k_lbs_mfma32<8>), grid, block, 0, st, h->d, M, nrg32, wpx, panel, atr, verts, cjv);
        else hipLaunchKernelGGL((k_lbs_mfma32<0>), grid, block, 0, st, h->d, M, nrg32, wpx, panel, atr, verts, cjv);
    } else {
        ProfScope ps(1, M, st);
        const int nrg = (M + kRowsPerWG - 1) / kRowsPerWG;
        const int T = nrg * kVTperX;
        int wpx = (T + 7) / 8;                      // at least one tile per wave
        wpx = std::max(1, std::min(32, wpx));       // 32 CUs per XCD
        const dim3 grid(8 * wpx), block(512);
#define HMF_LBS(S, ABL) hipLaunchKernelGGL((k_lbs_mfma<S, ABL>), grid, block, 0, st, h->d, M, nrg, wpx, panel, atr, verts, g_lbs_cycles, lbs_cjv)
        float* lbs_cjv = cjv;
#ifdef HMF_EXPERIMENTS
        if ((impl >> 4) & 16) lbs_cjv = nullptr;            // timing: no compact joint-vertex stores in the epilogue (joints are then stale)
#endif
        if (h->d.skin_slots == 4) {
#ifdef HMF_EXPERIMENTS
            switch (impl >> 4) {          // impl = 16 * ABL: ablation variants for tools/lbs_tune.py; 0 = product kernel
                case 1: HMF_LBS(4, 1); break;        // no skinning epilogue
                case 3: HMF_LBS(4, 3); break;        // ... and no B-operand stream
                case 7: HMF_LBS(4, 7); break;        // ... and no A-operand LDS reads: bare MFMA loop
                case 16: HMF_LBS(4, 16); break;      // next k-group's operand requests NOT pinned (the round 1-5 form)
                case 32: HMF_LBS(4, 32); break;      // skinning transforms by LDS-DMA under the first main loops of a segment
                case 48: HMF_LBS(4, 48); break;
                case 8: case 24: HMF_LBS(4, 8); break;   // non-temporal staging loads of the panel + transforms (rejected)
                default: HMF_LBS(4, 0); break;
            }
#else
            HMF_LBS(4, 0);
#endif
        } else if (h->d.skin_slots == 8) {
            HMF_LBS(8, 0);
        } else {
            HMF_LBS(0, 0);
        }
#undef HMF_LBS
    }
    HMF_LAUNCH_CHECK();
    if (joints) {
        const int n = M * (kOutJ - kJ);
        ProfScope ps(3, M, st);
        if (cjv) hipLaunchKernelGGL(k_joints_compact, dim3((n + 255) / 256), dim3(256), 0, st, h->d, cjv, M, joints);
        else hipLaunchKernelGGL(k_joints, dim3((n + 255) / 256), dim3(256), 0, st, h->d, verts, M, joints);
        HMF_LAUNCH_CHECK();
    }
    return HMF_OK;
}

int hmf_smpl_lbs(const hmf_smpl_t* h, const float* global_orient, int Mg, const float* body_pose, const float* betas,
                 int Mb, int M, float* verts, float* joints, void* workspace, size_t workspace_bytes, int impl,
                 void* stream) {
    return smpl_lbs_impl(h, global_orient, Mg, 0, body_pose, betas, Mb, M, verts, joints, nullptr, workspace, workspace_bytes, impl, stream);
}

int hmf_smpl_lbs_packed(const hmf_smpl_t* h, const float* glob_R, int B, int N, const float* body_pose, const float* betas, float* verts,
                        float* joints, float* full_pose, void* workspace, size_t workspace_bytes, int impl, void* stream) {
    if (B <= 0 || N <= 0) return fail(HMF_ERR_INVALID, "hmf_smpl_lbs_packed: B=%d N=%d", B, N);
    if ((long long)B * N + B > (1LL << 30)) return fail(HMF_ERR_INVALID, "hmf_smpl_lbs_packed: B*N too large");
    const int M = B * N + B;
    return smpl_lbs_impl(h, glob_R, B, N, body_pose, betas, M, M, verts, joints, full_pose, workspace, workspace_bytes, impl, stream);
}

int hmf_smpl_tpose(const hmf_smpl_t* h, const float* betas, int M, float* verts, float* joints, void* stream) {
    if (M == 0) return HMF_OK;
    if (!h || !betas || !verts) return fail(HMF_ERR_INVALID, "hmf_smpl_tpose: null argument");
    if (M < 0) return fail(HMF_ERR_INVALID, "hmf_smpl_tpose: M=%d", M);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((kV * 3 / 2 + 255) / 256, (M + kTpRows - 1) / kTpRows);
    {
        ProfScope ps(4, M, st);
        hipLaunchKernelGGL(k_tpose, grid, dim3(256), 0, st, h->d, betas, M, verts);
    }
    HMF_LAUNCH_CHECK();
    if (joints) {   // identity pose: posed joints == rest joints (0..23); 24..89 gathered from the vertices
        hipLaunchKernelGGL(k_rest_joints, dim3((M * kJ + 255) / 256), dim3(256), 0, st, h->d, betas, M, joints);
        HMF_LAUNCH_CHECK();
        const int n = M * (kOutJ - kJ);
        hipLaunchKernelGGL(k_joints, dim3((n + 255) / 256), dim3(256), 0, st, h->d, verts, M, joints);
        HMF_LAUNCH_CHECK();
    }
    return HMF_OK;
}

int hmf_smpl_joints(const hmf_smpl_t* h, const float* global_orient, const float* body_pose, const float* betas, int M, float* joints, void* stream) {
    if (M == 0) return HMF_OK;
    if (!h || !global_orient || !body_pose || !betas || !joints || M < 0) return fail(HMF_ERR_INVALID, "hmf_smpl_joints: bad args");
    hipLaunchKernelGGL(k_smpl_joints_fwd, dim3(M), dim3(256), 0, (hipStream_t)stream, h->d, global_orient, body_pose, betas, M, joints);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

int hmf_smpl_joints_backward(const hmf_smpl_t* h, const float* global_orient, const float* body_pose, const float* betas, int M,
                             const float* g_joints, float* g_global_orient, float* g_body_pose, float* g_betas, void* stream) {
    if (M == 0) return HMF_OK;
    if (!h || !global_orient || !body_pose || !betas || !g_joints || !g_global_orient || !g_body_pose || !g_betas || M < 0)
        return fail(HMF_ERR_INVALID, "hmf_smpl_joints_backward: bad args");
    hipLaunchKernelGGL(k_smpl_joints_bwd, dim3(M), dim3(256), 0, (hipStream_t)stream, h->d, global_orient, body_pose, betas, M, g_joints,
                       g_global_orient, g_body_pose, g_betas);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

long long hmf_smpl_tpose_errors_workspace_bytes(int npairs) { return (long long)align_up((size_t)std::max(npairs, 1) * kTeChunks * 2 * sizeof(float), 256); }

static int tpose_errors_impl(const hmf_smpl_t* h, const float* betas_pred, const float* betas_target, int nimg, int group, int extra, float* out, void* ws,
                             long long ws_bytes, void* stream, const char* who) {
    const long long np = (long long)nimg * (group + extra);
    if (np == 0) return HMF_OK;
    if (!h || !betas_pred || !betas_target || !out || nimg < 0 || group < 0 || group + extra < 1 || np >= (1LL << 28)) return fail(HMF_ERR_INVALID, "%s: bad args", who);
    if (!ws || ws_bytes < hmf_smpl_tpose_errors_workspace_bytes((int)np)) return fail(HMF_ERR_WORKSPACE, "%s: workspace too small", who);
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)ws;
    const int npairs = (int)np;
    switch (h->d.num_betas) {
#define HMF_TE(NB_) case NB_: launch_tpose_errors<NB_>(h->d, betas_pred, betas_target, nimg, group, extra, npairs, partial, st); break;
        HMF_TE(1) HMF_TE(2) HMF_TE(3) HMF_TE(4) HMF_TE(5) HMF_TE(6) HMF_TE(7) HMF_TE(8) HMF_TE(9) HMF_TE(10) HMF_TE(11) HMF_TE(12) HMF_TE(13) HMF_TE(14)
        HMF_TE(15) HMF_TE(16)
#undef HMF_TE
        default: return fail(HMF_ERR_UNSUPPORTED, "%s: num_betas=%d", who, h->d.num_betas);
    }
    HMF_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_tpose_errors_finish, dim3((npairs + 63) / 64), dim3(64), 0, st, partial, npairs, out);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

int hmf_smpl_tpose_errors(const hmf_smpl_t* h, const float* betas_pred, const float* betas_target, int npairs, int group, float* out, void* ws,
                          long long ws_bytes, void* stream) {
    if (npairs < 0 || group < 1 || npairs % group) return fail(HMF_ERR_INVALID, "hmf_smpl_tpose_errors: npairs must be a multiple of group >= 1");
    return tpose_errors_impl(h, betas_pred, betas_target, npairs / group, group, 0, out, ws, ws_bytes, stream, "hmf_smpl_tpose_errors");
}

int hmf_smpl_tpose_errors_packed(const hmf_smpl_t* h, const float* betas_packed, const float* betas_target, int B, int N, float* out, void* ws,
                                 long long ws_bytes, void* stream) {
    return tpose_errors_impl(h, betas_packed, betas_target, B, N, 1, out, ws, ws_bytes, stream, "hmf_smpl_tpose_errors_packed");
}

int hmf_vertex_variance(const float* verts, int N, int V, float* avg_l2, float* dir_std, void* stream) {
    if (!verts || !avg_l2 || !dir_std || N <= 0 || V <= 0) return fail(HMF_ERR_INVALID, "hmf_vertex_variance: bad args");
    hipLaunchKernelGGL(k_vertex_variance, dim3((V + 255) / 256), dim3(256), 0, (hipStream_t)stream, verts, N, V, avg_l2, dir_std);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

}  // extern "C"
