// Host side of the flow: validation, repacking of the reference's Linear weights into MFMA A-operand
// order (flow_common.h "ARL"), handle creation.  Pure host code except the upload.
#include <vector>
#include <algorithm>
#include "flow_common.h"
#include "flow_handle.h"

namespace hmf {

static int validate(const hmf_flow_host_t* w) {
    if (!w || !w->parents_host || !w->w_isgc_host || !w->b_isgc_host || !w->w_ctx_host || !w->b_ctx_host ||
        !w->w_nn_host || !w->b_nn_host)
        return fail(HMF_ERR_INVALID, "hmf_flow: null weight array");
    if (w->isgc_dim != kIsgc || w->ctx_dim != kCtx || w->num_transforms != 2 || w->hidden[0] != 64 ||
        w->hidden[1] != 32 || w->hidden[2] != 32 || w->count_bins != kBins)
        return fail(HMF_ERR_UNSUPPORTED,
                    "hmf_flow: only the default flow (isgc 256, ctx 64, 2 spline couplings, hidden [64,32,32], 8 bins) "
                    "is built; got isgc=%d ctx=%d T=%d hidden=[%d,%d,%d] bins=%d",
                    w->isgc_dim, w->ctx_dim, w->num_transforms, w->hidden[0], w->hidden[1], w->hidden[2], w->count_bins);
    if (w->num_betas < 1 || w->num_betas > 16) return fail(HMF_ERR_UNSUPPORTED, "hmf_flow: num_betas=%d", w->num_betas);
    if (w->feat_dim < 4 || (w->feat_dim % 256) != 0)
        return fail(HMF_ERR_UNSUPPORTED, "hmf_flow: feat_dim=%d must be a multiple of 256", w->feat_dim);
    if (w->parents_host[0] != -1) return fail(HMF_ERR_INVALID, "hmf_flow: parents[0] must be -1");
    for (int j = 1; j < kJ; ++j)
        if (w->parents_host[j] < 0 || w->parents_host[j] >= j)
            return fail(HMF_ERR_INVALID, "hmf_flow: parents[%d]=%d is not an earlier joint", j, w->parents_host[j]);
    for (int j = 0; j < kP; ++j)
        if (!w->w_ctx_host[j] || !w->b_ctx_host[j]) return fail(HMF_ERR_INVALID, "hmf_flow: null ctx layer %d", j);
    for (int i = 0; i < kP * 8; ++i)
        if (!w->w_nn_host[i] || !w->b_nn_host[i]) return fail(HMF_ERR_INVALID, "hmf_flow: null hyper-net layer %d", i);
    return HMF_OK;
}

// models/humaniflow_model.py:16-30 -- ordered [parent, grand-parent, ...], root excluded
static void ancestors(const int* parents, int j, std::vector<int>& out) {
    out.clear();
    int p = parents[j + 1] - 1;
    while (p >= 0) {
        out.push_back(p);
        p = parents[p + 1] - 1;
    }
}

struct Packer {
    std::vector<float> buf;
    size_t place(size_t n) {
        size_t o = (buf.size() + 63) / 64 * 64;   // 256-byte aligned blocks
        buf.resize(o + n, 0.f);
        return o;
    }
};

// A-operand block for `ntiles` output tiles of 32 channels and `ks` k-steps, float4-grouped:
// layout [tile][ceil(ks/4)][lane 64][4];  value(tile, s, lane) = W(out_channel(tile, lane&31), kidx(s, lane>>5)) or 0.
template <class F>
static size_t pack_a(Packer& P, int ntiles, int ks, F value) {
    const int g4 = (ks + 3) / 4;
    const size_t o = P.place((size_t)ntiles * g4 * 64 * 4);
    for (int t = 0; t < ntiles; ++t)
        for (int s = 0; s < g4 * 4; ++s)
            for (int lane = 0; lane < 64; ++lane)
                P.buf[o + (((size_t)t * g4 + (s >> 2)) * 64 + lane) * 4 + (s & 3)] = s < ks ? value(t, s, lane) : 0.f;
    return o;
}

int build_flow_pack(const hmf_flow_host_t* w, FlowPack& fp) {
    int rc = validate(w);
    if (rc) return rc;
    Packer P;
    const int F = w->feat_dim, nb = w->num_betas, in_dim = F + nb + 12;
    fp.feat_dim = F; fp.nb = nb; fp.in_dim = in_dim; fp.radius = w->radius; fp.base_std = w->base_std;
    fp.tab.assign(kP, PartTab());
    fp.anc_off.assign((size_t)kP * 64, kAncZero);
    // global (non-part) blocks
    fp.o_wisgc = (int)P.place((size_t)kIsgc * in_dim);
    std::copy(w->w_isgc_host, w->w_isgc_host + (size_t)kIsgc * in_dim, P.buf.begin() + fp.o_wisgc);
    fp.o_bisgc = (int)P.place(kIsgc);
    std::copy(w->b_isgc_host, w->b_isgc_host + kIsgc, P.buf.begin() + fp.o_bisgc);
    // rank-nb shape update of the isgc layer as an A operand: 8 tiles, k-steps = ceil(nb/2), kidx = 2 s + q
    fp.o_wbeta = (int)pack_a(P, 8, (nb + 1) / 2, [&](int tl, int s, int lane) {
        const int k = 2 * s + (lane >> 5);
        return k < nb ? w->w_isgc_host[(size_t)(32 * tl + (lane & 31)) * in_dim + F + k] : 0.f;
    });

    std::vector<int> anc;
    int depth[kP];
    for (int j = 0; j < kP; ++j) {
        PartTab& t = fp.tab[j];
        ancestors(w->parents_host, j, anc);
        if ((int)anc.size() > kMaxAnc) return fail(HMF_ERR_UNSUPPORTED, "hmf_flow: part %d has %zu ancestors (> %d)", j, anc.size(), kMaxAnc);
        t.n_anc = (int)anc.size();
        depth[j] = t.n_anc;
        for (int i = 0; i < kMaxAnc; ++i) t.anc[i] = i < t.n_anc ? anc[i] : -1;
        const int ka = 9 * t.n_anc, kin = kIsgc + ka;
        t.kas = (ka + 1) / 2;
        t.kag = (t.kas + 3) / 4;
        for (int k = 0; k < ka; ++k) fp.anc_off[(size_t)j * 64 + k] = anc[k / 9] * 9 + k % 9;
        const float* wc = w->w_ctx_host[j];
        // context layer, isgc half: 2 tiles x 128 k-steps over the 256 isgc channels held in ARL (8 tiles)
        t.o_wc0 = (int)pack_a(P, 2, 128, [&](int tl, int s, int lane) {
            return wc[(size_t)(32 * tl + (lane & 31)) * kin + arl_channel(s >> 4, s & 15, lane >> 5)];
        });
        t.o_bctx = (int)P.place(kCtx);
        std::copy(w->b_ctx_host[j], w->b_ctx_host[j] + kCtx, P.buf.begin() + t.o_bctx);
        // context layer, ancestor half: kidx = 256 + 2 s + q
        t.o_wa = (int)pack_a(P, 2, t.kas, [&](int tl, int s, int lane) {
            const int k = 2 * s + (lane >> 5);
            return k < ka ? wc[(size_t)(32 * tl + (lane & 31)) * kin + kIsgc + k] : 0.f;
        });
        const float* const* wn = w->w_nn_host + (size_t)j * 8;
        const float* const* bn = w->b_nn_host + (size_t)j * 8;
        // first hyper-network layer of BOTH couplings: 4 tiles (c = tile/2), K = 64 context channels in ARL order
        t.o_w1 = (int)pack_a(P, 4, 32, [&](int tl, int s, int lane) {
            const float* W = wn[(tl >> 1) * 4 + 0];   // (64 x 65), input = [context(64), x1]
            return W[(size_t)(32 * (tl & 1) + (lane & 31)) * 65 + arl_channel(s >> 4, s & 15, lane >> 5)];
        });
        // extra k-step of the first layer: A[n][k=0] = W1[n][64] (the x1 input column), A[n][k=1] = bias[n];
        // the kernel supplies B = (q == 0 ? x1 : 1)
        t.o_w1e = (int)P.place(4 * 64);
        for (int tl = 0; tl < 4; ++tl)
            for (int lane = 0; lane < 64; ++lane) {
                const int c = tl >> 1, n = 32 * (tl & 1) + (lane & 31);
                P.buf[t.o_w1e + tl * 64 + lane] = (lane >> 5) == 0 ? wn[c * 4 + 0][(size_t)n * 65 + 64] : bn[c * 4 + 0][n];
            }
        for (int c = 0; c < 2; ++c) {
            const float* W2 = wn[c * 4 + 1];   // (32 x 64)
            const float* W3 = wn[c * 4 + 2];   // (32 x 32)
            const float* W4 = wn[c * 4 + 3];   // (62 x 32)
            t.o_w2[c] = (int)pack_a(P, 1, 32, [&](int, int s, int lane) { return W2[(size_t)(lane & 31) * 64 + arl_channel(s >> 4, s & 15, lane >> 5)]; });
            t.o_w3[c] = (int)pack_a(P, 1, 16, [&](int, int s, int lane) { return W3[(size_t)(lane & 31) * 32 + arl_channel(0, s & 15, lane >> 5)]; });
            t.o_w4[c] = (int)pack_a(P, 2, 16, [&](int tl, int s, int lane) {
                const int o = spline_orig_index(32 * tl + (lane & 31));
                return o < 0 ? 0.f : W4[(size_t)o * 32 + arl_channel(0, s & 15, lane >> 5)];
            });
            // bias k-steps (B operand = (q == 0 ? 1 : 0)): rows 0,1 = layers 2,3; rows 2,3 = the two tiles of layer 4
            t.o_we[c] = (int)P.place(4 * 64);
            for (int lane = 0; lane < 32; ++lane) {
                P.buf[t.o_we[c] + 0 * 64 + lane] = bn[c * 4 + 1][lane];
                P.buf[t.o_we[c] + 1 * 64 + lane] = bn[c * 4 + 2][lane];
                for (int tl = 0; tl < 2; ++tl) {
                    const int o = spline_orig_index(32 * tl + lane);
                    P.buf[t.o_we[c] + (2 + tl) * 64 + lane] = o < 0 ? 0.f : bn[c * 4 + 3][o];
                }
            }
        }
        // natural transposed copies for the teacher-forced kernels (lane = output channel)
        t.o_wctxT = (int)P.place((size_t)kin * kCtx);
        for (int k = 0; k < kin; ++k)
            for (int n = 0; n < kCtx; ++n) P.buf[t.o_wctxT + (size_t)k * kCtx + n] = wc[(size_t)n * kin + k];
        static const int Kd[4] = {65, 64, 32, 32}, Nd[4] = {64, 32, 32, 62};
        for (int c = 0; c < 2; ++c)
            for (int l = 0; l < 4; ++l) {
                t.o_nnT[c][l] = (int)P.place((size_t)Kd[l] * 64);      // padded to 64 columns
                for (int k = 0; k < Kd[l]; ++k)
                    for (int n = 0; n < Nd[l]; ++n) P.buf[t.o_nnT[c][l] + (size_t)k * 64 + n] = wn[c * 4 + l][(size_t)n * Kd[l] + k];
                t.o_nnb[c][l] = (int)P.place(64);
                std::copy(bn[c * 4 + l], bn[c * 4 + l] + Nd[l], P.buf.begin() + t.o_nnb[c][l]);
            }
    }
    // Rounds: greedy list schedule of the kinematic tree, <= kChainWaves independent parts per round, all ancestors in
    // earlier rounds, longest-remaining-chain first (so leaf parts fill the idle waves of later rounds).
    Levels& L = fp.lev;
    int height[kP], done[kP], placed = 0;
    for (int j = kP - 1; j >= 0; --j) {
        height[j] = 0;
        for (int k = j + 1; k < kP; ++k)
            if (w->parents_host[k + 1] - 1 == j) height[j] = std::max(height[j], height[k] + 1);
        done[j] = -1;
    }
    L.nlev = 0;
    while (placed < kP) {
        L.start[L.nlev] = placed;
        int chosen[kChainWaves], nc = 0;
        for (int slot = 0; slot < kChainWaves; ++slot) {
            int best = -1;
            for (int j = 0; j < kP; ++j) {
                if (done[j] >= 0) continue;
                const int par = w->parents_host[j + 1] - 1;
                if (par >= 0 && (done[par] < 0 || done[par] == L.nlev)) continue;       // parent not in an EARLIER round
                bool taken = false;
                for (int i = 0; i < nc; ++i) taken |= chosen[i] == j;
                if (!taken && (best < 0 || height[j] > height[best])) best = j;
            }
            if (best < 0) break;
            chosen[nc++] = best;
        }
        for (int i = 0; i < nc; ++i) { done[chosen[i]] = L.nlev; L.parts[placed++] = chosen[i]; }
        ++L.nlev;
    }
    for (int d2 = L.nlev; d2 <= kP; ++d2) L.start[d2] = placed;
    (void)depth;
    fp.arena.swap(P.buf);
    return HMF_OK;
}

}  // namespace hmf

using namespace hmf;

extern "C" {

size_t hmf_flow_packed_floats(const hmf_flow_host_t* w) {
    FlowPack fp;
    if (build_flow_pack(w, fp)) return 0;
    return fp.arena.size() + (size_t)kP * kPartTabInts + (size_t)kP * 64 + (1 + kP + 1 + kP) + 4;
}

// out = [arena floats | PartTab[23] as raw int32 bit patterns | anc_off[23][64] as raw int32 |
//        Levels {nlev, start[24], parts[23]} | o_wisgc, o_bisgc, o_wbeta, arena_floats as raw int32]
int hmf_flow_pack_host(const hmf_flow_host_t* w, float* out) {
    if (!out) return fail(HMF_ERR_INVALID, "hmf_flow_pack_host: null output");
    FlowPack fp;
    int rc = build_flow_pack(w, fp);
    if (rc) return rc;
    memcpy(out, fp.arena.data(), fp.arena.size() * 4);
    memcpy(out + fp.arena.size(), fp.tab.data(), (size_t)kP * sizeof(PartTab));
    memcpy(out + fp.arena.size() + (size_t)kP * kPartTabInts, fp.anc_off.data(), (size_t)kP * 64 * 4);
    float* pos = out + fp.arena.size() + (size_t)kP * kPartTabInts + (size_t)kP * 64;
    memcpy(pos, &fp.lev, sizeof(Levels));                     // nlev, start[24], parts[23]
    pos += sizeof(Levels) / 4;
    const int tail[4] = {fp.o_wisgc, fp.o_bisgc, fp.o_wbeta, (int)fp.arena.size()};
    memcpy(pos, tail, sizeof(tail));
    return HMF_OK;
}

int hmf_flow_create(hmf_flow_t** out, const hmf_flow_host_t* w) {
    if (!out) return fail(HMF_ERR_INVALID, "hmf_flow_create: null out");
    *out = nullptr;
    FlowPack fp;
    int rc = build_flow_pack(w, fp);
    if (rc) return rc;
    if (hmf_device_count() <= 0) return fail(HMF_ERR_NO_DEVICE, "hmf_flow_create: no HIP device visible");
    const size_t b_arena = align_up(fp.arena.size() * 4, 256), b_tab = align_up((size_t)kP * sizeof(PartTab), 256);
    const size_t b_anc = align_up((size_t)kP * 64 * 4, 256), b_ids = kP * 4;
    void* mem = nullptr;
    HMF_HIP(hipMalloc(&mem, b_arena + b_tab + b_anc + b_ids));
    char* base = (char*)mem;
    hipError_t e = hipMemcpy(base, fp.arena.data(), fp.arena.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(base + b_arena, fp.tab.data(), (size_t)kP * sizeof(PartTab), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(base + b_arena + b_tab, fp.anc_off.data(), (size_t)kP * 64 * 4, hipMemcpyHostToDevice);
    int ids[kP];
    for (int j = 0; j < kP; ++j) ids[j] = j;
    if (e == hipSuccess) e = hipMemcpy(base + b_arena + b_tab + b_anc, ids, b_ids, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(mem);
        return fail(HMF_ERR_HIP, "hmf_flow_create: upload failed: %s", hipGetErrorString(e));
    }
    hmf_flow* h = new hmf_flow();
    h->mem = mem;
    h->host_tab = fp.tab;
    h->d_ids = (const int*)(base + b_arena + b_tab + b_anc);
    FlowDev& d = h->d;
    d.feat_dim = fp.feat_dim; d.nb = fp.nb; d.in_dim = fp.in_dim; d.radius = fp.radius; d.base_std = fp.base_std;
    d.arena = (const float*)base;
    d.tab = (const PartTab*)(base + b_arena);
    d.anc_off = (const int*)(base + b_arena + b_tab);
    d.o_wisgc = fp.o_wisgc; d.o_bisgc = fp.o_bisgc; d.o_wbeta = fp.o_wbeta;
    d.lev = fp.lev;
    *out = h;
    return HMF_OK;
}

void hmf_flow_destroy(hmf_flow_t* h) {
    if (!h) return;
    (void)hipFree(h->mem);
    delete h;
}

}  // extern "C"
