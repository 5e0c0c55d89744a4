// Host side of the flow: validation, repacking of the reference's Linear weights into MFMA A-operand
// order (flow_common.h "ARL"), handle creation.  Pure host code except the upload.
#include <vector>
#include <algorithm>
#include "flow_common.h"
#include "flow_handle.h"

namespace hmf {

// default configuration = what the MFMA kernels are built for; anything else (or force_generic) goes to flow_generic.hip
static bool is_generic(const hmf_flow_host_t* w) {
    if (w->force_generic || w->transform_type != 0 || w->permute_type != 0 || w->num_transforms != 2) return true;
    if (w->num_hidden != 0) return !(w->num_hidden == 3 && w->hidden_dims[0] == 64 && w->hidden_dims[1] == 32 && w->hidden_dims[2] == 32);
    return !(w->hidden[0] == 64 && w->hidden[1] == 32 && w->hidden[2] == 32);
}
static const int* gen_hidden(const hmf_flow_host_t* w, int& nh) {
    if (w->num_hidden != 0) { nh = w->num_hidden; return w->hidden_dims; }
    nh = 3;
    return w->hidden;
}

static int validate(const hmf_flow_host_t* w) {
    if (!w || !w->parents_host || !w->w_isgc_host || !w->b_isgc_host || !w->w_ctx_host || !w->b_ctx_host ||
        !w->w_nn_host || !w->b_nn_host)
        return fail(HMF_ERR_INVALID, "hmf_flow: null weight array");
    if (w->isgc_dim != kIsgc || w->ctx_dim != kCtx || w->count_bins != kBins)
        return fail(HMF_ERR_UNSUPPORTED, "hmf_flow: INPUT_SHAPE_GLOB_CAM_FEATS_DIM 256, CONTEXT_DIM 64 and 8 spline bins are built; got isgc=%d ctx=%d bins=%d",
                    w->isgc_dim, w->ctx_dim, w->count_bins);
    if (is_generic(w)) {
        if (w->num_transforms < 1 || w->num_transforms > kMaxT) return fail(HMF_ERR_UNSUPPORTED, "hmf_flow: NUM_TRANSFORMS=%d outside [1,%d]", w->num_transforms, kMaxT);
        if (w->transform_type < 0 || w->transform_type > 1 || w->permute_type < 0 || w->permute_type > 2)
            return fail(HMF_ERR_UNSUPPORTED, "hmf_flow: transform_type=%d permute_type=%d", w->transform_type, w->permute_type);
        int nh;
        const int* hd = gen_hidden(w, nh);
        if (nh < 1 || nh > kMaxLayers - 1) return fail(HMF_ERR_UNSUPPORTED, "hmf_flow: %d hidden layers (1..%d are built)", nh, kMaxLayers - 1);
        for (int i = 0; i < nh; ++i)
            if (hd[i] < 1 || hd[i] > 64) return fail(HMF_ERR_UNSUPPORTED, "hmf_flow: hidden width %d outside [1,64]", hd[i]);
        if (w->permute_type == 2) {
            if (w->plu_num_hidden < 1 || w->plu_num_hidden > kMaxLayers - 1 || !w->w_plu_host || !w->b_plu_host)
                return fail(HMF_ERR_INVALID, "hmf_flow: conditional_linear_plu needs PERMUTE_NN_HIDDEN_DIMS (1..%d layers) and its weights", kMaxLayers - 1);
            for (int i = 0; i < w->plu_num_hidden; ++i)
                if (w->plu_hidden_dims[i] < 1 || w->plu_hidden_dims[i] > 64) return fail(HMF_ERR_UNSUPPORTED, "hmf_flow: PLU hidden width %d", w->plu_hidden_dims[i]);
        }
        if (w->permute_type == 1 && !w->lu_host) return fail(HMF_ERR_INVALID, "hmf_flow: linear_plu needs the LU parameters");
    } else if (w->num_transforms != 2 || w->hidden[0] != 64 || w->hidden[1] != 32 || w->hidden[2] != 32) {
        return fail(HMF_ERR_UNSUPPORTED, "hmf_flow: inconsistent default configuration T=%d hidden=[%d,%d,%d]", w->num_transforms, w->hidden[0], w->hidden[1], w->hidden[2]);
    }
    if (w->num_betas < 1 || w->num_betas > 16) return fail(HMF_ERR_UNSUPPORTED, "hmf_flow: num_betas=%d", w->num_betas);
    if (w->feat_dim < 4 || (w->feat_dim % 256) != 0)
        return fail(HMF_ERR_UNSUPPORTED, "hmf_flow: feat_dim=%d must be a multiple of 256", w->feat_dim);
    if (w->parents_host[0] != -1) return fail(HMF_ERR_INVALID, "hmf_flow: parents[0] must be -1");
    for (int j = 1; j < kJ; ++j)
        if (w->parents_host[j] < 0 || w->parents_host[j] >= j)
            return fail(HMF_ERR_INVALID, "hmf_flow: parents[%d]=%d is not an earlier joint", j, w->parents_host[j]);
    for (int j = 0; j < kP; ++j)
        if (!w->w_ctx_host[j] || !w->b_ctx_host[j]) return fail(HMF_ERR_INVALID, "hmf_flow: null ctx layer %d", j);
    int nh;
    gen_hidden(w, nh);
    const int per_part = is_generic(w) ? w->num_transforms * (nh + 1) : 8;
    for (int i = 0; i < kP * per_part; ++i)
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
    fp.generic = is_generic(w);
    fp.T = w->num_transforms; fp.transform_type = w->transform_type; fp.permute_type = w->permute_type;
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
        t.o_bctx = (int)P.place(kCtx);
        std::copy(w->b_ctx_host[j], w->b_ctx_host[j] + kCtx, P.buf.begin() + t.o_bctx);
        // natural transposed copy of the context layer (lane = output channel): teacher-forced kernels, backward, generic variants
        // ... followed by the NATURAL copy of its ancestor columns [64 out][64: lane = ancestor entry k < 9a, zero beyond] for the backward sweep
        // (d L / d ancestors = W_anc^T . delta as a lane-parallel sum over outputs: no cross-lane reduction); offset = o_wctxT + kin * 64
        t.o_wctxT = (int)P.place((size_t)kin * kCtx + (size_t)kCtx * 64);
        for (int k = 0; k < kin; ++k)
            for (int n = 0; n < kCtx; ++n) P.buf[t.o_wctxT + (size_t)k * kCtx + n] = wc[(size_t)n * kin + k];
        for (int n = 0; n < kCtx; ++n)
            for (int k = 0; k < ka; ++k) P.buf[t.o_wctxT + (size_t)kin * kCtx + (size_t)n * 64 + k] = wc[(size_t)n * kin + kIsgc + k];
        if (fp.generic) continue;
        // context layer, isgc half: 2 tiles x 128 k-steps over the 256 isgc channels held in ARL (8 tiles)
        t.o_wc0 = (int)pack_a(P, 2, 128, [&](int tl, int s, int lane) {
            return wc[(size_t)(32 * tl + (lane & 31)) * kin + arl_channel(s >> 4, s & 15, lane >> 5)];
        });
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
        // ---- the same layers for 16-row tiles (flow_common.h "ARL16"): A operand lane = (i = lane & 15 out channel of the 16-tile, g = lane >> 4 the
        // k index inside the step)
        {
            t.kas16 = (ka + 3) / 4;
            t.kag16 = (t.kas16 + 3) / 4;
            t.o16_wa = (int)pack_a(P, 4, t.kas16, [&](int tl, int s, int lane) {
                const int k = 4 * s + (lane >> 4);
                return k < ka ? wc[(size_t)(16 * tl + (lane & 15)) * kin + kIsgc + k] : 0.f;
            });
            t.o16_w1 = (int)pack_a(P, 8, 16, [&](int tl, int s, int lane) {
                const float* W = wn[(tl >> 2) * 4 + 0];   // (64 x 65), input = [context(64), x1]
                return W[(size_t)(16 * (tl & 3) + (lane & 15)) * 65 + arl16_channel(s, lane >> 4)];
            });
            t.o16_w1e = (int)P.place(8 * 64);
            for (int tl = 0; tl < 8; ++tl)
                for (int lane = 0; lane < 64; ++lane) {
                    const int c = tl >> 2, n = 16 * (tl & 3) + (lane & 15), g = lane >> 4;
                    P.buf[t.o16_w1e + tl * 64 + lane] = g == 0 ? wn[c * 4 + 0][(size_t)n * 65 + 64] : (g == 1 ? bn[c * 4 + 0][n] : 0.f);
                }
            for (int c = 0; c < 2; ++c) {
                const float* W2 = wn[c * 4 + 1];   // (32 x 64)
                const float* W3 = wn[c * 4 + 2];   // (32 x 32)
                const float* W4 = wn[c * 4 + 3];   // (62 x 32)
                t.o16_w2[c] = (int)pack_a(P, 2, 16, [&](int tl, int s, int lane) { return W2[(size_t)(16 * tl + (lane & 15)) * 64 + arl16_channel(s, lane >> 4)]; });
                t.o16_w3[c] = (int)pack_a(P, 2, 8, [&](int tl, int s, int lane) { return W3[(size_t)(16 * tl + (lane & 15)) * 32 + arl16_channel(s, lane >> 4)]; });
                t.o16_w4[c] = (int)pack_a(P, 4, 8, [&](int tl, int s, int lane) {
                    const int o = spline_orig_index16(tl, lane & 15);
                    return o < 0 ? 0.f : W4[(size_t)o * 32 + arl16_channel(s, lane >> 4)];
                });
                t.o16_we[c] = (int)P.place(8 * 64);
                for (int lane = 0; lane < 16; ++lane) {                                   // lane group 0 carries the bias (B operand = 1 there, 0 elsewhere)
                    for (int tl = 0; tl < 2; ++tl) {
                        P.buf[t.o16_we[c] + (0 + tl) * 64 + lane] = bn[c * 4 + 1][16 * tl + lane];
                        P.buf[t.o16_we[c] + (2 + tl) * 64 + lane] = bn[c * 4 + 2][16 * tl + lane];
                    }
                    for (int tl = 0; tl < 4; ++tl) {
                        const int o = spline_orig_index16(tl, lane);
                        P.buf[t.o16_we[c] + (4 + tl) * 64 + lane] = o < 0 ? 0.f : bn[c * 4 + 3][o];
                    }
                }
            }
        }
        // natural transposed copies of the hyper-networks for the teacher-forced / backward kernels (lane = output channel)
        static const int Kd[4] = {65, 64, 32, 32}, Nd[4] = {64, 32, 32, 62};
        for (int c = 0; c < 2; ++c)
            for (int l = 0; l < 4; ++l) {
                // [Kd][64] transposed (lane = output), then the natural copy [64 rows = output][64: lane = input k < min(Kd, 64)] at + Kd * 64
                t.o_nnT[c][l] = (int)P.place((size_t)Kd[l] * 64 + 64 * 64);
                for (int k = 0; k < Kd[l]; ++k)
                    for (int n = 0; n < Nd[l]; ++n) P.buf[t.o_nnT[c][l] + (size_t)k * 64 + n] = wn[c * 4 + l][(size_t)n * Kd[l] + k];
                for (int n = 0; n < Nd[l]; ++n)
                    for (int k = 0; k < std::min(Kd[l], 64); ++k) P.buf[t.o_nnT[c][l] + (size_t)Kd[l] * 64 + (size_t)n * 64 + k] = wn[c * 4 + l][(size_t)n * Kd[l] + k];
                t.o_nnb[c][l] = (int)P.place(64);
                std::copy(bn[c * 4 + l], bn[c * 4 + l] + Nd[l], P.buf.begin() + t.o_nnb[c][l]);
            }
    }
    if (fp.generic) {
        int nh;
        const int* hd = gen_hidden(w, nh);
        const int out_dim = w->transform_type == 0 ? 62 : 2;
        auto pack_net = [&](GenNet& g, const float* const* W, const float* const* Bv, int in_dim, const int* hidden, int nhid, int odim) {
            g.nl = nhid + 1;
            int k = in_dim;
            for (int l = 0; l <= nhid; ++l) {
                const int n = l < nhid ? hidden[l] : odim;
                g.K[l] = k; g.N[l] = n;
                // W^T [K][64] (lane = output), then the NATURAL copy [64 rows = output][64: lane = input k < min(K, 64)] at + K * 64 for the reverse sweeps
                g.o_w[l] = (int)P.place((size_t)k * 64 + 64 * 64);
                for (int kk = 0; kk < k; ++kk)
                    for (int nn = 0; nn < n; ++nn) P.buf[g.o_w[l] + (size_t)kk * 64 + nn] = W[l][(size_t)nn * k + kk];
                for (int nn = 0; nn < n; ++nn)
                    for (int kk = 0; kk < std::min(k, 64); ++kk) P.buf[g.o_w[l] + (size_t)k * 64 + (size_t)nn * 64 + kk] = W[l][(size_t)nn * k + kk];
                g.o_b[l] = (int)P.place(64);
                std::copy(Bv[l], Bv[l] + n, P.buf.begin() + g.o_b[l]);
                k = n;
            }
        };
        fp.gtab.assign(kP, GenPartTab());
        for (int j = 0; j < kP; ++j)
            for (int t = 0; t < fp.T; ++t) {
                GenPartTab& g = fp.gtab[j];
                const size_t base = ((size_t)j * fp.T + t) * (nh + 1);
                pack_net(g.tr[t], w->w_nn_host + base, w->b_nn_host + base, kCtx + 1, hd, nh, out_dim);
                g.plu[t].nl = 0;
                g.o_lu[t] = -1;
                if (w->permute_type == 2) {
                    const size_t pb = ((size_t)j * fp.T + t) * (w->plu_num_hidden + 1);
                    for (int l = 0; l <= w->plu_num_hidden; ++l)
                        if (!w->w_plu_host[pb + l] || !w->b_plu_host[pb + l]) return fail(HMF_ERR_INVALID, "hmf_flow: null PLU layer");
                    pack_net(g.plu[t], w->w_plu_host + pb, w->b_plu_host + pb, kCtx, w->plu_hidden_dims, w->plu_num_hidden, 9);
                } else if (w->permute_type == 1) {
                    if (!w->lu_host[(size_t)j * fp.T + t]) return fail(HMF_ERR_INVALID, "hmf_flow: null LU parameter");
                    g.o_lu[t] = (int)P.place(16);
                    std::copy(w->lu_host[(size_t)j * fp.T + t], w->lu_host[(size_t)j * fp.T + t] + 9, P.buf.begin() + g.o_lu[t]);
                }
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

// shadow of `w` whose tensors point into `idx` (values = flat position + 1) at the same offsets they have inside the caller's flat buffer
static int shadow_struct(const hmf_flow_host_t* w, const float* flat_base, size_t flat_count, const float* idx, hmf_flow_host_t& sh,
                         std::vector<const float*>& store) {
    bool ok = true;
    auto tr = [&](const float* p) -> const float* {
        if (!p) return nullptr;
        if (p < flat_base || p >= flat_base + flat_count) { ok = false; return nullptr; }
        return idx + (p - flat_base);
    };
    sh = *w;
    sh.w_isgc_host = tr(w->w_isgc_host);
    sh.b_isgc_host = tr(w->b_isgc_host);
    int nh;
    gen_hidden(w, nh);
    const bool generic = is_generic(w);
    const size_t n_nn = (size_t)kP * (generic ? w->num_transforms * (nh + 1) : 8);
    const size_t n_plu = w->permute_type == 2 ? (size_t)kP * w->num_transforms * (w->plu_num_hidden + 1) : 0;
    const size_t n_lu = w->permute_type == 1 ? (size_t)kP * w->num_transforms : 0;
    store.assign(2 * kP + 2 * n_nn + 2 * n_plu + n_lu, nullptr);
    const float** q = store.data();
    auto arr = [&](const float* const* src, size_t n) -> const float* const* {
        if (!src || n == 0) return nullptr;
        const float** dst = q;
        for (size_t i = 0; i < n; ++i) dst[i] = tr(src[i]);
        q += n;
        return dst;
    };
    sh.w_ctx_host = arr(w->w_ctx_host, kP);
    sh.b_ctx_host = arr(w->b_ctx_host, kP);
    sh.w_nn_host = arr(w->w_nn_host, n_nn);
    sh.b_nn_host = arr(w->b_nn_host, n_nn);
    sh.w_plu_host = arr(w->w_plu_host, n_plu);
    sh.b_plu_host = arr(w->b_plu_host, n_plu);
    sh.lu_host = arr(w->lu_host, n_lu);
    return ok ? HMF_OK : fail(HMF_ERR_INVALID, "hmf_flow_create_updatable: a weight tensor lies outside the flat host buffer");
}

static int create_impl(hmf_flow_t** out, const hmf_flow_host_t* w, const float* flat_base, size_t flat_count) {
    if (!out) return fail(HMF_ERR_INVALID, "hmf_flow_create: null out");
    *out = nullptr;
    FlowPack fp;
    int rc = build_flow_pack(w, fp);
    if (rc) return rc;
    if (hmf_device_count() <= 0) return fail(HMF_ERR_NO_DEVICE, "hmf_flow_create: no HIP device visible");
    // index map for device-side re-packing: every packer only COPIES weights, so packing an array whose values are their own flat position
    // (+1; 0 = padding) yields, per arena slot, where its value comes from
    std::vector<int> map;
    if (flat_base) {
        if (flat_count == 0 || flat_count >= (1u << 24)) return fail(HMF_ERR_UNSUPPORTED, "hmf_flow_create_updatable: %zu parameters (must be < 2^24)", flat_count);
        std::vector<float> idx(flat_count);
        for (size_t i = 0; i < flat_count; ++i) idx[i] = (float)(i + 1);
        hmf_flow_host_t sh;
        std::vector<const float*> store;
        rc = shadow_struct(w, flat_base, flat_count, idx.data(), sh, store);
        if (rc) return rc;
        FlowPack fi;
        rc = build_flow_pack(&sh, fi);
        if (rc) return rc;
        if (fi.arena.size() != fp.arena.size()) return fail(HMF_ERR_INVALID, "hmf_flow_create_updatable: index pack has a different size");
        map.resize(fi.arena.size());
        for (size_t i = 0; i < fi.arena.size(); ++i) map[i] = (int)fi.arena[i];
    }
    const size_t b_arena = align_up(fp.arena.size() * 4, 256), b_tab = align_up((size_t)kP * sizeof(PartTab), 256);
    const size_t b_anc = align_up((size_t)kP * 64 * 4, 256), b_ids = align_up(kP * 4, 256);
    const size_t b_gen = align_up(fp.generic ? (size_t)kP * sizeof(GenPartTab) : 0, 256);
    const size_t b_map = map.size() * 4;
    void* mem = nullptr;
    HMF_HIP(hipMalloc(&mem, b_arena + b_tab + b_anc + b_ids + b_gen + b_map));
    char* base = (char*)mem;
    hipError_t e = hipMemcpy(base, fp.arena.data(), fp.arena.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(base + b_arena, fp.tab.data(), (size_t)kP * sizeof(PartTab), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(base + b_arena + b_tab, fp.anc_off.data(), (size_t)kP * 64 * 4, hipMemcpyHostToDevice);
    int ids[kP];
    for (int j = 0; j < kP; ++j) ids[j] = j;
    if (e == hipSuccess) e = hipMemcpy(base + b_arena + b_tab + b_anc, ids, kP * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && fp.generic) e = hipMemcpy(base + b_arena + b_tab + b_anc + b_ids, fp.gtab.data(), (size_t)kP * sizeof(GenPartTab), hipMemcpyHostToDevice);
    if (e == hipSuccess && b_map) e = hipMemcpy(base + b_arena + b_tab + b_anc + b_ids + b_gen, map.data(), b_map, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(mem);
        return fail(HMF_ERR_HIP, "hmf_flow_create: upload failed: %s", hipGetErrorString(e));
    }
    hmf_flow* h = new hmf_flow();
    h->mem = mem;
    h->host_tab = fp.tab;
    h->d_ids = (const int*)(base + b_arena + b_tab + b_anc);
    h->arena_floats = fp.arena.size();
    h->flat_count = flat_base ? flat_count : 0;
    h->d_map = b_map ? (const int*)(base + b_arena + b_tab + b_anc + b_ids + b_gen) : nullptr;
    FlowDev& d = h->d;
    d.feat_dim = fp.feat_dim; d.nb = fp.nb; d.in_dim = fp.in_dim; d.radius = fp.radius; d.base_std = fp.base_std;
    d.arena = (const float*)base;
    d.tab = (const PartTab*)(base + b_arena);
    d.anc_off = (const int*)(base + b_arena + b_tab);
    d.o_wisgc = fp.o_wisgc; d.o_bisgc = fp.o_bisgc; d.o_wbeta = fp.o_wbeta;
    d.lev = fp.lev;
    h->g.enabled = fp.generic ? 1 : 0;
    if (fp.generic) {
        int nh_;
        gen_hidden(w, nh_);
        h->gen_layers_tr = nh_ + 1;
        h->gen_layers_plu = w->permute_type == 2 ? w->plu_num_hidden + 1 : 0;
    }
    h->g.T = fp.T; h->g.transform_type = fp.transform_type; h->g.permute_type = fp.permute_type;
    h->g.tab = fp.generic ? (const GenPartTab*)(base + b_arena + b_tab + b_anc + b_ids) : nullptr;
    *out = h;
    return HMF_OK;
}

int hmf_flow_create(hmf_flow_t** out, const hmf_flow_host_t* w) { return create_impl(out, w, nullptr, 0); }

int hmf_flow_create_updatable(hmf_flow_t** out, const hmf_flow_host_t* w, const float* flat_base, size_t flat_count) {
    if (!flat_base) return fail(HMF_ERR_INVALID, "hmf_flow_create_updatable: null flat buffer");
    return create_impl(out, w, flat_base, flat_count);
}

__global__ void k_flow_repack(float* __restrict__ arena, const int* __restrict__ map, const float* __restrict__ flat, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int m = map[i];
        arena[i] = m > 0 ? flat[m - 1] : 0.f;
    }
}

int hmf_flow_update_from_device(hmf_flow_t* h, const float* flat_dev, size_t flat_count, void* stream) {
    if (!h || !flat_dev) return fail(HMF_ERR_INVALID, "hmf_flow_update_from_device: null argument");
    if (!h->d_map || flat_count != h->flat_count)
        return fail(HMF_ERR_INVALID, "hmf_flow_update_from_device: handle was not created with hmf_flow_create_updatable for %zu parameters", flat_count);
    const size_t n = h->arena_floats;
    hipLaunchKernelGGL(k_flow_repack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (float*)h->d.arena, h->d_map, flat_dev, n);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}

void hmf_flow_destroy(hmf_flow_t* h) {
    if (!h) return;
    (void)hipFree(h->mem);
    delete h;
}

}  // extern "C"
