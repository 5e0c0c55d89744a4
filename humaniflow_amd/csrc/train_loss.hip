// Training loss of one step (train/train_humaniflow.py:283-343 through losses/humaniflow_loss.py:25-79 with LOSS.J2D_LOSS_ON = 'point_est+samples', no
// point-estimate 3D terms) and its gradients in TWO launches.  As torch operators the same arithmetic is ~150 launches of a few hundred elements each
// per step (select the 17 COCO joints, flip about x, orthographic projection, normalise the targets, masked MSE, Normal log-density, MSE of the global
// rotation, the weighted sum -- and every one of them again in the backward pass): ~0.6 ms of a 13.8 ms step spent on launch boundaries.
//   pose      -sum_{b,p} lp[b][p]                                   / (B P)            (mean)
//   shape     sum_b sum_d [ (t - mu)^2 / (2 sigma^2) + log sigma + log sqrt(2 pi) ]  / B
//   joints2D  sum_{b,s,j,c} vis[b][j] (s_b (q + t_b) - (2 T / wh - 1))^2            / max(1, S sum vis * 2),   q = (x, -y) of COCO joint j, S = N + 1
//   glob      sum_{b,e} (R - R*)^2                                                   / (9 B)
// One work-group per image writes that image's gradients (they depend on the other images only through the visible-joint count, which every
// work-group recomputes from the 17 B flags) and four partial sums; a one-wave kernel adds the partial sums in a fixed order.  Deterministic.
#include "hmf_common.h"

namespace hmf {

struct TrainLossArgs {
    const float* lp;            // (B, P)
    const float* mode;          // (B, nb)
    const float* log_std;       // (B, nb)
    const float* shape_t;       // (B, nb)
    const float* joints;        // (B N + B, 90, 3): rows b N + n = sample n of image b, rows B N + b = the point estimate
    const float* cam;           // (B, 3): s, tx, ty
    const float* j2d_t;         // (B, 17, 2) pixels
    const unsigned char* vis;   // (B, 17)
    const float* glob;          // (B, 9)
    const float* glob_t;        // (B, 9)
    int coco[17];
    int B, N, P, nb, mean;
    float img_wh, w_pose, w_shape, w_j2d, w_glob;
    float* g_lp; float* g_mode; float* g_log_std; float* g_joints; float* g_cam; float* g_glob;
    float* partial;             // (B, 4)
    float* loss;                // (1)
};

__device__ __forceinline__ float block_sum(float v, float* red, int tid) {        // 256 threads; every thread gets the sum
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void k_train_loss(const TrainLossArgs a) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int S = a.N + 1, BN = a.B * a.N;
    float nv = 0.f;
    for (int i = tid; i < a.B * 17; i += 256) nv += a.vis[i] ? 1.f : 0.f;
    nv = block_sum(nv, red, tid);
    const float den_p = a.mean ? (float)(a.B * a.P) : 1.f, den_s = a.mean ? (float)a.B : 1.f, den_g = a.mean ? (float)(a.B * 9) : 1.f;
    const float den_j = a.mean ? fmaxf(1.f, (float)S * nv * 2.f) : 1.f;
    // joint gradients of this image's rows: zero, then the 17 COCO joints' x / y
    for (int i = tid; i < S * 270; i += 256) {
        const int s = i / 270, e = i - s * 270;
        const int row = s == 0 ? BN + b : b * a.N + (s - 1);
        a.g_joints[(size_t)row * 270 + e] = 0.f;
    }
    __syncthreads();
    const float cs = a.cam[b * 3], ct0 = a.cam[b * 3 + 1], ct1 = a.cam[b * 3 + 2];
    const float coef_j = a.w_j2d / den_j;
    float lj = 0.f, ds = 0.f, dt0 = 0.f, dt1 = 0.f;
    for (int it = tid; it < S * 34; it += 256) {
        const int s = it / 34, r = it - s * 34, j = r >> 1, c = r & 1;
        const int row = s == 0 ? BN + b : b * a.N + (s - 1);
        const size_t o = (size_t)row * 270 + a.coco[j] * 3 + c;
        const float sign = c ? -1.f : 1.f, tc = c ? ct1 : ct0;
        const float q = a.joints[o] * sign;
        const float p = cs * (q + tc);
        const float tn = 2.f * a.j2d_t[(b * 17 + j) * 2 + c] / a.img_wh - 1.f;
        const float d = p - tn;
        const bool v = a.vis[b * 17 + j] != 0;
        const float dp = v ? 2.f * d * coef_j : 0.f;
        lj += v ? d * d : 0.f;
        a.g_joints[o] = dp * cs * sign;
        ds += dp * (q + tc);
        if (c) dt1 += dp * cs; else dt0 += dp * cs;
    }
    lj = block_sum(lj, red, tid);
    ds = block_sum(ds, red, tid);
    dt0 = block_sum(dt0, red, tid);
    dt1 = block_sum(dt1, red, tid);
    if (tid == 0) { a.g_cam[b * 3] = ds; a.g_cam[b * 3 + 1] = dt0; a.g_cam[b * 3 + 2] = dt1; }
    // shape
    float ls = 0.f;
    for (int d = tid; d < a.nb; d += 256) {
        const int o = b * a.nb + d;
        const float lsd = a.log_std[o], inv = __expf(-2.f * lsd), e = a.shape_t[o] - a.mode[o];
        ls += 0.5f * e * e * inv + lsd + 0.91893853320467274178f;
        a.g_mode[o] = a.w_shape / den_s * (-e * inv);
        a.g_log_std[o] = a.w_shape / den_s * (1.f - e * e * inv);
    }
    ls = block_sum(ls, red, tid);
    // pose
    float lpose = 0.f;
    for (int p = tid; p < a.P; p += 256) {
        lpose -= a.lp[b * a.P + p];
        a.g_lp[b * a.P + p] = -a.w_pose / den_p;
    }
    lpose = block_sum(lpose, red, tid);
    // global rotation
    float lg = 0.f;
    if (tid < 9) {
        const float e = a.glob[b * 9 + tid] - a.glob_t[b * 9 + tid];
        lg = e * e;
        a.g_glob[b * 9 + tid] = a.w_glob / den_g * 2.f * e;
    }
    lg = block_sum(lg, red, tid);
    if (tid == 0) {
        a.partial[b * 4] = lpose; a.partial[b * 4 + 1] = ls; a.partial[b * 4 + 2] = lj; a.partial[b * 4 + 3] = lg;
        if (b == 0) a.partial[a.B * 4] = nv;
    }
}

__global__ __launch_bounds__(64) void k_train_loss_final(const TrainLossArgs a) {
    const int l = threadIdx.x;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (int b = 0; b < a.B; ++b)                                    // fixed order, in double: every lane computes the same totals
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] += (double)a.partial[b * 4 + k];
    if (l == 0) {
        const double nv = a.partial[a.B * 4], S = a.N + 1;
        const double den_p = a.mean ? (double)a.B * a.P : 1.0, den_s = a.mean ? (double)a.B : 1.0, den_g = a.mean ? 9.0 * a.B : 1.0;
        const double den_j = a.mean ? fmax(1.0, S * nv * 2.0) : 1.0;
        a.loss[0] = (float)(a.w_pose * s[0] / den_p + a.w_shape * s[1] / den_s + a.w_j2d * s[2] / den_j + a.w_glob * s[3] / den_g);
    }
}

}  // namespace hmf

using namespace hmf;

extern "C" long long hmf_train_loss_workspace_bytes(int B) { return B > 0 ? (4LL * B + 4) * (long long)sizeof(float) : 0; }

extern "C" int hmf_train_loss(const float* lp, int P, const float* shape_mode, const float* shape_log_std, const float* shape_target, int nb, const float* joints,
                              const float* cam, const int* coco17_host, const float* joints2D_target, const unsigned char* joints2D_vis, const float* glob_R,
                              const float* glob_target, int B, int N, float img_wh, float w_pose, float w_shape, float w_joints2D, float w_glob, int reduction_mean,
                              float* loss, float* g_lp, float* g_shape_mode, float* g_shape_log_std, float* g_joints, float* g_cam, float* g_glob_R, void* ws,
                              long long ws_bytes, void* stream) {
    if (!lp || !shape_mode || !shape_log_std || !shape_target || !joints || !cam || !coco17_host || !joints2D_target || !joints2D_vis || !glob_R || !glob_target ||
        !loss || !g_lp || !g_shape_mode || !g_shape_log_std || !g_joints || !g_cam || !g_glob_R || !ws)
        return fail(HMF_ERR_INVALID, "hmf_train_loss: null pointer");
    if (B <= 0 || N < 0 || P <= 0 || nb <= 0 || img_wh <= 0.f) return fail(HMF_ERR_INVALID, "hmf_train_loss: bad sizes");
    if (ws_bytes < hmf_train_loss_workspace_bytes(B)) return fail(HMF_ERR_INVALID, "hmf_train_loss: workspace smaller than hmf_train_loss_workspace_bytes");
    TrainLossArgs a;
    a.lp = lp; a.mode = shape_mode; a.log_std = shape_log_std; a.shape_t = shape_target; a.joints = joints; a.cam = cam; a.j2d_t = joints2D_target;
    a.vis = joints2D_vis; a.glob = glob_R; a.glob_t = glob_target;
    for (int j = 0; j < 17; ++j) {
        if (coco17_host[j] < 0 || coco17_host[j] >= 90) return fail(HMF_ERR_INVALID, "hmf_train_loss: joint index %d out of the 90 output joints", coco17_host[j]);
        for (int i = 0; i < j; ++i)          // the kernel writes g_joints with plain stores, one per (sample, COCO joint, x/y): a repeated joint would overwrite
            if (coco17_host[i] == coco17_host[j]) return fail(HMF_ERR_INVALID, "hmf_train_loss: output joint %d appears twice in the COCO map", coco17_host[j]);
        a.coco[j] = coco17_host[j];
    }
    a.B = B; a.N = N; a.P = P; a.nb = nb; a.mean = reduction_mean; a.img_wh = img_wh;
    a.w_pose = w_pose; a.w_shape = w_shape; a.w_j2d = w_joints2D; a.w_glob = w_glob;
    a.g_lp = g_lp; a.g_mode = g_shape_mode; a.g_log_std = g_shape_log_std; a.g_joints = g_joints; a.g_cam = g_cam; a.g_glob = g_glob_R;
    a.partial = (float*)ws; a.loss = loss;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_train_loss, dim3(B), dim3(256), 0, st, a);
    HMF_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_train_loss_final, dim3(1), dim3(64), 0, st, a);
    HMF_LAUNCH_CHECK();
    return HMF_OK;
}
