/*
 * humaniflow_amd.h -- C ABI of libhumaniflow_amd.so: the MI355X (gfx950) kernels of the
 * HuManiFlow per-image inference hot path.
 *
 * The reference (akashsengupta1997/HuManiFlow) is pure Python and has NO FFI: its boundary for
 * this path is the nn.Module API (models/humaniflow_model.py:33-37,188-200, models/smpl.py:13-41).
 * This library sits UNDER a Python mirror of that API (humaniflow_amd/humaniflow_model.py,
 * humaniflow_amd/smpl.py).  Each entry point names the reference op chain it replaces (file:line,
 * relative to the reference root); INTEGRATION.md shows the ctypes stub a reference maintainer
 * would add.
 *
 * Conventions
 *  - plain C types only; every `const float*` / `float*` / `double*` below is a DEVICE pointer owned
 *    by the caller (e.g. a torch tensor's data_ptr) unless the name ends in `_host`;
 *  - row-major contiguous arrays; shapes are written `rows x cols` in comments;
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); all compute entry points
 *    are asynchronous on it and never allocate or synchronise;
 *  - return value: 0 = success, negative = error (HMF_ERR_*); hmf_last_error() gives a message
 *    for the calling thread.  No C++ exception crosses the ABI;
 *  - thread-safe for distinct streams; handles are immutable after creation.
 */
#ifndef HUMANIFLOW_AMD_H
#define HUMANIFLOW_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HMF_ABI_VERSION 4

#define HMF_OK 0
#define HMF_ERR_INVALID (-1)     /* bad argument / shape                                  */
#define HMF_ERR_UNSUPPORTED (-2) /* configuration outside the default-config hot path      */
#define HMF_ERR_HIP (-3)         /* a HIP runtime call failed (message has hipGetErrorString) */
#define HMF_ERR_WORKSPACE (-4)   /* workspace too small                                    */
#define HMF_ERR_NO_DEVICE (-5)   /* no HIP device visible                                  */

#define HMF_NUM_VERTS 6890
#define HMF_NUM_SMPL_JOINTS 24
#define HMF_NUM_OUT_JOINTS 90 /* 24 + 21 vertex picks + 9 + 19 + 17 regressed (models/smpl.py:30-34) */
#define HMF_NUM_PARTS 23

int hmf_abi_version(void);
const char* hmf_last_error(void);
int hmf_device_count(void);

/* ------------------------------------------------------------------------------------------ */
/* SO(3) closed-form maps  (utils/rigid_transform_utils.py)                                    */
/* ------------------------------------------------------------------------------------------ */
/* rot6d_to_rotmat :86-100.  x6: n x 6 (row-major pairs R11,R12,R21,R22,R31,R32) -> R: n x 9. */
int hmf_rot6d_to_rotmat(const float* x6, int n, float* R, void* stream);
/* Pull-back of that map (training): gx6 (n,6) = dL/dx6 from gR (n,3,3) = dL/dR, through both F.normalize calls and the cross product. */
int hmf_rot6d_to_rotmat_backward(const float* x6, const float* gR, int n, float* gx6, void* stream);
/* so3_exp :182-201 (fp64 Rodrigues incl. the theta<=1e-10 branch).  v: n x 3 -> R: n x 9.    */
int hmf_so3_exp(const double* v, int n, double* R, void* stream);
/* so3_vee(so3_log(R)) :204-279,168-179 incl. the near-pi sign search.  R: n x 9 -> v: n x 3.  */
int hmf_so3_log(const double* R, int n, double* v, void* stream);
/* so3_log_abs_det_jacobian :298-314 (incl. the log(11/12) value at |v|<=1e-10).  v: n x 3.    */
int hmf_so3_log_abs_det_jacobian(const double* v, int n, double* out, void* stream);
/* smplx.lbs.batch_rodrigues (fp32, angle = |v + 1e-8|), call sites models/humaniflow_model.py:299,
 * evaluate/evaluate_humaniflow.py:83.  v: n x 3 -> R: n x 9.                                   */
int hmf_batch_rodrigues(const float* v, int n, float* R, void* stream);
/* The same for M whole SMPL poses (M x 72 axis-angle, global orientation first: data/pw3d_eval_dataset.py `pose`; evaluate/evaluate_humaniflow.py:83-87),
 * written as the two tensors hmf_smpl_lbs takes: global_orient M x 9, body_pose M x 23 x 9. */
int hmf_batch_rodrigues_pose(const float* aa, int M, float* global_orient, float* body_pose, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* SMPL: shape/pose blend-shapes + LBS + 90 joints  (models/smpl.py:27-41 -> smplx lbs)         */
/* ------------------------------------------------------------------------------------------ */
typedef struct hmf_smpl hmf_smpl_t; /* device-resident constants, repacked for the kernels */

typedef struct {
    int num_betas;                     /* <= 16                                                    */
    const float* v_template_host;      /* 6890 x 3                                                 */
    const float* shapedirs_host;       /* 6890 x 3 x num_betas                                     */
    const float* posedirs_host;        /* 207 x 20670  (smplx buffer layout: (P, V*3))             */
    const float* J_regressor_host;     /* 24 x 6890 dense                                          */
    const float* lbs_weights_host;     /* 6890 x 24 dense (zeros are skipped: exact)               */
    const int32_t* parents_host;       /* 24, parents[0] = -1                                      */
    const int32_t* extra_vertex_ids_host; /* 21 vertex picks (smplx VertexJointSelector order)     */
    const float* J_regressor_more_host;   /* 45 x 6890 dense: extra(9) | cocoplus(19) | h36m(17)   */
} hmf_smpl_host_t;

int hmf_smpl_create(hmf_smpl_t** out, const hmf_smpl_host_t* model);
void hmf_smpl_destroy(hmf_smpl_t* h);

size_t hmf_smpl_lbs_workspace_bytes(int M);
/* Posed bodies.  global_orient: Mg x 9, body_pose: M x 23 x 9 (rotation matrices, pose2rot=False
 * layout; use hmf_batch_rodrigues first for axis-angle input); betas: Mb x num_betas with Mb in
 * {1, M} (smplx broadcast rule); Mg in {1, M}.  verts: M x 6890 x 3; joints: M x 90 x 3 (nullable).
 * impl 4 (round 5) = the fp32 MFMA kernel on 32-row groups (64 KB of LDS, two work-groups per CU; bitwise the same result; measured 7 % slower alone and
 * 0.2-0.6 % slower in the three-stream step: opt-in, SMPL.lbs_rows = 32).
 * impl: 0 = fp32 MFMA kernel (product path), 1 = scalar cross-check kernel (tests only), 2 / 3 = BASELINE configs[2] variants:
 * blend-shape GEMM operands in bf16 on v_mfma_f32_32x32x16_bf16 with fp32 accumulate, fp32 template add and fp32 skinning (profiler
 * tag 12).  2: single bf16 operands (vertices within ~1.5e-3 m of the fp32 path for SMPL-sized bases).  3: both operands as bf16 hi + lo
 * pairs, hi.hi + hi.lo + lo.hi per product (vertices within 1e-4 m: the mode SMPL(blend_dtype='bf16') uses).                        */
int hmf_smpl_lbs(const hmf_smpl_t* h, const float* global_orient, int Mg, const float* body_pose,
                 const float* betas, int Mb, int M, float* verts, float* joints, void* workspace,
                 size_t workspace_bytes, int impl, void* stream);
/* The two SMPL calls of one HumaniflowModel.forward(num_samples = N) result as ONE launch (evaluate/evaluate_humaniflow.py:124-127,168-172;
 * predict/predict_humaniflow.py:128-131,160-163): M = B*N + B rows = the B*N sampled bodies (row b*N + n), then the B point-estimate bodies,
 * exactly the row order hmf_flow_forward writes with HMF_FLOW_PACKED_ROWS.  glob_R B x 9 is indexed per image inside the kernel (no expanded
 * copy: the reference's `glob_rotmat[:, None].expand(-1, N, ...)`), body_pose M x 23 x 9, betas M x nb.  full_pose (nullable) M x 24 x 9 =
 * smplx's `full_pose = cat([global_orient, body_pose], dim=1)`.                                                                          */
int hmf_smpl_lbs_packed(const hmf_smpl_t* h, const float* glob_R, int B, int N, const float* body_pose, const float* betas, float* verts,
                        float* joints, float* full_pose, void* workspace, size_t workspace_bytes, int impl, void* stream);
/* T-pose bodies (evaluate/evaluate_humaniflow.py:131-133,186-192; predict_humaniflow.py:147):
 * v_template + shapedirs . beta.  joints nullable.                                                */
int hmf_smpl_tpose(const hmf_smpl_t* h, const float* betas, int M, float* verts, float* joints,
                   void* stream);
/* The T-pose metrics of the evaluation loop without the T-pose bodies (evaluate/evaluate_humaniflow.py:131-133,186-192 build
 * `v_template + shapedirs . beta` for predictions and targets only to feed metrics/eval_metrics_tracker.py:200-222 -- PVE-T and, through
 * utils/eval_utils.py:107-125, PVE-T-SC): pair p = (betas_pred[p], betas_target[p / group]), both npairs (resp. npairs / group) x nb;
 * out npairs x 3 = [mean |P - T|, mean |sc(P) - T|, 0] in hmf_pointset_errors' layout.  The moments of the scale + translation correction
 * are closed forms of beta (T is affine in beta), the errors take one pass over the model's 6890 x 3 x (nb + 1) rows: no vertex tensor is
 * written or read.  ws: hmf_smpl_tpose_errors_workspace_bytes(npairs).  Deterministic.                                                */
long long hmf_smpl_tpose_errors_workspace_bytes(int npairs);
int hmf_smpl_tpose_errors(const hmf_smpl_t* h, const float* betas_pred, const float* betas_target, int npairs, int group, float* out, void* ws,
                          long long ws_bytes, void* stream);
/* The same for the packed rows of one evaluation step in ONE launch: betas_packed (B N + B) x nb = sample row b N + n, then the
 * point-estimate row B N + b (the `betas` input of hmf_smpl_lbs_packed); both kinds of row of image b are scored against betas_target[b].
 * out (B N + B) x 3 in the same row order.  ws: hmf_smpl_tpose_errors_workspace_bytes(B N + B).                                        */
int hmf_smpl_tpose_errors_packed(const hmf_smpl_t* h, const float* betas_packed, const float* betas_target, int B, int N, float* out, void* ws,
                                 long long ws_bytes, void* stream);
/* Backward of the 90 joints of hmf_smpl_lbs w.r.t. its inputs (the 2D reprojection loss of the training step,
 * losses/humaniflow_loss.py:44-60): g_joints M x 90 x 3 -> g_global_orient M x 9, g_body_pose M x 23 x 9, g_betas M x nb.
 * All three inputs must be per-row (no broadcast).  Vertex gradients are not built (LOSS.APPLY_POINT_EST_LOSS = False).          */
int hmf_smpl_joints_backward(const hmf_smpl_t* h, const float* global_orient, const float* body_pose, const float* betas, int M,
                             const float* g_joints, float* g_global_orient, float* g_body_pose, float* g_betas, void* stream);
/* The 90 joints of hmf_smpl_lbs WITHOUT its vertices (round 5): the training loss reads joints only (train/train_humaniflow.py:283-318), and only the
 * <= 320 vertices the joint tables touch are posed.  global_orient M x 9, body_pose M x 23 x 9, betas M x nb, all per row (the layout of
 * hmf_smpl_joints_backward) -> joints M x 90 x 3; equal to hmf_smpl_lbs's joints up to fp32 summation order (the blend sum runs in k order on one
 * thread here, through the MFMA accumulators there).                                                                                              */
int hmf_smpl_joints(const hmf_smpl_t* h, const float* global_orient, const float* body_pose, const float* betas, int M, float* joints, void* stream);

/* The training loss of one step and its gradients in two launches (train/train_humaniflow.py:283-343 through losses/humaniflow_loss.py:25-79 with
 * LOSS.J2D_LOSS_ON = 'point_est+samples' and APPLY_POINT_EST_LOSS off): pose NLL from the log-densities lp (B,P), shape NLL of Normal(shape_mode,
 * exp(shape_log_std)) at shape_target (B,nb), the visible-2D-joint MSE of the orthographic projection s ((x, -y) + t) of the 17 COCO joints
 * (coco17_host: their indices among the 90 output joints, HOST array) of the packed joints (B N + B, 90, 3) [rows b N + n = sample n of image b, rows
 * B N + b = the point estimate] against joints2D_target (B,17,2) pixels normalised by img_wh, and the MSE of glob_R (B,9); weights and the
 * reduction ('mean' = 1: the reference's denominators, 'sum' = 0) as LOSS.*.  Writes loss (1) and d loss / d {lp, shape_mode, shape_log_std, joints, cam,
 * glob_R}.  ws: hmf_train_loss_workspace_bytes(B).  Deterministic.  The 17 COCO indices must be distinct (HMF_ERR_INVALID otherwise: the joint gradients
 * are written, not accumulated).  Deliberate difference: an image with NO visible joint contributes 0 to the 2D term (denominator max(1, #visible)),
 * where the reference's MSELoss over an empty selection is NaN (losses/humaniflow_loss.py:48-60) and the step is then rolled back. */
long long hmf_train_loss_workspace_bytes(int B);
int hmf_train_loss(const float* lp, int P, const float* shape_mode, const float* shape_log_std, const float* shape_target, int nb, const float* joints,
                   const float* cam, const int* coco17_host, const float* joints2D_target, const unsigned char* joints2D_vis, const float* glob_R,
                   const float* glob_target, int B, int N, float img_wh, float w_pose, float w_shape, float w_joints2D, float w_glob, int reduction_mean,
                   float* loss, float* g_lp, float* g_shape_mode, float* g_shape_log_std, float* g_joints, float* g_cam, float* g_glob_R, void* ws,
                   long long ws_bytes, void* stream);

/* utils/sampling_utils.py:22-33 over N samples of one image.  verts: N x V x 3 ->
 * avg_l2: V, dir_std: V x 3.                                                                      */
int hmf_vertex_variance(const float* verts, int N, int V, float* avg_l2, float* dir_std, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Ancestor-conditioned SO(3) flow  (models/humaniflow_model.py:116-186,260-320, norm_flows/)  */
/* ------------------------------------------------------------------------------------------ */
typedef struct hmf_flow hmf_flow_t; /* device-resident weights, repacked into MFMA operand order */

typedef struct {
    int feat_dim;                 /* 2048 (ResNet-50) or 512 (ResNet-18)                          */
    int num_betas;                /* 10                                                          */
    int isgc_dim;                 /* 256  INPUT_SHAPE_GLOB_CAM_FEATS_DIM                          */
    int ctx_dim;                  /* 64   NORM_FLOW.CONTEXT_DIM                                   */
    int num_transforms;           /* 2    NORM_FLOW.NUM_TRANSFORMS                                */
    int hidden[3];                /* 64,32,32 NORM_FLOW.TRANSFORM_NN_HIDDEN_DIMS                  */
    int count_bins;               /* 8    NORM_FLOW.NUM_SPLINE_SEGMENTS                           */
    float radius;                 /* 1.5*pi NORM_FLOW.COMPACT_SUPPORT_RADIUS (spline bound too)    */
    float base_std;               /* 0.6  NORM_FLOW.BASE_DIST_STD                                 */
    const int32_t* parents_host;  /* 24 SMPL parents                                             */
    const float* w_isgc_host;     /* isgc_dim x (feat_dim + num_betas + 9 + 3)                    */
    const float* b_isgc_host;     /* isgc_dim                                                    */
    const float* const* w_ctx_host; /* [23]: ctx_dim x (isgc_dim + 9*num_ancestors(j))            */
    const float* const* b_ctx_host; /* [23]: ctx_dim                                             */
    const float* const* w_nn_host;  /* [23*num_transforms*4]: (64x65),(32x64),(32x32),(62x32)     */
    const float* const* b_nn_host;  /* same indexing                                             */
    /* ---- non-default flow variants (models/norm_flows/pyro_conditional_norm_flow.py:60-106).  All zero / NULL = the default
     * configuration above (spline_coupling + permute, hidden [64,32,32], 2 transforms), which runs on the MFMA kernels; anything else
     * runs on the generic lane-per-channel kernels (csrc/flow_generic.hip).  With variants, w_nn_host / b_nn_host hold
     * 23 * num_transforms * (num_hidden + 1) layers: (h0 x (ctx_dim + 1)), (h1 x h0), ..., (out x h_last), out = 62 (spline) or 2 (additive). */
    int transform_type;             /* 0 spline_coupling (transforms/conditional_spline_coupling_transform.py), 1 additive_coupling
                                       (transforms/additive_coupling_transform.py:12-85: y2 = x2 + nn([context, x1]), log-det 0)            */
    int permute_type;               /* 0 permute, 1 linear_plu, 2 conditional_linear_plu (transforms/linearPLU_transform.py:12-150)        */
    int num_hidden;                 /* 0 = use hidden[3] above; else 1..4 entries of hidden_dims                                          */
    int hidden_dims[4];             /* NORM_FLOW.TRANSFORM_NN_HIDDEN_DIMS, each <= 64                                                    */
    int plu_num_hidden;             /* conditional_linear_plu: entries of NORM_FLOW.PERMUTE_NN_HIDDEN_DIMS (1..4, each <= 64)             */
    int plu_hidden_dims[4];
    const float* const* w_plu_host; /* conditional_linear_plu: [23 * num_transforms * (plu_num_hidden + 1)] DenseNN layers ctx -> ... -> 9  */
    const float* const* b_plu_host;
    const float* const* lu_host;    /* linear_plu: [23 * num_transforms] 3x3 row-major `LU` parameters (L strictly below, U on/above)     */
    int force_generic;              /* tests: run the default configuration through the generic kernels too                              */
} hmf_flow_host_t;

int hmf_flow_create(hmf_flow_t** out, const hmf_flow_host_t* weights);
/* Training: the weights change every optimiser step.  If every tensor of `weights` is a slice of ONE contiguous host buffer
 * (flat_base, flat_count floats, e.g. the concatenation of the model's flow parameters), the handle also records, per packed slot, which
 * flat position it copies; hmf_flow_update_from_device then re-packs from the SAME concatenation held in device memory with one gather
 * kernel (no host round trip, no reallocation), asynchronously on `stream`.                                                              */
int hmf_flow_create_updatable(hmf_flow_t** out, const hmf_flow_host_t* weights, const float* flat_base, size_t flat_count);
int hmf_flow_update_from_device(hmf_flow_t* h, const float* flat_dev, size_t flat_count, void* stream);
void hmf_flow_destroy(hmf_flow_t* h);

/* flags of hmf_flow_forward */
#define HMF_FLOW_POINT_EST 1    /* also push z = 0 through every flow (compute_point_est, :294-300)                      */
#define HMF_FLOW_PACKED_ROWS 2  /* shape_samples has B*N + B rows: the point-estimate rows' shape (= shape_mode) is appended, so that
                                 * (rotmats_samples | rotmats_point) and shape_samples feed ONE hmf_smpl_lbs call of M = B*N + B rows */
#define HMF_FLOW_SHAPE_MODE 4   /* use_shape_mode_for_samples (:255-256): no shape noise, whatever eps_shape / rng say                  */
#define HMF_FLOW_CHAIN32 8      /* force the kinematic loop onto the 32-row-tile kernel (k_flow_chain; the default above 1,024 rows: throughput form) */
#define HMF_FLOW_CHAIN16 16     /* force it onto the 16-row-tile kernel (k_flow_chain16; the default up to 1,024 rows: latency form)             */
size_t hmf_flow_forward_workspace_bytes(const hmf_flow_t* h, int B, int N, int flags);

/* Base noise drawn INSIDE the kernels from the device generator's Philox4x32-10 stream, in the reference's draw order (SURVEY 8b "RNG
 * contract"; models/humaniflow_model.py:258 then :309): one normal_() call of shape (N, B, nb), then one of shape (B, N, 3) per body part --
 * bit for bit what torch.randn would have written on the same generator state (csrc/rng_philox.h restates ATen's launch geometry).
 *   seed, offset    the generator's state at the start of the pass (torch.cuda.default_generators[i]: initial_seed(), get_offset()); the caller
 *                   advances its generator by hmf_flow_rng_increment(...) afterwards;
 *   state_dev       NULL, or a DEVICE pointer to {uint64 seed, uint64 offset}: then seed / offset above are ignored, the pass takes them from
 *                   there and advances the offset itself with one atomic add (a pass captured into a HIP graph: every replay draws fresh noise);
 *   mirror_dev      NULL, or a DEVICE pointer to {uint64 seed, uint64 offset} (normally the same 16 bytes a later captured pass names as
 *                   state_dev): a pass that runs on the immediates above (state_dev == NULL) stores {seed, offset + increment} there, i.e. the
 *                   generator's state after itself, so that captured passes continue the stream where the eager passes stopped (ABI 4);
 *   grid_cap_blocks #CU * (max threads per CU / 256): the grid cap of ATen's launch on this device (2048 on MI355X).                       */
typedef struct {
    uint64_t seed, offset;
    void* state_dev;
    void* mirror_dev;
    int grid_cap_blocks;
} hmf_rng_t;
/* generator offset one sampling pass consumes (draw_shape = 0 with use_shape_mode_for_samples or injected shape noise) */
unsigned long long hmf_flow_rng_increment(int B, int N, int num_betas, int draw_shape, unsigned int grid_cap_blocks);

/* The same generator calls materialised as tensors by one launch (training keeps the noise for its reverse sweep, train/train_humaniflow.py:248-256;
 * tests compare with torch.randn): eps_shape N x B x nb (NULL = not drawn: use_shape_mode_for_samples), eps_pose 23 x B x N x 3.
 * workspace: >= 256 bytes of device memory.                                                                                               */
int hmf_flow_draw_noise(const hmf_rng_t* rng, int B, int N, int num_betas, float* eps_shape, float* eps_pose, void* workspace,
                        size_t workspace_bytes, void* stream);

/* Kinematic-chain pass: N pose samples per image and/or the z=0 point estimate, all 23 parts in
 * dependency order inside one launch sequence (replaces the Python loop :286-311).
 *   feats B x feat_dim; glob_R B x 9; cam B x 3; shape_mode/shape_log_std B x nb;
 *   base noise, either INJECTED (exact-parity tests): eps_shape N x B x nb, eps_pose 23 x B x N x 3 standard normals in the reference's
 *   draw order (:258, :309), or DRAWN in the kernels: eps_pose = NULL and rng != NULL (then eps_shape = NULL draws the shape noise too,
 *   unless HMF_FLOW_SHAPE_MODE).  Without rng, eps_shape = NULL means use_shape_mode_for_samples (:255-256).
 * outputs (nullable when not requested):
 *   shape_samples B x N x nb (+ B x nb with HMF_FLOW_PACKED_ROWS); rotmats_samples B x N x 23 x 9 (fp64 exp rounded to fp32, :311);
 *   rotmats_point B x 23 x 9 (batch_rodrigues, :299); so3_point B x 23 x 3 (:300).               */
int hmf_flow_forward(const hmf_flow_t* h, const float* feats, const float* glob_R, const float* cam,
                     const float* shape_mode, const float* shape_log_std, const float* eps_shape,
                     const float* eps_pose, const hmf_rng_t* rng, int B, int N, int flags, float* shape_samples,
                     float* rotmats_samples, float* rotmats_point, float* so3_point, void* workspace,
                     size_t workspace_bytes, void* stream);

/* compute_input_shape_glob_cam_feats :116-150 (2-D variant): ELU(W [f, shape, vec(R), cam] + b).
 * feats B x feat_dim, shape B x nb, glob_R B x 9, cam B x 3 -> isgc B x isgc_dim.               */
int hmf_flow_isgc(const hmf_flow_t* h, const float* feats, const float* shape, const float* glob_R,
                  const float* cam, int B, float* isgc, void* stream);
/* The same layer for the training step, where autograd differentiates it (train/train_humaniflow.py:351 through :116-150), for BOTH of its uses in one
 * call: pre (2 x B x 256) = W [f, 0, vec(R), cam] + b without shape and ELU -- [0] with glob_R (the predicted rotation: sample path), [1] with glob_R_ll (the
 * target rotation: teacher-forced path, :279-283); H ((B N + B) x 256) = ELU(pre[use][image] + W_beta shape[row]) for the B N per-sample shapes
 * (shape_rows, row b N + n) followed by the B teacher-forced shapes (shape_ll).  Three launches.  N >= 1. */
int hmf_flow_isgc_train(const hmf_flow_t* h, const float* feats, const float* glob_R, const float* glob_R_ll, const float* cam, const float* shape_rows,
                        const float* shape_ll, int B, int N, float* pre, float* H, void* stream);
/* Its backward below the dense products: from d loss / d H (g_H_rows (B N) x 256 and g_H_ll B x 256, either may be NULL = zero) to
 *   delta ((B N + 2 B) x 256): g_H * ELU'(H) of the sample rows, of the teacher-forced rows, then g_pre_s[b] = the sum of image b's sample rows (fixed order);
 *   g_pre_f (B x 256) = g_pre_s + delta_ll (what reaches the shared image-feature product and the bias), g_pre_f_t / g_pre_s_t (256 x B): transposes;
 *   g_shape ((B N + B) x nb) = delta W_beta.
 * The caller finishes with hmf_rows_atb_batched: d W[:, F:F+nb] = delta[:B N + B]^T shapes, d W[:, :F] = g_pre_f^T feats (+ column sums = d bias),
 * d W[:, F+nb:] = delta[B N:]^T [vec(R_ll), cam | vec(R), cam], g_feats = g_pre_f W[:, :F], g_vec(R) = g_pre_s W[:, F+nb:F+nb+9], g_cam = g_pre_f W[:, F+nb+9:]. */
int hmf_flow_isgc_train_backward(const hmf_flow_t* h, const float* H, const float* g_H_rows, const float* g_H_ll, int B, int N, float* delta, float* g_pre_f,
                                 float* g_pre_f_t, float* g_pre_s_t, float* g_shape, void* stream);
/* compute_flow_context :152-186 for all 23 parts with GIVEN (teacher-forced) rotations.
 * isgc B x isgc_dim, pose_R B x 23 x 9 fp32 -> ctx B x 23 x ctx_dim.                            */
int hmf_flow_context(const hmf_flow_t* h, const float* isgc, const float* pose_R, int B, float* ctx,
                     void* stream);
/* LocalDiffeoTransformedDistribution.log_prob (local_diffeo_transformed_distribution.py:84-142):
 * log p(R | ctx) on SO(3) with the logsumexp over the 2*pi pre-images.
 * ctx B x P x ctx_dim; part_ids P ints on the HOST, a contiguous ascending range of part indices
 * (NULL = 0..22, requires P == 23);
 * R B x P x 9 fp64 -> logp B x P fp32.                                                            */
int hmf_flow_log_prob_SO3(const hmf_flow_t* h, const float* ctx, const int32_t* part_ids_host, int P,
                          const double* R, int B, float* logp, void* stream);
/* Density on the algebra (conditioned_pose_so3flow_dists_for_loglik, :318): v B x P x 3 fp32.    */
int hmf_flow_log_prob_so3(const hmf_flow_t* h, const float* ctx, const int32_t* part_ids_host, int P,
                          const float* v, int B, float* logp, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Training step (BASELINE configs[4]): vector-Jacobian products of the flow       */
/* train/train_humaniflow.py:248-330 differentiates forward(compute_for_loglik=True, num_samples=NUM_J2D_SAMPLES)                 */
/* ------------------------------------------------------------------------------------------ */
/* Per-(part, row) records written by the backward kernels.  The weight gradients are dense reductions over the row axis,
 *   dW_l[part, coupling] = D_l^T . A_{l-1},   db_l = sum_rows D_l,   with A_0 = [ACTX | XIN]
 * which the caller does with library GEMMs.  Every array is part-major: [P][2 couplings][rows][width] (ACTX, DCTX: [P][rows][64]).  */
typedef struct {
    float* D1;   /* [P][2][rows][64]  delta of hyper-network layer 1 (65 -> 64)                         */
    float* D2;   /* [P][2][rows][32]  layer 2 (64 -> 32)                                               */
    float* D3;   /* [P][2][rows][32]  layer 3 (32 -> 32)                                               */
    float* D4;   /* [P][2][rows][64]  layer 4 (32 -> 62; columns 62, 63 are zero)                      */
    float* A1;   /* [P][2][rows][64]  ReLU output of layer 1                                           */
    float* A2;   /* [P][2][rows][32]                                                                   */
    float* A3;   /* [P][2][rows][32]                                                                   */
    float* XIN;  /* [P][2][rows]      the scalar x1 fed to the hyper-network next to the context        */
    float* ACTX; /* [P][rows][64]     context vector                                                   */
    float* DCTX; /* [P][rows][64]     delta of the context layer (hmf_flow_sample_backward only; else NULL) */
    int rows;
} hmf_flow_grad_records_t;

/* Round 6: reverse sweep of hmf_flow_log_prob_so3 (the density on the ALGEBRA, local_diffeo_transformed_distribution.py:84-142 below the exp map) w.r.t. its
 * ARGUMENT and the context: g_logp B x P -> g_ctx B x P x 64, g_v B x P x 3.  No weight-gradient records (the flow-as-prior call pattern of
 * optimise/optimise_humaniflow.py:96-109 evaluates a frozen flow); the pull-back from v to the rotation R (so3_log, the 2 pi k pre-images, the exp map's
 * log-determinant) is fp64 host-side composition (humaniflow_amd/autograd.py).  Default flow configuration only (HMF_ERR_UNSUPPORTED otherwise). */
int hmf_flow_log_prob_so3_vjp(const hmf_flow_t* h, const float* ctx, const int32_t* part_ids_host, int P, const float* v, int B, const float* g_logp,
                              float* g_ctx, float* g_v, void* stream);

/* Reverse sweep of hmf_flow_log_prob_SO3 (losses/humaniflow_loss.py:26-31 pose NLL): g_logp B x P upstream gradients ->
 * g_ctx B x P x 64 and the records with rows = 3 B (row = pre-image term * B + b; masked pre-images contribute zeros).
 * The targets R are constants (teacher forcing).                                                                      */
int hmf_flow_log_prob_SO3_backward(const hmf_flow_t* h, const float* ctx, const int32_t* part_ids_host, int P, const double* R, int B,
                                   const float* g_logp, float* g_ctx, const hmf_flow_grad_records_t* rec, void* stream);
/* Reverse sweep of the reparameterised samples of hmf_flow_forward (models/humaniflow_model.py:302-311): rounds of the kinematic
 * schedule in reverse order, one launch each.  isgc_rows (B N) x 256 = ELU(fc_input_shape_glob_cam_feats(...)) per sample row,
 * rotmats_samples (B N) x 23 x 9 the forward result, eps_pose as in the forward; g_rotmats (B N) x 23 x 9 holds d L / d samples on entry
 * and additionally every descendant's contribution on exit (in/out).  Records: rows = B N, P = 23, DCTX filled.
 * Point-estimate rows carry no gradient in the reference (torch.set_grad_enabled(grad_for_pose_point_est), :294).          */
/* Dense weight-gradient reductions of the training step (what loss.backward() does for every Linear of the flow, train/train_humaniflow.py:351):
 * C[b] = A[b]^T . B[b] over the row axis, b = 0..batch-1.  A[b] (rows, m) fp32 with row pitch lda at A + b * batch_stride_a, B[b] (rows, n) with
 * pitch ldb at B + b * batch_stride_b (a stride of 0 shares the operand), C (batch, m, n) contiguous; a_colsum (batch, m) or NULL receives the
 * column sums of A (bias gradients).  Device pointers. */
int hmf_rows_atb_batched(const float* A, const float* B, float* C, float* a_colsum, int batch, int rows, int m, int n, int lda, int ldb,
                         long long batch_stride_a, long long batch_stride_b, void* stream);

int hmf_flow_sample_backward(const hmf_flow_t* h, const float* isgc_rows, const float* rotmats_samples, const float* eps_pose, int B, int N,
                             float* g_rotmats, const hmf_flow_grad_records_t* rec, void* stream);

/* The same two reverse sweeps for the flow VARIANTS (PERMUTE_TYPE linear_plu / conditional_linear_plu, TRANSFORM_TYPE
 * additive_coupling, other NUM_TRANSFORMS / hidden sizes -- every handle that runs the generic kernels, models/norm_flows/
 * pyro_conditional_norm_flow.py:60-106; the reference trains them through the same loss.backward()).  Records per (part, transform, row);
 * T = NUM_TRANSFORMS, L = len(TRANSFORM_NN_HIDDEN_DIMS) + 1, LP = len(PERMUTE_NN_HIDDEN_DIMS) + 1.  Weight gradients, as above:
 *   dW_l[part, t] = D[part, t, l]^T . (l == 0 ? [ACTX[part] | XIN[part, t]] : A[part, t, l]),  db_l = column sums of D[part, t, l]
 *   conditional_linear_plu: the same with DP / AP (layer 0 reads ACTX[part]);  linear_plu: d LU[part, t] = sum over rows of GLU[part, t].
 * The CALLER ZERO-FILLS every array (rows of masked pre-images and channels beyond a layer's width are not written). */
typedef struct {
    float* D;    /* [P][T][L][rows][64]   deltas of the transform hyper-network's layers                                      */
    float* A;    /* [P][T][L][rows][64]   inputs of layers 1 .. L-1 (slot 0 unused)                                          */
    float* XIN;  /* [P][T][rows]          the scalar x1 fed to the hyper-network next to the context                         */
    float* DP;   /* [P][T][LP][rows][64]  conditional_linear_plu: deltas of the PLU network (linearPLU_transform.py:123-150); else NULL */
    float* AP;   /* [P][T][LP][rows][64]  inputs of its layers 1 .. LP-1; else NULL                                          */
    float* GLU;  /* [P][T][rows][9]       linear_plu: d L / d LU (row-major 3 x 3) of the row; else NULL                     */
    float* ACTX; /* [P][rows][64]         context vector                                                                      */
    float* DCTX; /* [P][rows][64]         delta of the context layer (hmf_flow_generic_sample_backward only; else NULL)       */
    int rows, L, LP;
} hmf_flow_generic_grad_records_t;
/* Arguments as hmf_flow_log_prob_SO3_backward / hmf_flow_sample_backward; HMF_ERR_UNSUPPORTED on a handle of the default configuration
 * (and the two entry points above return it on a variant handle). */
int hmf_flow_generic_log_prob_SO3_backward(const hmf_flow_t* h, const float* ctx, const int32_t* part_ids_host, int P, const double* R, int B,
                                           const float* g_logp, float* g_ctx, const hmf_flow_generic_grad_records_t* rec, void* stream);
int hmf_flow_generic_sample_backward(const hmf_flow_t* h, const float* isgc_rows, const float* rotmats_samples, const float* eps_pose, int B, int N,
                                     float* g_rotmats, const hmf_flow_generic_grad_records_t* rec, void* stream);


/* ------------------------------------------------------------------------------------------ */
/* Proxy-representation front end: models/canny_edge_detector.py:104-166 +          */
/* utils/label_conversions.py:106-125, as composed at predict/predict_humaniflow.py:101-109         */
/* ------------------------------------------------------------------------------------------ */
/* img B x 3 x S x S; joints2D B x J x 2 (column, row) and visib B x J (both nullable) ->
 * proxy B x (1 + J) x S x S: channel 0 = thresholded (thin) edges, channels 1.. = Gaussian heat-maps (* visib).
 * Optional extra outputs (B x 1 x S x S, nullable): grad_magnitude, grad_orientation (degrees, binned to 45), thin_edges.
 * Only edge_gaussian_size == 5 is built (the default EDGE_GAUSSIAN_SIZE).                                                  */
int hmf_proxy_rep(const float* img, const float* joints2D, const float* visib, int B, int S, int num_joints,
                  int edge_nms, float edge_threshold, float edge_gaussian_std, int edge_gaussian_size, float heatmap_std,
                  float* proxy, float* grad_magnitude, float* grad_orientation, float* thin_edges, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Encoder convolutions (models/resnet.py:102-122,202-217)                                          */
/* ------------------------------------------------------------------------------------------ */
/* Fused convolution of the encoder (models/resnet.py:24-31 conv3x3/conv1x1 + :72-76,111-127 BatchNorm/ReLU/residual of
 * BasicBlock/Bottleneck.forward, with eval-mode BN folded into w and bias): NHWC fp32 implicit GEMM on the fp32 MFMA,
 *   y = relu?(conv(x, w, stride, padding = ks/2) + bias (+ residual)).
 * x (B,H,W,Cin), w (Cout,ks,ks,Cin) [= a channels_last (Cout,Cin,ks,ks) tensor], bias (Cout) or NULL, residual/y (B,OH,OW,Cout);
 * ks in {1,3}, stride in {1,2}, Cin % 32 == 0, Cout % 64 == 0.  tile: 0 = pick by work-group count, else BM*1000+BN in
 * {64128, 64064}.  Profiler tag 8. */
int hmf_conv2d_nhwc(const float* x, const float* w, const float* bias, const float* residual, float* y, int B, int H, int W,
                    int Cin, int Cout, int ks, int stride, int relu, int tile, void* stream);

/* Last 1x1 convolution of a bottleneck + its downsample branch (models/resnet.py:111-127 with self.downsample) as ONE GEMM over the
 * concatenated K: y = relu?(x . W[:, :Cin] + x2[:, ::stride2, ::stride2, :] . W[:, Cin:] + bias).  x (B,OH,OW,Cin), x2 (B,H2,W2,Cin2),
 * w (Cout, Cin+Cin2) row-major, OH = (H2-1)/stride2+1; Cin, Cin2 % 32 == 0, Cout % 64 == 0.  Profiler tag 8. */
int hmf_conv1x1_dual_nhwc(const float* x, const float* x2, const float* w, const float* bias, float* y, int B, int OH, int OW, int Cin,
                          int H2, int W2, int Cin2, int stride2, int Cout, int relu, void* stream);

/* 1x1 convolution (models/resnet.py:102-122 conv1 / conv3 of a Bottleneck with the eval-mode BatchNorm folded in, optionally + the block's
 * downsample branch :111-127 as a second K range, + identity, + ReLU) as an LDS-DMA pipelined fp32-MFMA GEMM (csrc/conv_gemm.hip):
 * y = relu?(x . W[:, :K1]^T (+ x2[:, ::stride2, ::stride2, :] . W[:, K1:]^T) + bias (+ residual)).  x (B,OH,OW,K1) NHWC, x2 (B,H2,W2,K2) or
 * NULL, w (N, K1+K2) row-major, y / residual (B,OH,OW,N); K1, K2 % 32 == 0, N % 64 == 0, every pointer 16-byte aligned.
 * tile: 0 = from the shape, else [ring stages * 1000000 +] BM*1000+BN in {64064, 3064064, 3064128, 2064064, 2064128}.  Same results as hmf_conv2d_nhwc(ks=1) /
 * hmf_conv1x1_dual_nhwc up to the fp32 accumulation order.  Profiler tag 8. */
int hmf_conv1x1_gemm(const float* x, const float* x2, const float* w, const float* bias, const float* residual, float* y, int B, int OH, int OW,
                     int K1, int H2, int W2, int K2, int stride2, int N, int relu, int tile, void* stream);
/* The encoder's last 1x1 convolution with the global average pool of models/resnet.py:214-215 (`avgpool` + flatten) in its epilogue:
 * pooled (B, N) = mean over the OH*OW = 64 pixels of relu?(x . W^T + bias (+ residual)); a 64-row GEMM tile is one image's 8 x 8 map, so the
 * pooled features are column means of finished tiles: no cross-work-group reduction, fixed summation order.  y (B,8,8,N) also receives the map
 * when non-NULL; the inference plan passes NULL (the map is neither written nor re-read).  Other arguments as hmf_conv1x1_gemm. */
int hmf_conv1x1_gemm_avgpool(const float* x, const float* w, const float* bias, const float* residual, float* y, float* pooled, int B, int OH,
                             int OW, int K1, int N, int relu, int tile, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Heads between the encoder and the flow (models/humaniflow_model.py:232-252, models/resnet.py:214-215)                              */
/* ------------------------------------------------------------------------------------------ */
/* F.adaptive_avg_pool2d(x, 1) + flatten (models/resnet.py:214-215) for maps the fused epilogue above does not cover: x (B, HW, C) NHWC -> (B, C). */
int hmf_avgpool_nhwc(const float* x, int B, int HW, int C, float* out, void* stream);
size_t hmf_heads_workspace_bytes(int B, int F, int N1);
/* models/humaniflow_model.py:232-253 in two launches and no vendor GEMM: x1 = ELU(fc1 feats) (fp32 MFMA over the live (N1, F) weight, K split in a
 * fixed order); cam = fc_cam x1 + init_cam (:237-238); glob_R = rot6d_to_rotmat(fc_glob x1 + init_glob) (:243-245); (shape_mode | shape_log_std) =
 * fc_shape x1 (:250-252); shape_std = exp(shape_log_std) (the scale of :253's Normal).  feats B x F; w1 N1 x F, b1 N1;
 * w_cam 3 x N1, w_glob 6 x N1, w_shape 2nb x N1 (+ biases); init_cam 3, init_glob 6.  Outputs cam B x 3, glob_R B x 9, shape_mode / shape_log_std /
 * shape_std B x nb each.  F % 32 == 0, N1 % 32 == 0, N1 <= 1024; feats and weight matrices 16-byte aligned.  Profiler tag 15. */
int hmf_heads(const float* feats, int B, int F, const float* w1, const float* b1, int N1, const float* w_cam, const float* b_cam, const float* init_cam,
              const float* w_glob, const float* b_glob, const float* init_glob, const float* w_shape, const float* b_shape, int nb, float* cam,
              float* glob_R, float* shape_mode, float* shape_log_std, float* shape_std, void* workspace, size_t workspace_bytes, void* stream);
/* The same forward for the training step: additionally x1 (B, N1) = ELU(fc1 feats + b1) and glob6 (B, 6) = the 6-D vector before Gram-Schmidt,
 * which hmf_heads_backward needs. */
int hmf_heads_train(const float* feats, int B, int F, const float* w1, const float* b1, int N1, const float* w_cam, const float* b_cam, const float* init_cam,
                    const float* w_glob, const float* b_glob, const float* init_glob, const float* w_shape, const float* b_shape, int nb, float* cam,
                    float* glob_R, float* shape_mode, float* shape_log_std, float* shape_std, float* x1, float* glob6, void* workspace, size_t workspace_bytes,
                    void* stream);
/* Backward of the heads (what autograd does for models/humaniflow_model.py:232-253 in train/train_humaniflow.py:351), the part below fc1's
 * pre-activation in ONE launch: from d loss / d {cam (B,3), glob_R (B,9), shape_mode (B,nb), shape_log_std (B,nb)} (any NULL = zero) to
 * g_pre (B, N1) = d loss / d (fc1 pre-activation), the same transposed (N1, B), and the weight / bias gradients of fc_cam, fc_glob, fc_shape.
 * The caller finishes with two hmf_rows_atb_batched calls: d fc1.weight = g_pre^T feats (+ column sums = d fc1.bias) and g_feats = g_pre W1
 * (rows = N1 over g_pre_t and fc1.weight).  B <= 128.  Deterministic. */
int hmf_heads_backward(const float* x1, const float* glob6, int B, int N1, int nb, const float* w_cam, const float* w_glob, const float* w_shape,
                       const float* g_cam, const float* g_glob_R, const float* g_shape_mode, const float* g_shape_log_std, float* g_pre, float* g_pre_t,
                       float* g_w_cam, float* g_b_cam, float* g_w_glob, float* g_b_glob, float* g_w_shape, float* g_b_shape, void* stream);

/* OPT-IN: the same GEMM on the bf16 matrix pipe with split fp32 operands (csrc/conv_gemm_split.hip; not the default of any plan).  Every fp32
 * value x = p0 + p1 (+ p2) in bf16 pieces (round-to-nearest remainders); terms = 3: x*w ~ p0 q0 + p0 q1 + p1 q0 (product error ~2^-16 relative),
 * terms = 6: + p0 q2 + p2 q0 + p1 q1 (dropped products <= 2^-24: fp32-level).  fp32 accumulation, bias, residual and ReLU.  The filter is split
 * once: hmf_conv1x1_split_pack(w_dev (N, K1+K2) row-major fp32 ON THE DEVICE, packed_dev of hmf_conv1x1_split_packed_bytes(N, K, terms) bytes);
 * activations are split in registers by the kernel (no second copy of a tensor).  Other arguments and constraints as hmf_conv1x1_gemm;
 * tile: 0 = from the shape, else ring stages * 1000 + BN in {3128, 4064}.  Profiler tag 8. */
long long hmf_conv1x1_split_packed_bytes(int N, int K, int terms);
int hmf_conv1x1_split_pack(const float* w_dev, void* packed_dev, int N, int K, int terms, void* stream);
int hmf_conv1x1_gemm_split(const float* x, const float* x2, const void* w_packed, const float* bias, const float* residual, float* y, int B, int OH,
                           int OW, int K1, int H2, int W2, int K2, int stride2, int N, int relu, int terms, int tile, void* stream);

/* The seam between two bottlenecks (models/resnet.py:102-127: block i's conv3 + bn3 (+ downsample branch) + identity + ReLU, then block
 * i+1's conv1 + bn1 + ReLU) as ONE launch: y = relu([x1 | x2(strided)] . W3^T + bias3 (+ residual)) is written (the next identity branch) and
 * its 64-channel chunks feed z = relu(y . W1^T + bias1) through LDS, so the C-channel tensor y is not read back from HBM.
 * x1 (B,OH,OW,K1); x2 (B,H2,W2,K2) or NULL (then K2 is ignored); K1 + K2 in {64, 128}; y/residual (B,OH,OW,C), C % 64 == 0; z (B,OH,OW,N2),
 * N2 in {64, 128, 256}; B*OH*OW % 64 == 0.  Filters packed by hmf_conv_pair_pack_host (w3 (C, K1+K2) and w1 (N2, C) row-major, BN folded) into
 * C*(K1+K2) and N2*C floats of MFMA lane order.  variant: 0 (others are timing experiments of tools/pair_probe.py that compute wrong results).
 * Profiler tag 8. */
int hmf_conv_pair_pack_host(const float* w3, const float* w1, float* w3_packed, float* w1_packed, int Kt, int C, int N2);
int hmf_conv_pair(const float* x1, const float* x2, const float* w3_packed, const float* bias3, const float* residual, float* y,
                  const float* w1_packed, const float* bias1, float* z, int B, int OH, int OW, int K1, int H2, int W2, int K2, int stride2,
                  int C, int N2, int variant, void* stream);

/* Patch-resident direct 3x3 convolution (stride 1 or 2; the encoder uses it where the Winograd kernel does not apply): the input patch of a
 * pixel tile lives in LDS for a whole slab of 64 input channels and the weights stream from L2 in MFMA lane order (no barrier inside a slab).
 * w_packed = hmf_conv_patch_pack_host(w (Cout,3,3,Cin) channels_last, ..., cfg) uploaded to the device (hmf_conv_patch_packed_floats floats).
 * cfg = TW*1000 + PG*100 + TN*10: 16420 = 8x16-pixel tile x 64 channels, 16410 = 8x16 x 32, 8210 = 8x8 x 64 (the only one for stride 2);
 * Cin % 64 == 0, Cout % (channels per tile) == 0.  Same epilogue as hmf_conv2d_nhwc; profiler tag 10. */
long long hmf_conv_patch_packed_floats(int Cin, int Cout, int ks);
int hmf_conv_patch_pack_host(const float* w, float* packed, int Cin, int Cout, int ks, int cfg);
int hmf_conv_patch(const float* x, const float* w_packed, const float* bias, const float* residual, float* y, int B, int H, int W,
                   int Cin, int Cout, int ks, int stride, int relu, int cfg, void* stream);
/* Small launches (one image per call): as hmf_conv_winograd_ws below -- with fewer than 128 work-groups the slab loop (Cin / 64) is split over up to 8
 * work-groups per tile, raw partial sums go to `ws` (hmf_conv_patch_workspace_bytes: 0 = the shape is not split, ws may be NULL) and a second small launch
 * adds them in a fixed order and applies bias / residual / ReLU.  With a split, y, bias and residual must be 16-byte aligned. */
long long hmf_conv_patch_workspace_bytes(int B, int H, int W, int Cin, int Cout, int ks, int stride, int cfg);
int hmf_conv_patch_ws(const float* x, const float* w_packed, const float* bias, const float* residual, float* y, int B, int H, int W,
                      int Cin, int Cout, int ks, int stride, int relu, int cfg, void* ws, long long ws_bytes, void* stream);

/* Winograd F(2x2,3x3) variant for stride-1 3x3 layers (same result as hmf_conv2d_nhwc with ks = 3, 2.25x fewer MFMA k-steps): 8x16-pixel
 * tiles x 32 channels (maps at most 8 pixels wide: 8x8 pixels of two consecutive images), input transform in registers, output transform through LDS, same fused epilogue.  w_packed =
 * hmf_conv_winograd_pack_host(w (Cout,3,3,Cin) channels_last, ...) [the transformed weights G g G^T in MFMA lane order,
 * hmf_conv_winograd_packed_floats = 16*Cout*Cin floats] uploaded to the device.  Cin % 64 == 0, Cout % 32 == 0; x, w_packed, y, residual and
 * bias 16-byte aligned (the output phase uses 16-byte buffer accesses).  Profiler tag 11. */
long long hmf_conv_winograd_packed_floats(int Cin, int Cout);
int hmf_conv_winograd_pack_host(const float* w, float* packed, int Cin, int Cout);
int hmf_conv_winograd(const float* x, const float* w_packed, const float* bias, const float* residual, float* y, int B, int H, int W,
                      int Cin, int Cout, int relu, void* stream);

/* Small batches (predict: one image per call, predict/predict_humaniflow.py:112-115): a launch then has 16-64 work-groups, each walking all Cin / 64
 * slabs, on 256 CUs.  hmf_conv_winograd_ws splits the slab loop over up to 8 work-groups per tile when fewer than 128 work-groups would be launched;
 * the raw partial outputs go to `ws` (hmf_conv_winograd_workspace_bytes: 0 when the shape is not split, then ws may be NULL) and a second small launch
 * adds them in a fixed order and applies bias / residual / ReLU.  Deterministic; rounding differs from the unsplit form by the summation order only. */
long long hmf_conv_winograd_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int hmf_conv_winograd_ws(const float* x, const float* w_packed, const float* bias, const float* residual, float* y, int B, int H, int W,
                         int Cin, int Cout, int relu, void* ws, long long ws_bytes, void* stream);

/* hmf_conv_winograd_pack_host on the device, from a parameter's native (out, in, 3, 3) storage (the training graph re-packs every step):
 * dgrad = 0 packs the layer's own filter (Cout = out, Cin = in); dgrad = 1 packs the filter of the input-gradient convolution (flipped taps,
 * in/out swapped: Cout = the parameter's in-channels, Cin = its out-channels).  packed: hmf_conv_winograd_packed_floats(Cin, Cout) floats. */
int hmf_conv_winograd_pack_device(const float* w_native, float* packed, int Cin, int Cout, int dgrad, void* stream);
/* The same for n filters in one launch (host arrays of n pointers / sizes / flags; the training graph packs every layer's forward and
 * input-gradient filter at the top of a step). */
int hmf_conv_winograd_pack_device_batch(const float* const* w_native, float* const* packed, const int* cin, const int* cout, const int* dgrad, int n,
                                        void* stream);


/* Encoder stem (models/resnet.py:141-144,203-206): conv 7x7 / stride 2 / pad 3 (18 -> 64 channels) + folded BatchNorm + ReLU in one
 * kernel that reads the NCHW proxy representation x (B,18,H,W) and writes the NHWC map y (B,OH,OW,64), OH = (H-1)/2+1.
 * w_packed: hmf_stem_pack_host(w (64,18,7,7) row-major [already multiplied by the BN scale], packed[hmf_stem_packed_floats()])
 * uploaded to the device; bias (64) = folded BN shift or NULL.  Profiler tag 9.
 * hmf_maxpool3x3s2_nhwc: the 3x3 / stride 2 / pad 1 max-pool that follows, NHWC (B,H,W,C) -> (B,(H-1)/2+1,(W-1)/2+1,C), C % 4 == 0. */
long long hmf_stem_packed_floats(void);
int hmf_stem_pack_host(const float* w, float* packed);
int hmf_stem_conv(const float* x, const float* w_packed, const float* bias, float* y, int B, int C, int H, int W, int relu, void* stream);

/* The stem as the inference plan runs it: conv 7x7/2 + folded-BN bias + ReLU + max-pool 3x3/2/1 (models/resnet.py:141-144,203-206) in ONE launch --
 * the pool is applied to each work-group's tile before it leaves the CU, windows shared between tiles are combined with an atomic max.
 * y_pooled (B, PH, PW, 64) NHWC, PH = ((H-1)/2)/2 + 1; zeroed inside.  Same filters as hmf_stem_conv.  Profiler tag 9. */
int hmf_stem_conv_pool(const float* x, const float* w_packed, const float* bias, float* y_pooled, int B, int C, int H, int W, void* stream);
int hmf_maxpool3x3s2_nhwc(const float* x, float* y, int B, int H, int W, int C, void* stream);

/* The same stem (same arguments, same outputs as hmf_stem_conv / hmf_stem_conv_pool) as a polyphase fast convolution (csrc/stem_wino.hip): the 7 taps of
 * a stride-2 filter split into a 4-tap and a 3-tap stride-1 filter on the even / odd samples, each evaluated with Winograd F(2,3) arithmetic (constants
 * 0, +-1, 1/2) -- 100 products per 2 x 2 output tile and input channel instead of 196, chained into 16 class accumulators on the MFMA pipe.  Results
 * differ from the direct form by fp32 rounding only (held to the same 2e-5 x scale bar against a float64 convolution).  The inference plan's stem.
 * w_packed: hmf_stem_wino_pack_host / hmf_stem_wino_pack_device(w (64,18,7,7) row-major, packed[hmf_stem_wino_packed_floats()]); the device form packs the
 * live parameter (training re-packs every step).  Profiler tag 9. */
long long hmf_stem_wino_packed_floats(void);
int hmf_stem_wino_pack_host(const float* w, float* packed);
int hmf_stem_wino_pack_device(const float* w, float* packed, void* stream);
int hmf_stem_wino(const float* x, const float* w_packed, const float* bias, float* y, int B, int C, int H, int W, int relu, void* stream);
/* hmf_stem_wino_pool: the max-pool is applied to each work-group's tile before it leaves the CU; pooled positions on a tile's rim go through per-tile
 * halo records in `ws` (hmf_stem_wino_pool_workspace_bytes(B, H, W) bytes, 16-byte aligned) and a second, small launch combines them -- no atomics and
 * no zero-filled map, every pooled value is written exactly once (hmf_stem_conv_pool uses integer atomic maxima on a pre-zeroed map instead). */
long long hmf_stem_wino_pool_workspace_bytes(int B, int H, int W);
int hmf_stem_wino_pool(const float* x, const float* w_packed, const float* bias, float* y_pooled, int B, int C, int H, int W, void* ws, long long ws_bytes,
                       void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Evaluation epilogue (metrics/eval_metrics_tracker.py:95-280,398-410; utils/eval_utils.py)     */
/* ------------------------------------------------------------------------------------------ */
/* Point-set error metrics of npairs (prediction, target) pairs of npts 3-D points.  P: (npairs, npts, 3); T: (npairs/group, npts, 3),
 * pair p is scored against target p / group (group = N for the N samples of an image, 1 for point estimates).
 * out: (npairs, 3) = mean |P-T| (PVE / MPJPE), mean |sc(P)-T| with scale_and_translation_transform_batch (eval_utils.py:107-125),
 * mean |pa(P)-T| with procrustes_analysis_batch (:62-104; 0 when want_pa = 0).  All on the device, fp64 statistics. */
int hmf_pointset_errors(const float* P, const float* T, int npairs, int group, int npts, int want_pa, float* out, void* stream);
/* Sample diversity (eval_metrics_tracker.py:398-410): S (B, N, npts, 3) -> partial (B, hmf_sample_diversity_chunks(npts)) partial sums over n
 * and 64 points each of |S - mean_n S|; the caller adds the chunks (deterministic) and divides by N * npts.  N <= 128: S is read ONCE (a
 * chunk's samples stay in registers between the mean and the distances). */
int hmf_sample_diversity_chunks(int npts);
int hmf_sample_diversity(const float* S, int B, int N, int npts, float* partial, void* stream);
/* The joint-side glue of one evaluation step (evaluate/evaluate_humaniflow.py:138-143,176-183,197-206) in one launch.  joints: the packed
 * (B N + B, 90, 3) output of hmf_smpl_lbs_packed (sample row b N + n, then point-estimate row B N + b); cam (B, 3) weak-perspective
 * [s, tx, ty].  Writes, per row: the 14 evaluation joints (idx14_host: ALL_JOINTS_TO_H36M_MAP o H36M_TO_J14), the 17 COCO joints
 * (idx17_host) and their 2-D pixels undo_keypoint_normalisation(orthographic_project(rotate pi about x)) for an img_wh-pixel image. */
int hmf_eval_joints(const float* joints, const float* cam, int B, int N, const int* idx14_host, const int* idx17_host, float img_wh, float* j14,
                    float* coco3d, float* joints2d, void* stream);
/* One batch of metrics/eval_metrics_tracker.py:95-433 after the point-set kernels, in two launches: minima over the N samples, the 2-D joint
 * errors over the visible joints, the per-frame arrays and the fp64 running sums.  Any input pointer may be NULL (that group of metrics is
 * skipped); slot[i] = position of metric / count i in `sums` or -1 (order: PVE, PVE-SC, PVE-PA, PVE-T, PVE-T-SC, MPJPE, MPJPE-SC, MPJPE-PA,
 * the eight *_samples_min in the same order, joints2D-L2E, joints2Dsamples-L2E, verts3D_sample_diversity, joints3D_sample_diversity,
 * num_total_test_data, num_vis_joints2D, num_vis_joints2Dsamples).  frames: (HMF_EVAL_NUM_SLOTS - 3, B) float32, row i = the batch's
 * per-frame values of metric i.  sums is accumulated into (+=), images in order: deterministic. */
#define HMF_EVAL_NUM_SLOTS 23
typedef struct {
    int B, N, num_verts, num_joints, chunks_v, chunks_j;
    const float *e_verts, *e_tpose, *e_joints;          /* (B, 3) [plain, SC, PA] of the point estimates (hmf_pointset_errors / hmf_smpl_tpose_errors) */
    const float *es_verts, *es_tpose, *es_joints;       /* (B, N, 3) of the samples                                                                  */
    const float *div_verts, *div_joints;                /* (B, chunks_v) / (B, chunks_j) partials of hmf_sample_diversity (joints: the 17 COCO joints) */
    const float *j2d_pred, *j2d_samples, *j2d_target;   /* (B, 17, 2), (B, N, 17, 2), (B, 17, 2) pixels                                                */
    const unsigned char* j2d_vis;                       /* (B, 17) or NULL = all visible                                                               */
    double* scratch;                                    /* hmf_eval_tracker_scratch_bytes(B)                                                           */
    int slot[HMF_EVAL_NUM_SLOTS];
} hmf_eval_batch_t;
long long hmf_eval_tracker_scratch_bytes(int B);
int hmf_eval_tracker_update(const hmf_eval_batch_t* batch, double* sums, float* frames, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Encoder, training mode (models/resnet.py under train(): nn.BatchNorm2d batch statistics) and its backward                 */
/* ------------------------------------------------------------------------------------------ */
/* *counters[i] += 1 for n device int64 scalars in one launch per 64: the num_batches_tracked buffers of the BatchNorm layers of one training pass
 * (torch.nn.BatchNorm2d semantics, models/resnet.py bn*). */
int hmf_i64_increment_batch(long long* const* counters, int n, void* stream);

/* All tensors NHWC rows [M][C], M = B*H*W, C % 64 == 0.  Forward of one conv+BN(+ReLU)(+residual) unit:
 *   y = hmf_conv2d_nhwc(x, w, bias = NULL, relu = 0)   (or hmf_stem_conv with relu = 0)
 *   hmf_bn_stats(y) -> acc[2][C] (double, zeroed by the caller); hmf_bn_finalize -> mean, rstd, scale = gamma rstd, shift = beta - mean scale,
 *       and the running statistics updated as torch does (momentum, unbiased variance);
 *   out = hmf_bn_apply(y, scale, shift, residual, relu).
 * Backward, given g = dL/d out:  gz = g * (out > 0)  [relu];  hmf_bn_bwd_reduce -> acc = (sum gz, sum gz xhat);  hmf_bn_bwd_finalize ->
 * g_gamma, g_beta and the per-channel coefficients of  gy = a gz + b y + c  (= gamma rstd (gz - mean(gz) - xhat mean(gz xhat)));
 * hmf_bn_bwd_apply writes gy (and optionally gz, the identity branch's gradient).                                                      */
int hmf_bn_stats(const float* y, int M, int C, double* acc, void* stream);
int hmf_bn_finalize(const double* acc, int M, int C, const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                    float* running_var, float* mean, float* rstd, float* scale, float* shift, void* stream);
int hmf_bn_apply(const float* y, const float* scale, const float* shift, const float* residual, float* out, int M, int C, int relu, void* stream);
int hmf_bn_bwd_reduce(const float* g, const float* out, const float* y, const float* mean, const float* rstd, int M, int C, int relu, double* acc,
                      void* stream);
int hmf_bn_bwd_finalize(const double* acc, int M, int C, const float* gamma, const float* mean, const float* rstd, float* g_gamma, float* g_beta,
                        float* ca, float* cb, float* cc, void* stream);
int hmf_bn_bwd_apply(const float* g, const float* out, const float* y, const float* ca, const float* cb, const float* cc, float* gy, float* gz_out,
                     int M, int C, int relu, void* stream);
/* Backward of hmf_maxpool3x3s2_nhwc: x (B,H,W,C) the pooled tensor's input, g_out (B,OH,OW,C) -> g_x; ties go to the first maximum in
 * (kh, kw) scan order, as torch's forward records them.                                                                               */
int hmf_maxpool3x3s2_nhwc_backward(const float* x, const float* g_out, float* g_x, int B, int H, int W, int C, void* stream);
/* The stem's tail in training mode (models/resnet.py:148-151: bn1 -> relu -> maxpool 3x3 / 2 / pad 1) as ONE pass over the convolution's output and
 * its two-pass backward, round 5: the activated (B,H,W,C) map is never written.
 *   forward     out (B,OH,OW,C) = maxpool(relu(scale[c] y + shift[c])),  argmax (B,OH,OW,C) bytes = tap 3 dh + dw of the FIRST maximum (torch's scan)
 *   bwd_reduce  acc[0][C] += sum gz, acc[1][C] += sum gz xhat over the INPUT elements (gz = pooled gradient at a window's recorded arg-max when the
 *               maximum is positive, else 0) -- feeds hmf_bn_bwd_finalize exactly as hmf_bn_bwd_reduce does (M = B H W there)
 *   bwd_apply   gy (B,H,W,C) = ca gz + cb y + cc: the convolution's output gradient (what hmf_maxpool3x3s2_nhwc_backward + hmf_bn_bwd_apply produce)
 * `out` is the forward's pooled map (its sign is the ReLU mask).  C % 64 == 0 for the reduction, C % 4 elsewhere; argmax 4-byte aligned.            */
int hmf_bn_relu_maxpool3x3s2_nhwc(const float* y, const float* scale, const float* shift, float* out, unsigned char* argmax, int B, int H, int W, int C,
                                  void* stream);
int hmf_bn_relu_maxpool3x3s2_nhwc_bwd_reduce(const float* g_out, const unsigned char* argmax, const float* out, const float* y, const float* mean,
                                             const float* rstd, int B, int H, int W, int C, double* acc, void* stream);
int hmf_bn_relu_maxpool3x3s2_nhwc_bwd_apply(const float* g_out, const unsigned char* argmax, const float* out, const float* y, const float* ca,
                                            const float* cb, const float* cc, float* gy, int B, int H, int W, int C, void* stream);
/* Weight gradient of a convolution: gy (B,OH,OW,Cout), x (B,H,W,Cin) NHWC -> dw [Cout][ks][ks][Cin] (channels_last filter layout), padding ks/2.
 * stem_nchw = 1: x is the NCHW proxy input (B,Cin,H,W) and dw comes out as [Cout][Cin][ks][ks] (torch's native layout).
 * dw is zeroed, then accumulated with float atomics over slices of the pixel axis (run-to-run differences at rounding level).            */
int hmf_conv_wgrad(const float* gy, const float* x, float* dw, int B, int H, int W, int Cin, int Cout, int ks, int stride, int stem_nchw, void* stream);
/* The same gradient for stride-1 3x3 layers as Winograd F(3x3, 2x2) (csrc/conv_wgrad_wino.hip: 16 instead of 36 multiplies per 2x2 tile of gy,
 * the sum over tiles as 16 MFMA GEMMs, one inverse transform at the end): gy (B,H,W,Cout), x (B,H,W,Cin) NHWC -> dw_native (Cout,Cin,3,3), the
 * parameter's own layout.  workspace: hmf_conv_wgrad_winograd_workspace_floats(Cin, Cout) floats of device memory (the work-groups' partial sums
 * + the Winograd-domain accumulator; no initialisation needed).  Built for the shapes hmf_conv_wgrad_winograd_supported() accepts (W a multiple of
 * 16 with H % 4 == 0, or W a multiple of 32 with H even, or W = 8 with H % 8 == 0; Cin % 64 == 0, Cout % 32 == 0).  The slices of the tile sum
 * are added by a reduce kernel in a fixed order: no atomics, reproducible run to run.                                                          */
size_t hmf_conv_wgrad_winograd_workspace_floats(int Cin, int Cout);
int hmf_conv_wgrad_winograd_supported(int H, int W, int Cin, int Cout);
int hmf_conv_wgrad_winograd(const float* gy, const float* x, float* dw_native, float* workspace, int B, int H, int W, int Cin, int Cout, void* stream);

/* Input gradient of a 3x3 / stride 2 / pad 1 convolution (models/resnet.py:53-67, :95-100 in the backward of train/train_humaniflow.py:287-290) as
 * four parity-class GEMMs in one launch (csrc/conv_dgrad_s2.hip): gx (B,H,W,Ci) = dgrad(gy (B,OH,OW,Co), w) [+ dgrad of the block's 1x1 / stride 2
 * downsample convolution of gy2 (B,OH,OW,Co2)] [+ residual (B,H,W,Ci)], OH = (H-1)/2+1.  wt (Ci,3,3,Co) = the (Co,Ci,3,3) parameter permuted (1,2,3,0);
 * wt2 (Ci,Co2) = the (Co2,Ci) parameter transposed; gy2 and wt2 both given or both NULL.  Co, Co2 % 32 == 0, Ci % 64 == 0, pointers 16-byte aligned.
 * Every element of gx is written (no zero fill).  Profiler tag 10. */
int hmf_conv_dgrad_s2(const float* gy, const float* wt, const float* gy2, const float* wt2, const float* residual, float* gx, int B, int H, int W,
                      int Co, int Co2, int Ci, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Per-kernel timing with HIP events on the launch stream (bench.py roofline figures)           */
/* ------------------------------------------------------------------------------------------ */
/* While enabled, every launch of the kernels tagged below is bracketed by hipEventRecord on ITS stream
 * (no synchronisation, <= 16384 launches kept; launches issued while their stream is being captured into a HIP graph are not timed).  hmf_profile_collect synchronises the recorded events and
 * copies out, per launch: duration in ms, kernel tag, and the launch's row count M (or B*N for flow).
 * Tags: 1 = k_lbs_mfma, 2 = k_rigid_chain, 3 = k_joints, 4 = k_tpose, 5 = k_flow_ctx0, 6 = k_flow_chain,
 * 7 = k_isgc, 8 = k_conv_igemm, 9 = k_stem_conv, 10 = k_conv_patch, 11 = k_conv_winograd (rows = output pixels B*OH*OW; stem: B).  Returns the number of launches copied (<= max), or a negative error.                */
int hmf_profile_enable(int enable);
int hmf_profile_collect(double* ms_host, int32_t* tag_host, int32_t* rows_host, int max);
/* One wave spins for ~spin_us (<= 2 s) on `stream` and writes {shader-clock cycles, 100 MHz ticks} elapsed to out_dev[0..1] (device memory):
 * cycles / ticks * 100 = the shader clock in MHz a CU sustained while whatever else was running on the device ran (bench.py records it
 * beside `roofline`: the 157.3 TFLOP/s fp32 MFMA peak assumes 2400 MHz). */
int hmf_profile_clock_probe(int spin_us, long long* out_dev, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Host-side packing probes (no GPU needed; used by the CPU test-suite to check operand layouts) */
/* ------------------------------------------------------------------------------------------ */
/* Number of floats / copy-out of the packed pose-blend operand (see DESIGN.md "hmf_smpl_lbs").  */
size_t hmf_smpl_packed_blend_floats(int num_betas);
int hmf_smpl_pack_blend_host(const hmf_smpl_host_t* model, float* out_host);
/* (k, vertex, component) -> offset into the packed pose-blend operand.                           */
size_t hmf_smpl_packed_blend_index(int k, int v, int c);
/* Packed flow weights: total float count and copy-out (layout: DESIGN.md "hmf_flow_forward").     */
size_t hmf_flow_packed_floats(const hmf_flow_host_t* weights);
int hmf_flow_pack_host(const hmf_flow_host_t* weights, float* out_host);

#ifdef __cplusplus
}
#endif
#endif /* HUMANIFLOW_AMD_H */
