// Base noise of the flow drawn INSIDE the kernels, in the reference's draw order and from the device generator's own stream.
//
// The reference draws (models/humaniflow_model.py:258, :309; SURVEY 8b "RNG contract"), per forward:
//     shape_dist.rsample([N])                 -> one standard normal of shape (N, B, nb)           (skipped with use_shape_mode_for_samples)
//     pose_SO3flow_dists[j].rsample([B, N])   -> one standard normal of shape (B, N, 3), j = 0..22
// i.e. 1 + 23 calls of Tensor.normal_() on the device generator.  On ROCm that kernel is ATen's distribution_nullary_kernel over
// rocrand's Philox4x32-10 (`hiprand_normal4`): with `grid = min(#CU * (maxThreadsPerCU / 256), ceil(numel / 256))` blocks of 256 threads,
// thread `idx` owns the Philox subsequence `idx`, starts at the generator's offset, draws one float4 of Box-Muller normals per loop
// iteration and stores component `ii` to element `idx + T * (4 * it + ii)`, T = 256 * grid; the call advances the generator's offset by
// 4 * ceil(numel / (4 T)).  normal_at() below returns element `e` of such a call from (seed, offset) alone -- a counter-based generator
// needs no state between elements -- so k_flow_ctx0 / k_flow_chain produce, in registers, exactly the values the 24 ATen launches would
// have written to HBM.  tests/test_gpu_flow.py checks this against torch.randn on the same generator state.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hmf {

// One forward pass' view of the generator: (seed, offset at the start of the pass) + the launch geometry ATen would use on this device.
struct RngPass {
    unsigned long long seed, offset;
    unsigned int cap_blocks;        // #CU * (maxThreadsPerCU / 256): the grid cap of ATen's calc_execution_policy
    unsigned int pad;
};

// Where a pass starts in the stream (one thread, first kernel of the pass).  `state` != NULL: a captured pass -- take (seed, offset) from the
// device-resident state and advance it atomically.  Otherwise the immediates the host read from the generator; `mirror` != NULL then receives
// the generator's state after this pass.
__device__ inline RngPass rng_begin_pass(RngPass imm, unsigned long long* state, unsigned long long inc, unsigned long long* mirror) {
    RngPass p = imm;
    if (state) {
        p.seed = state[0];
        p.offset = atomicAdd(state + 1, inc);
    } else if (mirror) {
        mirror[0] = imm.seed;
        mirror[1] = imm.offset + inc;
    }
    return p;
}

__host__ __device__ inline unsigned int rng_threads(unsigned long long numel, unsigned int cap_blocks) {
    unsigned long long g = (numel + 255) / 256;
    if (g > cap_blocks) g = cap_blocks;
    if (g < 1) g = 1;
    return (unsigned int)g * 256u;
}
// generator offset consumed by one normal_() call of `numel` elements
__host__ __device__ inline unsigned long long rng_call_increment(unsigned long long numel, unsigned int cap_blocks) {
    if (numel == 0) return 0;
    const unsigned long long T = rng_threads(numel, cap_blocks);
    return ((numel - 1) / (T * 4) + 1) * 4;
}

__device__ __forceinline__ void philox_round(unsigned int (&c)[4], unsigned int k0, unsigned int k1) {
    const unsigned long long m0 = (unsigned long long)0xD2511F53u * c[0];
    const unsigned long long m1 = (unsigned long long)0xCD9E8D57u * c[2];
    const unsigned int hi0 = (unsigned int)(m0 >> 32), lo0 = (unsigned int)m0, hi1 = (unsigned int)(m1 >> 32), lo1 = (unsigned int)m1;
    const unsigned int n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
}

// Philox4x32-10 block of (seed, subsequence, counter): what rocrand's engine holds as `result` after rocrand_init(seed, subsequence, 4 * counter)
__device__ __forceinline__ void philox4x32_10(unsigned long long seed, unsigned long long subsequence, unsigned long long counter, unsigned int (&out)[4]) {
    unsigned int c[4] = {(unsigned int)counter, (unsigned int)(counter >> 32), (unsigned int)subsequence, (unsigned int)(subsequence >> 32)};
    unsigned int k0 = (unsigned int)seed, k1 = (unsigned int)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    philox_round(c, k0, k1);
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

// rocrand's Box-Muller pair (rocrand_normal.h box_muller): same expressions, same device math functions
__device__ __forceinline__ float box_muller_component(unsigned int x, unsigned int y, int second) {
    const float u = 2.3283064e-10f + (x * 2.3283064e-10f);
    const float v = 1.46291807e-09f + (y * 1.46291807e-09f);
    const float s = sqrtf(-2.0f * logf(u));
    float sn, cs;
    __sincosf(v, &sn, &cs);
    return (second ? cs : sn) * s;
}

// element `e` of a normal_() call of `numel` elements that starts at generator offset `offset` (a multiple of 4)
__device__ __forceinline__ float normal_at(unsigned long long seed, unsigned long long offset, unsigned int T, unsigned int e) {
    const unsigned int idx = e % T, r = e / T;                  // calls of < 2^32 elements (checked on the host)
    unsigned int p[4];
    philox4x32_10(seed, idx, (offset >> 2) + (r >> 2), p);
    const int ii = (int)(r & 3);
    return box_muller_component(ii < 2 ? p[0] : p[2], ii < 2 ? p[1] : p[3], ii & 1);
}

// Where a kernel takes the base noise from: injected tensors (exact-parity tests, `noise=` of HumaniflowModel.forward) or the generator.
struct NoiseSrc {
    const float* eps_shape;          // N x B x nb, or NULL
    const float* eps_pose;           // 23 x B x N x 3, or NULL
    const RngPass* pass;             // this pass' (seed, offset) snapshot in the workspace (written by the first kernel of the pass), or NULL
    int draw_shape;                  // 1: the (N, B, nb) shape noise is the pass' first generator call
    unsigned int T_shape, T_pose;    // ATen launch geometry of the two call shapes
    unsigned long long inc_shape, inc_pose;     // generator offset consumed per call
};
__device__ __forceinline__ bool has_shape_noise(const NoiseSrc& s) { return s.eps_shape != nullptr || s.draw_shape != 0; }
// element e = (n * B + b) * nb + i of the (N, B, nb) call
__device__ __forceinline__ float shape_noise(const NoiseSrc& s, unsigned int e) {
    if (s.eps_shape) return s.eps_shape[e];
    return normal_at(s.pass->seed, s.pass->offset, s.T_shape, e);
}
// element e = (b * N + n) * 3 + c of body part j's (B, N, 3) call; per_part = B * N * 3
__device__ __forceinline__ float pose_noise(const NoiseSrc& s, int j, unsigned int e, unsigned long long per_part) {
    if (s.eps_pose) return s.eps_pose[(unsigned long long)j * per_part + e];
    return normal_at(s.pass->seed, s.pass->offset + (s.draw_shape ? s.inc_shape : 0ull) + (unsigned long long)j * s.inc_pose, s.T_pose, e);
}

}  // namespace hmf
