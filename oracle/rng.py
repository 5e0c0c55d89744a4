"""Oracle: the device generator's normal stream as the flow kernels draw it (csrc/rng_philox.h).  Test infrastructure only.

The reference draws its base noise with Tensor.normal_() on the device generator (models/humaniflow_model.py:258, :309 through
torch.distributions; SURVEY 8b "RNG contract").  On ROCm that is ATen's distribution_nullary_kernel over rocrand's Philox4x32-10
(hiprand_normal4): thread `idx` of a grid of T = 256 * min(cap_blocks, ceil(numel / 256)) threads owns Philox subsequence `idx`, starts at
the generator's offset and writes component ii of its it-th float4 of Box-Muller normals to element idx + T * (4 it + ii); the call
advances the offset by 4 * ceil(numel / (4 T)).

Pinned by (tests/test_oracle_rng.py) the Random123 known-answer vectors of philox4x32-10 and, on the GPU (tests/test_gpu_heads_rng.py),
by torch.randn on the same generator state; the kernels are then held to this restatement bit for bit on the Philox words.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter, key):
    """counter (..., 4) uint32, key (..., 2) uint32 -> (..., 4) uint32 (Salmon et al. 2011, ten rounds, key bumped between rounds)."""
    c = [np.asarray(counter[..., i], dtype=np.uint32).copy() for i in range(4)]
    k0, k1 = np.asarray(key[..., 0], dtype=np.uint32).copy(), np.asarray(key[..., 1], dtype=np.uint32).copy()
    with np.errstate(over='ignore'):
        for r in range(10):
            p0 = M0 * c[0].astype(np.uint64)
            p1 = M1 * c[2].astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & MASK).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & MASK).astype(np.uint32)
            c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
            if r < 9:
                k0 = (k0 + W0).astype(np.uint32)
                k1 = (k1 + W1).astype(np.uint32)
    return np.stack(c, axis=-1)


def threads(numel, cap_blocks):
    g = min(max((numel + 255) // 256, 1), cap_blocks)
    return g * 256


def call_increment(numel, cap_blocks):
    if numel == 0:
        return 0
    T = threads(numel, cap_blocks)
    return ((numel - 1) // (T * 4) + 1) * 4


def philox_words(seed, offset, numel, cap_blocks):
    """The (numel, 4) Philox words behind the elements of one normal_() call and the component ii each element takes."""
    T = threads(numel, cap_blocks)
    e = np.arange(numel, dtype=np.uint64)
    idx, r = e % np.uint64(T), e // np.uint64(T)
    ctr = np.uint64(offset // 4) + (r >> np.uint64(2))
    counter = np.stack([(ctr & MASK).astype(np.uint32), (ctr >> np.uint64(32)).astype(np.uint32),
                        (idx & MASK).astype(np.uint32), (idx >> np.uint64(32)).astype(np.uint32)], axis=-1)
    key = np.broadcast_to(np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32), (numel, 2))
    return philox4x32_10(counter, key), (r & np.uint64(3)).astype(np.int64)


def normal_call(seed, offset, numel, cap_blocks):
    """One normal_() call of `numel` elements: rocrand's Box-Muller (rocrand_normal.h) in float32."""
    w, ii = philox_words(seed, offset, numel, cap_blocks)
    x = np.where(ii < 2, w[:, 0], w[:, 2])
    y = np.where(ii < 2, w[:, 1], w[:, 3])
    f = np.float32
    u = f(2.3283064e-10) + x.astype(f) * f(2.3283064e-10)
    v = f(1.46291807e-09) + y.astype(f) * f(1.46291807e-09)
    s = np.sqrt(f(-2.0) * np.log(u, dtype=f), dtype=f)
    return np.where((ii & 1) == 1, np.cos(v, dtype=f), np.sin(v, dtype=f)).astype(f) * s


def flow_noise(seed, offset, B, N, nb, cap_blocks, draw_shape=True):
    """The base noise of one forward(num_samples = N) in the reference's draw order: (eps_shape (N, B, nb) or None, eps_pose (23, B, N, 3),
    offset after the pass)."""
    eps_shape = None
    if draw_shape:
        eps_shape = normal_call(seed, offset, N * B * nb, cap_blocks).reshape(N, B, nb)
        offset += call_increment(N * B * nb, cap_blocks)
    parts = []
    for _ in range(23):
        parts.append(normal_call(seed, offset, B * N * 3, cap_blocks).reshape(B, N, 3))
        offset += call_increment(B * N * 3, cap_blocks)
    return eps_shape, np.stack(parts), offset
