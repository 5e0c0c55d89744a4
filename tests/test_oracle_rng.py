"""Pin the oracle's restatement of the device generator's normal stream (oracle/rng.py) -- what csrc/rng_philox.h evaluates inside the flow
kernels -- with the Random123 known-answer vectors of philox4x32-10 and the bookkeeping rules of ATen's launch geometry."""
import numpy as np

from oracle import rng


def _kat(counter, key):
    c = np.array([counter], dtype=np.uint32)
    k = np.array([key], dtype=np.uint32)
    return [int(v) for v in rng.philox4x32_10(c, k)[0]]


def test_philox4x32_10_known_answer_vectors():
    # Random123 kat_vectors: philox4x32 10  <counter x4> <key x2>  ->  <output x4>
    assert _kat([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert _kat([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert _kat([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_launch_geometry_and_offset_bookkeeping():
    cap = 2048                                              # MI355X: 256 CUs x (2048 threads / 256)
    assert rng.threads(9600, cap) == 38 * 256 and rng.call_increment(9600, cap) == 4
    assert rng.threads(32000, cap) == 125 * 256 and rng.call_increment(32000, cap) == 4
    assert rng.threads(10 ** 7, cap) == cap * 256 and rng.call_increment(10 ** 7, cap) == 4 * 5      # 1e7 / (4 * 524288) -> 5 float4 per thread
    assert rng.call_increment(0, cap) == 0
    # element e of a call that wraps the grid: thread e % T, its (e / T / 4)-th float4, component (e / T) % 4
    T = rng.threads(10 ** 7, cap)
    w, ii = rng.philox_words(7, 16, 10 ** 7, cap)
    e = 5 * T + 123
    ref = rng.philox4x32_10(np.array([[16 // 4 + 1, 0, 123, 0]], dtype=np.uint32), np.array([[7, 0]], dtype=np.uint32))[0]
    assert np.array_equal(w[e], ref) and ii[e] == 1
    # a pass = one (N,B,nb) call + 23 (B,N,3) calls
    s, p, off = rng.flow_noise(3, 40, B=2, N=5, nb=10, cap_blocks=cap)
    assert s.shape == (5, 2, 10) and p.shape == (23, 2, 5, 3) and off == 40 + 24 * 4
    s2, p2, off2 = rng.flow_noise(3, 40, B=2, N=5, nb=10, cap_blocks=cap, draw_shape=False)
    assert s2 is None and off2 == 40 + 23 * 4 and np.array_equal(p2[0], rng.normal_call(3, 40, 30, cap).reshape(2, 5, 3))
    assert np.array_equal(p[0], rng.normal_call(3, 44, 30, cap).reshape(2, 5, 3))


def test_normals_are_standard():
    z = rng.normal_call(1234, 0, 200000, 2048)
    assert z.dtype == np.float32 and np.isfinite(z).all()
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1.0) < 0.01 and abs((z ** 3).mean()) < 0.03 and abs((z ** 4).mean() - 3.0) < 0.1
