"""Kernel times of the evaluation epilogue at the BASELINE configs[3] sizes (B = 32, N = 100).   python tools/eval_probe.py"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from humaniflow_amd import _lib
lib = _lib.load()
dev = 'cuda:0'


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


B, N = 32, 100
st = _lib.current_stream()
for npts in (6890, 90, 17, 14):
    for npairs, group in ((B * N, N), (B, 1)):
        P = torch.randn(npairs, npts, 3, device=dev)
        T = torch.randn(npairs // group, npts, 3, device=dev)
        out = torch.empty(npairs, 3, device=dev)
        for pa in (0, 1):
            t = timeit(lambda: _lib.check(lib.hmf_pointset_errors(P.data_ptr(), T.data_ptr(), npairs, group, npts, pa, out.data_ptr(), st), 'pe'))
            print('pointset_errors npairs=%5d npts=%5d want_pa=%d  %7.1f us  (%.0f MB -> %.2f TB/s)' % (npairs, npts, pa, t, P.numel() * 4 / 1e6, P.numel() * 4 / t / 1e6), flush=True)
for npts in (6890, 17):
    S = torch.randn(B, N, npts, 3, device=dev)
    part = torch.empty(B, (npts + 255) // 256, device=dev)
    t = timeit(lambda: _lib.check(lib.hmf_sample_diversity(S.data_ptr(), B, N, npts, part.data_ptr(), st), 'div'))
    print('sample_diversity npts=%5d  %7.1f us  (%.0f MB -> %.2f TB/s)' % (npts, t, S.numel() * 4 / 1e6, S.numel() * 4 / t / 1e6), flush=True)
