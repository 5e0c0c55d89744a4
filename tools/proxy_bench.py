"""Time hmf_proxy_rep (fused Canny + heat-maps) at the benchmark shape and compare with its HBM roofline and with the eager oracle on the GPU."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import humaniflow_amd as H
from humaniflow_amd.proxy import proxy_representation
dev = torch.device('cuda:0')
B, S = 32, 256
g = torch.Generator().manual_seed(0)
img = torch.nn.functional.interpolate(torch.rand(B, 3, 16, 16, generator=g), size=(S, S), mode='bicubic', align_corners=False).contiguous().to(dev)
j2d = (torch.rand(B, 17, 2, generator=g) * S).to(dev)
vis = torch.ones(B, 17, device=dev)
cfg = H.get_humaniflow_cfg_defaults().DATA
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
t = timeit(lambda: proxy_representation(img, j2d, vis, cfg))
nbytes = B * S * S * 4 * (3 + 18)
print('hmf_proxy_rep B=%d S=%d: %.1f us  -> %.2f TB/s algorithmic (%.0f MB), %.2f of 8 TB/s' % (B, S, t, nbytes / t / 1e6, nbytes / 1e6, nbytes / t / 1e6 / 8))
# the same op chain as eager torch ops on the GPU (what the reference would launch), from the oracle restatement
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import proxy as op
import oracle.proxy
_orig = oracle.proxy.gaussian_taps
oracle.proxy.gaussian_taps = lambda *a, **k: _orig(*a, **k).to(dev)
def eager():
    e = op.canny(img, True, 1.0, 5, 0.0)['thresholded_thin_edges']
    r = torch.arange(S, device=dev).float()
    hm = torch.exp(-(((r[None, None, :, None] - j2d[:, :, 1, None, None]) / 4.0) ** 2) / 2 - (((r[None, None, None, :] - j2d[:, :, 0, None, None]) / 4.0) ** 2) / 2)
    return torch.cat([e, hm * vis[:, :, None, None]], 1)
try:
    te = timeit(eager, reps=5)
    print('eager torch op chain on the same GPU: %.1f us  (%.1fx)' % (te, te / t))
except Exception as ex:
    print('eager comparison skipped:', str(ex)[:200])
