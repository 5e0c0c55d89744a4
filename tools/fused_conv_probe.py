"""Is MIOpen's fused conv+bias+ReLU (torch.ops.aten.miopen_convolution_relu) faster than conv + hmf_bias_act for the encoder's shapes?"""
import sys, os, time
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import humaniflow_amd as H
from humaniflow_amd import _lib
lib = _lib.load()
dev = torch.device('cuda:0')
torch.backends.cudnn.benchmark = True
shapes = [  # cin, cout, H, k, stride
    (64, 64, 64, 1, 1), (64, 256, 64, 1, 1), (256, 64, 64, 1, 1), (64, 64, 64, 3, 1), (256, 128, 64, 1, 1), (128, 128, 64, 3, 2),
    (128, 512, 32, 1, 1), (512, 128, 32, 1, 1), (128, 128, 32, 3, 1), (256, 256, 16, 3, 1), (1024, 256, 16, 1, 1), (256, 1024, 16, 1, 1),
    (512, 512, 8, 3, 1), (2048, 512, 8, 1, 1), (512, 2048, 8, 1, 1)]
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for fmt_name, fmt in (('nhwc', torch.channels_last), ('nchw', torch.contiguous_format)):
    print('==', fmt_name)
    for cin, cout, Hh, k, s in shapes:
        x = torch.randn(32, cin, Hh, Hh, device=dev).contiguous(memory_format=fmt)
        w = (torch.randn(cout, cin, k, k, device=dev) * 0.05).contiguous(memory_format=fmt)
        b = torch.randn(cout, device=dev)
        pad = k // 2
        def plain():
            y = F.conv2d(x, w, None, s, pad)
            if not y.is_contiguous(memory_format=fmt): y = y.contiguous(memory_format=fmt)
            N_, C_, H_, W_ = y.shape
            lib.hmf_bias_act(y.data_ptr(), b.data_ptr(), None, N_, C_, H_ * W_, int(fmt_name == 'nhwc'), 1, _lib.current_stream())
            return y
        def fused():
            return torch.ops.aten.miopen_convolution_relu(x, w, b, [s, s], [pad, pad], [1, 1], 1)
        try:
            yf = fused(); yp = plain()
            err = (yf - yp).abs().max().item()
            tf_, tp_ = timeit(fused), timeit(plain)
            tc = timeit(lambda: F.conv2d(x, w, None, s, pad))
            print('cin %4d cout %4d H %3d k%d s%d: conv %.1f us  conv+epilogue %.1f us  miopen fused %.1f us  (diff %.1e)' % (cin, cout, Hh, k, s, tc, tp_, tf_, err), flush=True)
        except Exception as e:
            print('cin %4d cout %4d H %3d k%d s%d: fused failed: %s' % (cin, cout, Hh, k, s, str(e)[:100]), flush=True)
