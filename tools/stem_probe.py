"""Per-phase timeline of the fused stem + max-pool kernel (experiments build: stamps per work-group).   python tools/stem_probe.py [B] [--polyphase]"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
from humaniflow_amd import _lib
from humaniflow_amd.synthetic import synthetic_proxy_input
lib = _lib.use_experiments_library()
POLY = '--polyphase' in sys.argv
args_ = [a for a in sys.argv[1:] if not a.startswith('--')]
B = int(args_[0]) if args_ else 32
st = _lib.current_stream()
x = synthetic_proxy_input(B, 256, seed=0).cuda()
w = torch.randn(64, 18, 7, 7) / 30.0
packed = torch.empty(int(lib.hmf_stem_packed_floats()))
_lib.check(lib.hmf_stem_pack_host(w.contiguous().data_ptr(), packed.data_ptr()), 'pack')
pd, bias = packed.cuda(), torch.randn(64).cuda() * 0.1
z = torch.zeros(B, 64, 64, 64, device='cuda').contiguous(memory_format=torch.channels_last)
fn = lambda: _lib.check(lib.hmf_stem_conv_pool(x.data_ptr(), pd.data_ptr(), bias.data_ptr(), z.data_ptr(), B, 18, 256, 256, st), 'stem')
for _ in range(3):
    fn()
torch.cuda.synchronize()
ts = []
for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        fn()
    b.record()
    torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) / 10 * 1e3)
n = min(4096, B * 128)
buf = np.zeros((n, 6), np.int64)
_lib.check(lib.hmf_debug_stem_stamps(buf.ctypes.data_as(ctypes.c_void_p), n), 'stamps')
wall = (buf[:, 5] - buf[:, 4]) / 100.0
cyc = (buf[:, 3] - buf[:, 0]).astype(np.float64)
print('stem B=%d: %.1f us/launch (incl. the zero-fill of the pooled map); %d work-groups; first start -> last end %.1f us; per work-group %.1f us at %.0f MHz: '
      'patch fetch %d cyc, main loop %d (882 MFMAs per wave: %d cycles of issue alone), pooling epilogue %d'
      % (B, float(np.median(ts)), n, (buf[:, 5].max() - buf[:, 4].min()) / 100.0, float(np.median(wall)), float(np.median(cyc / wall)),
         (buf[:, 1] - buf[:, 0]).mean(), (buf[:, 2] - buf[:, 1]).mean(), 882 * 64, (buf[:, 3] - buf[:, 2]).mean()))
starts = np.sort(buf[:, 4])
print('work-group start times (us after the first): 10/25/50/75/90 %%: %s' % np.percentile((starts - starts[0]) / 100.0, [10, 25, 50, 75, 90]).round(1))

if POLY:
    packed = torch.empty(int(lib.hmf_stem_wino_packed_floats()))
    _lib.check(lib.hmf_stem_wino_pack_host(w.contiguous().data_ptr(), packed.data_ptr()), 'pack')
    pd = packed.cuda()
    ws = torch.empty(int(lib.hmf_stem_wino_pool_workspace_bytes(B, 256, 256)), dtype=torch.uint8, device='cuda')
    fn = lambda: _lib.check(lib.hmf_stem_wino_pool(x.data_ptr(), pd.data_ptr(), bias.data_ptr(), z.data_ptr(), B, 18, 256, 256, ws.data_ptr(), ws.numel(), st), 'stem')
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 10 * 1e3)
    buf = np.zeros((n, 12), np.int64)
    _lib.check(lib.hmf_debug_stem_wino_stamps(buf.ctypes.data_as(ctypes.c_void_p), n), 'stamps')
    wall = (buf[:, 6] - buf[:, 5]) / 100.0
    cyc = (buf[:, 4] - buf[:, 0]).astype(np.float64)
    print('polyphase stem B=%d: %.1f us/launch (incl. the zero-fill); per work-group %.1f us at %.0f MHz: main loop with the streamed patch %d cyc (450 MFMAs per wave: %d '
          'cycles of issue alone), class sums + exchange %d, pooling %d'
          % (B, float(np.median(ts)), float(np.median(wall)), float(np.median(cyc / wall)), (buf[:, 2] - buf[:, 0]).mean(), 450 * 64,
             (buf[:, 3] - buf[:, 2]).mean(), (buf[:, 4] - buf[:, 3]).mean()))
    print('   wave 1 (vh = 1) leaves its main loop %d cycles after wave 0 (vh = 0) on average (min %d, max %d)' % ((buf[:, 7] - buf[:, 2]).mean(), (buf[:, 7] - buf[:, 2]).min(), (buf[:, 7] - buf[:, 2]).max()))
    hw, xcc = buf[:, 8], buf[:, 9] & 0xf
    cu = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)      # (xcc, se, sh, cu)
    tg = (hw >> 16) & 0xf
    t0 = buf[:, 5].min()
    ids = np.unique(cu)
    print('   %d distinct CUs seen; timeline of one CU (us: start-end [slot]):' % len(ids))
    for c in ids[:2]:
        sel = np.where(cu == c)[0]
        sel = sel[np.argsort(buf[sel, 5])]
        print('   CU %06x: ' % c + '  '.join('%.1f-%.1f[%d]' % ((buf[i, 5] - t0) / 100.0, (buf[i, 6] - t0) / 100.0, tg[i]) for i in sel[:16]))
