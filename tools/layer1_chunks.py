"""Experiment: layer1 (blocks 0-2 + the seam into layer2.0's first 1x1) on the stem's output in batch chunks, so that the 256-channel
tensors between its kernels (134 MB at B = 32) are small enough to be found in the 256 MB Infinity Cache by their consumer.
    python tools/layer1_chunks.py [B]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from humaniflow_amd.encoder import ResNetEncoder
from humaniflow_amd.synthetic import synthetic_proxy_input

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
enc = ResNetEncoder(50, 18).eval().cuda()
x = synthetic_proxy_input(B, 256, seed=0).cuda()
with torch.no_grad():
    enc(x)
    p = enc._plan
    z = enc._stem_hip(x, p['stem_packed'], p['stem'][1])
    for nblk in (3, 7):
        for chunks in (1, 2, 4, 8):
            zs = z.split(B // chunks)

            def run():
                return [enc._run_blocks(p, zc, None, 0, nblk) for zc in zs]
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                run()
            b.record()
            torch.cuda.synchronize()
            print('blocks 0..%d in %d chunk(s) of %d images: %.1f us' % (nblk - 1, chunks, B // chunks, a.elapsed_time(b) / 20 * 1e3), flush=True)
