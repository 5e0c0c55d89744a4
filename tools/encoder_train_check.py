"""Whole-encoder training graph vs float64 module autograd, per-parameter relative gradient error (and ATen float32's for comparison).
    python tools/encoder_train_check.py [layers] [B] [size]"""
import copy
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from humaniflow_amd.encoder import ResNetEncoder

layers, B, size = (int(v) for v in (sys.argv[1:4] + ['50', '4', '128'][len(sys.argv) - 1:]))
rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp(min=1e-30)).item()      # relative L2 error
torch.manual_seed(layers)
enc = ResNetEncoder(layers, 18).cuda().train()
if os.environ.get('BNINIT'):
    with torch.no_grad():
        for m in enc.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.2)
ref, ref32 = copy.deepcopy(enc).double(), copy.deepcopy(enc)
x = torch.randn(B, 18, size, size, device='cuda')
w = torch.randn(B, enc.out_dim, device='cuda')
f_ref = ref.forward_module(x.double()); (f_ref * w.double()).sum().backward()
f32 = ref32.forward_module(x); (f32 * w).sum().backward()
feats = enc(x); (feats * w).sum().backward()
print('feats: ours %.2e  aten32 %.2e' % (rel(feats, f_ref), rel(f32, f_ref)))
rows = []
for (n, p), (_, pr), (_, p32) in zip(enc.named_parameters(), ref.named_parameters(), ref32.named_parameters()):
    rows.append((rel(p.grad, pr.grad), rel(p32.grad, pr.grad), n))
for r in rows[:20]:
    print('   %.2e  (aten32 %.2e)  %s' % r)
for r in sorted(rows, reverse=True)[:6]:
    print('%.2e  (aten32 %.2e)  %s' % r)
