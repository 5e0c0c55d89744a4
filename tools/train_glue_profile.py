"""Which torch (ATen) operators does one training step still launch, and from which source lines?  CPU-side operator profile of two eager
training steps at the configs[4] shape (torch.profiler, CPU activity, grouped by input shapes: every leaf operator below launches one or more device
kernels; Python stacks are not recorded by this torch build, the shapes are what identifies the source: (72, 10) = the shape prior / Normal log-prob of the loss,
(256, 534) = the isgc layer's weight gradient assembled from slices, (72, 9, 17, 2) = the 2D-joint loss over point estimate + 8 samples, ...).
    python tools/train_glue_profile.py [B] > gpurun_out/train_glue.txt"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import collections
import torch
import humaniflow_amd as H
from humaniflow_amd import train as T
from humaniflow_amd.losses import HumaniflowLoss
from humaniflow_amd.synthetic import init_benchmark_weights

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 72
N = 8
cfg = H.get_humaniflow_cfg_defaults()
cfg.MODEL.NUM_RESNET_LAYERS = 18
torch.manual_seed(0)
model = init_benchmark_weights(H.HumaniflowModel(torch.device('cpu'), cfg.MODEL, H.SMPL_PARENTS), seed=0).to(dev)
model.train()
smpl = H.SMPL(model_data=H.synthetic_smpl_model(seed=0), batch_size=B, create_transl=False).to(dev)
for p in smpl.parameters():
    p.requires_grad_(False)
criterion = HumaniflowLoss(cfg.LOSS, cfg.DATA.PROXY_REP_SIZE)
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.Adam(params, lr=cfg.TRAIN.LR, fused=True, capturable=True)
red = T.GradBucketReducer(params, None)
batch = T.synthetic_train_batch(smpl, B, cfg.DATA.PROXY_REP_SIZE, dev, seed=1000, data_cfg=cfg.DATA)
for _ in range(3):
    T.train_step(model, smpl, criterion, opt, red, *batch, N)
torch.cuda.synchronize()
STEPS = 2
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    for _ in range(STEPS):
        T.train_step(model, smpl, criterion, opt, red, *batch, N)
    torch.cuda.synchronize()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LEAF = ('aten::fill_', 'aten::zero_', 'aten::add', 'aten::add_', 'aten::copy_', 'aten::mul', 'aten::mul_', 'aten::sub', 'aten::div', 'aten::div_', 'aten::neg',
        'aten::flip', 'aten::index', 'aten::index_put_', 'aten::sum', 'aten::mean', 'aten::cat', 'aten::stack', 'aten::exp', 'aten::log', 'aten::sqrt', 'aten::pow',
        'aten::where', 'aten::clamp', 'aten::clamp_min', 'aten::abs', 'aten::norm', 'aten::linalg_vector_norm', 'aten::isfinite', 'aten::eq', 'aten::ne', 'aten::lt',
        'aten::gt', 'aten::ge', 'aten::le', 'aten::bitwise_and', 'aten::bitwise_not', 'aten::mm', 'aten::addmm', 'aten::bmm', 'aten::matmul', 'aten::_foreach_norm',
        'aten::_fused_adam_', 'aten::gather', 'aten::scatter_', 'aten::index_select', 'aten::masked_fill_', 'aten::elu', 'aten::elu_backward', 'aten::sigmoid',
        'aten::tanh', 'aten::cos', 'aten::sin', 'aten::rsqrt', 'aten::reciprocal', 'aten::square', 'aten::all', 'aten::any', 'aten::max', 'aten::min', 'aten::normal_',
        'aten::_to_copy', 'aten::sum_to_size', 'aten::threshold_backward', 'aten::linalg_norm', 'aten::nll_loss', 'aten::mse_loss')
counts = collections.Counter()
for ev in prof.key_averages(group_by_stack_n=12, group_by_input_shape=True):
    if ev.key not in LEAF:
        continue
    where = 'autograd engine / unknown'
    for fr in (ev.stack or []):
        if ('humaniflow_amd/' in fr or 'bench.py' in fr) and '/tools/' not in fr:
            where = fr[fr.index('humaniflow_amd/'):] if 'humaniflow_amd/' in fr else fr
            break
    shp = str(ev.input_shapes)[:60] if ev.input_shapes else ''
    counts[(ev.key, where.strip(), shp)] += ev.count
print('leaf ATen operators per training step (B = %d, N = %d, ResNet-18), by the first frame inside the package:' % (B, N))
tot = 0
by_where = collections.Counter()
for (name, where, shp), c in sorted(counts.items(), key=lambda kv: -kv[1]):
    print('%6.1f  %-24s %-80s %s' % (c / STEPS, name, where[:80], shp))
    tot += c
    by_where[where.split('(')[0]] += c
print('total %.1f leaf operators per step' % (tot / STEPS))
print('by file:', [(k, round(v / STEPS, 1)) for k, v in by_where.most_common(12)])
