"""Oracle: functional ResNet-18/50 encoder forward from a state_dict. Test infrastructure only.

Follows models/resnet.py:202-217 (conv7x7/2 -> BN -> ReLU -> maxpool3x3/2 -> 4 stages ->
global avg-pool -> flatten, no fc), Bottleneck :102-122 (stride on conv2, :91) and
BasicBlock :59-76, eval-mode BatchNorm (running statistics).
PINNED by tests/golden/resnet_golden.npz (reference module imported as is).
"""
import torch
import torch.nn.functional as F

LAYERS = {18: ('basic', [2, 2, 2, 2]), 50: ('bottleneck', [3, 4, 6, 3])}


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'], sd[p + '.weight'], sd[p + '.bias'],
                        training=False, eps=1e-5)


def _block(x, sd, p, kind, stride):
    idt = x
    if kind == 'bottleneck':
        out = F.relu(_bn(F.conv2d(x, sd[p + '.conv1.weight']), sd, p + '.bn1'))
        out = F.relu(_bn(F.conv2d(out, sd[p + '.conv2.weight'], stride=stride, padding=1), sd, p + '.bn2'))
        out = _bn(F.conv2d(out, sd[p + '.conv3.weight']), sd, p + '.bn3')
    else:
        out = F.relu(_bn(F.conv2d(x, sd[p + '.conv1.weight'], stride=stride, padding=1), sd, p + '.bn1'))
        out = _bn(F.conv2d(out, sd[p + '.conv2.weight'], padding=1), sd, p + '.bn2')
    if (p + '.downsample.0.weight') in sd:
        idt = _bn(F.conv2d(x, sd[p + '.downsample.0.weight'], stride=stride), sd, p + '.downsample.1')
    return F.relu(out + idt)


def resnet_forward(sd, x, num_layers=50, prefix=''):
    kind, layers = LAYERS[num_layers]
    sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    x = F.relu(_bn(F.conv2d(x, sd['conv1.weight'], stride=2, padding=3), sd, 'bn1'))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for li, n in enumerate(layers):
        for bi in range(n):
            stride = 2 if (li > 0 and bi == 0) else 1
            x = _block(x, sd, 'layer%d.%d' % (li + 1, bi), kind, stride)
    return torch.flatten(F.adaptive_avg_pool2d(x, (1, 1)), 1)
