"""Numerics of Winograd F(4x4,3x3) against F(2x2,3x3) in float32 on the encoder's stride-1 3x3 shapes (round-3 review item 4: "measure F(4x4,3x3) for
layer3/4 against the 2e-5 * scale bar before rejecting it").  CPU only: both algorithms restated with float32 tensor ops (transforms, 36 / 16
position-wise channel contractions with float32 accumulation, inverse transforms), error vs the float64 direct convolution in units of the largest
output value -- the bar of tests/test_gpu_encoder.py.   python tools/wino44_numerics.py"""
import torch
import torch.nn.functional as F

torch.manual_seed(0)


def mats(kind, dt):
    if kind == 2:
        Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=dt)
        G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=dt)
        At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=dt)
        return Bt, G, At, 2
    Bt = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=dt)
    G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=dt)
    At = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=dt)
    return Bt, G, At, 4


def winograd(x, w, kind):
    """x (B, C, H, W) float32, w (N, C, 3, 3) float32, H, W multiples of the output tile; all arithmetic in float32 (U from float64, rounded once,
    as the plan-time packer does)."""
    Bt, G, At, m = mats(kind, torch.float32)
    _, G64, _, _ = mats(kind, torch.float64)
    U = (G64 @ w.double() @ G64.T).float()                                   # (N, C, a, a)
    a = m + 2
    B_, C, H, W = x.shape
    xp = F.pad(x, (1, 1, 1, 1))
    tiles = xp.unfold(2, a, m).unfold(3, a, m)                               # (B, C, th, tw, a, a)
    V = Bt @ tiles @ Bt.T                                                    # float32
    M = torch.einsum('bchwij,ncij->bnhwij', V, U)                            # float32 accumulation over channels
    Y = At @ M @ At.T                                                        # (B, N, th, tw, m, m)
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(B_, w.shape[0], H, W)


for name, C, hw in (('layer1', 64, 64), ('layer2', 128, 32), ('layer3', 256, 16), ('layer4', 512, 8)):
    B_ = 2
    x = F.relu(torch.randn(B_, C, hw, hw)) * 1.3                             # post-ReLU activations
    w = torch.randn(C, C, 3, 3) * (2.0 / (9 * C)) ** 0.5 * (torch.rand(C, 1, 1, 1) + 0.5)      # kaiming filter x a folded BatchNorm scale
    ref = F.conv2d(x.double(), w.double(), padding=1)
    scale = ref.abs().max().item()
    e2 = (winograd(x, w, 2).double() - ref).abs().max().item() / scale
    e4 = (winograd(x, w, 4).double() - ref).abs().max().item() / scale
    ed = (F.conv2d(x, w, padding=1).double() - ref).abs().max().item() / scale
    print('%s %3d ch @%2d: max error / largest output  direct fp32 %.2e | F(2x2,3x3) %.2e | F(4x4,3x3) %.2e   (bar 2e-5)' % (name, C, hw, ed, e2, e4))
