"""Randomised fuzz of hmf_pointset_errors / hmf_sample_diversity against the float64 torch restatement (GPU).   python tools/eval_fuzz.py [cases]"""
import os
import random
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from humaniflow_amd import evaluate as ev
from oracle import metrics as om

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rnd = random.Random(0)
trk = ev.EvalMetricsTracker(torch.device('cuda:0'))
worst = 0.0
for it in range(cases):
    g = torch.Generator().manual_seed(rnd.randrange(1 << 30))
    Bt, N, npts = rnd.randint(1, 6), rnd.randint(1, 9), rnd.choice([3, 4, 14, 17, 100, 431, 6890])
    T = torch.randn(Bt, npts, 3, generator=g) * rnd.choice([0.01, 0.3, 5.0]) + torch.randn(Bt, 1, 3, generator=g)
    S = T[:, None] + torch.randn(Bt, N, npts, 3, generator=g) * rnd.choice([1e-3, 0.05, 1.0])
    e = trk._pointset(S.cuda(), T.cuda(), True).cpu().double()
    Sd, Td = S.double(), T.double()[:, None].expand(-1, N, -1, -1)
    nrm = lambda a, b: torch.linalg.norm(a - b, dim=-1).mean(-1)
    ref = torch.stack([nrm(Sd, Td), nrm(om.scale_and_translation_transform_batch(Sd, Td), Td),
                       nrm(om._pa_many(Sd, Td[:, :1].reshape(Bt, 1, npts, 3)), Td)], -1)
    # pass 2 subtracts fp32 coordinates: absolute error ~ 1e-7 x coordinate magnitude, plus fp32 rounding of the mean itself
    err = ((e - ref).abs() / (1e-4 * ref.abs() + 2e-6 * (T.abs().max() + 1.0))).max().item()
    worst = max(worst, err)
    assert err < 1.0, (Bt, N, npts, err, e[0, 0], ref[0, 0])
    d = trk._diversity(S.cuda()).cpu()
    refd = torch.linalg.norm(Sd - Sd.mean(1, keepdim=True), dim=-1).sum(dim=(1, 2))
    assert torch.allclose(d, refd, rtol=1e-4), (Bt, N, npts)
print('%d cases passed; worst error / tolerance %.2f' % (cases, worst))
