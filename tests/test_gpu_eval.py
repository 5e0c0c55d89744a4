"""Evaluation epilogue on the device (hmf_pointset_errors / hmf_sample_diversity through the tracker) vs the reference tracker's own
running sums (tests/golden/tracker_golden.npz, produced by the reference's numpy EvalMetricsTracker) and vs the torch restatement."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN
from humaniflow_amd import evaluate as ev
from oracle import metrics as om

pytestmark = pytest.mark.gpu


def _golden_batches():
    z = np.load(os.path.join(GOLDEN, 'tracker_golden.npz'))
    out = []
    for bi in range(2):
        pred = {k[len('b%d_pred_' % bi):]: torch.from_numpy(z[k]) for k in z.files if k.startswith('b%d_pred_' % bi)}
        targ = {k[len('b%d_targ_' % bi):]: torch.from_numpy(z[k]) for k in z.files if k.startswith('b%d_targ_' % bi)}
        out.append((pred, targ))
    return z, out


def test_device_tracker_matches_reference_tracker_sums():
    z, batches = _golden_batches()
    names = [str(k) for k in z['final_names']]
    trk = ev.EvalMetricsTracker(torch.device('cuda:0'), names)
    for pred, targ in batches:
        trk.update_per_batch({k: v.cuda() for k, v in pred.items()}, {k: v.cuda() for k, v in targ.items()})
    sums, frames = trk.reduce()
    for i, k in enumerate(names):
        ref = z['sum_values'][i]
        assert abs(sums[i].item() - ref) < 2e-5 * max(1.0, abs(ref)), (k, sums[i].item(), ref)          # fp32 points, fp64 statistics
        assert np.allclose(frames[k].cpu().numpy(), z['frames_' + k], rtol=2e-5, atol=1e-6), k


def test_pointset_errors_properties():
    """Similarity-transformed copies have zero PA error and (for pure scale+translation) zero SC error; reflections are not undone
    (the reference's det-sign fix); degenerate identical sets give exact zeros."""
    from humaniflow_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    T = torch.randn(3, 6890, 3, generator=g)
    from oracle import so3
    R = so3.so3_exp(torch.randn(3, 3, generator=g).double()).float()
    P = torch.stack([2.5 * T[0] @ R[0].T + 0.3, 0.7 * T[1] - 1.0, T[2] * torch.tensor([1.0, 1.0, -1.0])])
    trk = ev.EvalMetricsTracker(torch.device('cuda:0'))
    e = trk._pointset(P.cuda(), T.cuda(), True).cpu()
    assert e[0, 2] < 1e-5 and e[0, 0] > 0.1                        # rotated + scaled + shifted: PA undoes it
    assert e[1, 1] < 1e-5 and e[1, 2] < 1e-5 and e[1, 0] > 0.1     # scale + shift: SC and PA undo it
    assert e[2, 2] > 0.1                                           # mirror image: no proper rotation aligns it
    ref = torch.linalg.norm(om.procrustes_analysis_batch(P.double(), T.double()) - T.double(), dim=-1).mean(-1)
    assert torch.allclose(e[:, 2].double(), ref, atol=1e-5)
    e = trk._pointset(T.cuda(), T.cuda(), True).cpu()
    assert e.abs().max() < 1e-6
    # samples layout: (B, N, npts, 3) against (B, npts, 3)
    S = T[:, None] + 0.01 * torch.randn(3, 5, 6890, 3, generator=g)
    e = trk._pointset(S.cuda(), T.cuda(), True).cpu()
    ref = torch.linalg.norm(S - T[:, None], dim=-1).mean(-1)
    assert e.shape == (3, 5, 3) and torch.allclose(e[..., 0], ref, rtol=1e-5, atol=1e-7)
    d = trk._diversity(S.cuda()).cpu()
    refd = torch.linalg.norm(S - S.mean(1, keepdim=True), dim=-1).double().sum(dim=(1, 2))
    assert torch.allclose(d, refd, rtol=1e-5)
