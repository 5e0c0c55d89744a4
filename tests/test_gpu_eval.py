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


@pytest.mark.parametrize('N', [5, 100, 130])
def test_sample_diversity_one_pass_and_fallback(N):
    """N <= 128: the register-resident one-pass kernel; above: the two-pass kernel.  Both vs the torch restatement, ragged point counts."""
    trk = ev.EvalMetricsTracker(torch.device('cuda:0'))
    g = torch.Generator().manual_seed(N)
    for npts in (17, 200, 6890):
        S = torch.randn(3, 1, npts, 3, generator=g) + 0.05 * torch.randn(3, N, npts, 3, generator=g)
        d = trk._diversity(S.cuda()).cpu()
        refd = om.sample_diversity(S.double())
        assert torch.allclose(d, refd, rtol=2e-5), (N, npts)


def _smpl():
    import humaniflow_amd as H
    return H.SMPL(model_data=H.synthetic_smpl_model(seed=0), create_transl=False).to('cuda:0')


def test_tpose_errors_from_the_shape_vectors_equal_the_materialised_route():
    """hmf_smpl_tpose_errors (no T-pose vertex tensor) vs hmf_pointset_errors on smpl.tpose() bodies vs the oracle metric functions on the same
    bodies: PVE-T and PVE-T-SC of (B, N) sample shapes and of B point-estimate shapes against B targets."""
    smpl = _smpl()
    trk = ev.EvalMetricsTracker(torch.device('cuda:0'))
    g = torch.Generator().manual_seed(3)
    for B, N in ((3, 7), (2, 100), (4, 130)):
        bt = (torch.randn(B, 10, generator=g) * 1.25).cuda()
        bs = (bt[:, None].cpu() + 0.8 * torch.randn(B, N, 10, generator=g)).cuda()
        bp = (bt.cpu() + 0.5 * torch.randn(B, 10, generator=g)).cuda()
        Tt, Ts, Tp = smpl.tpose(bt), smpl.tpose(bs.reshape(B * N, 10)).reshape(B, N, -1, 3), smpl.tpose(bp)
        for got, verts in ((smpl.tpose_errors(bs, bt), Ts), (smpl.tpose_errors(bp, bt), Tp)):
            via_verts = trk._pointset(verts, Tt, False)
            ref = om.pointset_errors(verts.double().cpu(), Tt.double().cpu(), False)
            assert got.shape == via_verts.shape
            assert (got[..., :2] - via_verts[..., :2]).abs().max() < 2e-7 + 1e-5 * via_verts.abs().max()
            assert torch.allclose(got[..., :2].double().cpu(), ref[..., :2], rtol=2e-5, atol=2e-7)
            assert (got[..., 2] == 0).all()
        packed = smpl.tpose_errors_packed(torch.cat([bs.reshape(B * N, 10), bp]), bt, B, N)          # one launch for both kinds of row
        assert torch.equal(packed[:B * N].reshape(B, N, 3), smpl.tpose_errors(bs, bt)) and torch.equal(packed[B * N:], smpl.tpose_errors(bp, bt))
    # identical shapes: exact zeros for the plain error, ~0 for the corrected one
    e = smpl.tpose_errors(bt, bt)
    assert e[:, 0].abs().max() == 0 and e[:, 1].abs().max() < 1e-6


def test_eval_joints_kernel_equals_the_torch_operator_chain():
    g = torch.Generator().manual_seed(5)
    B, N = 3, 6
    joints = torch.randn(B * N + B, 90, 3, generator=g).cuda()
    cam = (torch.tensor([0.9, 0.0, 0.0]) + 0.1 * torch.randn(B, 3, generator=g)).cuda()
    j14, jc, j2d = ev.eval_joints(joints, cam, B, N, 256)
    assert torch.equal(j14, ev.select_joints(joints, 'h36m_j14')) and torch.equal(jc, ev.select_joints(joints, 'coco'))
    cam_rows = torch.cat([cam[:, None].expand(-1, N, -1).reshape(B * N, 3), cam])
    ref = ev.undo_keypoint_normalisation(ev.orthographic_project_torch(ev.flip_about_x(jc), cam_rows), 256)
    assert (j2d - ref).abs().max() < 1e-4                                   # pixels


def test_fused_tracker_update_equals_the_operator_path():
    """The default device path (point-set kernels + ONE k_eval_tracker_update launch) vs the same accumulation rules as torch operators
    (`fused = False`, what the CPU-shim tracker of tests/test_distributed.py drives): running sums, counts and every per-frame array --
    on the reference-generated golden batches and on a synthetic batch that carries shape vectors instead of T-pose bodies."""
    z, batches = _golden_batches()
    names = [str(k) for k in z['final_names']]
    dev = torch.device('cuda:0')
    smpl = _smpl()
    g = torch.Generator().manual_seed(11)
    B, N = 4, 9
    tv = torch.randn(B, 6890, 3, generator=g)
    syn_pred = {'verts3D': tv + 0.02 * torch.randn(B, 6890, 3, generator=g), 'joints3D': torch.randn(B, 14, 3, generator=g),
                'joints2D': torch.rand(B, 17, 2, generator=g) * 256, 'verts3D_samples': tv[:, None] + 0.03 * torch.randn(B, N, 6890, 3, generator=g),
                'joints3D_samples': torch.randn(B, N, 14, 3, generator=g), 'joints3D_coco_samples': torch.randn(B, N, 17, 3, generator=g),
                'joints2Dsamples': torch.rand(B, N, 17, 2, generator=g) * 256, 'tpose_betas': torch.randn(B, 10, generator=g),
                'tpose_betas_samples': torch.randn(B, N, 10, generator=g)}
    syn_targ = {'verts3D': tv, 'joints3D': torch.randn(B, 14, 3, generator=g), 'tpose_betas': torch.randn(B, 10, generator=g),
                'joints2D': torch.rand(B, 17, 2, generator=g) * 256, 'joints2D_vis': torch.rand(B, 17, generator=g) > 0.3}
    for metrics, todo in ((names, batches), (ev.METRICS, [(syn_pred, syn_targ)]), (['PVE', 'MPJPE_samples_min', 'joints2D-L2E'], [(syn_pred, syn_targ)])):
        a, b = ev.EvalMetricsTracker(dev, metrics, smpl=smpl), ev.EvalMetricsTracker(dev, metrics, smpl=smpl)
        b.fused = False
        for pred, targ in todo:
            pc, tc = {k: v.cuda() for k, v in pred.items()}, {k: v.cuda() for k, v in targ.items()}
            assert a._fused_ok(pc, tc)
            a.update_per_batch(pc, tc)
            b.update_per_batch(pc, tc)
        assert torch.allclose(a.sums, b.sums, rtol=1e-6, atol=1e-9), (a.sums, b.sums)
        fa, fb = a.reduce()[1], b.reduce()[1]
        for k in metrics:
            assert fa[k].shape == fb[k].shape, k
            assert torch.allclose(fa[k], fb[k], rtol=2e-6, atol=1e-7), k
        ma, mb = a.compute_final_metrics()[0], b.compute_final_metrics()[0]
        assert set(ma) == set(mb) and all(abs(ma[k] - mb[k]) <= 1e-6 * max(1.0, abs(mb[k])) for k in ma)
