"""Generate the golden vectors under tests/golden/ from the REFERENCE's own Python.

Run in the build container only (needs /root/reference; nothing of the reference is copied -- only
inputs and the outputs its functions produce are stored):

    python tests/golden/make_golden.py [/root/reference]

Importable reference pieces: utils/rigid_transform_utils.py (with an empty `cv2` stub),
models/resnet.py, the radial-tanh / to / so3-exp transforms and LocalDiffeoTransformedDistribution (with
empty package modules for models.norm_flows so their pyro-importing __init__ files are bypassed, and a
two-class stub for pyro.distributions.conditional).  pyro-ppl and smplx are absent, so the spline coupling
and LBS have no golden vectors ("parity unpinned", oracle/__init__.py).
"""
import importlib.util
import math
import os
import sys
import types

import numpy as np
import torch

REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(OUT, '..'))
from helpers import deterministic_state_dict  # noqa: E402


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def import_reference():
    sys.modules['cv2'] = types.ModuleType('cv2')
    sys.path.insert(0, REF)
    import utils.rigid_transform_utils as rtu
    import models.resnet as resnet
    for pkg in ('models.norm_flows', 'models.norm_flows.transforms'):
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(REF, *pkg.split('.'))]
        sys.modules[pkg] = m
    tdir = os.path.join(REF, 'models', 'norm_flows', 'transforms')
    ld = _load('models.norm_flows.transforms.local_diffeo_transform', os.path.join(tdir, 'local_diffeo_transform.py'))
    sys.modules['models.norm_flows.transforms'].LocalDiffeoTransform = ld.LocalDiffeoTransform
    rt = _load('models.norm_flows.transforms.scaled_radial_tanh_transform', os.path.join(tdir, 'scaled_radial_tanh_transform.py'))
    to = _load('models.norm_flows.transforms.to_transform', os.path.join(tdir, 'to_transform.py'))
    ex = _load('models.norm_flows.transforms.so3_exp_transform', os.path.join(tdir, 'so3_exp_transform.py'))
    pyro = types.ModuleType('pyro')
    pd = types.ModuleType('pyro.distributions')
    pc = types.ModuleType('pyro.distributions.conditional')

    class ConditionalDistribution:
        pass

    class ConstantConditionalDistribution(ConditionalDistribution):
        def __init__(self, base):
            self.base = base

        def condition(self, ctx):
            return self.base
    pc.ConditionalDistribution, pc.ConstantConditionalDistribution = ConditionalDistribution, ConstantConditionalDistribution
    sys.modules.update({'pyro': pyro, 'pyro.distributions': pd, 'pyro.distributions.conditional': pc})
    dist = _load('models.norm_flows.local_diffeo_transformed_distribution',
                 os.path.join(REF, 'models', 'norm_flows', 'local_diffeo_transformed_distribution.py'))
    return rtu, resnet, rt, to, ex, dist


def rotations_with_angles(rtu, angles, gen):
    axes = torch.randn(len(angles), 3, generator=gen, dtype=torch.float64)
    axes = axes / axes.norm(dim=-1, keepdim=True)
    return rtu.so3_exp(axes * torch.tensor(angles, dtype=torch.float64)[:, None])


def main():
    rtu, resnet, rt, to, ex, dist = import_reference()
    g = torch.Generator().manual_seed(1234)
    # ---------------- SO(3) closed forms ----------------------------------------------------------
    v = torch.randn(64, 3, generator=g, dtype=torch.float64) * 1.5
    edge = torch.tensor([[0, 0, 0], [1e-12, 0, 0], [0, 1e-3, 0], [math.pi / 2, 0, 0], [0, 0, math.pi - 5e-3],
                         [0, math.pi - 1e-4, 0], [2.0, -2.0, 1.0], [1e-11, 1e-11, 0]], dtype=torch.float64)
    v = torch.cat([v, edge])
    R = rtu.so3_exp(v)
    angles = [0.0, 1e-12, 1e-3, math.pi / 2, math.pi - 5e-3, math.pi - 1e-4, math.pi - 9e-3, math.pi - 1.1e-2, 3.0, 0.5]
    R_log_in = torch.cat([rotations_with_angles(rtu, angles, g), R[:32], torch.eye(3, dtype=torch.float64)[None],
                          torch.diag(torch.tensor([-1.0, 1.0, -1.0], dtype=torch.float64))[None]])
    log_v = rtu.so3_log(R_log_in, return_axis_angle=True)
    xset = rtu.so3_xset(v, 1)
    logdet = rtu.so3_log_abs_det_jacobian(v)
    x6 = torch.randn(64, 6, generator=g)
    x6 = torch.cat([x6, torch.tensor([[1, 0, 0, 1, 0, 0], [2, 0, 0, 3, 0, 0], [1e-3, 1, 0, 1, 1, 0.5]], dtype=torch.float32)])
    np.savez_compressed(os.path.join(OUT, 'so3_golden.npz'), v=v.numpy(), R=R.numpy(), R_log_in=R_log_in.numpy(),
                        log_v=log_v.numpy(), xset=xset.numpy(), logdet=logdet.numpy(), x6=x6.numpy(),
                        R6=rtu.rot6d_to_rotmat(x6).numpy(), rot6d_of_R=rtu.rotmat_to_rot6d(R[:8].float()).numpy())
    # ---------------- radial tanh + push-forward log-prob plumbing ------------------------------------
    radius = 1.5 * math.pi
    T = rt.ScaledRadialTanhTransform(radius=radius)
    xs = torch.randn(48, 3, generator=g) * torch.tensor([0.0, 1e-8, 0.1, 3.0, 30.0, 1.0]).repeat(8)[:, None]
    ys = T._call(xs)
    ys_in = torch.randn(48, 3, generator=g)
    ys_in = ys_in / ys_in.norm(dim=-1, keepdim=True) * (torch.rand(48, 1, generator=g) * radius * 0.999)
    ys_in[0] = 0.0
    xinv = T._inverse(ys_in)
    ld = T.log_abs_det_jacobian(xinv, ys_in)
    base = torch.distributions.TransformedDistribution(
        torch.distributions.Independent(torch.distributions.Normal(torch.zeros(3), torch.ones(3) * 0.6, validate_args=False), 1),
        [rt.ScaledRadialTanhTransform(radius=radius)], validate_args=False)
    SO3 = dist.LocalDiffeoTransformedDistribution(base, [to.ToTransform(dict(dtype=torch.float32), dict(dtype=torch.float64)),
                                                         ex.SO3ExpCompactTransform(support_radius=radius)], validate_args=False)
    Rq = torch.cat([rotations_with_angles(rtu, [0.1, 1.0, 1.6, 2.0, 2.5, 3.0, math.pi - 5e-3, math.pi - 1e-4, 1e-3], g),
                    torch.eye(3, dtype=torch.float64)[None], R[:22]])
    lp = SO3.log_prob(Rq)
    x_, xset_, mask_ = ex.SO3ExpCompactTransform(support_radius=radius)._inverse_set(Rq)
    np.savez_compressed(os.path.join(OUT, 'flow_golden.npz'), radius=radius, xs=xs.numpy(), ys=ys.numpy(), ys_in=ys_in.numpy(),
                        xinv=xinv.numpy(), ld=ld.numpy(), Rq=Rq.numpy(), lp=lp.numpy(), inv_x=x_.numpy(),
                        inv_xset=np.nan_to_num(xset_.numpy()), inv_mask=mask_.numpy())
    # ---------------- ResNet encoders --------------------------------------------------------------------
    out = {}
    for layers, ctor in ((50, resnet.resnet50), (18, resnet.resnet18)):
        net = ctor(in_channels=18)
        sd = deterministic_state_dict({k: tuple(t.shape) for k, t in net.state_dict().items()}, seed=7)
        net.load_state_dict(sd, strict=True)
        net.eval()
        x = torch.randn(2, 18, 96, 96, generator=torch.Generator().manual_seed(5))
        with torch.no_grad():
            y = net(x)
        out['keys%d' % layers] = np.array(sorted(net.state_dict().keys()))
        out['feats%d' % layers] = y.numpy()
    np.savez_compressed(os.path.join(OUT, 'resnet_golden.npz'), **out)
    # ---------------- eval metric helpers (numpy, importable as is) -----------------------------------------
    import utils.eval_utils as eu
    rng = np.random.RandomState(3)
    P = rng.randn(6, 50, 3)
    T = P @ np.linalg.qr(rng.randn(3, 3))[0] * 1.3 + rng.randn(6, 1, 3) * 0.2 + rng.randn(6, 50, 3) * 0.05
    # the reference's EvalMetricsTracker itself on two seeded batches (hard-codes 6890 vertices / 14 joints / 17 COCO joints)
    import io, contextlib
    import metrics.eval_metrics_tracker as emt
    names = ['PVE', 'PVE-SC', 'PVE-PA', 'PVE-T', 'PVE-T-SC', 'MPJPE', 'MPJPE-SC', 'MPJPE-PA', 'PVE_samples_min', 'PVE-SC_samples_min',
             'PVE-PA_samples_min', 'PVE-T_samples_min', 'PVE-T-SC_samples_min', 'MPJPE_samples_min', 'MPJPE-SC_samples_min',
             'MPJPE-PA_samples_min', 'joints2D-L2E', 'joints2Dsamples-L2E', 'verts3D_sample_diversity', 'joints3D_sample_diversity']
    NS = 3
    batches = []
    with contextlib.redirect_stdout(io.StringIO()):
        trk = emt.EvalMetricsTracker(names, num_samples_for_prob_metrics=NS)
        trk.initialise_metric_sums()
        trk.initialise_per_frame_metric_lists()
        for bsz in (2, 1):
            tv = rng.randn(bsz, 431, 3) * 0.3
            pred = {'verts3D': tv + rng.randn(bsz, 431, 3) * 0.02, 'tpose_verts3D': rng.randn(bsz, 431, 3) * 0.3,
                    'joints3D': rng.randn(bsz, 14, 3) * 0.3, 'joints2D': rng.rand(bsz, 17, 2) * 256,
                    'verts3D_samples': tv[:, None] + rng.randn(bsz, NS, 431, 3) * 0.03, 'tpose_verts3D_samples': rng.randn(bsz, NS, 431, 3) * 0.3,
                    'joints3D_samples': rng.randn(bsz, NS, 14, 3) * 0.3, 'joints2Dsamples': rng.rand(bsz, NS, 17, 2) * 256,
                    'joints3D_coco_samples': rng.randn(bsz, NS, 17, 3) * 0.3}
            targ = {'verts3D': tv, 'tpose_verts3D': pred['tpose_verts3D'] + rng.randn(bsz, 431, 3) * 0.01, 'joints3D': pred['joints3D'] + rng.randn(bsz, 14, 3) * 0.02,
                    'joints2D': pred['joints2D'] + rng.randn(bsz, 17, 2) * 3, 'joints2D_vis': (rng.rand(bsz, 17) > 0.3)}
            pred = {k: v.astype(np.float32).astype(np.float64) for k, v in pred.items()}       # the fixture stores fp32
            targ = {k: (v.astype(np.float32).astype(np.float64) if v.dtype != bool else v) for k, v in targ.items()}
            trk.update_per_batch(pred, targ, bsz)
            batches.append((pred, targ))
        trk.compute_final_metrics()          # returns nothing (prints); the golden holds the tracker's own running sums
        printed = None
    tr = {}
    for bi, (pred, targ) in enumerate(batches):
        for k, v in pred.items():
            tr['b%d_pred_%s' % (bi, k)] = v.astype(np.float32)
        for k, v in targ.items():
            tr['b%d_targ_%s' % (bi, k)] = v.astype(np.float32) if v.dtype != bool else v
    tr['final_names'] = np.array(names)
    tr['sum_values'] = np.array([float(trk.metric_sums[k]) for k in names])
    tr['num_total_test_data'] = np.array(trk.num_total_test_data)
    tr['num_vis_joints2D'] = np.array(float(trk.metric_sums['num_vis_joints2D']))
    tr['num_vis_joints2Dsamples'] = np.array(float(trk.metric_sums['num_vis_joints2Dsamples']))
    for k in names:
        tr['frames_' + k] = np.concatenate(trk.per_frame_metrics[k])
    np.savez_compressed(os.path.join(OUT, 'tracker_golden.npz'), **tr)
    # sampling helpers that import here: quaternion -> rotation, heat-map arg-max (joints2D_error_sorted_verts_sampling itself needs pytorch3d)
    import utils.label_conversions as lc
    import utils.rigid_transform_utils as rtu
    quat = torch.from_numpy(rng.randn(16, 4))
    hm = torch.from_numpy(rng.rand(2, 17, 24, 24)).float()
    hm[0, 3] = 0.0                                    # an undetected joint
    hm[1, 5] *= 1e-8
    j2d, vis = lc.convert_heatmaps_to_2Djoints_coordinates_torch(hm, eps=1e-6)
    np.savez_compressed(os.path.join(OUT, 'sampling_golden.npz'), quat=quat.numpy(), rotmat=rtu.quat_to_rotmat(quat).numpy(),
                        heatmaps=hm.numpy(), joints2D=j2d.numpy(), vis=vis.numpy())
    np.savez_compressed(os.path.join(OUT, 'metrics_golden.npz'), P=P, T=T, pa=eu.procrustes_analysis_batch(P, T),
                        sc=eu.scale_and_translation_transform_batch(P, T), cam_pts=rng.randn(4, 17, 3), cam=rng.randn(4, 3))
    # ---------------- proxy-representation front end (Canny + heat-maps; both importable as they are) ----------
    import models.canny_edge_detector as ced
    import utils.label_conversions as lc
    gp = torch.Generator().manual_seed(11)
    base = torch.rand(2, 3, 12, 12, generator=gp)
    img = torch.nn.functional.interpolate(base, size=(48, 48), mode='bilinear', align_corners=False)      # smooth blobs with real edges
    img[:, :, 10:30, 14:20] += 0.8                                                                         # a hard-edged rectangle
    img = img.contiguous()
    outs = {}
    for nms, thr in ((True, 0.0), (True, 0.2), (False, 0.1)):
        det = ced.CannyEdgeDetector(non_max_suppression=nms, gaussian_filter_std=1.0, gaussian_filter_size=5, threshold=thr)
        with torch.no_grad():
            o = det(img)
        for k, v in o.items():
            outs['canny_%d_%g_%s' % (int(nms), thr, k)] = v.numpy()
    j2d = torch.rand(2, 17, 2, generator=gp) * 60 - 6
    hm = lc.convert_2Djoints_to_gaussian_heatmaps_torch(j2d, 48, std=4.0)
    np.savez_compressed(os.path.join(OUT, 'proxy_golden.npz'), img=img.numpy(), joints2D=j2d.numpy(), heatmaps=hm.numpy(), **outs)
    print('golden vectors written to', OUT)


if __name__ == '__main__':
    main()
