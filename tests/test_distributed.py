"""Host logic of the data-parallel path on CPU: metric functions vs the reference's numpy (golden), sharding, and the one
exchange step (all-reduce of packed sums + all-gather of per-frame arrays) with gloo at world_size 2."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import GOLDEN, ROOT, CpuShimTracker
from humaniflow_amd import evaluate as ev
from oracle import metrics as om


def test_metric_helpers_match_reference_numpy():
    z = np.load(os.path.join(GOLDEN, 'metrics_golden.npz'))
    P, T = torch.from_numpy(z['P']), torch.from_numpy(z['T'])
    assert np.abs(om.procrustes_analysis_batch(P, T).numpy() - z['pa']).max() < 1e-10
    assert np.abs(om.scale_and_translation_transform_batch(P, T).numpy() - z['sc']).max() < 1e-12
    assert ev.ALL_JOINTS_TO_H36M_MAP == list(range(73, 90)) and len(ev.H36M_TO_J14) == 14 and len(ev.ALL_JOINTS_TO_COCO_MAP) == 17


def test_projection_helpers():
    z = np.load(os.path.join(GOLDEN, 'metrics_golden.npz'))
    pts, cam = torch.from_numpy(z['cam_pts']), torch.from_numpy(z['cam'])
    p2 = ev.orthographic_project_torch(pts, cam)
    assert torch.allclose(p2, cam[:, None, :1] * (pts[..., :2] + cam[:, None, 1:]))
    assert torch.allclose(ev.undo_keypoint_normalisation(torch.tensor([-1.0, 0.0, 1.0]), 256), torch.tensor([0.0, 128.0, 256.0]))
    # 180 degrees about x is (x, -y, -z): the axis-angle rotation the reference applies before projecting
    from oracle import so3
    R = so3.so3_exp(torch.tensor([[np.pi, 0.0, 0.0]], dtype=torch.float64))[0]
    assert torch.allclose(ev.flip_about_x(pts), pts @ R.T, atol=1e-12)


def test_tracker_matches_reference_tracker_sums():
    """The reference's own EvalMetricsTracker (numpy) was run on two seeded batches (tests/golden/make_golden.py); the product tracker's
    accumulation rules (driven on CPU tensors through the test-only shim: point-set kernels replaced by oracle.metrics) must reproduce
    every running sum, every per-frame array and therefore every final ratio.  The device kernels themselves: tests/test_gpu_eval.py."""
    z = np.load(os.path.join(GOLDEN, 'tracker_golden.npz'))
    names = [str(k) for k in z['final_names']]
    trk = CpuShimTracker(torch.device('cpu'), names)
    for bi in range(2):
        pred = {k[len('b%d_pred_' % bi):]: torch.from_numpy(z[k]).double() for k in z.files if k.startswith('b%d_pred_' % bi)}
        targ = {k[len('b%d_targ_' % bi):]: torch.from_numpy(z[k]) for k in z.files if k.startswith('b%d_targ_' % bi)}
        targ = {k: (v.double() if v.dtype != torch.bool else v) for k, v in targ.items()}
        trk.update_per_batch(pred, targ)
    sums, frames = trk.reduce()
    for i, k in enumerate(names):
        assert abs(sums[i].item() - z['sum_values'][i]) < 1e-7 * max(1.0, abs(z['sum_values'][i])), (k, sums[i].item(), z['sum_values'][i])
        assert np.allclose(frames[k].numpy(), z['frames_' + k], rtol=1e-7, atol=1e-9), k
    assert sums[trk.slots.index('num_vis_joints2D')].item() == z['num_vis_joints2D']
    assert sums[trk.slots.index('num_vis_joints2Dsamples')].item() == z['num_vis_joints2Dsamples']
    final, _ = trk.compute_final_metrics(num_verts=6890)
    n = int(z['num_total_test_data'])
    assert final['num_total_test_data'] == n
    ref = dict(zip(names, z['sum_values']))
    assert abs(final['PVE'] - ref['PVE'] / (n * 6890)) < 1e-12 and abs(final['MPJPE-PA_samples_min'] - ref['MPJPE-PA_samples_min'] / (n * 14)) < 1e-12
    assert abs(final['joints2D-L2E'] - ref['joints2D-L2E'] / z['num_vis_joints2D']) < 1e-9
    assert abs(final['joints2Dsamples-L2E'] - ref['joints2Dsamples-L2E'] / z['num_vis_joints2Dsamples']) < 1e-9
    assert abs(final['verts3D_sample_diversity'] - ref['verts3D_sample_diversity'] / (n * 3 * 6890)) < 1e-12
    assert abs(final['joints3D_sample_diversity'] - ref['joints3D_sample_diversity'] / (n * 3 * 17)) < 1e-12


def test_sharding_covers_every_item_once():
    for n, w in [(10, 1), (10, 2), (7, 4), (64, 8), (3, 8)]:
        got = sorted(i for r in range(w) for i in ev.shard_indices(n, r, w))
        assert got == list(range(n))


class FakeSet:
    """Deterministic per-index items; `batch` has the SyntheticEvalSet interface."""

    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def batch(self, indices, device):
        return {'index': torch.tensor(indices)}


def _fake(batch, salt, samples):
    outs = []
    for i in batch['index'].tolist():
        g = torch.Generator().manual_seed(1000 * salt + i)
        d = {'verts3D': torch.randn(40, 3, generator=g), 'tpose_verts3D': torch.randn(40, 3, generator=g),
             'joints3D': torch.randn(14, 3, generator=g), 'joints2D': torch.rand(17, 2, generator=g) * 256}
        if samples:
            d['verts3D_samples'] = torch.randn(5, 40, 3, generator=g)
            d['tpose_verts3D_samples'] = torch.randn(5, 40, 3, generator=g)
            d['joints3D_samples'] = torch.randn(5, 14, 3, generator=g)
            d['joints3D_coco_samples'] = torch.randn(5, 17, 3, generator=g)
            d['joints2Dsamples'] = torch.rand(5, 17, 2, generator=g) * 256
            # the model outputs the reference saves per frame (evaluate/evaluate_humaniflow.py:260-268)
            d.update(glob_rotmat=torch.randn(3, 3, generator=g), pose_rotmats_point_est=torch.randn(23, 3, 3, generator=g),
                     shape_mode=torch.randn(10, generator=g), cam_wp=torch.randn(3, generator=g))
        else:
            d['joints2D_vis'] = torch.rand(17, generator=g) > 0.3
        outs.append(d)
    return {k: torch.stack([o[k] for o in outs]) for k in outs[0]}


def _run_eval(rank, world, n, dist_mod):
    return ev.evaluate_humaniflow(FakeSet(n), 3, torch.device('cpu'), lambda b: _fake(b, 1, True), lambda b: _fake(b, 2, False),
                                  dist=dist_mod, rank=rank, world_size=world, tracker=CpuShimTracker(torch.device('cpu')))


class _CountingDist:
    """torch.distributed with every collective call counted (the evaluation's exchange step must be exactly two)."""
    COLLECTIVES = ('all_reduce', 'all_gather', 'all_gather_into_tensor', 'all_gather_object', 'broadcast', 'reduce', 'gather', 'scatter',
                   'all_to_all', 'reduce_scatter', 'barrier')

    def __init__(self):
        self.calls = []

    def __getattr__(self, name):
        fn = getattr(dist, name)
        if name in self.COLLECTIVES:
            def counted(*a, **k):
                self.calls.append(name)
                return fn(*a, **k)
            return counted
        return fn


def _worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    cd = _CountingDist()
    metrics, frames = _run_eval(rank, world, n, cd)
    assert cd.calls == ['all_reduce', 'all_gather_into_tensor'], cd.calls       # ~20 metrics, two collectives
    q.put((rank, metrics, {k: v.numpy() for k, v in frames.items()}))
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize('n', [11, 1])                 # 11: ragged, 6 + 5 items, last batches of different sizes; 1: rank 1 owns no item at all
def test_two_rank_gloo_eval_equals_single_process(n):
    single, frames1 = _run_eval(0, 1, n, None)
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, metrics, frames in res:
        assert metrics['num_total_test_data'] == n
        for k, v in single.items():
            assert abs(metrics[k] - v) < 1e-9 * max(1.0, abs(v)), (k, metrics[k], v)       # sums are order-independent up to fp64
        # every per-frame array -- metric frames AND the reference's pose / shape / cam per-frame outputs -- comes back in DATASET order
        # (evaluate/evaluate_humaniflow.py:260-286), i.e. equal to the single-process run, on every rank
        assert set(frames) == set(frames1) and {'index', 'pose_per_frame', 'shape_per_frame', 'cam_per_frame'} <= set(frames)
        assert np.array_equal(frames['index'], np.arange(n))
        assert frames['pose_per_frame'].shape == (n, 24, 3, 3) and frames['shape_per_frame'].shape == (n, 10) and frames['cam_per_frame'].shape == (n, 3)
        for k in frames1:
            assert frames[k].shape == frames1[k].shape
            assert np.allclose(frames[k], frames1[k].numpy(), atol=1e-6), k
        for i in range(n):                                               # the outputs of frame i are the ones the model produced for item i
            d = _fake({'index': torch.tensor([i])}, 1, True)
            assert np.array_equal(frames['pose_per_frame'][i, 0], d['glob_rotmat'][0].numpy())
            assert np.array_equal(frames['pose_per_frame'][i, 1:], d['pose_rotmats_point_est'][0].numpy())
            assert np.array_equal(frames['shape_per_frame'][i], d['shape_mode'][0].numpy()) and np.array_equal(frames['cam_per_frame'][i], d['cam_wp'][0].numpy())


def test_save_per_frame_outputs_writes_the_reference_files(tmp_path):
    _, frames = _run_eval(0, 1, 5, None)
    ev.save_per_frame_outputs(frames, str(tmp_path), fnames=['img_%02d.png' % i for i in range(5)])
    assert np.load(str(tmp_path / 'pose_per_frame.npy')).shape == (5, 24, 3, 3)
    assert np.load(str(tmp_path / 'shape_per_frame.npy')).shape == (5, 10) and np.load(str(tmp_path / 'cam_per_frame.npy')).shape == (5, 3)
    assert list(np.load(str(tmp_path / 'fname_per_frame.npy'))) == ['img_%02d.png' % i for i in range(5)]
    assert np.load(str(tmp_path / 'PVE_per_frame.npy')).shape == (5,) and np.load(str(tmp_path / 'MPJPE-PA_samples_min_per_frame.npy')).shape == (5,)


def test_sampling_helpers_match_reference():
    """quat_to_rotmat and the heat-map arg-max against the reference's own functions (golden); the error-sorted sample selection against
    a direct restatement (the reference function needs pytorch3d for its 180-degree flip and cannot run here)."""
    from humaniflow_amd import sampling_utils as su
    z = np.load(os.path.join(GOLDEN, 'sampling_golden.npz'))
    assert np.abs(su.quat_to_rotmat(torch.from_numpy(z['quat'])).numpy() - z['rotmat']).max() < 1e-14
    j2d, vis = su.convert_heatmaps_to_2Djoints_coordinates_torch(torch.from_numpy(z['heatmaps']))
    assert np.array_equal(j2d.numpy(), z['joints2D']) and np.array_equal(vis.numpy(), z['vis'])
    R = su.so3_uniform_sampling_torch(64, generator=torch.Generator().manual_seed(0))
    assert torch.allclose(R @ R.transpose(1, 2), torch.eye(3).expand(64, 3, 3), atol=1e-5) and torch.allclose(torch.linalg.det(R), torch.ones(64), atol=1e-5)
    g = torch.Generator().manual_seed(3)
    N, S = 7, 64
    verts, joints = torch.randn(N, 50, 3, generator=g), torch.randn(N, 90, 3, generator=g) * 0.3
    cam = torch.tensor([[0.9, 0.05, -0.02]])
    hm = torch.zeros(1, 17, S, S)
    tgt = torch.randint(8, S - 8, (17, 2), generator=g)
    for k in range(17):
        if k not in (2, 9):                                    # two invisible joints
            hm[0, k, tgt[k, 1], tgt[k, 0]] = 1.0
    out = su.joints2D_error_sorted_verts_sampling(verts, joints, hm, cam)
    jc = joints[:, ev.ALL_JOINTS_TO_COCO_MAP] * torch.tensor([1.0, -1.0, -1.0])
    p2 = (cam[:, None, :1] * (jc[..., :2] + cam[:, None, 1:]) + 1) * (S / 2.0)
    keep = [k for k in range(17) if k not in (2, 9)]
    worst = torch.stack([torch.norm(p2[n, keep] - tgt[keep].float(), dim=-1).max() for n in range(N)])
    assert torch.equal(out, verts[torch.argsort(worst)])


def test_bench_rank_plumbing_two_gloo_ranks():
    """bench.py's multi-rank scaffolding (env parsing, barrier + max-over-ranks timing, the metric all-gather, rank-0-only JSON line) launched the
    way the driver launches it -- `python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2` -- with the CPU stand-in step."""
    import json
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '1', '--dry-run-cpu']
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1                                           # rank 0 prints the one JSON line
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 4 and d['warmup'] == 1 and d['scaling'] == 'weak' and d['higher_is_better'] is True
    assert d['gathered_rank_sums'] == [4.0, 8.0]                     # rank r contributed (r+1) * ones(4)
    assert abs(d['value'] - 2 * 32 * 100 * 4 / (d['ms_per_step'] * 4 / 1e3)) < 1e-6 * d['value']     # whole-job aggregate over both ranks


def _reducer_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from humaniflow_amd.train import GradBucketReducer
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(40, 300), torch.nn.ReLU(), torch.nn.Linear(300, 300), torch.nn.ReLU(), torch.nn.Linear(300, 5),
                              torch.nn.Linear(5, 5))        # the last layer is NOT used below: its gradient never becomes ready
    params = list(net.parameters())
    red = GradBucketReducer(params, dist, bucket_mb=0.04)    # several buckets
    x = torch.randn(16, 40, generator=torch.Generator().manual_seed(100 + rank))
    for step in range(2):                                    # twice: the bookkeeping must reset
        for p in params:
            p.grad = None
        red.begin()
        net[:5](x).square().sum().backward()
        red.finish()
        with red.disabled():                                 # a backward that is not a step (measurement pass): no collective, no bookkeeping
            net[:3](x).square().sum().backward()
        assert not red.inflight and not any(red.launched)
    # the skip decision is shared: rank 1's non-finite flag makes BOTH ranks skip (one 4-byte all_reduce, same order on every rank)
    ok = red.all_finite(torch.tensor(rank == 0))
    assert bool(ok) is False
    assert bool(red.all_finite(torch.tensor(True))) is True
    # an armed backward that nobody finishes is caught at the next begin()
    for p in params:
        p.grad = None
    red.begin()
    net[:5](x).square().sum().backward()
    try:
        red.begin()
        caught = False
    except RuntimeError:
        caught = True
    assert caught
    red.finish()
    q.put((rank, len(red.buckets), [None if p.grad is None else p.grad.numpy().copy() for p in params]))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_bucket_reducer_two_gloo_ranks_equals_mean_of_local_gradients():
    """The training step's exchange (humaniflow_amd.train.GradBucketReducer): bucketed asynchronous all-reduce launched from gradient-ready
    hooks, incl. a parameter that receives no gradient; every rank ends with the mean of the per-rank gradients."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_reducer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(40, 300), torch.nn.ReLU(), torch.nn.Linear(300, 300), torch.nn.ReLU(), torch.nn.Linear(300, 5),
                              torch.nn.Linear(5, 5))
    local = []
    for rank in range(2):
        net.zero_grad()
        x = torch.randn(16, 40, generator=torch.Generator().manual_seed(100 + rank))
        net[:5](x).square().sum().backward()
        local.append([None if p.grad is None else p.grad.clone() for p in net.parameters()])
    assert res[0][1] >= 2, res[0][1]
    for rank, _, grads in res:
        for i, g in enumerate(grads):
            if local[0][i] is None:
                assert g is not None and np.abs(g).max() == 0          # unused parameter: exchanged as zeros so the ranks stay in step
            else:
                assert np.allclose(g, ((local[0][i] + local[1][i]) / 2).numpy(), rtol=1e-5, atol=1e-6)


def test_bench_rank_plumbing_eight_gloo_ranks():
    """The driver's 8-GPU launch line with the CPU stand-in step: 8 ranks rendezvous, the gathered tensor proves 8 ranks took part."""
    import json
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8', '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '3', '--warmup', '1', '--dry-run-cpu']
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 8 and d['rccl_ranks_seen'] == 8 and d['gathered_rank_sums'] == [4.0 * (r_ + 1) for r_ in range(8)]


def test_bench_gpus_2_without_a_launcher_starts_two_ranks():
    """`python bench.py --gpus 2 ...` with no WORLD_SIZE in the environment (the shape of the driver's 1-GPU command with another N) must not print
    an n_gpus = 1 line: bench.py starts the ranks itself and the line carries n_gpus == rccl_ranks_seen == 2."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--dry-run-cpu'],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['rccl_ranks_seen'] == 2 and d['gathered_rank_sums'] == [4.0, 8.0]


def test_bench_refuses_a_gpus_flag_that_disagrees_with_the_launcher():
    """--gpus 4 under a 1-rank environment (WORLD_SIZE=1 set by a launcher) exits non-zero instead of reporting another N than asked."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4', '--steps', '2', '--warmup', '1', '--dry-run-cpu'],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and b'--gpus 4' in r.stderr
    assert not [l for l in r.stdout.decode().splitlines() if l.startswith('{')]
