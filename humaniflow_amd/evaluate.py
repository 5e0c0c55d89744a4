"""Data-parallel evaluation of the hot path: the counterpart of evaluate/evaluate_humaniflow.py:18-286 restricted to
what the path produces (3D point-estimate and sample-minimum metrics), one process per GPU.

Sharding: images are independent, so rank r evaluates items r, r+W, r+2W, ... with B images per step; the only
exchange is the metric state -- running SUMS and counts (metrics/eval_metrics_tracker.py:53-93) -- packed into one fp64 vector and
all-reduced ONCE at the end (final metrics are ratios of sums, :437-494), plus an all-gather of the per-frame arrays.
`torch.distributed` backend "nccl" is RCCL over xGMI on ROCm; the same code runs on gloo (CPU tests).

The metric functions are torch restatements of utils/eval_utils.py:62-125 (Procrustes, scale+translation correction) and of the
tracker's update rules; they run on the device, so the 0.53 GB/batch device->host copy of the reference (:227-252) disappears.
"""
import torch

# utils/label_conversions.py:17-21
ALL_JOINTS_TO_COCO_MAP = [24, 26, 25, 28, 27, 16, 17, 18, 19, 20, 21, 1, 2, 4, 5, 7, 8]
ALL_JOINTS_TO_H36M_MAP = list(range(73, 90))
H36M_TO_J17 = [6, 5, 4, 1, 2, 3, 16, 15, 14, 11, 12, 13, 8, 10, 0, 7, 9]
H36M_TO_J14 = H36M_TO_J17[:14]

METRICS = ['PVE', 'PVE-SC', 'PVE-PA', 'PVE-T', 'PVE-T-SC', 'MPJPE', 'MPJPE-SC', 'MPJPE-PA',
           'PVE_samples_min', 'PVE-SC_samples_min', 'MPJPE_samples_min']


def procrustes_analysis_batch(S1, S2):
    """utils/eval_utils.py:62-104: similarity-align S1 (B,N,3) onto S2."""
    S1t, S2t = S1.transpose(1, 2), S2.transpose(1, 2)
    mu1, mu2 = S1t.mean(dim=2, keepdim=True), S2t.mean(dim=2, keepdim=True)
    X1, X2 = S1t - mu1, S2t - mu2
    var1 = (X1 ** 2).sum(dim=(1, 2))
    K = X1 @ X2.transpose(1, 2)
    U, s, Vh = torch.linalg.svd(K)
    V = Vh.transpose(1, 2)
    Z = torch.eye(3, dtype=S1.dtype, device=S1.device).repeat(S1.shape[0], 1, 1)
    Z[:, -1, -1] *= torch.sign(torch.linalg.det(U @ Vh))
    R = V @ Z @ U.transpose(1, 2)
    trace = torch.diagonal(R @ K, dim1=-2, dim2=-1).sum(-1)
    scale = (trace / var1)[:, None, None]
    t = mu2 - scale * (R @ mu1)
    return (scale * (R @ S1t) + t).transpose(1, 2)


def scale_and_translation_transform_batch(P, T):
    """utils/eval_utils.py:107-125."""
    P_mean = P.mean(dim=-2, keepdim=True)
    P_trans = P - P_mean
    P_scale = torch.sqrt((P_trans ** 2).sum(dim=(-2, -1), keepdim=True) / P.shape[-2])
    T_mean = T.mean(dim=-2, keepdim=True)
    T_scale = torch.sqrt(((T - T_mean) ** 2).sum(dim=(-2, -1), keepdim=True) / T.shape[-2])
    return P_trans / P_scale * T_scale + T_mean


class EvalMetricsTracker:
    """Running sums of metrics/eval_metrics_tracker.py:95-433 for the 3D metrics of the path, kept on the device in fp64."""

    def __init__(self, device, metrics=METRICS):
        self.metrics = list(metrics)
        self.sums = torch.zeros(len(self.metrics) + 1, dtype=torch.float64, device=device)      # [..., num_total_test_data]
        self.per_frame = {m: [] for m in self.metrics}

    def _add(self, name, err):               # err (B, P) per-point L2 errors
        if name in self.metrics:
            self.sums[self.metrics.index(name)] += err.double().sum()
            self.per_frame[name].append(err.mean(dim=-1))

    def update_per_batch(self, pred, target):
        """pred: verts3D (B,V,3), tpose_verts3D, joints3D (B,14,3), optional verts3D_samples (B,N,V,3), joints3D_samples (B,N,14,3);
        target: verts3D, tpose_verts3D, joints3D."""
        B = pred['verts3D'].shape[0]
        self.sums[-1] += B
        nrm = lambda a, b: torch.linalg.norm(a - b, dim=-1)
        self._add('PVE', nrm(pred['verts3D'], target['verts3D']))
        self._add('PVE-SC', nrm(scale_and_translation_transform_batch(pred['verts3D'], target['verts3D']), target['verts3D']))
        self._add('PVE-PA', nrm(procrustes_analysis_batch(pred['verts3D'], target['verts3D']), target['verts3D']))
        self._add('PVE-T', nrm(pred['tpose_verts3D'], target['tpose_verts3D']))
        self._add('PVE-T-SC', nrm(scale_and_translation_transform_batch(pred['tpose_verts3D'], target['tpose_verts3D']),
                                  target['tpose_verts3D']))
        self._add('MPJPE', nrm(pred['joints3D'], target['joints3D']))
        self._add('MPJPE-SC', nrm(scale_and_translation_transform_batch(pred['joints3D'], target['joints3D']), target['joints3D']))
        self._add('MPJPE-PA', nrm(procrustes_analysis_batch(pred['joints3D'], target['joints3D']), target['joints3D']))
        if 'verts3D_samples' in pred:
            ar = torch.arange(B, device=self.sums.device)
            tv = target['verts3D'][:, None]
            e = nrm(pred['verts3D_samples'], tv)                                                     # (B, N, V)
            self._add('PVE_samples_min', e[ar, e.mean(-1).argmin(-1)])
            e = nrm(scale_and_translation_transform_batch(pred['verts3D_samples'], tv.expand_as(pred['verts3D_samples'])), tv)
            self._add('PVE-SC_samples_min', e[ar, e.mean(-1).argmin(-1)])
            e = nrm(pred['joints3D_samples'], target['joints3D'][:, None])
            self._add('MPJPE_samples_min', e[ar, e.mean(-1).argmin(-1)])

    def reduce(self, dist=None):
        """The ONE exchange step: all-reduce(SUM) of the packed sums; all-gather of per-frame arrays (padded to equal length)."""
        sums = self.sums.clone()
        frames = {m: (torch.cat(v) if v else torch.zeros(0, device=self.sums.device)) for m, v in self.per_frame.items()}
        if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(sums, op=dist.ReduceOp.SUM)
            W = dist.get_world_size()
            n_local = torch.tensor([len(next(iter(frames.values())))], device=sums.device)
            counts = [torch.zeros_like(n_local) for _ in range(W)]
            dist.all_gather(counts, n_local)
            nmax = int(max(c.item() for c in counts))
            for m in list(frames):
                pad = torch.zeros(nmax, dtype=frames[m].dtype, device=sums.device)
                pad[:len(frames[m])] = frames[m]
                got = [torch.zeros_like(pad) for _ in range(W)]
                dist.all_gather(got, pad)
                frames[m] = torch.cat([g[:int(c.item())] for g, c in zip(got, counts)])
        return sums, frames

    def compute_final_metrics(self, dist=None, num_verts=6890, num_joints=14):
        """metrics/eval_metrics_tracker.py:437-494: ratios of sums, in millimetres."""
        sums, frames = self.reduce(dist)
        n = sums[-1].item()
        out = {}
        for i, m in enumerate(self.metrics):
            pts = num_joints if m.startswith('MPJPE') else num_verts
            out[m] = sums[i].item() / (n * pts) * 1000.0
        out['num_total_test_data'] = int(n)
        return out, frames


def shard_indices(num_items, rank, world_size):
    return list(range(rank, num_items, world_size))


class SyntheticEvalSet:
    """3DPW-shaped synthetic items (data/pw3d_eval_dataset.py:96-103 keys used by the path): proxy input, axis-angle pose (72),
    shape (10).  Deterministic per index, so any sharding sees the same items."""

    def __init__(self, num_items, size=256, seed=0):
        self.num_items, self.size, self.seed = num_items, size, seed

    def __len__(self):
        return self.num_items

    def batch(self, indices, device):
        from .synthetic import synthetic_proxy_input
        xs, poses, shapes = [], [], []
        for i in indices:
            g = torch.Generator().manual_seed(self.seed * 1000003 + i)
            xs.append(synthetic_proxy_input(1, self.size, seed=self.seed * 1000003 + i))
            poses.append(torch.randn(72, generator=g) * 0.3)
            shapes.append(torch.randn(10, generator=g) * 1.25)
        return {'input': torch.cat(xs).to(device), 'pose': torch.stack(poses).to(device), 'shape': torch.stack(shapes).to(device),
                'index': torch.tensor(indices)}


@torch.no_grad()
def predict_batch(model, smpl, batch, num_samples, noise=None):
    """evaluate/evaluate_humaniflow.py:113-192 on the device: model -> SMPL point estimate + T-pose -> SMPL on the B*N samples."""
    N = num_samples
    pred = model(batch['input'], num_samples=N, noise=noise)
    g = pred['glob_rotmat']
    pe = smpl(body_pose=pred['pose_rotmats_point_est'], global_orient=g[:, None], betas=pred['shape_mode'], pose2rot=False)
    out = {'verts3D': pe.vertices, 'joints3D': pe.joints[:, ALL_JOINTS_TO_H36M_MAP][:, H36M_TO_J14],
           'tpose_verts3D': smpl.tpose(pred['shape_mode'])}
    if N > 0:
        B = g.shape[0]
        sa = smpl(body_pose=pred['pose_rotmats_samples'].reshape(-1, 23, 3, 3),
                  global_orient=g[:, None].expand(-1, N, -1, -1).reshape(-1, 1, 3, 3),
                  betas=pred['shape_samples'].reshape(-1, pred['shape_samples'].shape[-1]), pose2rot=False)
        out['verts3D_samples'] = sa.vertices.reshape(B, N, -1, 3)
        out['joints3D_samples'] = sa.joints[:, ALL_JOINTS_TO_H36M_MAP][:, H36M_TO_J14].reshape(B, N, 14, 3)
    return out


@torch.no_grad()
def target_batch(smpl, batch):
    """evaluate/evaluate_humaniflow.py:83-110 (single body model: the synthetic set has no gender split)."""
    t = smpl(body_pose=batch['pose'][:, 3:], global_orient=batch['pose'][:, :3], betas=batch['shape'])
    return {'verts3D': t.vertices, 'joints3D': t.joints[:, ALL_JOINTS_TO_H36M_MAP][:, H36M_TO_J14],
            'tpose_verts3D': smpl.tpose(batch['shape'])}


def evaluate_humaniflow(dataset, batch_size, device, predict_fn, target_fn, dist=None, rank=0, world_size=1):
    """Sharded evaluation loop.  predict_fn(batch) / target_fn(batch) return the dicts EvalMetricsTracker consumes."""
    tracker = EvalMetricsTracker(device)
    mine = shard_indices(len(dataset), rank, world_size)
    for s in range(0, len(mine), batch_size):
        batch = dataset.batch(mine[s:s + batch_size], device)
        tracker.update_per_batch(predict_fn(batch), target_fn(batch))
    return tracker.compute_final_metrics(dist)
