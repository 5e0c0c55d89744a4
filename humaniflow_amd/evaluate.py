"""Data-parallel evaluation of the hot path: the counterpart of evaluate/evaluate_humaniflow.py:18-286 restricted to
what the path produces (3D point-estimate and sample-minimum metrics), one process per GPU.

Sharding: images are independent, so rank r evaluates items r, r+W, r+2W, ... with B images per step; the only
exchange is the metric state -- running SUMS and counts (metrics/eval_metrics_tracker.py:53-93) -- packed into one fp64 vector and
all-reduced ONCE at the end (final metrics are ratios of sums, :437-494), plus an all-gather of the per-frame arrays.
`torch.distributed` backend "nccl" is RCCL over xGMI on ROCm; the same code runs on gloo (CPU tests).

The point-set errors (plain / scale+translation-corrected / Procrustes-aligned, utils/eval_utils.py:62-125) and the sample diversity
run in the fused HIP epilogue of the library (csrc/eval_kernels.hip), so the 0.53 GB/batch device->host copy of the reference
(:227-252) disappears; the tracker keeps the reference's running sums (metrics/eval_metrics_tracker.py:95-433) on the device in fp64.
The numpy/torch restatement of those metric functions lives in oracle/metrics.py (checker only).
"""
import torch

# utils/label_conversions.py:17-21
ALL_JOINTS_TO_COCO_MAP = [24, 26, 25, 28, 27, 16, 17, 18, 19, 20, 21, 1, 2, 4, 5, 7, 8]
ALL_JOINTS_TO_H36M_MAP = list(range(73, 90))
H36M_TO_J17 = [6, 5, 4, 1, 2, 3, 16, 15, 14, 11, 12, 13, 8, 10, 0, 7, 9]
H36M_TO_J14 = H36M_TO_J17[:14]

METRICS = ['PVE', 'PVE-SC', 'PVE-PA', 'PVE-T', 'PVE-T-SC', 'MPJPE', 'MPJPE-SC', 'MPJPE-PA',
           'PVE_samples_min', 'PVE-SC_samples_min', 'PVE-PA_samples_min', 'PVE-T_samples_min', 'PVE-T-SC_samples_min',
           'MPJPE_samples_min', 'MPJPE-SC_samples_min', 'MPJPE-PA_samples_min',
           'joints2D-L2E', 'joints2Dsamples-L2E', 'verts3D_sample_diversity', 'joints3D_sample_diversity']
# not built: silhouette IOUs (need the pytorch3d renderer) and the input_joints2D / (in)visible-joint variants (need HRNet heat-maps)


_INDEX_CACHE = {}


def joint_index(name, device):
    """Device-resident index tensors of the joint maps (built once per device: indexing with a Python list would upload the indices on
    every call, which also cannot be captured into a HIP graph).  'h36m_j14' = ALL_JOINTS_TO_H36M_MAP then H36M_TO_J14, composed."""
    key = (name, str(device))
    t = _INDEX_CACHE.get(key)
    if t is None:
        idx = {'coco': ALL_JOINTS_TO_COCO_MAP, 'h36m_j14': [ALL_JOINTS_TO_H36M_MAP[i] for i in H36M_TO_J14]}[name]
        t = _INDEX_CACHE[key] = torch.tensor(idx, dtype=torch.long, device=device)
    return t


def select_joints(joints, name):
    return joints.index_select(1, joint_index(name, joints.device))


def orthographic_project_torch(points3D, cam_params):
    """utils/cam_utils.py:9-17: s * (xy + t)."""
    return cam_params[:, None, 0:1] * (points3D[:, :, :2] + cam_params[:, None, 1:])


def undo_keypoint_normalisation(normalised_keypoints, img_wh):
    """utils/joints2d_utils.py:5-10."""
    return (normalised_keypoints + 1) * (img_wh / 2.0)


def flip_about_x(points):
    """aa_rotate_translate_points_pytorch3d(points, axes=[1,0,0], angles=pi, translations=0) (utils/rigid_transform_utils.py:67-83,
    call sites evaluate_humaniflow.py:141-143,197-203): a 180-degree rotation about x, i.e. (x, -y, -z)."""
    return torch.stack([points[..., 0], -points[..., 1], -points[..., 2]], dim=-1)


class EvalMetricsTracker:
    """Running sums of metrics/eval_metrics_tracker.py:95-433 for the metrics the path can feed, kept on the device in fp64.
    Final metrics are ratios of sums (:437-494), in the reference's units (metres / pixels)."""

    COUNTS = ['num_total_test_data', 'num_vis_joints2D', 'num_vis_joints2Dsamples']

    def __init__(self, device, metrics=METRICS, smpl=None):
        """smpl: the body model of the predictions (humaniflow_amd.SMPL).  With it, batches that carry `tpose_betas` instead of T-pose vertex
        tensors (predict_batch / target_batch with materialise_tpose=False) get their PVE-T metrics from hmf_smpl_tpose_errors."""
        self.smpl = smpl
        self.metrics = list(metrics)
        self.slots = self.metrics + self.COUNTS
        self.sums = torch.zeros(len(self.slots), dtype=torch.float64, device=device)
        self.per_frame = {m: [] for m in self.metrics}
        self.frame_index, self.frame_outputs = [], []             # record_frames(): dataset indices and the reference's per-frame model outputs
        self._pending, self._consts = None, {}

    def record_frames(self, index, pred=None):
        """Dataset indices of the frames of the batch just scored and, with `pred` carrying glob_rotmat / pose_rotmats_point_est / shape_mode /
        cam_wp, the per-frame outputs evaluate/evaluate_humaniflow.py:260-268 saves: pose (24,3,3) = [global | 23 body rotations], shape (nb),
        camera (3) -- one fixed-width float32 row per frame, kept on the device.  reduce() returns every per-frame array in DATASET order through
        these indices, whatever the sharding."""
        dev = self.sums.device
        idx = torch.as_tensor(index, dtype=torch.float32).to(dev)     # exact below 2**24 frames; rides in the float32 all-gather
        assert idx.numel() == 0 or float(idx.max()) < 2 ** 24
        self.frame_index.append(idx)
        if pred is not None and all(k in pred for k in ('glob_rotmat', 'pose_rotmats_point_est', 'shape_mode', 'cam_wp')):
            B = idx.numel()
            self.frame_outputs.append(torch.cat([pred['glob_rotmat'].reshape(B, 9), pred['pose_rotmats_point_est'].reshape(B, -1),
                                                 pred['shape_mode'].reshape(B, -1), pred['cam_wp'].reshape(B, 3)], dim=1).float())

    def _acc(self, name, value):
        if torch.is_tensor(value) and self._pending is not None:          # inside update_per_batch: one index_add_ for the whole batch (_flush)
            self._pending[1].append((self.slots.index(name), value))
        else:
            self.sums[self.slots.index(name)] += value

    def _flush(self):
        """Adds everything update_per_batch queued to the running sums in a handful of launches: the ~20 per-frame error vectors are stacked, summed
        in fp64 and weighted in one pass each, the scalar terms stacked, and ONE index_add_ updates the sums (per metric the reference's
        `sum += err.sum()` is four small launches; 20 metrics per batch were ~100 of the eval step's launches)."""
        frames, scalars = self._pending
        self._pending = None
        idx, vals = [], []
        if frames:
            F = torch.stack([f.reshape(-1).to(torch.float32) for _, f, _ in frames]).double().sum(dim=1)
            key = ('w', tuple(n for _, _, n in frames), str(F.device))
            w = self._consts.get(key)
            if w is None:
                w = self._consts[key] = torch.tensor([float(n) for _, _, n in frames], dtype=torch.float64, device=F.device)
            idx += [i for i, _, _ in frames]
            vals.append(F * w)
        if scalars:
            idx += [i for i, _ in scalars]
            vals.append(torch.stack([v.reshape(()).double() for _, v in scalars]))
        if idx:
            key = ('i', tuple(idx), str(self.sums.device))
            it = self._consts.get(key)
            if it is None:
                it = self._consts[key] = torch.tensor(idx, dtype=torch.long, device=self.sums.device)
            self.sums.index_add_(0, it, vals[0] if len(vals) == 1 else torch.cat(vals))

    def _add(self, name, err, frame=None):               # err (B, ...) per-point errors
        if name in self.metrics:
            self._acc(name, err.double().sum())
            self.per_frame[name].append(err.flatten(1).mean(dim=-1) if frame is None else frame)

    def update_per_batch(self, pred, target):
        """pred: verts3D (B,V,3), tpose_verts3D, joints3D (B,14,3) [+ joints2D (B,17,2)] and optionally verts3D_samples (B,N,V,3),
        tpose_verts3D_samples, joints3D_samples (B,N,14,3), joints2Dsamples (B,N,17,2), joints3D_coco_samples (B,N,17,3);
        target: verts3D, tpose_verts3D, joints3D [+ joints2D, joints2D_vis].  Device tensors only: the point-set errors run in the fused
        HIP epilogue (hmf_pointset_errors / hmf_sample_diversity); there is no host path."""
        from . import _lib
        _lib.require_cuda(pred['verts3D'], "pred['verts3D']")
        if 'tpose_verts3D' not in pred or 'tpose_verts3D' not in target:
            # batches built with materialise_tpose=False carry shape vectors instead of T-pose meshes: both sides must, and the tracker needs the body model
            if ('tpose_verts3D' in pred) != ('tpose_verts3D' in target) or 'tpose_betas' not in pred or 'tpose_betas' not in target:
                raise ValueError("EvalMetricsTracker.update_per_batch: pred and target must BOTH carry 'tpose_verts3D', or BOTH carry 'tpose_betas' "
                                 '(predict_batch / target_batch with the same materialise_tpose)')
            if self.smpl is None:
                raise ValueError("EvalMetricsTracker.update_per_batch: the batch carries 'tpose_betas' instead of T-pose vertices; construct the tracker "
                                 'with smpl=<the humaniflow_amd.SMPL of the predictions> so that hmf_smpl_tpose_errors can score them')
        if self.fused and self._fused_ok(pred, target):
            return self._update_fused(pred, target)
        self._pending = ([], [])
        try:
            self._update(pred, target)
        finally:
            if self._pending is not None:
                self._flush()

    # ---- the default device path: point-set kernels, then ONE launch for everything else (csrc/eval_kernels.hip k_eval_tracker_update) ----
    fused = True           # False: the same rules as torch operators (`_update`; also what the CPU-shim tracker of the tests drives)

    def _fused_ok(self, pred, target):
        ts = [v for v in list(pred.values()) + list(target.values()) if torch.is_tensor(v)]
        if not all(t.is_cuda for t in ts):
            return False
        if ('tpose_verts3D' in pred) != ('tpose_verts3D' in target):
            return False
        if 'tpose_verts3D' not in pred and not ('tpose_betas' in pred and 'tpose_betas' in target and self.smpl is not None):
            return False
        j2 = [pred[k] for k in ('joints2D', 'joints2Dsamples') if k in pred] + ([target['joints2D']] if 'joints2D' in target else [])
        if any(t.shape[-2] != 17 for t in j2):
            return False
        if 'joints3D_coco_samples' in pred and pred['joints3D_coco_samples'].shape[-2] != 17:
            return False
        return all(k in pred for k in ('verts3D', 'joints3D')) and all(k in target for k in ('verts3D', 'joints3D'))

    def _update_fused(self, pred, target):
        from . import _lib
        lib = _lib.load()
        dev = self.sums.device
        f32 = lambda t: t.float().contiguous()
        tv, tj = target['verts3D'], target['joints3D']
        B, V, J = pred['verts3D'].shape[0], tv.shape[1], tj.shape[1]
        self.num_verts = V
        need_pa = any(m in self.metrics for m in ('PVE-PA', 'PVE-PA_samples_min'))
        by_betas = 'tpose_verts3D' not in pred
        keep = []                                                     # tensors the launch reads: alive until it is issued
        eb = _lib.EvalBatch()
        eb.B, eb.num_verts, eb.num_joints = B, V, J

        def put(field, t):
            keep.append(t)
            setattr(eb, field, t.data_ptr())
        put('e_verts', self._pointset(pred['verts3D'], tv, need_pa))
        packed = None                                                 # T-pose errors of the sample rows and the point-estimate rows in one launch
        if by_betas and 'tpose_betas_packed' in pred and 'verts3D_samples' in pred:
            Ns = pred['verts3D_samples'].shape[1]
            packed = self.smpl.tpose_errors_packed(pred['tpose_betas_packed'], target['tpose_betas'], B, Ns)
            put('e_tpose', packed[B * Ns:])
        else:
            put('e_tpose', self.smpl.tpose_errors(pred['tpose_betas'], target['tpose_betas']) if by_betas
                else self._pointset(pred['tpose_verts3D'], target['tpose_verts3D'], False))
        put('e_joints', self._pointset(pred['joints3D'], tj, True))
        N = 0
        if 'verts3D_samples' in pred:
            vs = pred['verts3D_samples']
            N = self.num_samples = vs.shape[1]
            put('es_verts', self._pointset(vs, tv, need_pa))
            if packed is not None:
                put('es_tpose', packed[:B * N])
            elif by_betas and 'tpose_betas_samples' in pred:
                put('es_tpose', self.smpl.tpose_errors(pred['tpose_betas_samples'], target['tpose_betas']))
            elif 'tpose_verts3D_samples' in pred:
                put('es_tpose', self._pointset(pred['tpose_verts3D_samples'], target['tpose_verts3D'], False))
            put('es_joints', self._pointset(pred['joints3D_samples'], tj, True))
            if 'verts3D_sample_diversity' in self.metrics:
                put('div_verts', self._diversity_partial(vs))
                eb.chunks_v = keep[-1].shape[1]
        if 'joints3D_coco_samples' in pred and 'joints3D_sample_diversity' in self.metrics:
            jc = pred['joints3D_coco_samples']
            N = N or jc.shape[1]
            put('div_joints', self._diversity_partial(jc))
            eb.chunks_j = keep[-1].shape[1]
        if 'joints2D' in target:
            put('j2d_target', f32(target['joints2D']))
            if 'joints2D' in pred and 'joints2D-L2E' in self.metrics:
                put('j2d_pred', f32(pred['joints2D']))
            if 'joints2Dsamples' in pred and 'joints2Dsamples-L2E' in self.metrics:
                put('j2d_samples', f32(pred['joints2Dsamples']))
                N = N or pred['joints2Dsamples'].shape[1]
            if 'joints2D_vis' in target:
                vis = target['joints2D_vis'].contiguous()
                put('j2d_vis', vis.view(torch.uint8) if vis.dtype == torch.bool else vis.to(torch.uint8))      # bool is one byte: no conversion launch
        eb.N = N
        names = METRICS + self.COUNTS
        for i, nm in enumerate(names):
            eb.slot[i] = self.slots.index(nm) if nm in self.slots else -1
        frames = torch.empty(len(METRICS), B, dtype=torch.float32, device=dev)
        put('scratch', torch.empty(int(lib.hmf_eval_tracker_scratch_bytes(B)), dtype=torch.uint8, device=dev))
        with torch.cuda.device(dev):
            _lib.check(lib.hmf_eval_tracker_update(eb, self.sums.data_ptr(), frames.data_ptr(), _lib.current_stream()), 'hmf_eval_tracker_update')
        have = {'pe': True, 's': bool(eb.es_verts), 'st': bool(eb.es_tpose), 'dv': bool(eb.div_verts), 'dj': bool(eb.div_joints), 'j2': bool(eb.j2d_pred), 'j2s': bool(eb.j2d_samples)}
        group = ['pe'] * 8 + ['s', 's', 's', 'st', 'st', 's', 's', 's', 'j2', 'j2s', 'dv', 'dj']
        for i, nm in enumerate(METRICS):
            if nm in self.metrics and have[group[i]]:
                self.per_frame[nm].append(frames[i])

    def _diversity_partial(self, S):
        """(B, chunks) float32 partial sums of hmf_sample_diversity."""
        from . import _lib
        lib = _lib.load()
        B, N, npts = S.shape[:3]
        Sc = S.float().contiguous()
        part = torch.empty(B, int(lib.hmf_sample_diversity_chunks(npts)), device=S.device, dtype=torch.float32)
        with torch.cuda.device(S.device):
            _lib.check(lib.hmf_sample_diversity(_lib.dptr(Sc), B, N, npts, _lib.dptr(part), _lib.current_stream()), 'hmf_sample_diversity')
        return part

    def _pointset(self, P, T, want_pa):
        """P (..., npts, 3) vs T (Bt, npts, 3) with P's leading dims = (Bt,) or (Bt, N): -> (..., 3) mean errors [plain, SC, PA]."""
        from . import _lib
        lib = _lib.load()
        lead, npts = P.shape[:-2], P.shape[-2]
        npairs = 1
        for d in lead:
            npairs *= d
        Pc, Tc = P.float().contiguous(), T.float().contiguous()
        out = torch.empty(npairs, 3, device=P.device, dtype=torch.float32)
        with torch.cuda.device(P.device):                  # the launch stream must be the one of the tensors' device, not of device 0
            _lib.check(lib.hmf_pointset_errors(_lib.dptr(Pc), _lib.dptr(Tc), npairs, npairs // Tc.shape[0], npts, int(want_pa), _lib.dptr(out),
                                               _lib.current_stream()), 'hmf_pointset_errors')
        return out.reshape(*lead, 3)

    def _diversity(self, S):
        return self._diversity_partial(S).double().sum(dim=1)             # (B,) sum over n, p of the distances

    def _addf(self, name, frame, npts):
        """frame (B,) per-frame MEAN error over npts points: running sum += sum_b frame * npts."""
        if name in self.metrics:
            if self._pending is not None:
                self._pending[0].append((self.slots.index(name), frame, npts))
            else:
                self._acc(name, frame.double().sum() * npts)
            self.per_frame[name].append(frame)

    def _update(self, pred, target):
        B, self.num_verts = pred['verts3D'].shape[:2]
        self._acc('num_total_test_data', B)
        tv, tj = target['verts3D'], target['joints3D']
        by_betas = 'tpose_verts3D' not in pred                             # T-pose metrics straight from the shape vectors (needs self.smpl)
        tt = target['tpose_betas'] if by_betas else target['tpose_verts3D']
        V, J = tv.shape[1], tj.shape[1]
        need_pa = any(m in self.metrics for m in ('PVE-PA', 'PVE-PA_samples_min'))
        e = self._pointset(pred['verts3D'], tv, need_pa)
        self._addf('PVE', e[:, 0], V); self._addf('PVE-SC', e[:, 1], V); self._addf('PVE-PA', e[:, 2], V)
        e = self.smpl.tpose_errors(pred['tpose_betas'], tt) if by_betas else self._pointset(pred['tpose_verts3D'], tt, False)
        self._addf('PVE-T', e[:, 0], V); self._addf('PVE-T-SC', e[:, 1], V)
        e = self._pointset(pred['joints3D'], tj, True)
        self._addf('MPJPE', e[:, 0], J); self._addf('MPJPE-SC', e[:, 1], J); self._addf('MPJPE-PA', e[:, 2], J)
        if 'verts3D_samples' in pred:
            vs, js = pred['verts3D_samples'], pred['joints3D_samples']
            self.num_samples = vs.shape[1]
            e = self._pointset(vs, tv, need_pa).min(dim=1).values            # per metric: the sample with the smallest mean error
            self._addf('PVE_samples_min', e[:, 0], V); self._addf('PVE-SC_samples_min', e[:, 1], V)
            self._addf('PVE-PA_samples_min', e[:, 2], V)
            if by_betas and 'tpose_betas_samples' in pred:
                e = self.smpl.tpose_errors(pred['tpose_betas_samples'], tt).min(dim=1).values
                self._addf('PVE-T_samples_min', e[:, 0], V); self._addf('PVE-T-SC_samples_min', e[:, 1], V)
            elif 'tpose_verts3D_samples' in pred:
                e = self._pointset(pred['tpose_verts3D_samples'], tt, False).min(dim=1).values
                self._addf('PVE-T_samples_min', e[:, 0], V); self._addf('PVE-T-SC_samples_min', e[:, 1], V)
            e = self._pointset(js, tj, True).min(dim=1).values
            self._addf('MPJPE_samples_min', e[:, 0], J); self._addf('MPJPE-SC_samples_min', e[:, 1], J)
            self._addf('MPJPE-PA_samples_min', e[:, 2], J)
            if 'verts3D_sample_diversity' in self.metrics:
                d = self._diversity(vs)
                self._acc('verts3D_sample_diversity', d.sum())
                self.per_frame['verts3D_sample_diversity'].append((d / (vs.shape[1] * V)).float())
        if 'joints3D_coco_samples' in pred and 'joints3D_sample_diversity' in self.metrics:
            jc = pred['joints3D_coco_samples']
            d = self._diversity(jc)
            self._acc('joints3D_sample_diversity', d.sum())
            self.per_frame['joints3D_sample_diversity'].append((d / (jc.shape[1] * jc.shape[2])).float())
        self._update_2d(pred, target)

    def _update_2d(self, pred, target):
        """joints2D-L2E / joints2Dsamples-L2E (eval_metrics_tracker.py:284-359): 17 points per body -- a handful of small device ops."""
        nrm = lambda a, b: torch.linalg.norm(a - b, dim=-1)
        if 'joints2D' in pred and 'joints2D' in target and 'joints2D-L2E' in self.metrics:
            e = nrm(pred['joints2D'], target['joints2D'])                                         # (B, 17)
            if 'joints2D_vis' in target:
                vis = target['joints2D_vis'].to(e.dtype)
                e = e * vis
                self._acc('num_vis_joints2D', vis.double().sum())
                frame = e.sum(-1) / vis.sum(-1)
            else:
                self._acc('num_vis_joints2D', e.numel())
                frame = e.mean(-1)
            self._add('joints2D-L2E', e, frame)
        if 'joints2Dsamples' in pred and 'joints2D' in target and 'joints2Dsamples-L2E' in self.metrics:
            ps = pred['joints2Dsamples']
            e = nrm(ps, target['joints2D'][:, None])                                              # (B, N, 17)
            if 'joints2D_vis' in target:
                vis = target['joints2D_vis'].to(e.dtype)
                e = e * vis[:, None]
                self._acc('num_vis_joints2Dsamples', vis.double().sum() * ps.shape[1])
                frame = e.sum(dim=(1, 2)) / (vis.sum(-1) * ps.shape[1])
            else:
                self._acc('num_vis_joints2Dsamples', e.numel())
                frame = e.mean(dim=(1, 2))
            self._add('joints2Dsamples-L2E', e, frame)

    def reduce(self, dist=None):
        """The ONE exchange step of the path, as exactly two collectives whatever the number of metrics:
          1. all_reduce(SUM) of one float64 vector = the packed metric sums | one frame-count slot per rank (only the own slot is non-zero),
             so every rank learns every rank's frame count in the same message that sums the metrics;
          2. all_gather of ONE (rows, nmax) float32 tensor holding every per-frame array -- the metric frames and, when record_frames() was fed,
             the frames' dataset indices and the reference's per-frame outputs (pose 216 | shape nb | cam 3) -- padded to the largest rank's count.
        Messages are bytes to KB: latency-bound, so the count of collectives is what matters on the xGMI ring.
        With recorded indices every returned array is in DATASET order (the reference's single-process order, evaluate_humaniflow.py:260-286);
        without them it is rank-major (rank 0's frames, then rank 1's, ...)."""
        sums = self.sums.clone()
        dev = sums.device
        frames = {m: (torch.cat(v) if v else torch.zeros(0, device=dev)) for m, v in self.per_frame.items()}
        index = torch.cat(self.frame_index) if self.frame_index else None
        outs = torch.cat(self.frame_outputs) if self.frame_outputs else None
        if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
            W, rank = dist.get_world_size(), dist.get_rank()
            names = [m for m in frames]                                       # same order on every rank (dict order of the constructor)
            n_local = max([len(v) for v in frames.values()] + [len(index) if index is not None else 0])
            present = torch.tensor([float(len(frames[m]) > 0) for m in names], dtype=sums.dtype, device=dev)
            # per-frame arrays of one rank must have one length.  A rank that breaks this must not raise BEFORE the collective (the others would
            # block in all_reduce for ever): it reports the fault in the packed vector and every rank raises together after collective 1.
            bad = [m for m in names if len(frames[m]) not in (0, n_local)]
            if index is not None and len(index) != n_local:
                bad.append('frame index')
            if outs is not None and (index is None or len(outs) != n_local):
                bad.append('per-frame outputs')
            TAIL = 5                                                            # [fault, num_verts, num_samples, ranks with indices, output row width]
            packed = torch.zeros(sums.numel() + len(names) + W + TAIL, dtype=sums.dtype, device=dev)
            packed[-5] = float(len(bad) > 0)
            packed[:sums.numel()] = sums
            packed[sums.numel():sums.numel() + len(names)] = present           # a metric present on ANY rank is gathered (a rank may own zero frames)
            packed[sums.numel() + len(names) + rank] = n_local
            if n_local > 0:                                                     # mesh size / sample count of the ranks that saw data (a rank that owns
                packed[-4] = getattr(self, 'num_verts', 0)                      # no item learns them here: the final ratios divide by them)
                packed[-3] = getattr(self, 'num_samples', 0)
                packed[-2] = float(index is not None)
                packed[-1] = float(outs.shape[1]) if outs is not None else 0.0
            dist.all_reduce(packed, op=dist.ReduceOp.SUM)                       # collective 1
            sums = packed[:sums.numel()].clone()
            host = packed[sums.numel():].tolist()                               # one D2H of n_metrics + W + TAIL numbers
            if host[-5] > 0:
                raise RuntimeError('EvalMetricsTracker.reduce: per-frame arrays of one rank must have one length (%d rank(s) report a mismatch%s)'
                                   % (int(host[-5]), '; here: ' + ', '.join('%s=%d' % (m, len(frames[m])) if m in frames else m for m in bad)
                                      + ' vs %d' % n_local if bad else ''))
            have, counts = [h > 0 for h in host[:len(names)]], [int(c) for c in host[len(names):-TAIL]]
            nmax = max(counts)
            with_data = sum(c > 0 for c in counts)
            all_idx, ow = False, 0
            if with_data:
                nv, ns = int(round(host[-4] / with_data)), int(round(host[-3] / with_data))
                if nv > 0:
                    self.num_verts = nv
                if ns > 0:
                    self.num_samples = ns
                all_idx = int(round(host[-2])) == with_data                     # every rank that saw data recorded its frames' indices
                ow = int(round(host[-1] / with_data)) if all_idx else 0
                if ow and abs(host[-1] - ow * with_data) > 0.5:
                    raise RuntimeError('EvalMetricsTracker.reduce: per-frame outputs recorded on some ranks only')
            sel = [m for m, h in zip(names, have) if h]
            rows = len(sel) + (1 if all_idx else 0) + ow
            if rows and nmax > 0:
                mine = torch.zeros((rows, nmax), dtype=torch.float32, device=dev)
                for i, m in enumerate(sel):
                    if len(frames[m]):
                        mine[i, :n_local] = frames[m].float()
                if all_idx and n_local:
                    mine[len(sel), :n_local] = index
                    if ow:
                        mine[len(sel) + 1:, :n_local] = outs.t()
                got = torch.empty((W, rows, nmax), dtype=torch.float32, device=dev)
                dist.all_gather_into_tensor(got.view(-1), mine.view(-1))       # collective 2
                for i, m in enumerate(sel):
                    frames[m] = torch.cat([got[r, i, :counts[r]] for r in range(W)])
                if all_idx:
                    index = torch.cat([got[r, len(sel), :counts[r]] for r in range(W)])
                    outs = torch.cat([got[r, len(sel) + 1:, :counts[r]].t() for r in range(W)]) if ow else None
                else:
                    index = outs = None
            else:
                index = outs = None
        if index is not None and index.numel():
            order = torch.argsort(index, stable=True)                          # the inverse of the sharding permutation
            n = index.numel()
            frames = {m: (v[order] if v.numel() == n else v) for m, v in frames.items()}
            frames['index'] = index[order].long()
            if outs is not None and outs.shape[0] == n:
                outs = outs[order]
                nb = outs.shape[1] - 219
                frames['pose_per_frame'] = outs[:, :216].reshape(n, 24, 3, 3)
                frames['shape_per_frame'] = outs[:, 216:216 + nb]
                frames['cam_per_frame'] = outs[:, 216 + nb:]
        return sums, frames

    def compute_final_metrics(self, dist=None, num_verts=None, num_joints=14, num_samples=None):
        """metrics/eval_metrics_tracker.py:437-494: ratios of sums, in the reference's units (metres, pixels)."""
        sums, frames = self.reduce(dist)
        get = lambda k: sums[self.slots.index(k)].item()
        n = get('num_total_test_data')
        ns = num_samples if num_samples is not None else getattr(self, 'num_samples', 1)
        num_verts = num_verts if num_verts is not None else getattr(self, 'num_verts', 6890)      # 6890 for every SMPL model
        out = {}
        for m in self.metrics:
            if not frames[m].numel():
                continue
            if m == 'joints2D-L2E':
                out[m] = get(m) / get('num_vis_joints2D')
            elif m == 'joints2Dsamples-L2E':
                out[m] = get(m) / get('num_vis_joints2Dsamples')
            elif m == 'verts3D_sample_diversity':
                out[m] = get(m) / (n * ns * num_verts)
            elif m == 'joints3D_sample_diversity':
                out[m] = get(m) / (n * ns * 17)
            else:
                out[m] = get(m) / (n * (num_joints if m.startswith('MPJPE') else num_verts))
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
        xs, poses, shapes, j2d, vis = [], [], [], [], []
        for i in indices:
            g = torch.Generator().manual_seed(self.seed * 1000003 + i)
            xs.append(synthetic_proxy_input(1, self.size, seed=self.seed * 1000003 + i))
            poses.append(torch.randn(72, generator=g) * 0.3)
            shapes.append(torch.randn(10, generator=g) * 1.25)
            j2d.append(torch.rand(17, 2, generator=g) * self.size)
            vis.append(torch.rand(17, generator=g) > 0.2)
        return {'input': torch.cat(xs).to(device), 'pose': torch.stack(poses).to(device), 'shape': torch.stack(shapes).to(device),
                'joints2D': torch.stack(j2d).to(device), 'joints2D_visib': torch.stack(vis).to(device), 'index': torch.tensor(indices)}


PER_FRAME_MODEL_KEYS = ('glob_rotmat', 'pose_rotmats_point_est', 'shape_mode', 'cam_wp')     # what evaluate_humaniflow.py:260-268 saves per frame


@torch.no_grad()
def predict_batch(model, smpl, batch, num_samples, noise=None, materialise_tpose=True):
    """evaluate/evaluate_humaniflow.py:113-192 on the device: model -> SMPL point estimate + samples (ONE launch of B*N + B bodies) ->
    the evaluation joints, COCO joints and their 2-D projection of all rows (one launch, hmf_eval_joints) -> T-pose bodies of both (one
    launch).  materialise_tpose=False: the T-pose bodies are not built; the dict carries `tpose_betas` (B, nb) / `tpose_betas_samples`
    (B, N, nb) instead and an EvalMetricsTracker(smpl=...) scores them with hmf_smpl_tpose_errors (no 2 x 267 MB vertex round trip)."""
    N = num_samples
    pred = model(batch['input'], num_samples=N, noise=noise)
    g = pred['glob_rotmat']
    B = g.shape[0]
    img_wh = batch['input'].shape[-1]
    proj = lambda j3d, cam: undo_keypoint_normalisation(orthographic_project_torch(flip_about_x(j3d), cam), img_wh)     # :138-143
    if N > 0:
        pe, sa, joints_packed = smpl.forward_samples_and_point_est(pred, return_packed=True)
        betas = pred['smpl_packed_inputs']['betas']                          # rows: B*N sampled shapes, then the B shape modes
        j14, jc, j2d = eval_joints(joints_packed, pred['cam_wp'], B, N, img_wh)
        out = {'verts3D': pe.vertices, 'joints3D': j14[B * N:], 'joints2D': j2d[B * N:],
               'verts3D_samples': sa.vertices.reshape(B, N, -1, 3), 'joints3D_samples': j14[:B * N].reshape(B, N, 14, 3),
               'joints3D_coco_samples': jc[:B * N].reshape(B, N, 17, 3),                       # :180-183
               'joints2Dsamples': j2d[:B * N].reshape(B, N, 17, 2)}                              # :197-206
        out.update({k: pred[k] for k in PER_FRAME_MODEL_KEYS})
        if materialise_tpose:
            tp = smpl.tpose(betas)
            out['tpose_verts3D'], out['tpose_verts3D_samples'] = tp[B * N:], tp[:B * N].reshape(B, N, -1, 3)                     # :186-192
        else:
            out['tpose_betas'], out['tpose_betas_samples'] = betas[B * N:], betas[:B * N].reshape(B, N, -1)
            out['tpose_betas_packed'] = betas                                  # both, as the one (B N + B, nb) tensor of the SMPL launch
        return out
    pe = smpl(body_pose=pred['pose_rotmats_point_est'], global_orient=g[:, None], betas=pred['shape_mode'], pose2rot=False)
    out = {'verts3D': pe.vertices, 'joints3D': select_joints(pe.joints, 'h36m_j14'), 'joints2D': proj(select_joints(pe.joints, 'coco'), pred['cam_wp'])}
    if materialise_tpose:
        out['tpose_verts3D'] = smpl.tpose(pred['shape_mode'])
    else:
        out['tpose_betas'] = pred['shape_mode']
    out.update({k: pred[k] for k in PER_FRAME_MODEL_KEYS})
    return out


def eval_joints(joints_packed, cam, B, N, img_wh):
    """hmf_eval_joints: packed (B N + B, 90, 3) joints + (B, 3) cameras -> (M, 14, 3) evaluation joints, (M, 17, 3) COCO joints, (M, 17, 2) pixels."""
    import ctypes as C
    from . import _lib
    _lib.require_cuda(joints_packed, 'joints')
    lib = _lib.load()
    jp, cm = joints_packed.float().contiguous(), cam.float().contiguous()
    M, dev = jp.shape[0], jp.device
    assert M == B * N + B and cm.shape == (B, 3)
    j14, jc, j2d = (torch.empty(M, k, c, dtype=torch.float32, device=dev) for k, c in ((14, 3), (17, 3), (17, 2)))
    i14 = (C.c_int * 14)(*[ALL_JOINTS_TO_H36M_MAP[i] for i in H36M_TO_J14])
    i17 = (C.c_int * 17)(*ALL_JOINTS_TO_COCO_MAP)
    with torch.cuda.device(dev):
        _lib.check(lib.hmf_eval_joints(_lib.dptr(jp), _lib.dptr(cm), B, N, i14, i17, float(img_wh), _lib.dptr(j14), _lib.dptr(jc), _lib.dptr(j2d),
                                       _lib.current_stream()), 'hmf_eval_joints')
    return j14, jc, j2d


@torch.no_grad()
def target_batch(smpl, batch, materialise_tpose=True):
    """evaluate/evaluate_humaniflow.py:83-110 (single body model: the synthetic set has no gender split)."""
    verts, joints = smpl.forward_axis_angle(batch['pose'], batch['shape'])           # = smpl(body_pose=pose[:, 3:], global_orient=pose[:, :3], betas=shape)
    B = verts.shape[0]
    key = ('zero_cam', str(joints.device), B)
    if key not in _INDEX_CACHE:
        _INDEX_CACHE[key] = joints.new_zeros(B, 3)
    j14, _, _ = eval_joints(joints, _INDEX_CACHE[key], B, 0, 2.0)                    # the 14 evaluation joints (the camera outputs are not used)
    out = {'verts3D': verts, 'joints3D': j14}
    if materialise_tpose:
        out['tpose_verts3D'] = smpl.tpose(batch['shape'])
    else:
        out['tpose_betas'] = batch['shape']
    if 'joints2D' in batch:
        out['joints2D'] = batch['joints2D']
        if 'joints2D_visib' in batch:
            out['joints2D_vis'] = batch['joints2D_visib']
    return out


def evaluate_humaniflow(dataset, batch_size, device, predict_fn, target_fn, dist=None, rank=0, world_size=1, tracker=None, overlap_targets=False,
                        save_per_frame=True):
    """Sharded evaluation loop.  predict_fn(batch) / target_fn(batch) return the dicts EvalMetricsTracker consumes.  overlap_targets: the
    target bodies depend on the labels only, so they CAN be built on a side stream while the model runs (their M = B launches are latency-bound);
    off by default: at B = 32, N = 100 the side branch made the three-stream step slower (3.40 vs 3.23 ms, bench.py --eval, round 5).
    save_per_frame: record every frame's dataset index and point-estimate outputs (record_frames), so the returned per-frame arrays are in
    dataset order and carry the reference's pose / shape / cam per-frame outputs (evaluate_humaniflow.py:260-286)."""
    tracker = EvalMetricsTracker(device) if tracker is None else tracker
    mine = shard_indices(len(dataset), rank, world_size)
    dev = torch.device(device)
    side = torch.cuda.Stream(device=dev) if (overlap_targets and dev.type == 'cuda') else None
    for s in range(0, len(mine), batch_size):
        batch = dataset.batch(mine[s:s + batch_size], device)
        if side is None:
            pred, targ = predict_fn(batch), target_fn(batch)
        else:
            cur = torch.cuda.current_stream(dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                targ = target_fn(batch)
            pred = predict_fn(batch)
            cur.wait_stream(side)
            for t in targ.values():
                if torch.is_tensor(t):
                    t.record_stream(cur)
        tracker.update_per_batch(pred, targ)
        if save_per_frame:
            tracker.record_frames(batch['index'] if 'index' in batch else mine[s:s + batch_size], pred)
    return tracker.compute_final_metrics(dist)


def save_per_frame_outputs(frames, save_path, fnames=None):
    """Writes what the reference writes with save_per_frame_metrics (evaluate/evaluate_humaniflow.py:271-286, metrics/eval_metrics_tracker.py:498-501):
    fname / pose / shape / cam `_per_frame.npy` in dataset order plus one `<metric>_per_frame.npy` per tracked metric.  `fnames`: the dataset's
    file names (indexed by frames['index']); without them the indices themselves are saved."""
    import os
    import numpy as np
    os.makedirs(save_path, exist_ok=True)
    idx = frames['index'].cpu().numpy()
    np.save(os.path.join(save_path, 'fname_per_frame.npy'), np.asarray(fnames)[idx] if fnames is not None else idx)
    for k, v in frames.items():
        if k != 'index' and torch.is_tensor(v) and v.numel():
            np.save(os.path.join(save_path, k + '.npy' if k.endswith('_per_frame') else k + '_per_frame.npy'), v.cpu().numpy())
