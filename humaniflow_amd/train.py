"""Training step of the hot path: the differentiable part of train/train_humaniflow.py:248-335, data-parallel.

  forward    HumaniflowModel(proxy, compute_point_est=True, num_samples=LOSS.NUM_J2D_SAMPLES, compute_for_loglik=True, targets...)   :250-256
  SMPL       joints of the point estimate and of the samples in ONE launch, flip about x, orthographic projection                     :283-318
  loss       HumaniflowLoss: pose NLL + shape NLL + visible-2D-joint MSE + global-rotation MSE                                        :323-343
  backward   HIP reverse sweeps of the flow / SMPL (autograd.py) and of the encoder (encoder_train.py), torch autograd for the four head GEMMs
  exchange   gradients bucketed (25 MB) and all-reduced over RCCL while the rest of the backward still runs (xGMI rings are per-link bound)
  update     Adam.  nan_policy = 'skip' (default): a non-finite loss or gradient skips the update on the device, no host read;
             nan_policy = 'rollback': the reference's own semantics (:261-278, :348-349) -- NaNs in the forward's outputs undo the LAST update
             (parameters, buffers, optimiser state as they were before it: LastBatchState) and the batch is run again

What the reference's loop does around this (pytorch3d rendering of the synthetic person, texture / background / occlusion
augmentation, :113-246) is data generation outside the path: `synthetic_train_batch` stands in with SMPL-consistent targets
(2D joints projected from the target body) and a proxy representation built by the library's front end (hmf_proxy_rep).
"""
import math
import time

import os

import torch

from . import _lib
from .evaluate import ALL_JOINTS_TO_COCO_MAP, flip_about_x, joint_index, orthographic_project_torch, select_joints
from .losses import HumaniflowLoss


_XY_SIGN = {}
FUSED_LOSS = True               # the step's loss and its gradients on hmf_train_loss (two launches); False: HumaniflowLoss on torch operators
JOINTS_ONLY = os.environ.get('HMF_TRAIN_JOINTS_ONLY', '1') != '0'    # fused-loss path: SMPL joints without the vertices (hmf_smpl_joints); the loss reads joints only


def _xy_sign(t):
    """(1, -1) on t's device: the x / y part of the 180-degree rotation about x (evaluate.flip_about_x), a cached device constant."""
    key = (str(t.device), t.dtype)
    v = _XY_SIGN.get(key)
    if v is None:
        v = _XY_SIGN[key] = torch.tensor([1.0, -1.0], device=t.device, dtype=t.dtype)
    return v


# Measured round 5 (bench.py --train, same box, x2): 13.64-13.81 ms with the density branch on a side stream vs 13.26-13.30 ms on one stream -- the
# forks and joins the captured graph gains (autograd adds an event pair per cross-stream edge) cost more than the 0.6 ms of latency-class
# kernels they overlap.  Off; HMF_TRAIN_SIDE=1 turns it on for experiments.
OVERLAP_NLL_BRANCH = os.environ.get('HMF_TRAIN_SIDE', '0') == '1'
_SIDE = {}


def _side_stream(device):
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


def train_forward_loss(model, smpl, criterion, proxy_input, targets, num_samples, noise=None, input_feats=None):
    """targets: pose_rotmats (B,23,3,3), glob_rotmats (B,3,3), shape (B,nb), joints2D (B,17,2) pixels, joints2D_vis (B,17) bool.
    Returns (loss, pred).  J2D_LOSS_ON = 'point_est+samples' | 'samples' | 'point_est' as LOSS.J2D_LOSS_ON."""
    N = num_samples
    pred = model(proxy_input, compute_point_est=True, num_samples=N, compute_for_loglik=True, shape_for_loglik=targets['shape'],
                 pose_R_for_loglik=targets['pose_rotmats'], glob_R_for_loglik=targets['glob_rotmats'], noise=noise, input_feats=input_feats)
    B = pred['glob_rotmat'].shape[0]
    cfg_l = criterion.loss_cfg
    dists = pred['conditioned_pose_SO3flow_dists_for_loglik']
    if (FUSED_LOSS and N > 0 and cfg_l.J2D_LOSS_ON == 'point_est+samples' and not cfg_l.APPLY_POINT_EST_LOSS and cfg_l.REDUCTION in ('mean', 'sum')
            and pred['glob_rotmat'].is_cuda and getattr(dists[0], 'parent', None) is not None):
        # :283-343 in two launches (csrc/train_loss.hip): joint selection, flip, projection and the four loss terms with their gradients; as torch
        # operators the same arithmetic is ~150 launches of a few hundred elements each (and the reference's own loop)
        from .autograd import FusedTrainLoss
        # The pose NLL (teacher-forced densities) and the reprojection term (samples -> SMPL joints) are independent branches between the heads and the
        # loss, and both are latency-class launches: the density branch CAN run on a side stream, forward and -- autograd replays a node on the stream
        # of its forward -- backward (k_flow_logprob / _bwd: 0.6 ms of a step).  Measured slower (see OVERLAP_NLL_BRANCH): off by default.
        if OVERLAP_NLL_BRANCH:
            cur = torch.cuda.current_stream(joints_dev := pred['glob_rotmat'].device)
            side = _side_stream(joints_dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                lp = criterion.pose_log_probs(dists, targets['pose_rotmats'].double())
            pe, sa, joints = smpl.forward_samples_and_point_est(pred, return_packed=True, joints_only=JOINTS_ONLY)
            cur.wait_stream(side)
            lp.record_stream(cur)
        else:
            pe, sa, joints = smpl.forward_samples_and_point_est(pred, return_packed=True, joints_only=JOINTS_ONLY)
            lp = criterion.pose_log_probs(dists, targets['pose_rotmats'].double())
        W = cfg_l.WEIGHTS
        loss = FusedTrainLoss.apply(lp, pred['shape_mode'], pred['shape_log_std'], joints, pred['cam_wp'], pred['glob_rotmat'], targets['shape'],
                                    targets['joints2D'], targets['joints2D_vis'], targets['glob_rotmats'], N, ALL_JOINTS_TO_COCO_MAP, criterion.img_wh,
                                    (W.POSE, W.SHAPE, W.JOINTS2D, W.GLOB_ROTMATS), cfg_l.REDUCTION == 'mean')
        return loss, pred
    pe, sa = smpl.forward_samples_and_point_est(pred)                                   # :283-287 and :300-304 in one launch
    cam = pred['cam_wp']
    # :289-297: orthographic_project_torch(flip_about_x(select_joints(joints, 'coco')), cam) = s * ((x, -y) + t) of the 17 COCO joints, written
    # as one gather, one sign multiply and the projection (4 launches, 8 in backward; the three helpers in a row are 8 + ~17)
    def proj(joints, c):
        xy = joints.index_select(1, joint_index('coco', joints.device))[..., :2] * _xy_sign(joints)
        return c[:, None, 0:1] * (xy + c[:, None, 1:])
    j2d_pe = proj(pe.joints, cam)[:, None]
    on = criterion.loss_cfg.J2D_LOSS_ON
    if 'samples' in on:
        j2d = proj(sa.joints, cam[:, None].expand(-1, N, -1).reshape(B * N, 3)).reshape(B, N, 17, 2)
        if on == 'point_est+samples':
            j2d = torch.cat([j2d_pe, j2d], dim=1)                                       # :312-314
    else:
        j2d = j2d_pe
    pred_for_loss = {'pose_dist': pred['conditioned_pose_SO3flow_dists_for_loglik'], 'shape_dist': pred['shape_dist_for_loglik'],
                     'joints2D': j2d, 'glob_rotmats': pred['glob_rotmat']}
    targ_for_loss = {'pose_params_rotmats': targets['pose_rotmats'], 'shape_params': targets['shape'], 'joints2D': targets['joints2D'],
                     'joints2D_vis': targets['joints2D_vis'], 'glob_rotmats': targets['glob_rotmats']}
    return criterion(targ_for_loss, pred_for_loss), pred


@torch.no_grad()
def synthetic_train_batch(smpl, B, size, device, seed, data_cfg=None, shape_std=1.25):
    """on_the_fly_smpl_train_dataset stand-in (train/train_humaniflow.py:113-246 without the renderer): random axis-angle poses and
    N(0, 1.25^2) shapes, targets from the body model, 2D joints = orthographic projection of the target COCO joints, proxy representation =
    Canny edges of a random image + heat-maps of those joints (hmf_proxy_rep)."""
    from .proxy import proxy_representation
    from .rigid_transform_utils import batch_rodrigues
    g = torch.Generator().manual_seed(seed)
    pose = (torch.randn(B, 72, generator=g) * 0.3).to(device)
    shape = (torch.randn(B, 10, generator=g) * shape_std).to(device)
    R = batch_rodrigues(pose.reshape(-1, 3)).view(B, 24, 3, 3)
    glob, body = R[:, 0].contiguous(), R[:, 1:].contiguous()
    out = smpl(body_pose=body, global_orient=glob[:, None], betas=shape, pose2rot=False)
    cam = torch.tensor([0.9, 0.0, 0.0], device=device).expand(B, 3)
    j2d_n = orthographic_project_torch(flip_about_x(select_joints(out.joints, 'coco')), cam)       # normalised [-1, 1]
    j2d = (j2d_n + 1.0) * (size / 2.0)
    vis = ((j2d > 0) & (j2d < size)).all(dim=-1) & (torch.rand(B, 17, generator=g) > 0.1).to(device)
    img = torch.rand(B, 3, size, size, generator=g).to(device)
    proxy = proxy_representation(img, j2d, vis.float(), data_cfg)
    return proxy, {'pose_rotmats': body, 'glob_rotmats': glob, 'shape': shape, 'joints2D': j2d, 'joints2D_vis': vis}


class GradBucketReducer:
    """Data-parallel gradient exchange: parameters are grouped (in reverse registration order, i.e. roughly the order their gradients
    become ready) into flat buckets of <= bucket_mb; a post-accumulate hook counts a bucket's parameters in, and the moment the last one
    arrives the bucket is flattened and all-reduced asynchronously (RCCL's own stream) while autograd keeps producing the earlier
    layers' gradients.  finish() waits, averages and scatters the results back into .grad.

    Life cycle of one step: begin() (counters reset, nothing may be in flight) -> backward (hooks) -> finish().  Every rank must issue the same
    collectives in the same order: a bucket is launched either by the hook of its last parameter or -- for buckets holding a parameter that
    received no gradient -- by finish(), which launches the stragglers in ascending bucket index; the all_reduce of the shared skip flag
    (all_finite) comes last.  Backward passes that are not a training step (measurement passes, gradient checks) run under `with
    reducer.disabled():` -- an armed hook would launch collectives nobody waits for."""

    def __init__(self, params, dist, bucket_mb=25.0):
        self.dist = dist
        self.world = dist.get_world_size() if dist is not None else 1
        self.params = [p for p in params if p.requires_grad]
        self.buckets, cur, size = [], [], 0
        for p in reversed(self.params):
            cur.append(p)
            size += p.numel() * 4
            if size >= bucket_mb * 2 ** 20:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        self.where = {id(p): bi for bi, b in enumerate(self.buckets) for p in b}
        self.pending = [len(b) for b in self.buckets]
        self.launched = [False] * len(self.buckets)
        self.inflight = []
        self.hooks = []
        self.enabled = True
        self.consecutive_skips = None          # device counter of update steps skipped in a row (train_step), see check_progress()
        if self.world > 1:
            for p in self.params:
                self.hooks.append(p.register_post_accumulate_grad_hook(self._ready))

    def begin(self):
        """Start of a step: nothing of the previous step may still be in flight (finish() was not called, or a stray backward ran armed)."""
        if self.inflight or any(self.launched):
            raise RuntimeError('GradBucketReducer.begin(): %d bucket(s) of a previous backward were never finished -- call finish() after every '
                               'backward of a step and wrap other backward passes in `with reducer.disabled():`' % (len(self.inflight) or sum(self.launched)))
        self.pending = [len(b) for b in self.buckets]

    def disabled(self):
        """Context manager: backward passes inside do not exchange gradients (measurement passes, gradient checks)."""
        red = self

        class _Off:
            def __enter__(self_):
                self_.prev, red.enabled = red.enabled, False

            def __exit__(self_, *exc):
                red.enabled = self_.prev
                return False
        return _Off()

    def _ready(self, p):
        if not self.enabled:
            return
        bi = self.where[id(p)]
        self.pending[bi] -= 1
        assert self.pending[bi] >= 0, 'GradBucketReducer: a parameter produced two gradients in one step (begin() not called?)'
        if self.pending[bi] == 0:
            self._launch(bi)

    def _launch(self, bi):
        assert not self.launched[bi], 'GradBucketReducer: bucket %d launched twice in one step' % bi
        self.launched[bi] = True
        ps = self.buckets[bi]
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in ps])
        work = self.dist.all_reduce(flat, op=self.dist.ReduceOp.SUM, async_op=True)
        self.inflight.append((bi, flat, work))

    def finish(self):
        """Call after backward(): launches buckets whose parameters never produced a gradient hook, waits, writes the averages back."""
        if self.world == 1:
            return
        for bi in range(len(self.buckets)):              # stragglers in ascending bucket index: the same order on every rank
            if not self.launched[bi]:                    # a parameter unused by this step's graph: still exchange (zeros) so ranks agree
                self.pending[bi] = 0
                self._launch(bi)
        for bi, flat, work in self.inflight:
            work.wait()
            flat.div_(self.world)
            o = 0
            for p in self.buckets[bi]:
                n = p.numel()
                if p.grad is None:
                    p.grad = flat[o:o + n].view_as(p).clone()
                else:
                    p.grad.copy_(flat[o:o + n].view_as(p))
                o += n
        self.inflight = []
        self.pending = [len(b) for b in self.buckets]
        self.launched = [False] * len(self.buckets)

    def all_finite(self, flag):
        """AND of a 0-dim boolean tensor over the ranks (one 4-byte all_reduce, after the gradient buckets): the skip decision of a step must be
        the same everywhere, or one rank skips an update the others apply and the replicas drift apart for good."""
        if self.world == 1:
            return flag
        bad = (~flag).to(torch.float32).reshape(1)
        self.dist.all_reduce(bad, op=self.dist.ReduceOp.MAX)
        return bad.reshape(()) == 0

    def check_progress(self, limit=20):
        """Host-side guard for the training loop (one device read; call it every few hundred steps): raises when the last `limit` or more
        updates in a row were skipped as non-finite -- a weight state that always yields NaN would otherwise stall training silently."""
        if self.consecutive_skips is not None and float(self.consecutive_skips) >= limit:
            raise RuntimeError('humaniflow_amd.train: the last %d optimiser updates were all skipped (non-finite loss or gradients)' % int(self.consecutive_skips))


def grads_finite(params):
    """Device-side finite check of all gradients: one stacked reduction, one host read."""
    norms = torch._foreach_norm([p.grad for p in params if p.grad is not None])
    return bool(torch.isfinite(torch.stack(norms)).all())


class LastBatchState:
    """The reference's `last_batch_model_state_dict` / `last_batch_opt_state_dict` (train/train_humaniflow.py:348-349: deep copies of both state
    dicts taken before every backward + step; :264-268 loads them back when the NEXT forward's outputs hold NaNs), as in-place device copies:
    save() mirrors every tensor of model.state_dict() and of the optimiser's per-parameter state into buffers allocated once (two multi-tensor
    copies per step instead of a few hundred tensor allocations), restore() copies them back INTO the live tensors -- parameter, moment and
    step-count storage keeps its address, so gradient buckets, a captured step and a fused optimiser stay valid, which load_state_dict (new
    state tensors for the optimiser) would not give.  Non-tensor optimiser state (python step counts of a non-fused / non-capturable optimiser)
    is kept by value.  Restoring to the state before the update that poisoned the weights and stepping again from the current batch is exactly
    the reference's trajectory."""

    def __init__(self, model, optimiser):
        self.model, self.optimiser = model, optimiser
        self._live, self._kept, self._plain, self._keys = None, None, None, None

    def _tensors(self):
        live = [t for t in self.model.state_dict(keep_vars=True).values() if isinstance(t, torch.Tensor)]
        plain = {}
        for gi, g in enumerate(self.optimiser.param_groups):
            for pi, p in enumerate(g['params']):
                for k, v in sorted(self.optimiser.state.get(p, {}).items()):
                    if isinstance(v, torch.Tensor):
                        live.append(v)
                    else:
                        plain[(gi, pi, k)] = v
        return live, plain

    def valid(self):
        return self._kept is not None

    @torch.no_grad()
    def save(self):
        live, plain = self._tensors()
        keys = [(t.data_ptr(), tuple(t.shape), t.dtype) for t in live]
        if self._keys != keys:                       # first call, or the optimiser created its state since (Adam does on its first step)
            self._kept = [torch.empty_like(t) for t in live]
            self._keys = keys
        self._live = live
        if live:
            torch._foreach_copy_(self._kept, [t.detach() for t in live])
        self._plain = dict(plain)

    @torch.no_grad()
    def restore(self):
        if self._kept is None:
            raise RuntimeError('LastBatchState.restore() before any save(): there is no earlier state to go back to')
        torch._foreach_copy_([t.detach() for t in self._live], self._kept)
        # state the optimiser created AFTER the snapshot (first update) has no earlier value: the reference's load_state_dict drops it
        kept_ptrs = {k[0] for k in self._keys}
        for gi, g in enumerate(self.optimiser.param_groups):
            for pi, p in enumerate(g['params']):
                st = self.optimiser.state.get(p)
                if not st:
                    continue
                for k in list(st.keys()):
                    v = st[k]
                    if isinstance(v, torch.Tensor):
                        if v.data_ptr() not in kept_ptrs:
                            del st[k]
                    elif (gi, pi, k) in self._plain:
                        st[k] = self._plain[(gi, pi, k)]
                    else:
                        del st[k]


def nans_in_output(pred):
    """utils/train_utils.py:4-12 (check_for_nans_in_output) with one host read: any NaN in any tensor value of the forward's return dict."""
    flags = [v.isnan().any() for v in pred.values() if isinstance(v, torch.Tensor) and v.is_floating_point()]
    return torch.stack(flags).any() if flags else torch.zeros((), dtype=torch.bool)


def train_step(model, smpl, criterion, optimiser, reducer, proxy_input, targets, num_samples, noise=None, nan_policy='skip', last_state=None):
    """One optimisation step.  Returns (loss tensor, skipped): a non-finite loss or gradient skips the update (the reference reloads the previous
    state dicts, train/train_humaniflow.py:264-278).  With a fused Adam on the device the decision never leaves the device: the finite flag is
    handed to the optimiser as `found_inf` (the fused kernel then leaves parameters, moments and step counts untouched), `skipped` is a 0-dim
    tensor and the host never waits for the step -- consecutive steps overlap launch with execution.  Other optimisers: one host read.
    The flag is AND-reduced over the ranks (reducer.all_finite): the local loss differs per rank, the decision must not.
    Difference from the reference's roll-back: the BatchNorm running statistics (and num_batches_tracked) of a skipped step's forward stay
    updated -- the reference restores the whole state dict including buffers; the trained parameters, Adam moments and step counts are
    untouched in both.  Skips in a row are counted on the device (reducer.consecutive_skips); reducer.check_progress() raises on a stall.

    nan_policy='rollback' (with last_state = LastBatchState(model, optimiser), kept across steps) is the reference's loop instead (:261-278,
    :348-349): NaNs in the forward's outputs -> print, restore the state from before the last update, forward again; the state is saved before
    every backward + step.  It costs one host read per step (the reference's check is a host read per output tensor) and one copy of the
    weights + moments (ResNet-18: ~140 MB, ~0.06 ms), and it cannot run inside a captured graph (the redo is host control flow).  The
    non-finite-gradient skip below stays armed in both modes (the reference has no such check; with finite gradients it never fires)."""
    if nan_policy not in ('skip', 'rollback'):
        raise ValueError("nan_policy must be 'skip' or 'rollback', got %r" % (nan_policy,))
    if nan_policy == 'rollback' and last_state is None:
        raise ValueError("nan_policy='rollback' needs last_state=LastBatchState(model, optimiser), kept across steps")
    optimiser.zero_grad(set_to_none=True)
    reducer.begin()
    loss, pred = train_forward_loss(model, smpl, criterion, proxy_input, targets, num_samples, noise=noise)
    if nan_policy == 'rollback':
        nan = nans_in_output(pred)
        if reducer.world > 1:                                           # every rank must take the same branch
            nan = ~reducer.all_finite(~nan)
        if bool(nan) and last_state.valid():
            print('Reloading model and optimiser state from before last SGD step.')
            last_state.restore()
            print('Re-doing current batch forward pass.')
            del loss, pred
            reducer.begin()
            loss, pred = train_forward_loss(model, smpl, criterion, proxy_input, targets, num_samples, noise=noise)
        last_state.save()
    loss.backward()
    reducer.finish()
    params = [p for g in optimiser.param_groups for p in g['params']]
    on_device = isinstance(optimiser, torch.optim.Adam) and loss.is_cuda and all(g.get('fused') for g in optimiser.param_groups)
    norms = torch._foreach_norm([p.grad for p in params if p.grad is not None])
    ok = reducer.all_finite(torch.isfinite(loss.detach()) & torch.isfinite(torch.stack(norms)).all())
    bad = (~ok).to(torch.float32)
    if reducer.consecutive_skips is None:                               # created by the first (eager, warm-up) step; updated in place afterwards so a
        reducer.consecutive_skips = torch.zeros((), device=loss.device)  # captured step keeps counting into the same tensor
    reducer.consecutive_skips.copy_((reducer.consecutive_skips + 1.0) * bad)
    if on_device:
        optimiser.grad_scale, optimiser.found_inf = None, bad
        optimiser.step()
        optimiser.found_inf = None
        skipped = bad
    else:
        if not bool(ok):
            optimiser.zero_grad(set_to_none=True)
            return loss.detach(), True
        optimiser.step()
        skipped = False
    for d in model.pose_SO3flow_dists:                                  # :353-354 (no caches here; kept for call parity)
        d.clear_cache()
    return loss.detach(), skipped


def bench_train(args, dev, dist, world, rank, log, dist_sync, dist_elapsed, dist_gather_metrics):
    """bench.py --train: BASELINE configs[4] per-GPU step (B = TRAIN.BATCH_SIZE = 72, N = LOSS.NUM_J2D_SAMPLES = 8, ResNet-18 by default,
    the reference's training configuration), weak scaling, gradients all-reduced over RCCL."""
    import json
    import humaniflow_amd as H
    from .synthetic import init_benchmark_weights
    cfg = H.get_humaniflow_cfg_defaults()
    cfg.MODEL.NUM_RESNET_LAYERS = args.layers
    torch.manual_seed(0)
    model = init_benchmark_weights(H.HumaniflowModel(torch.device('cpu'), cfg.MODEL, H.SMPL_PARENTS), seed=0).to(dev)
    model.train()
    model.pose_so3flow_transform_modules.train()                        # :103 call parity
    smpl = H.SMPL(model_data=H.synthetic_smpl_model(seed=0), batch_size=args.batch, create_transl=False).to(dev)
    for p in smpl.parameters():
        p.requires_grad_(False)
    criterion = HumaniflowLoss(cfg.LOSS, cfg.DATA.PROXY_REP_SIZE)
    params = [p for p in model.parameters() if p.requires_grad]
    use_graph = world == 1 and not args.no_graph                        # one captured HIP graph per distinct batch (single GPU; the RCCL hooks stay eager)
    optimiser = torch.optim.Adam(params, lr=cfg.TRAIN.LR, fused=True, capturable=use_graph)   # one multi-tensor launch instead of ~10 tiny ones per parameter
    reducer = GradBucketReducer(params, dist if world > 1 else None)
    B, N = args.batch, args.samples
    NB = max(1, args.batches)
    batches = [synthetic_train_batch(smpl, B, cfg.DATA.PROXY_REP_SIZE, dev, seed=1000 + rank * NB + i, data_cfg=cfg.DATA) for i in range(NB)]
    torch.manual_seed(1234 + rank)
    losses, skipped = [], torch.zeros((), device=dev)
    log('train flavour: B=%d, N=%d, ResNet-%d, %d parameters (%.1f MB of gradients per step), %d ranks'
        % (B, N, args.layers, sum(p.numel() for p in params), sum(p.numel() for p in params) * 4 / 2 ** 20, world))
    graphs = None
    if use_graph:
        # whole-step capture: forward + loss + backward + device-side finite check + fused Adam are ONE graph per batch (the step has no host
        # synchronisation, so its ~1,000 launches can be replayed instead of issued)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for k in range(max(3, NB)):
                train_step(model, smpl, criterion, optimiser, reducer, *batches[k % NB], N)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        try:
            graphs = []
            for i in range(NB):
                g = torch.cuda.CUDAGraph()
                optimiser.zero_grad(set_to_none=True)
                with torch.cuda.graph(g, capture_error_mode='thread_local'):
                    out = train_step(model, smpl, criterion, optimiser, reducer, *batches[i], N)
                graphs.append((g, out))
            torch.cuda.synchronize()
            log('captured %d training-step graphs' % NB)
        except Exception as e:                                          # eager issue instead, and say so in the JSON line
            log('training-step graph capture failed (%r): eager steps' % (e,))
            graphs = None
            torch.cuda.synchronize()

    def run_step(k):
        if graphs is not None:
            g, out = graphs[k % NB]
            g.replay()
            return out[0].clone(), out[1].clone()
        return train_step(model, smpl, criterion, optimiser, reducer, *batches[k % NB], N)
    for k in range(args.warmup):
        run_step(k)
    dist_sync(dist, dev)
    t0 = time.perf_counter()
    for k in range(args.steps):
        loss, sk = run_step(args.warmup + k)
        losses.append(loss)
        skipped = skipped + sk
    metric_vec = torch.stack([torch.stack(losses).double().mean(), skipped.double().reshape(())])
    gathered = dist_gather_metrics(metric_vec, dist, world)
    elapsed = dist_elapsed(t0, dist, dev)
    # stage split of one step (separate, untimed): forward+loss / backward / exchange+update
    stages = {}

    def tm(fn):
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) * 1e3, r
    if graphs is not None:
        model.mark_params_changed()          # the replays stepped the parameters without bumping their version counters: re-pack before eager passes
    optimiser.zero_grad(set_to_none=True)
    reducer.begin()
    stages['forward+loss_ms'], (loss, _) = tm(lambda: train_forward_loss(model, smpl, criterion, *batches[0], N))
    stages['backward_ms'], _ = tm(lambda: loss.backward())
    stages['exchange+update_ms'], _ = tm(lambda: (reducer.finish(), optimiser.step()))
    with reducer.disabled():                 # measurement passes: no collectives
        enc_ms, feats = tm(lambda: model.image_encoder(batches[0][0]))
        stages['of_forward: encoder training graph (batch-statistics BN)_ms'] = enc_ms
        w_ = torch.randn_like(feats)
        stages['of_backward: encoder_ms'], _ = tm(lambda: (feats * w_).sum().backward())
    optimiser.zero_grad(set_to_none=True)
    # ---- like-for-like note: the default step builds SMPL JOINTS only (the loss reads nothing else); the reference's step always builds
    # the vertices too (train/train_humaniflow.py:262-283, used by its metric tracking).  Same eager issue for both, a few steps each, untimed region.
    global JOINTS_ONLY
    like = None
    try:
        def eager_ms(k_steps=4):
            with reducer.disabled():
                train_step(model, smpl, criterion, optimiser, reducer, *batches[0], N)
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for k in range(k_steps):
                    train_step(model, smpl, criterion, optimiser, reducer, *batches[k % NB], N)
                b.record()
                torch.cuda.synchronize()
            return a.elapsed_time(b) / k_steps
        keep_jo = JOINTS_ONLY
        t_jo = eager_ms()
        JOINTS_ONLY = False
        t_full = eager_ms()
        JOINTS_ONLY = keep_jo
        like = {'eager_ms_per_step_smpl_joints_only': round(t_jo, 3), 'eager_ms_per_step_full_smpl_vertices': round(t_full, 3),
                'note': 'the timed step builds SMPL joints without the vertices (HMF_TRAIN_JOINTS_ONLY=1, default: the loss reads joints only); the reference step '
                        'also builds the 6890-vertex meshes of the point estimate and the samples for its metric tracking.  Both forms issued eagerly on one '
                        'stream here: the difference is what the like-for-like step would add to ms_per_step'}
    except Exception as e:
        JOINTS_ONLY = os.environ.get('HMF_TRAIN_JOINTS_ONLY', '1') != '0'
        log('like-for-like pass failed: %r' % (e,))
    optimiser.zero_grad(set_to_none=True)
    # ---- per-kernel-family measurement pass: HIP event pair around every convolution / weight-gradient launch of 2 more steps ----
    roof = None
    try:
        import numpy as np
        enc = model.image_encoder
        import copy
        lib = _lib.load()
        menc = copy.deepcopy(enc).to('meta')                            # shapes only: a meta copy, no kernel runs
        convs = [m for m in menc.modules() if isinstance(m, torch.nn.Conv2d)]
        shapes = {}
        hooks = [m.register_forward_hook(lambda mod, i, o: shapes.__setitem__(mod, o.shape)) for m in convs]
        with torch.no_grad():
            menc.forward_module(torch.empty((1,) + tuple(batches[0][0].shape[1:]), device='meta'))
        for h in hooks:
            h.remove()
        per_img = {m: 2.0 * shapes[m][2] * shapes[m][3] * m.out_channels * m.in_channels * m.kernel_size[0] ** 2 for m in convs}
        f_all = B * sum(per_img.values())
        f_dgrad = f_all - B * per_img[menc.conv1]                      # the proxy input needs no gradient
        # what the matrix pipe EXECUTES: the stride-1 3x3 layers run as Winograd (F(2x2,3x3) forward / input gradient, F(3x3,2x2) weight gradient):
        # 16 multiplies per 2x2 outputs instead of 36 -> 4/9 of the direct-form FLOPs, in all three passes
        wino = lambda m: m.kernel_size[0] == 3 and m.stride[0] == 1 and m.in_channels % 64 == 0 and m.out_channels % 32 == 0
        x_all = B * sum(v * (4.0 / 9.0 if wino(m) else 1.0) for m, v in per_img.items())
        x_dgrad = x_all - B * per_img[menc.conv1]
        P = 2
        lib.hmf_profile_enable(1)
        with reducer.disabled():
            for k in range(P):
                optimiser.zero_grad(set_to_none=True)
                loss, _ = train_forward_loss(model, smpl, criterion, *batches[k % len(batches)], N)
                loss.backward()
        torch.cuda.synchronize()
        cap = 65536
        ms, tag, rows = np.zeros(cap), np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        n = lib.hmf_profile_collect(ms.ctypes.data_as(_lib.c_double_p), _lib.iptr(tag), _lib.iptr(rows), cap)
        lib.hmf_profile_enable(0)
        ms, tag = ms[:n], tag[:n]
        t_w = float(ms[tag == 13].sum()) / P * 1e-3
        t_c = float(ms[(tag >= 8) & (tag <= 11)].sum()) / P * 1e-3
        fam = {'weight gradients (k_stem_wgrad, k_wgw, k_conv_wgrad3, k_conv_wgrad)': (f_all, t_w, int((tag == 13).sum()) // P, x_all),
               'forward + input-gradient convolutions (k_stem_conv, k_conv_winograd, k_conv_patch, k_conv_dgrad_s2, k_conv_gemm, k_conv_igemm)': (f_all + f_dgrad, t_c, int(((tag >= 8) & (tag <= 11)).sum()) // P, x_all + x_dgrad)}
        name = max(fam, key=lambda k: fam[k][1])
        f, t, nl, fx = fam[name]
        # HBM bytes per launch of that family from the separate rocprofv3 --pmc passes (tools/profile_round.sh), quoted only when the summary was
        # collected at this library source hash on this workload
        traffic, traffic_note = None, 'null: no profiles/*_pmc.json at this library source hash for the train workload'
        try:
            import json as _json
            import os as _os
            from .build import source_hash
            pdir = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'profiles')
            keys = ('k_stem_wgrad', 'k_conv_wgrad') if name.startswith('weight') else ('k_stem_conv', 'k_conv_winograd', 'k_conv_patch', 'k_conv_igemm', 'k_conv_gemm')
            for fn in sorted(_os.listdir(pdir)):
                if fn.endswith('_pmc.json'):
                    d = _json.load(open(_os.path.join(pdir, fn)))
                    if d.get('lib_source_hash') == source_hash() and d.get('workload') == 'train,B=%d,N=%d' % (B, N):
                        cand = [v for k, v in d['kernels'].items() if any(kk in k for kk in keys)]
                        if cand:
                            traffic = int(sum(c['hbm_bytes_per_launch'] * c['launches'] for c in cand) / sum(c['launches'] for c in cand))
                            traffic_note = 'mean HBM bytes/launch over this family (profiles/%s)' % fn
        except Exception:
            pass
        # `achieved` / `frac` = EXECUTED matrix-pipe FLOPs over the family's kernel time against the fp32 MFMA peak (round 3 quoted the direct-form
        # figure there: 0.96 while MfmaUtil was ~50 %); the direct-form equivalent is kept beside it as `direct_form_*`
        roof = {'kernel': name + ': the kernel family with the largest share of the training step (%d launches/step)' % nl, 'bound': 'mfma',
                'achieved': round(fx / t / 1e12, 2), 'peak': 157.3, 'unit': 'TFLOP/s', 'frac': round(fx / t / 1e12 / 157.3, 4), 'traffic': traffic, 'traffic_note': traffic_note,
                'direct_form_tflops': round(f / t / 1e12, 2), 'direct_form_frac': round(f / t / 1e12 / 157.3, 4),
                'ms_per_step': round(t * 1e3, 3), 'algorithmic_gflop_per_step': round(f / 1e9, 1), 'executed_mfma_gflop_per_step': round(fx / 1e9, 1),
                'how': 'algorithmic 2*M*N*K FLOPs of the family / its kernel execution time (HIP event pair around every launch of %d extra steps); '
                       'the stride-1 3x3 layers run as Winograd F(2x2,3x3) and are booked at direct-form FLOPs (the MFMA pipe executes 4/9 of them), '
                       'the input gradients of the stride-2 3x3 layers run as parity-class GEMMs (hmf_conv_dgrad_s2) and are booked at the FLOPs of the layer; the stride-1 3x3 '
                       'weight gradients run as Winograd F(3x3,2x2) and are booked at direct-form FLOPs' % P,
                'families': {k: {'gflop_per_step': round(v[0] / 1e9, 1), 'executed_mfma_gflop_per_step': round(v[3] / 1e9, 1), 'ms_per_step': round(v[1] * 1e3, 3),
                                 'tflops_direct_form': round(v[0] / v[1] / 1e12, 2), 'tflops_executed_mfma': round(v[3] / v[1] / 1e12, 2),
                                 'frac_executed_mfma': round(v[3] / v[1] / 1e12 / 157.3, 4), 'launches_per_step': v[2]} for k, v in fam.items()}}
    except Exception as e:                                              # the measurement pass must not cost the bench line
        log('train roofline pass failed: %r' % (e,))
    optimiser.zero_grad(set_to_none=True)
    res = None
    if rank == 0:
        value = world * B * N * args.steps / elapsed
        res = ({
            'metric': 'images*samples/sec, training step (B=%d, N=%d)' % (B, N), 'value': round(value, 1), 'unit': 'images*samples/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'configs[4] train step: B=%d synthetic on-the-fly SMPL batch, N=%d reparameterised samples, ResNet-%d + flow + SMPL joints, '
                                   'HumaniflowLoss (pose NLL + shape NLL + 2D reprojection + global rotation), Adam' % (B, N, args.layers),
                       'global_batch': world * B, 'samples_per_image': N, 'parallelism': 'dp%d' % world, 'issue': 'HIP graph replay (one captured training-step graph per batch)' if graphs is not None else 'eager, no host synchronisation inside a step',
                       'exchange': ('gradient all-reduce over RCCL, %d buckets of <= 25 MB, launched from gradient-ready hooks' % len(reducer.buckets))
                       if world > 1 else 'none'},
            'rccl_ranks_seen': int(gathered.shape[0]), 'mean_loss': float(gathered[:, 0].mean()), 'skipped_steps': int(gathered[:, 1].sum()),
            'train_images_per_s': round(world * B * args.steps / elapsed, 1),
            'smpl_work_vs_reference_step': like,
            'stages_ms_eager_synchronised': {k: round(v, 3) for k, v in stages.items()},
            'stages_note': 'one EAGER pass with a host synchronise around every stage (launch-bound: ~500 launches issued from Python); NOT a breakdown of '
                           'ms_per_step, which is the graph-replayed step',
            'hand_written': 'encoder training graph (conv kernels incl. Winograd with device-side filter transform + hmf_bn_* + hmf_conv_wgrad + dgrad through the forward kernels), flow forward/backward '
                            '(hmf_flow_forward, hmf_flow_log_prob_SO3[_backward], hmf_flow_sample_backward), SMPL forward + joints backward, proxy front end; '
                            'library GEMMs (torch) for the heads and the dense weight-gradient reductions of the flow',
            'roofline': roof, 'cpu_baseline': None})
    return res
