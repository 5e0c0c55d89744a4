#!/usr/bin/env python
"""bench.py -- images.samples/sec of the HuManiFlow inference hot path on MI355X (BASELINE.json metric).

One step = one batch of B synthetic 256x256 proxy inputs through
    ResNet-50 encoder -> heads -> flow (N pose samples + point estimate) -> SMPL on samples AND point estimate (one launch, M = B*N + B)
all resident in HBM; fp32 throughout (BASELINE configs[1]); outputs stay on the device.
    --eval   BASELINE configs[3] per-GPU step: + T-pose bodies, target bodies from the 3DPW-style labels and the device metric-tracker update
    --train  BASELINE configs[4] per-GPU step: forward under grad (pose/shape NLL + 2D reprojection + global-rotation losses), backward,
             gradient all-reduce (RCCL when WORLD_SIZE > 1), Adam step

    python bench.py                     # 1 GPU, B=32, N=100
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
        bench.py --gpus 8 --steps 20 --warmup 5

How the step is issued: NB >= 3 DISTINCT input batches live in HBM; the step on batch i is captured once into a HIP graph on stream
i % S and the timed region replays the graphs round-robin (every one of the K steps completes inside it; no per-launch events, no
profiler in the timed region).  Per-kernel figures come from a separate pass after the timed region: the same step eagerly on ONE stream
with a HIP event pair around every launch (hmf_profile_*), i.e. kernel EXECUTION time with nothing else on the device.

Prints ONE JSON line (rank 0): throughput, `roofline` of the dominant kernel family (algorithmic FLOPs / execution time, vs the dense
fp32 MFMA peak), and `cpu_baseline` (the oracle in reference mode on a bounded sample, rank 0 at N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic FLOPs per image.sample of the SMPL stage: shape blend 0.413 + pose blend 8.557 +
# 4-nnz skinning blend 0.66 + transform apply 0.145 MFLOP
LBS_FLOP_PER_UNIT = 9.775e6
FLOW_FLOP_PER_UNIT = 1.72e6              # 23 parts x (context layer 0.84 + two hyper-networks 0.87 MFLOP) per sampled row
HBM_PEAK_GBS = 8000.0                  # guides/MI355X_MICROARCH.md: HBM3E ~8 TB/s
FP32_MFMA_PEAK_TFLOPS = 157.3          # guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
BF16_MFMA_PEAK_TFLOPS = 2516.6         # v_mfma_f32_32x32x16_bf16: 16x the k per instruction at half the issue time (the guide's ~2.5 PFLOP/s dense)
WINOGRAD_MFMA_FRACTION = 4.0 / 9.0     # F(2x2,3x3): 16 multiplies per 2x2 outputs instead of 36 -> executed MFMA FLOPs = 4/9 of direct
STEM_POLYPHASE_MFMA_FRACTION = 100.0 / 196.0   # csrc/stem_wino.hip: 100 products per 2x2 outputs and channel instead of 4 * 49


def encoder_conv_flops(enc, B, size, with_bytes=False):
    """Algorithmic FLOPs (2*M*N*K, no padding taps removed) of every convolution launch of the encoder's inference plan, in launch order.
    A fused seam (hmf_conv_pair: block i's last 1x1 + block i+1's first 1x1, `enc._plan['pair']`) is one launch carrying both layers' FLOPs.
    with_bytes: also the ALGORITHMIC HBM bytes of every launch -- each tensor the launch must read (input, filter, identity branch, strided
    source of a fused downsample branch) or write (output; the stem writes only the pooled map) counted once, fp32; a seam does not re-read its
    intermediate tensor.  `roofline.traffic` (PMC) divided by this is the re-read factor."""
    flops, kind, role = [2.0 * B * (size // 2) ** 2 * 64 * enc.conv1.in_channels * 49], ['7x7/2'], [None]
    pooled = (size // 4) ** 2 if getattr(enc, 'fuse_stem_pool', False) else (size // 2) ** 2
    nbytes = [4.0 * (B * size * size * enc.conv1.in_channels + 64 * enc.conv1.in_channels * 49 + B * pooled * 64)]
    hw = size // 4
    bi = 0
    for li in range(1, 5):
        for blk in getattr(enc, 'layer%d' % li):
            stages = [c for c, _ in blk.stages()]
            fused = blk.downsample is not None and enc.fuse_downsample and stages[-1].kernel_size[0] == 1     # conv3 + downsample: one launch
            seq = ([blk.downsample[0]] if (blk.downsample is not None and not fused) else []) + stages
            cur = hw
            for ci, c in enumerate(seq):
                is_ds = blk.downsample is not None and not fused and ci == 0
                inhw = hw if is_ds else cur
                oh = inhw // c.stride[0]
                f = 2.0 * B * oh * oh * c.out_channels * c.in_channels * c.kernel_size[0] ** 2
                nb = B * inhw * inhw * c.in_channels + c.out_channels * c.in_channels * c.kernel_size[0] ** 2 + B * oh * oh * c.out_channels
                if c is stages[-1] and not is_ds and (blk.downsample is None or not fused):
                    nb += B * oh * oh * c.out_channels                        # + the identity branch (block input, or the downsample launch's output)
                if fused and c is stages[-1]:
                    f += 2.0 * B * oh * oh * c.out_channels * blk.downsample[0].in_channels
                    nb += B * oh * oh * blk.downsample[0].in_channels + c.out_channels * blk.downsample[0].in_channels   # the pixels the strided branch reads
                flops.append(f)
                nbytes.append(4.0 * nb)
                kind.append('%dx%d' % (c.kernel_size[0], c.kernel_size[0]))
                role.append((bi, 'first' if c is stages[0] else ('last' if c is stages[-1] else 'mid'), B * oh * oh))
                if not is_ds:
                    cur = oh
            hw = cur
            bi += 1
    plan = getattr(enc, '_plan', None)
    pairs = set(plan['pair']) if plan and plan.get('pair') else set()
    out_f, out_k, out_b = [], [], []
    for f, k, r, nb in zip(flops, kind, role, nbytes):
        if r is not None and r[1] == 'first' and (r[0] - 1) in pairs and r[2] % 64 == 0:      # the plan fuses a seam only on a multiple of 64 pixels
            li_ = max(i for i in range(len(out_f)) if out_k[i].startswith('1x1'))          # the seam launch = the previous block's last 1x1
            out_f[li_] += f
            out_k[li_] = '1x1+1x1 (seam)'
            blk = [b_ for l_ in range(1, 5) for b_ in getattr(enc, 'layer%d' % l_)][r[0]]
            c1 = blk.stages()[0][0]
            out_b[li_] += nb - 4.0 * r[2] * c1.in_channels                                # the seam's second GEMM reads its input from LDS, not HBM
            continue
        out_f.append(f)
        out_k.append(k)
        out_b.append(nb)
    return (out_f, out_k, out_b) if with_bytes else (out_f, out_k)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=6)
    ap.add_argument('--batch', type=int, default=None, help='images per step and GPU (default 32; --train: 72 = TRAIN.BATCH_SIZE)')
    ap.add_argument('--samples', type=int, default=None, help='pose samples per image (default 100; --train: 8 = LOSS.NUM_J2D_SAMPLES)')
    ap.add_argument('--layers', type=int, default=None, help='ResNet depth (default 50; --train: 18, the reference training default)')
    ap.add_argument('--eval', action='store_true', help='BASELINE configs[3] flavour: + T-pose bodies, target bodies, device tracker update')
    ap.add_argument('--train', action='store_true', help='BASELINE configs[4] flavour: one training step (forward under grad, losses, backward, all-reduce, Adam)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='skip the configs[2]/[3]/[4] lines a default single-GPU run appends as `secondary`')
    ap.add_argument('--cpu-images', type=int, default=32, help='images of the bounded CPU-baseline sample')
    ap.add_argument('--blend-dtype', choices=['fp32', 'bf16'], default='fp32',
                    help="'bf16': BASELINE configs[2] (use with --batch 64): SMPL blend-shape GEMM operands in bf16, fp32 accumulate -- a parity-test "
                         'configuration, not the headline line')
    ap.add_argument('--gemm-terms', type=int, choices=[0, 3, 6], default=0,
                    help='OPT-IN flavour, never the headline line: the encoder 1x1 GEMMs on the bf16 matrix pipe with split fp32 operands '
                         '(hmf_conv1x1_gemm_split): 3 = hi/lo pieces, 3 bf16 MFMAs per product (~2^-16 relative per product); 6 = three pieces, 6 MFMAs '
                         '(fp32-level products).  0 (default) = fp32 MFMA everywhere')
    ap.add_argument('--dry-run-cpu', action='store_true', help='exercise only the rank plumbing (gloo, CPU, stand-in step): used by the CPU tests')
    ap.add_argument('--streams', type=int, default=3, help='HIP streams the steps are round-robined over (batches are independent: the '
                    'low-occupancy flow kernels of step i overlap the encoder convolutions of step i+1)')
    ap.add_argument('--batches', type=int, default=3, help='distinct input batches rotated through (>= 3: the inputs do not sit in the 256 MB Infinity Cache)')
    ap.add_argument('--no-graph', action='store_true', help='issue the steps eagerly instead of replaying one captured HIP graph per batch')
    ap.add_argument('--profile-steps', type=int, default=5, help='steps of the per-kernel measurement pass after the timed region (0 = skip)')
    a = ap.parse_args()
    a.batch = a.batch if a.batch is not None else (72 if a.train else 32)
    a.samples = a.samples if a.samples is not None else (8 if a.train else 100)
    a.layers = a.layers if a.layers is not None else (18 if a.train else 50)
    return a


def parity_noise(B, N, seed=0):
    """The base noise both legs of the parity block consume: the reference's draw shapes (models/humaniflow_model.py:258,309), one (N,B,10)
    standard normal for the shape samples and one (B,N,3) per body part."""
    g = torch.Generator().manual_seed(seed)
    return {'shape': torch.randn(N, B, 10, generator=g), 'pose': torch.randn(23, B, N, 3, generator=g)}


def parity_targets(B, seed=11):
    """Teacher-forced targets for the log-prob leg: random rotations (angles up to ~2.5 rad, so some rows have a second pre-image in the
    flow's support) with the reference's edge cases planted: R = I (utils/rigid_transform_utils.py:246-262 zero-angle branch) and theta = pi - 1e-4
    (the near-pi branch of so3_log)."""
    import math
    from oracle import so3 as oso3
    g = torch.Generator().manual_seed(seed)
    ax = torch.randn(B * 23, 3, generator=g, dtype=torch.float64)
    ax = ax / ax.norm(dim=-1, keepdim=True)
    ang = torch.rand(B * 23, 1, generator=g, dtype=torch.float64) * 2.5
    R = oso3.so3_exp(ax * ang).float().view(B, 23, 3, 3)
    R[0, 0] = torch.eye(3)
    R[1 % B, 1] = oso3.so3_exp(torch.tensor([[0.6, 0.0, 0.8]], dtype=torch.float64) * (math.pi - 1e-4))[0].float()
    R[2 % B, 2] = oso3.so3_exp(torch.tensor([[0.0, 1.0, 0.0]], dtype=torch.float64) * 2.0)[0].float()
    ax_g = torch.randn(B, 3, generator=g, dtype=torch.float64)
    glob = oso3.so3_exp(ax_g / ax_g.norm(dim=-1, keepdim=True) * torch.rand(B, 1, generator=g, dtype=torch.float64) * 3.0).float()
    return {'pose_R': R, 'shape': torch.randn(B, 10, generator=g), 'glob_R': glob}


def cpu_baseline(model_sd, cfg, smpl_model_dict, images, N, seed=0):
    """The oracle in reference mode (Python loop over 23 parts, per-call hyper-nets, fp64 exp, smplx-style dense LBS with
    the (M,6890,4,4) temporary) on `images` synthetic inputs x N samples, all host cores.  Returns (cpu_baseline dict, reference outputs of the
    last timed iteration + the teacher-forced log-probs): the second half is what the `parity` block of the JSON line checks the HIP path against."""
    from oracle import model as omodel
    from oracle import smpl as osmpl
    from humaniflow_amd.synthetic import synthetic_proxy_input
    ocfg = omodel.OracleCfg(num_resnet_layers=cfg.MODEL.NUM_RESNET_LAYERS, num_betas=cfg.MODEL.NUM_SMPL_BETAS)
    x = synthetic_proxy_input(images, 256, seed=seed)
    sm = {k: (v.long() if k == 'parents' else v.float()) for k, v in smpl_model_dict.items()}
    noise_full = parity_noise(images, N)
    noise = noise_full
    keep = {}

    def run():
        out = omodel.forward(model_sd, ocfg, input=x, num_samples=N, noise=noise)
        g = out['glob_rotmat']
        vp, jp = osmpl.smpl_forward(sm, out['shape_mode'], out['pose_rotmats_point_est'], g[:, None], pose2rot=False)
        vs, js = osmpl.smpl_forward(sm, out['shape_samples'].reshape(-1, 10), out['pose_rotmats_samples'].reshape(-1, 23, 3, 3),
                                    g[:, None].expand(-1, N, -1, -1).reshape(-1, 1, 3, 3), pose2rot=False)
        keep.update(out=out, verts_point_est=vp, joints_point_est=jp, verts_samples=vs, joints_samples=js)
    # Host-thread choice: the path is thousands of small ops, so "all cores" is NOT the fastest setting on a many-core
    # host (measured on the 256-thread bench box: 16 threads beat 128 by 8x).  Give the baseline its best shot: probe a
    # few thread counts on a 4-image slice, then time the full sample with the winner.
    avail = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    xs_full = x
    best_nt, best_t = None, None
    ns = min(4, images)
    for nt in sorted({min(avail, c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(nt)
        x = xs_full[:ns]
        noise = {'shape': noise_full['shape'][:, :ns].contiguous(), 'pose': noise_full['pose'][:, :ns].contiguous()}
        run()
        t0 = time.perf_counter()
        run()
        dt = time.perf_counter() - t0
        if best_t is None or dt < best_t:
            best_nt, best_t = nt, dt
    torch.set_num_threads(best_nt)
    x, noise = xs_full, noise_full
    run()                                           # warm-up
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        run()
        ts.append(time.perf_counter() - t0)
    t = float(np.median(ts))
    # teacher-forced log-probabilities (losses/humaniflow_loss.py:30-31 call pattern) on the reference features of the same images: not timed
    tg = parity_targets(images)
    ll = omodel.forward(model_sd, ocfg, input_feats=keep['out']['input_feats'], compute_point_est=False, compute_for_loglik=True,
                        shape_for_loglik=tg['shape'], pose_R_for_loglik=tg['pose_R'], glob_R_for_loglik=tg['glob_R'])
    keep.update(x=xs_full, noise=noise_full, targets=tg, flow_context_for_loglik=ll['flow_context_for_loglik'],
                log_prob=omodel.log_prob_SO3(model_sd, ocfg, ll['flow_context_for_loglik'], tg['pose_R'].double()))
    return {'value': images * N / t, 'unit': 'images*samples/s', 'cores': best_nt, 'host_threads_available': avail, 'kind': 'port',
            'sample': '%d images x %d samples (ResNet-50 + 23-part flow loop + smplx-style LBS), best of {8,16,32,64} host threads, median of 3 timed iterations after 1 warm-up, %.2f s/iter'
                      % (images, N, t)}, keep


PARITY_TOL = {'vertex': 1e-4, 'joint': 1e-4, 'logprob': 1e-3}         # BASELINE.json north_star: vertices/joints within 1e-4, log-probs within 1e-3


def parity_block(model, smpl, ref, N, dev):
    """The second half of BASELINE.json's metric ("vertex L-inf vs ref") at the bench's own size: the product path END TO END (hand-written
    encoder -> heads -> flow -> one fused SMPL launch) on the same weights, inputs and base noise as the CPU oracle iteration `ref` that
    cpu_baseline() kept, outside every timed region.  Reference op chain: models/humaniflow_model.py:188-340, models/smpl.py:27-41."""
    o = ref['out']
    B = ref['x'].shape[0]
    cu = lambda d: {k: v.to(dev) for k, v in d.items()}
    with torch.no_grad():
        out = model(ref['x'].to(dev), num_samples=N, noise=cu(ref['noise']), return_input_feats=True)
        pe, sa = smpl.forward_samples_and_point_est(out)
        tg = ref['targets']
        ll = model(None, input_feats=out['input_feats'], compute_point_est=False, compute_for_loglik=True, shape_for_loglik=tg['shape'].to(dev),
                   pose_R_for_loglik=tg['pose_R'].to(dev), glob_R_for_loglik=tg['glob_R'].to(dev))
        lp = model.log_prob_all_parts(ll['flow_context_for_loglik'], tg['pose_R'].to(dev)).cpu()
    linf = lambda a, b: float((a.detach().cpu().double() - b.double()).abs().max())
    fscale = max(1.0, float(o['input_feats'].abs().max()))
    p = {'what': 'product path end to end vs the CPU oracle (reference mode) on the same weights, %d synthetic inputs and base noise; '
                 'log-probs teacher-forced on %d x 23 random target rotations incl. R = I, theta = pi - 1e-4 and theta > pi/2' % (B, B),
         'rows': B * N + B, 'images': B, 'samples_per_image': N,
         'vertex_linf': max(linf(sa.vertices, ref['verts_samples']), linf(pe.vertices, ref['verts_point_est'])),
         'joint_linf': max(linf(sa.joints, ref['joints_samples']), linf(pe.joints, ref['joints_point_est'])),
         'rotmat_linf': max(linf(out['pose_rotmats_samples'], o['pose_rotmats_samples']), linf(out['pose_rotmats_point_est'], o['pose_rotmats_point_est'])),
         'shape_samples_linf': linf(out['shape_samples'], o['shape_samples']),
         'glob_rotmat_linf': linf(out['glob_rotmat'], o['glob_rotmat']), 'cam_wp_linf': linf(out['cam_wp'], o['cam_wp']),
         'encoder_feats_rel_linf': linf(out['input_feats'], o['input_feats']) / fscale,
         'flow_context_linf': linf(ll['flow_context_for_loglik'], ref['flow_context_for_loglik']),
         'logprob_max_abs': linf(lp, ref['log_prob']), 'logprob_rows': int(lp.numel()), 'logprob_finite': bool(torch.isfinite(lp).all()),
         'vertex_abs_max': float(ref['verts_samples'].abs().max()), 'tolerances': dict(PARITY_TOL)}
    p['ok'] = bool(p['vertex_linf'] <= PARITY_TOL['vertex'] and p['joint_linf'] <= PARITY_TOL['joint'] and p['logprob_max_abs'] <= PARITY_TOL['logprob']
                   and p['logprob_finite'])
    return p


_T0 = time.time()


def log(msg):
    if int(os.environ.get('RANK', '0')) == 0:
        print('[bench %7.1fs] %s' % (time.time() - _T0, msg), file=sys.stderr, flush=True)


# ---- multi-rank scaffolding (one process per GPU), shared by the real run and by --dry-run-cpu (gloo; tests/test_distributed.py) ----
def dist_setup(dev, backend):
    """RANK / WORLD_SIZE / MASTER_* come from the launcher (torch.distributed.run).  Returns (dist module or None, world, rank)."""
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    if world == 1:
        return None, 1, 0
    import torch.distributed as dist
    if backend == 'nccl':
        dist.init_process_group('nccl', device_id=dev)          # RCCL on ROCm
    else:
        dist.init_process_group(backend)
    return dist, world, rank


def dist_sync(dist, dev):
    """barrier + device synchronise on both sides of the timed region."""
    if dev.type == 'cuda':
        torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        if dev.type == 'cuda':
            torch.cuda.synchronize()


def dist_elapsed(t0, dist, dev):
    """Elapsed time of the timed region: sync + barrier, then the MAX over ranks."""
    dist_sync(dist, dev)
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def dist_gather_metrics(vec, dist, world):
    """The one exchange step of the data-parallel path: every rank's metric vector, all-gathered (RCCL over xGMI on the GPU box)."""
    if dist is None:
        return vec[None]
    out = torch.empty(world * vec.numel(), dtype=vec.dtype, device=vec.device)
    dist.all_gather_into_tensor(out, vec)
    return out.reshape(world, -1)


def dry_run_cpu(args):
    """The rank plumbing of main() with a stand-in CPU step (no model, no HIP): what tests/test_distributed.py launches with gloo ranks."""
    dev = torch.device('cpu')
    dist, world, rank = dist_setup(dev, 'gloo')
    check_ranks(args, world, rank)
    B, N = args.batch, args.samples
    step = lambda: torch.full((4,), float(rank + 1), dtype=torch.float64) * torch.randn(1000).square().mean().clamp(1, 1)
    for _ in range(args.warmup):
        step()
    dist_sync(dist, dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        vec = step()
    gathered = dist_gather_metrics(vec, dist, world)
    elapsed = dist_elapsed(t0, dist, dev)
    if rank == 0:
        check_ranks(args, world, rank, int(gathered.shape[0]))
        print(json.dumps({'metric': 'images*samples/sec (B=%d, N=%d)' % (B, N), 'value': world * B * N * args.steps / elapsed, 'unit': 'images*samples/s',
                          'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
                          'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dry_run': True,
                          'rccl_ranks_seen': int(gathered.shape[0]), 'gathered_rank_sums': gathered.sum(dim=1).tolist()}))
    if dist is not None:
        dist.destroy_process_group()


KERNEL_TAGS = {1: 'k_lbs_mfma', 12: 'k_lbs_mfma16', 2: 'k_rigid_chain', 3: 'k_joints', 4: 'k_tpose', 5: 'k_flow_ctx0', 6: 'k_flow_chain', 7: 'k_isgc', 15: 'k_heads_fc1 + k_heads_out'}
CONV_TAGS = {9: 'k_stem_wino | k_stem_conv (7x7/2)', 10: 'k_conv_patch (3x3)', 8: 'k_conv_gemm + k_conv_pair (1x1)', 11: 'k_conv_winograd (3x3)'}


def collect_profile(lib, _lib, cap=65536):
    ms = np.zeros(cap)
    tag = np.zeros(cap, np.int32)
    rows = np.zeros(cap, np.int32)
    n = lib.hmf_profile_collect(ms.ctypes.data_as(_lib.c_double_p), _lib.iptr(tag), _lib.iptr(rows), cap)
    return ms[:n], tag[:n], rows[:n]


def conv_families(ms_, tag_, reps, conv_flops, conv_bytes=None, stem_polyphase=False):
    """Per-family table of the encoder convolutions from `reps` recorded encoder passes (one record per launch, plan order)."""
    csel = (tag_ >= 8) & (tag_ <= 11)
    if csel.sum() != reps * len(conv_flops) or reps == 0:
        return None
    cms = ms_[csel].reshape(reps, len(conv_flops)).mean(0)                      # ms per launch, per layer
    ctag = tag_[csel].reshape(reps, len(conv_flops))[0]
    fam = {}
    for i, (f, t, tg) in enumerate(zip(conv_flops, cms, ctag)):
        d = fam.setdefault(CONV_TAGS[int(tg)], {'launches_per_step': 0, 'us_per_step': 0.0, 'flop': 0.0, 'algorithmic_bytes_per_step': 0.0})
        d['launches_per_step'] += 1
        d['us_per_step'] += t * 1e3
        d['flop'] += f
        d['algorithmic_bytes_per_step'] += conv_bytes[i] if conv_bytes else 0.0
    for name, d in fam.items():
        d['tflops'] = round(d['flop'] / (d['us_per_step'] * 1e-6) / 1e12, 2)
        if 'winograd' in name:               # what the MFMA pipe executes: 4/9 of the direct-form multiplies
            d['tflops_executed_mfma'] = round(d['tflops'] * WINOGRAD_MFMA_FRACTION, 2)
        if 'stem' in name and stem_polyphase:
            d['tflops_executed_mfma'] = round(d['tflops'] * STEM_POLYPHASE_MFMA_FRACTION, 2)
        d['us_per_step'] = round(d['us_per_step'], 1)
        d['gflop_per_step'] = round(d.pop('flop') / 1e9, 2)
    return fam


def load_pmc_traffic(source_hash, workload):
    """HBM bytes per launch from the separate `rocprofv3 --pmc` passes summarised by tools/pmc_summary.py.  PMC counters cannot be read
    from inside the process; the file is stamped with the library source hash it was profiled at and is IGNORED when that differs from
    the library being benchmarked, or when it was collected on another workload (batch, samples, blend dtype): no silent stale figure."""
    best = None
    pdir = os.path.join(ROOT, 'profiles')
    for f in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if f.endswith('_pmc.json'):
            try:
                d = json.load(open(os.path.join(pdir, f)))
            except Exception:
                continue
            if isinstance(d, dict) and d.get('lib_source_hash') == source_hash and d.get('workload', 'B=32,N=100,fp32') == workload:
                best = (f, d)
    return best


_STREAMS = {}


def _stream_pool(dev, n):
    pool = _STREAMS.setdefault(dev.index, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=dev))
    return pool[:n]


def run_inference(args, dev, dist, world, rank):
    """One inference flavour (configs[1] by default; --eval = configs[3]; --batch 64 --blend-dtype bf16 = configs[2]): timed region, clock probe,
    per-kernel measurement pass.  Returns (JSON dict on rank 0 else None, what cpu_baseline() needs)."""
    import humaniflow_amd as H
    from humaniflow_amd import _lib, evaluate as ev
    from humaniflow_amd.build import source_hash
    from humaniflow_amd.synthetic import synthetic_proxy_input, init_benchmark_weights, calibrate_benchmark_weights
    lib = _lib.load()
    cfg = H.get_humaniflow_cfg_defaults()
    cfg.MODEL.NUM_RESNET_LAYERS = args.layers
    torch.manual_seed(0)
    model = init_benchmark_weights(H.HumaniflowModel(torch.device('cpu'), cfg.MODEL, H.SMPL_PARENTS), seed=0).eval()
    model = model.to(dev)
    # metre-sized bodies from random weights: heads scaled by 1 / rms(encoder features) of one synthetic batch (same batch on every rank)
    feat_scale = calibrate_benchmark_weights(model, synthetic_proxy_input(8, 256, seed=99).to(dev))
    model_sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()} if rank == 0 else None
    model.image_encoder.gemm_terms = args.gemm_terms
    for kv in filter(None, os.environ.get('HMF_ENC_OPTS', '').split(',')):          # experiments only: e.g. HMF_ENC_OPTS=fuse_stem_pool=0,fuse_pairs=0
        k_, v_ = kv.split('=')
        cur_ = getattr(model.image_encoder, k_)
        val_ = int(v_) if (isinstance(cur_, int) and not isinstance(cur_, bool)) else bool(int(v_))
        setattr(model.image_encoder, k_, val_)
        log('encoder option %s = %s' % (k_, val_))
    smpl = H.SMPL(model_data=H.synthetic_smpl_model(seed=0), batch_size=args.batch, create_transl=False, blend_dtype=args.blend_dtype).to(dev)
    B, N = args.batch, args.samples
    NB = max(1, args.batches)
    S = max(1, args.streams)
    # each rank its own shard of the synthetic set, NB distinct batches per rank
    if args.eval:
        dset = ev.SyntheticEvalSet(world * NB * B, size=256, seed=7)
        batches = [dset.batch(ev.shard_indices(len(dset), rank, world)[i * B:(i + 1) * B], dev) for i in range(NB)]
        trackers = [ev.EvalMetricsTracker(dev, smpl=smpl) for _ in range(NB)]
        eval_side = os.environ.get('HMF_EVAL_SIDE', '0') == '1'        # experiments only
        side_streams = [torch.cuda.Stream(device=dev) if eval_side else None for _ in range(NB)]
    else:
        batches = [{'input': synthetic_proxy_input(B, 256, seed=100 + rank * NB + i).to(dev)} for i in range(NB)]
    torch.manual_seed(1234 + rank)

    def step(i):
        """One step on batch i; returns a small tuple of device tensors (kept alive as the graph's outputs)."""
        with torch.no_grad():
            if args.eval:
                # the target bodies depend on the labels only, so a side stream (a parallel branch of the captured graph) could build them under the
                # encoder -- measured round 5, same box, 60 steps x 3: 3.397 ms with the side branch vs 3.233 ms without (more co-resident kernels evict
                # each other's LDS allocations, as with the split encoder / tail graphs of round 4): off; HMF_EVAL_SIDE=1 turns it on
                cur = torch.cuda.current_stream()
                if eval_side:
                    side_streams[i].wait_stream(cur)
                with torch.cuda.stream(side_streams[i] if eval_side else cur):
                    targ = ev.target_batch(smpl, batches[i], materialise_tpose=False)
                pred = ev.predict_batch(model, smpl, batches[i], N, materialise_tpose=False)       # T-pose metrics from the shape vectors (hmf_smpl_tpose_errors)
                if eval_side:
                    cur.wait_stream(side_streams[i])
                    for t_ in targ.values():
                        t_.record_stream(cur)
                trackers[i].update_per_batch(pred, targ)
                return pred['verts3D_samples'], pred['joints3D_samples'], pred['verts3D'], trackers[i].sums
            pred = model(batches[i]['input'], num_samples=N)
            pe, sa = smpl.forward_samples_and_point_est(pred)
            return sa.vertices, sa.joints, pe.vertices, pred['pose_rotmats_samples']

    streams = _stream_pool(dev, S)                              # one set of streams per process: the secondary flavours reuse the headline's
    use_graph = not args.no_graph
    log('model + SMPL on device; %d distinct batches, %d streams, %s' % (NB, S, 'HIP-graph replay' if use_graph else 'eager'))
    outs = [None] * NB
    for i in range(NB):                                         # eager warm-up of every batch on its stream (builds plans, workspaces)
        with torch.cuda.stream(streams[i % S]):
            outs[i] = step(i)
    torch.cuda.synchronize()
    graphs = [None] * NB
    if use_graph:
        for i in range(NB):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=streams[i % S], capture_error_mode='thread_local'):
                outs[i] = step(i)
            graphs[i] = g
        torch.cuda.synchronize()
        log('captured %d step graphs' % NB)

    def issue(k0, n):
        cur = torch.cuda.current_stream()
        for s_ in streams:
            s_.wait_stream(cur)
        for k in range(k0, k0 + n):
            i = k % NB
            with torch.cuda.stream(streams[i % S]):
                if use_graph:
                    graphs[i].replay()
                else:
                    outs[i] = step(i)
        for s_ in streams:
            cur.wait_stream(s_)

    issue(0, args.warmup)
    dist_sync(dist, dev)
    log('timing %d steps' % args.steps)
    t0 = time.perf_counter()
    issue(args.warmup, args.steps)
    host_issue = time.perf_counter() - t0          # host time to ISSUE the steps (no sync yet): CPU-bound if ~ elapsed
    # the one exchange step of the data-parallel eval path: per-rank metric sums gathered over RCCL/xGMI
    if args.eval:
        metric_vec = torch.stack([t.sums for t in trackers]).sum(0)
    else:
        o = outs[(args.warmup + args.steps - 1) % NB]
        metric_vec = torch.stack([o[0].abs().mean(), o[1].abs().mean(), o[2].abs().mean(), o[3].mean()]).double()
    gathered = dist_gather_metrics(metric_vec, dist, world)
    elapsed = dist_elapsed(t0, dist, dev)
    log('timed region done: %.3f ms/step' % (elapsed / args.steps * 1e3))

    # ---- shader clock under this load (separate pass after the timed region): the same round-robin replay of the step graphs with a one-wave probe
    # kernel on a side stream that reads the shader-clock counter against the constant 100 MHz counter (hmf_profile_clock_probe).  The fp32 MFMA
    # peak of 157.3 TFLOP/s that `roofline.frac` is quoted against assumes 2400 MHz; what the DVFS governor grants under MFMA load is recorded here.
    clock = None
    if args.profile_steps > 0 and not args.dry_run_cpu:
        cbuf = torch.zeros(2, dtype=torch.int64, device=dev)
        side = _stream_pool(dev, S + 1)[S]
        nrep = max(12, args.steps)
        issue(0, 3)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            _lib.check(lib.hmf_profile_clock_probe(int(max(2000, 0.8 * nrep * elapsed / args.steps * 1e6)), cbuf.data_ptr(), _lib.current_stream()), 'clock probe')
        issue(0, nrep)
        torch.cuda.synchronize()
        c_ = cbuf.tolist()
        clock = {'shader_mhz_during_step_replay': round(c_[0] / max(c_[1], 1) * 100.0, 1), 'peak_assumes_mhz': 2400,
                 'how': 'one-wave probe kernel on a side stream for ~80 %% of %d replayed steps: shader-clock counter (s_memtime) delta / 100 MHz counter '
                        '(s_memrealtime) delta on the CU it ran on' % nrep}

    # ---- per-kernel measurement pass: eager, ONE stream, HIP events around every launch (execution time, nothing else running) ----
    conv_flops, conv_kind, conv_bytes = encoder_conv_flops(model.image_encoder, B, 256, with_bytes=True)
    P = args.profile_steps
    kern, conv, stages = {}, None, {}
    lbs_rows = B * N + B
    lbs_ms = float('nan')
    if P > 0:
        lib.hmf_profile_enable(1)
        for k in range(P):
            step(k % NB)
        ms, tag, rows = collect_profile(lib, _lib)
        lib.hmf_profile_enable(0)
        for tg, nm in KERNEL_TAGS.items():
            for r in sorted(set(rows[tag == tg].tolist())):
                sel = (tag == tg) & (rows == r)
                kern['%s[M=%d]' % (nm, r)] = {'us': round(float(ms[sel].mean()) * 1e3, 2), 'launches_per_step': int(sel.sum()) // P}
        conv = conv_families(ms, tag, P, conv_flops, conv_bytes, stem_polyphase=getattr(model.image_encoder, 'stem_polyphase', False))
        ltag = 12 if args.blend_dtype == 'bf16' else 1
        lsel = (tag == ltag) & (rows == lbs_rows)
        lbs_ms = float(ms[lsel].mean()) if lsel.any() else float('nan')

        def timed(fn, reps=5):
            fn()                                            # warm the caching allocator for this call's output sizes
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                r = fn()
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / reps, r
        with torch.no_grad():
            x0 = batches[0]['input']
            stages['encoder_resnet%d' % args.layers], feats = timed(lambda: model(x0, return_input_feats_only=True)['input_feats'])
            stages['heads+flow'], pred = timed(lambda: model(None, input_feats=feats, num_samples=N))
            stages['smpl_samples+point_est'], _ = timed(lambda: smpl.forward_samples_and_point_est(pred))
            stages['smpl_tpose(B*N+B)'], _ = timed(lambda: smpl.tpose(pred['smpl_packed_inputs']['betas']))
            stages['whole_step_one_stream_eager'], _ = timed(lambda: step(0))
        stages = {k: round(v, 3) for k, v in stages.items()}

    if rank == 0:
        value = world * B * N * args.steps / elapsed
        flavour = 'configs[3] eval step' if args.eval else ('configs[1]' if args.blend_dtype == 'fp32' else 'configs[2]')
        if args.gemm_terms:
            flavour += ' with the opt-in split-operand 1x1 GEMMs (gemm_terms=%d)' % args.gemm_terms
        dtype = 'f32' if args.blend_dtype == 'fp32' else 'bf16 blend-shape GEMM operands as hi+lo pairs, 3 bf16 MFMAs per product (f32 accumulate), f32 elsewhere'
        if args.gemm_terms:
            dtype += '; encoder 1x1 GEMM operands as %d bf16 pieces each, %d bf16 MFMAs per product, f32 accumulate' % (2 if args.gemm_terms == 3 else 3, args.gemm_terms)
        res = {
            'metric': 'images*samples/sec (B=%d, N=%d)' % (B, N), 'value': round(value, 1), 'unit': 'images*samples/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': dtype, 'data': 'synthetic',
            'config': {'workload': flavour + ': B=%d synthetic 256x256 18-ch proxy inputs, N=%d samples, ResNet-%d + flow + SMPL %s, random-init weights, '
                                   'synthetic SMPL-shaped body model%s'
                                   % (B, N, args.layers, 'fp32' if args.blend_dtype == 'fp32' else 'fp32 with bf16 blend-shape GEMM operands',
                                      '; + T-pose bodies, target bodies, device metric-tracker update' if args.eval else ''),
                       'weights': 'default PyTorch init (seed 0), random BatchNorm statistics; fc1 / isgc feature columns scaled by 1/rms(encoder features) = %.3e '
                                  '(metre-sized bodies)' % feat_scale,
                       'global_batch': world * B, 'samples_per_image': N, 'parallelism': 'dp%d' % world, 'streams_per_gpu': S,
                       'distinct_input_batches': NB, 'issue': 'HIP graph replay (one captured graph per batch)' if use_graph else 'eager',
                       'exchange': 'one RCCL all_gather of per-rank metric sums per run' if world > 1 else 'none'},
            'rccl_ranks_seen': int(gathered.shape[0]),
            'roofline': None,
            'shader_clock': clock,
            'host_issue_ms_per_step': round(host_issue / args.steps * 1e3, 3),
            'conv_kernels': conv, 'stages_ms': stages, 'kernels_us': kern,
            'measurement_pass': '%d eager steps on one stream after the timed region, HIP event pair around every launch (kernel execution time with nothing '
                                'else on the device); the timed region itself carries no events' % P,
        }
        step_flops = sum(conv_flops) + LBS_FLOP_PER_UNIT * B * (N + 1)
        agg = step_flops / (elapsed / args.steps) / 1e12
        # stride-1 3x3 layers run as Winograd F(2x2,3x3) on maps >= 16 pixels wide and on 8-pixel maps (encoder._conv_hip): 4/9 of the direct-form multiplies
        enc_ = model.image_encoder
        wino_flops = sum(f for f, k in zip(conv_flops, conv_kind) if k == '3x3') - sum(
            2.0 * B * (256 // (4 * 2 ** li)) ** 2 * c.out_channels * c.in_channels * 9
            for li, l in enumerate([enc_.layer1, enc_.layer2, enc_.layer3, enc_.layer4]) for blk in l for c, _ in blk.stages()
            if c.kernel_size[0] == 3 and c.stride[0] == 2)
        stem_flops = conv_flops[0] if getattr(enc_, 'stem_polyphase', False) else 0.0          # the polyphase stem executes 100/196 of its direct-form multiplies
        agg_exec = (step_flops - wino_flops * (1.0 - WINOGRAD_MFMA_FRACTION) - stem_flops * (1.0 - STEM_POLYPHASE_MFMA_FRACTION)) / (elapsed / args.steps) / 1e12
        pmc = load_pmc_traffic(source_hash(), 'B=%d,N=%d,%s%s%s' % (args.batch, args.samples, args.blend_dtype, ',eval' if args.eval else '',
                                                                     ',x%d' % args.gemm_terms if args.gemm_terms else ''))

        def traffic_of(key, largest_grid=False):
            """Mean HBM bytes per launch over the PMC rows of a kernel family.  The rows are keyed by (kernel, grid): `largest_grid` keeps only the
            largest launch of a kernel name (the eval flavour also runs k_lbs_mfma on the M = B target bodies: averaging that small launch in made
            `traffic` smaller than the full launch's own algorithmic bytes)."""
            if pmc is None:
                return None
            keys = key if isinstance(key, tuple) else (key,)
            cand = [(k, v) for k, v in pmc[1].get('kernels', {}).items() if any(kk in k for kk in keys)]
            if largest_grid and cand:
                grid_of = lambda k: int(k.rsplit('[grid=', 1)[1].rstrip(']')) if '[grid=' in k else 0
                gmax = max(grid_of(k) for k, _ in cand)
                cand = [(k, v) for k, v in cand if grid_of(k) == gmax]
            cand = [v for _, v in cand]
            if not cand:
                return None
            return int(sum(c['hbm_bytes_per_launch'] * c['launches'] for c in cand) / sum(c['launches'] for c in cand))
        if conv:
            dom_name = max(conv, key=lambda k: conv[k]['us_per_step'])
            dom = conv[dom_name]
            per_launch_flop = dom['gflop_per_step'] * 1e9 / dom['launches_per_step']
            key = {'k_conv_gemm + k_conv_pair (1x1)': ('k_conv_gemm', 'k_conv_igemm', 'k_conv_pair'), 'k_conv_patch (3x3)': 'k_conv_patch', 'k_conv_winograd (3x3)': 'k_conv_winograd',
                   'k_stem_wino | k_stem_conv (7x7/2)': ('k_stem_wino', 'k_stem_conv')}[dom_name]
            res['roofline'] = {
                'kernel': '%s convolutions of the encoder: %d launches/step, the kernel family with the largest share of the step' % (dom_name, dom['launches_per_step']),
                'bound': 'mfma', 'achieved': dom['tflops'], 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': round(dom['tflops'] / FP32_MFMA_PEAK_TFLOPS, 4),
                'traffic': traffic_of(key),
                'traffic_note': ('mean HBM bytes/launch over this family from separate rocprofv3 --pmc passes (profiles/%s, same library source hash)' % pmc[0]) if pmc
                                else 'null: no profiles/*_pmc.json was collected at this library source hash on this workload',
                'avg_launch_us': round(dom['us_per_step'] / dom['launches_per_step'], 2), 'launches_measured': dom['launches_per_step'] * P,
                'algorithmic_flop_per_launch': per_launch_flop,
                'algorithmic_bytes_per_launch': round(dom['algorithmic_bytes_per_step'] / dom['launches_per_step']),
                'traffic_over_algorithmic_bytes': None,
                'how': 'algorithmic 2*M*N*K FLOPs of the family / its kernel execution time (measurement pass: HIP events on the launch stream, one stream)',
                'step_aggregate': {'what': 'algorithmic FLOPs of every convolution (Winograd layers at direct-form FLOPs) + SMPL blend GEMMs of a step / step time of the timed region',
                                   'achieved': round(agg, 2), 'frac': round(agg / FP32_MFMA_PEAK_TFLOPS, 4), 'peak': FP32_MFMA_PEAK_TFLOPS,
                                   'executed_mfma': {'what': 'the same with the Winograd layers (4/9 of direct form) and the polyphase stem (100/196) at the FLOPs the matrix pipe executes',
                                                     'achieved': round(agg_exec, 2), 'frac': round(agg_exec / FP32_MFMA_PEAK_TFLOPS, 4)}}}
            if args.gemm_terms:
                res['roofline']['step_aggregate']['peak_note'] = ('OPT-IN split-operand flavour: the 1x1 GEMMs run on the bf16 matrix pipe, so `frac` against the fp32 MFMA '
                                                                  'peak is not bounded by 1 here; it is quoted for comparison with the fp32 line only')
            if args.gemm_terms and dom_name.startswith('k_conv_gemm'):
                # the family runs on the bf16 pipe: its ceiling is the dense bf16 peak over the MFMAs spent per product
                pk = BF16_MFMA_PEAK_TFLOPS / args.gemm_terms
                res['roofline'].update({'peak': round(pk, 1), 'frac': round(dom['tflops'] / pk, 4),
                                        'how': res['roofline']['how'] + '; peak = dense bf16 MFMA peak %.0f / %d MFMAs per product' % (BF16_MFMA_PEAK_TFLOPS, args.gemm_terms)})
            if res['roofline']['traffic']:
                res['roofline']['traffic_over_algorithmic_bytes'] = round(res['roofline']['traffic'] / res['roofline']['algorithmic_bytes_per_launch'], 3)
        lbs_bytes = lbs_rows * (82680 + 1080 + 868) + 19.3e6          # vertices + joints written, rotations + betas read, constants once
        if args.blend_dtype == 'fp32':
            ach = LBS_FLOP_PER_UNIT * lbs_rows / (lbs_ms * 1e-3) / 1e12 if lbs_ms == lbs_ms else None
            res['roofline_lbs'] = {'kernel': 'k_lbs_mfma (SMPL pose-blend GEMM + skinning, M=%d rows)' % lbs_rows, 'bound': 'mfma',
                                   'achieved': round(ach, 2) if ach else None, 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                   'frac': round(ach / FP32_MFMA_PEAK_TFLOPS, 4) if ach else None}
        else:                                                          # bf16 operands: 1/16 of the MFMA time -> bound by the vertex write
            ach = lbs_bytes / (lbs_ms * 1e-3) / 1e9 if lbs_ms == lbs_ms else None
            res['roofline_lbs'] = {'kernel': 'k_lbs_mfma16 (SMPL pose-blend GEMM in bf16 + fp32 skinning, M=%d rows)' % lbs_rows, 'bound': 'hbm',
                                   'achieved': round(ach, 1) if ach else None, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                   'frac': round(ach / HBM_PEAK_GBS, 4) if ach else None}
        res['roofline_lbs'].update({'avg_launch_us': round(lbs_ms * 1e3, 2) if lbs_ms == lbs_ms else None, 'traffic': traffic_of('k_lbs_mfma', largest_grid=True),
                                    'algorithmic_flop_per_launch': LBS_FLOP_PER_UNIT * lbs_rows, 'algorithmic_bytes_per_launch': lbs_bytes})
        # K3: the 23-part kinematic loop against the fp32 FMA peak -- expected low: the bound is the depth-8 dependency chain, so the
        # time per dependency wave (round of <= 4 independent parts) is reported beside it
        fl_us = {k.split('[')[0]: v['us'] for k, v in kern.items() if k.startswith(('k_flow_ctx0', 'k_flow_chain', 'k_isgc')) and '[M=' in k}
        if 'k_flow_chain' in fl_us:
            t_flow = sum(fl_us.values()) * 1e-6
            ach = FLOW_FLOP_PER_UNIT * lbs_rows / t_flow / 1e12
            res['roofline_flow'] = {'kernel': 'k_isgc + k_flow_ctx0 + k_flow_chain (23 ancestor-conditioned SO(3) flows in kinematic order, %d rows incl. the point estimate)' % lbs_rows,
                                    'bound': 'fp32 FMA peak nominally; in fact the latency of 8 dependent rounds x ~10 dependent small GEMMs + 2 splines per part',
                                    'achieved': round(ach, 2), 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(ach / FP32_MFMA_PEAK_TFLOPS, 4),
                                    'algorithmic_flop_per_launch_group': FLOW_FLOP_PER_UNIT * lbs_rows, 'us_per_step': round(t_flow * 1e6, 1), 'kernels_us': fl_us,
                                    'us_per_dependency_wave': round(fl_us['k_flow_chain'] / 8.0, 2),
                                    'how': 'SURVEY 8d K3: 1.72 MFLOP per sampled row (0.84 context + 0.87 hyper-networks) x rows / kernel execution time of the three flow kernels '
                                           '(measurement pass, one stream); peak = the fp32 vector/matrix rate of the guide; us_per_dependency_wave = k_flow_chain / 8 rounds'}
        if res['roofline'] is None:
            res['roofline'] = res['roofline_lbs']
        res['cpu_baseline'] = None
        return res, {'model_sd': model_sd, 'cfg': cfg, 'smpl_model': {k: v.detach().cpu() for k, v in smpl.model_dict().items()}, 'model': model, 'smpl': smpl}
    return None, None


SECONDARY = [   # the other BASELINE configurations, timed in-process after the headline so that the driver's one default run carries them
    ('configs2', 'B=64, N=100, bf16 hi+lo blend-shape GEMM operands', {'batch': 64, 'blend_dtype': 'bf16'}),
    ('configs3', 'per-GPU eval step B=32, N=100 (+ T-pose bodies, target bodies, device metric tracker)', {'eval': True}),
    ('configs4', 'per-GPU train step B=72, N=8, ResNet-18 (forward, HumaniflowLoss, backward, Adam)', {'train': True, 'batch': 72, 'samples': 8, 'layers': 18}),
]


def run_secondary(args, dev, steps=20, warmup=6):
    """configs[2] / [3] / [4] on this GPU, `steps` timed steps each with the same barrier + synchronise bracketing as the headline (world = 1).
    Compact lines: value, ms_per_step, workload, dtype and the roofline fraction of the flavour's dominant kernel family."""
    import copy
    import gc
    out = {}
    for key, what, over in SECONDARY:
        a = copy.copy(args)
        a.eval = a.train = False
        a.blend_dtype, a.batch, a.samples, a.layers, a.gemm_terms = 'fp32', 32, 100, 50, 0
        for k, v in over.items():
            setattr(a, k, v)
        a.steps, a.warmup, a.profile_steps, a.no_cpu_baseline = steps, warmup, 2, True
        t0 = time.perf_counter()
        try:
            if a.train:
                from humaniflow_amd.train import bench_train
                r = bench_train(a, dev, None, 1, 0, log, dist_sync, dist_elapsed, dist_gather_metrics)
            else:
                r, _ = run_inference(a, dev, None, 1, 0)
            roof = r.get('roofline') or {}
            line = {'workload': r['config']['workload'], 'metric': r['metric'], 'value': r['value'], 'unit': r['unit'], 'ms_per_step': r['ms_per_step'],
                    'steps': steps, 'warmup': warmup, 'dtype': r['dtype'],
                    'roofline': {k: roof.get(k) for k in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac')}}
            if r.get('roofline_lbs'):
                line['roofline_lbs'] = {k: r['roofline_lbs'].get(k) for k in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'avg_launch_us')}
            if r.get('roofline_flow'):
                line['roofline_flow'] = {k: r['roofline_flow'].get(k) for k in ('bound', 'achieved', 'peak', 'unit', 'frac')}
            if a.train:
                line['train_images_per_s'] = r.get('train_images_per_s')
                line['smpl_work_vs_reference_step'] = r.get('smpl_work_vs_reference_step')
        except Exception as e:                                          # a secondary flavour must not cost the headline line
            line = {'workload': what, 'error': repr(e)}
        line['wall_s'] = round(time.perf_counter() - t0, 1)
        out[key] = line
        log('secondary %s: %s' % (key, {k: line.get(k) for k in ('value', 'ms_per_step', 'wall_s', 'error') if k in line}))
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    return out


def self_launch(args):
    """`python bench.py --gpus N` WITHOUT a launcher (no WORLD_SIZE in the environment) and N > 1: start the N ranks ourselves, exactly as the
    driver's documented form does (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>),
    and return the launcher's exit code.  The alternative -- one rank printing an n_gpus = 1 line under a --gpus N command -- would let a scaling
    run read N = 1 throughput N times."""
    import socket
    import subprocess
    if not args.dry_run_cpu:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            sys.exit('bench.py: --gpus %d but only %d HIP device(s) visible; refusing to fall back to fewer ranks' % (args.gpus, have))
    with socket.socket() as s_:
        s_.bind(('127.0.0.1', 0))
        port = s_.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log('no launcher in the environment: starting %d ranks: %s' % (args.gpus, ' '.join(cmd)))
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '8')
    return subprocess.call(cmd, env=env)


def check_ranks(args, world, rank, seen=None):
    """--gpus N must be what runs: N ranks from the launcher, and (after the exchange step) N rows in the gathered metric tensor."""
    if args.gpus != world:
        sys.exit('bench.py: --gpus %d but the launcher started WORLD_SIZE = %d rank(s)' % (args.gpus, world))
    if seen is not None and rank == 0 and seen != world:
        sys.exit('bench.py: the metric all-gather returned %d rank rows, expected %d' % (seen, world))


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args))
    if args.dry_run_cpu:
        return dry_run_cpu(args)
    local = int(os.environ.get('LOCAL_RANK', '0'))
    assert torch.cuda.is_available(), 'bench.py needs a HIP device (the hot path has no CPU fallback)'
    if local >= torch.cuda.device_count():
        sys.exit('bench.py: LOCAL_RANK %d but only %d HIP device(s) visible' % (local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist, world, rank = dist_setup(dev, 'nccl')
    parity_failed = False
    check_ranks(args, world, rank)
    if args.train:
        from humaniflow_amd.train import bench_train
        res = bench_train(args, dev, dist, world, rank, log, dist_sync, dist_elapsed, dist_gather_metrics)
    else:
        res, ctx = run_inference(args, dev, dist, world, rank)
        headline = not (args.eval or args.gemm_terms or args.blend_dtype != 'fp32' or args.batch != 32 or args.samples != 100 or args.layers != 50)
        if rank == 0 and world == 1 and headline and not args.no_secondary:
            res['secondary'] = run_secondary(args, dev)
            res['secondary_note'] = ('BASELINE configs[2] / [3] / [4] timed in this process after the headline region: %d steps each after %d warm-up steps; '
                                     'their full lines (tools/profile_round.sh) are under profiles/' % (20, 6))
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            log('timing the CPU baseline (oracle, %d images x %d samples)' % (args.cpu_images, args.samples))
            res['cpu_baseline'], ref = cpu_baseline(ctx['model_sd'], ctx['cfg'], ctx['smpl_model'], args.cpu_images, args.samples)
            res['speedup_vs_cpu_baseline'] = round(res['value'] / res['cpu_baseline']['value'], 1)
            # the second half of BASELINE.json's metric: full-size end-to-end parity against the oracle iteration the baseline leg just ran
            res['parity'] = parity_block(ctx['model'], ctx['smpl'], ref, args.samples, dev)
            parity_failed = not res['parity']['ok']
            log('parity (%d rows): vertex Linf %.3g, joint Linf %.3g, rotmat Linf %.3g, log-prob max abs %.3g -> %s'
                % (res['parity']['rows'], res['parity']['vertex_linf'], res['parity']['joint_linf'], res['parity']['rotmat_linf'],
                   res['parity']['logprob_max_abs'], 'ok' if res['parity']['ok'] else 'OUT OF TOLERANCE'))
    if rank == 0:
        check_ranks(args, world, rank, res.get('rccl_ranks_seen'))
        assert res['n_gpus'] == args.gpus
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()
    if parity_failed:
        sys.exit('bench.py: parity out of tolerance (see the "parity" object of the JSON line)')


if __name__ == '__main__':
    main()
