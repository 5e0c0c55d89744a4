"""GPU parity of the stage between the encoder and the flow: global average pool in the last convolution's epilogue, the hand-written heads
(models/humaniflow_model.py:232-253) vs oracle.model.heads, the base noise drawn INSIDE the flow kernels vs torch.randn on the same generator
in the reference's draw order (models/humaniflow_model.py:258, :309; SURVEY 8b "RNG contract"), and the packed SMPL launch."""
import ctypes as C

import pytest
import torch

import humaniflow_amd as H
from humaniflow_amd import _lib
from helpers import init_model_for_tests, oracle_cfg_from
from oracle import model as omodel

pytestmark = pytest.mark.gpu


def _model(layers, seed=3):
    cfg = H.get_humaniflow_cfg_defaults()
    cfg.MODEL.NUM_RESNET_LAYERS = layers
    m = init_model_for_tests(H.HumaniflowModel(torch.device('cpu'), cfg.MODEL, H.SMPL_PARENTS), seed=seed).eval()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    return m.to('cuda:0'), sd, cfg


@pytest.mark.parametrize('layers,B', [(50, 32), (50, 1), (50, 45), (18, 7), (18, 72)])
def test_heads_kernel_matches_oracle_heads(layers, B):
    model, sd, cfg = _model(layers)
    F = 512 if layers == 18 else 2048
    feats = torch.randn(B, F, generator=torch.Generator().manual_seed(B)) * 0.7
    cam, glob_R, mode, lstd = omodel.heads(sd, oracle_cfg_from(cfg.MODEL), feats)
    with torch.no_grad():
        out = model(None, input_feats=feats.cuda(), compute_point_est=False)
    for k, ref in (('cam_wp', cam), ('glob_rotmat', glob_R), ('shape_mode', mode), ('shape_log_std', lstd)):
        assert out[k].shape == ref.shape and (out[k].cpu() - ref).abs().max() < 1e-5, k
    d = out['shape_dist_for_loglik']
    assert (d.loc.cpu() - mode).abs().max() < 1e-5 and (d.scale.cpu() - torch.exp(lstd)).abs().max() < 1e-5
    # the torch-op heads (the differentiable path of the training step) agree with the kernel
    model.heads_kernel = False
    with torch.no_grad():
        out2 = model(None, input_feats=feats.cuda(), compute_point_est=False)
    for k in ('cam_wp', 'glob_rotmat', 'shape_mode', 'shape_log_std'):
        assert (out[k] - out2[k]).abs().max() < 1e-5, k


def test_fused_average_pool_equals_mean_of_the_map():
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    for B, K, N in ((32, 512, 2048), (3, 64, 128), (5, 256, 64)):
        x = torch.randn(B, 8, 8, K, generator=g).cuda()
        w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
        bias = torch.randn(N, generator=g).cuda()
        res = torch.randn(B, 8, 8, N, generator=g).cuda()
        y = torch.empty(B, 8, 8, N, device='cuda')
        y2 = torch.empty_like(y)
        pooled, pooled2 = torch.empty(B, N, device='cuda'), torch.empty(B, N, device='cuda')
        st = _lib.current_stream()
        _lib.check(lib.hmf_conv1x1_gemm(x.data_ptr(), None, w.data_ptr(), bias.data_ptr(), res.data_ptr(), y.data_ptr(), B, 8, 8, K, 0, 0, 0, 1, N, 1, 0, st))
        _lib.check(lib.hmf_conv1x1_gemm_avgpool(x.data_ptr(), w.data_ptr(), bias.data_ptr(), res.data_ptr(), y2.data_ptr(), pooled.data_ptr(), B, 8, 8, K, N, 1, 0, st))
        _lib.check(lib.hmf_conv1x1_gemm_avgpool(x.data_ptr(), w.data_ptr(), bias.data_ptr(), res.data_ptr(), None, pooled2.data_ptr(), B, 8, 8, K, N, 1, 0, st))
        assert torch.equal(y, y2)                                        # same tile, same arithmetic
        assert torch.equal(pooled, pooled2)
        ref = y.double().mean(dim=(1, 2))
        assert (pooled.double() - ref).abs().max() < 2e-6 * max(1.0, ref.abs().max().item())
        p3 = torch.empty(B, N, device='cuda')
        _lib.check(lib.hmf_avgpool_nhwc(y.data_ptr(), B, 64, N, p3.data_ptr(), st))
        assert (p3.double() - ref).abs().max() < 2e-6 * max(1.0, ref.abs().max().item())
    # a map size the fused epilogue is not built for is refused (the plan then pools with hmf_avgpool_nhwc)
    x = torch.randn(2, 4, 4, 64).cuda()
    rc = lib.hmf_conv1x1_gemm_avgpool(x.data_ptr(), w[:, :64].contiguous().data_ptr(), None, None, None, pooled.data_ptr(), 2, 4, 4, 64, 64, 1, 0, st)
    assert rc == -2


def _reference_draws(B, N, nb, draw_shape=True):
    """What the reference's forward draws from the device generator, call for call (models/humaniflow_model.py:258, :309)."""
    shape = torch.randn(N, B, nb, device='cuda') if draw_shape else None
    pose = torch.stack([torch.randn(B, N, 3, device='cuda') for _ in range(23)])
    return shape, pose


@pytest.mark.parametrize('B,N', [(2, 5), (32, 100), (1, 10), (3, 4000)])
def test_in_kernel_noise_is_the_device_generators_stream_in_the_reference_order(B, N):
    model, sd, cfg = _model(18)
    torch.manual_seed(1234)
    torch.randn(7, device='cuda')                                   # the generator is somewhere in its stream, not at offset 0
    state = torch.cuda.get_rng_state()
    ref_shape, ref_pose = _reference_draws(B, N, 10)
    after_ref = torch.randn(5, device='cuda')
    torch.cuda.set_rng_state(state)
    eps_shape, eps_pose = model.draw_noise(torch.device('cuda:0'), B, N)
    after = torch.randn(5, device='cuda')
    assert torch.equal(after, after_ref)                            # the generator was advanced exactly as the reference's 24 calls advance it
    # part j's noise is the j-th (B,N,3) draw; bit-equal when this build's device math library matches torch's, else to the last ulp
    exact = torch.equal(eps_pose, ref_pose) and torch.equal(eps_shape, ref_shape)
    assert exact or ((eps_pose - ref_pose).abs().max() < 1e-6 and (eps_shape - ref_shape).abs().max() < 1e-6)
    for j in (0, 11, 22):
        assert (eps_pose[j] - ref_pose[j]).abs().max() < 1e-6
    # use_shape_mode_for_samples: the shape call is skipped, the parts follow immediately
    torch.cuda.set_rng_state(state)
    _, ref_pose2 = _reference_draws(B, N, 10, draw_shape=False)
    torch.cuda.set_rng_state(state)
    s2, p2 = model.draw_noise(torch.device('cuda:0'), B, N, draw_shape=False)
    assert s2 is None and (p2 - ref_pose2).abs().max() < 1e-6


@pytest.mark.parametrize('B,N', [(2, 5), (32, 100)])
def test_in_kernel_noise_equals_the_oracle_restatement_of_the_generator(B, N):
    """The third leg: csrc/rng_philox.h vs oracle/rng.py (numpy Philox4x32-10 pinned by the Random123 known-answer vectors + rocrand's Box-Muller
    in float32) from the generator's (seed, offset) -- independent of torch's own kernel."""
    from oracle import rng as orng
    model, sd, cfg = _model(18)
    torch.manual_seed(99)
    torch.randn(3, device='cuda')
    gen = torch.cuda.default_generators[0]
    seed, off = gen.initial_seed(), gen.get_offset()
    props = torch.cuda.get_device_properties(0)
    cap = props.multi_processor_count * (props.max_threads_per_multi_processor // 256)
    eps_shape, eps_pose = model.draw_noise(torch.device('cuda:0'), B, N)
    ref_shape, ref_pose, off_after = orng.flow_noise(seed, off, B, N, 10, cap)
    assert gen.get_offset() == off_after
    assert (eps_shape.cpu().numpy() - ref_shape).__abs__().max() < 2e-6 and (eps_pose.cpu().numpy() - ref_pose).__abs__().max() < 2e-6


@pytest.mark.parametrize('layers,B,N,use_mode', [(18, 3, 17, False), (50, 2, 100, False), (18, 4, 9, True)])
def test_seeded_forward_equals_forward_with_the_reference_draws_injected(layers, B, N, use_mode):
    model, sd, cfg = _model(layers)
    F = 512 if layers == 18 else 2048
    feats = (torch.randn(B, F, generator=torch.Generator().manual_seed(5)) * 0.5).cuda()
    torch.manual_seed(77)
    state = torch.cuda.get_rng_state()
    with torch.no_grad():
        out = model(None, input_feats=feats, num_samples=N, use_shape_mode_for_samples=use_mode)       # noise drawn inside the kernels
    after = torch.randn(3, device='cuda')
    torch.cuda.set_rng_state(state)
    shape, pose = _reference_draws(B, N, 10, draw_shape=not use_mode)
    after_ref = torch.randn(3, device='cuda')
    assert torch.equal(after, after_ref)
    with torch.no_grad():
        ref = model(None, input_feats=feats, num_samples=N, use_shape_mode_for_samples=use_mode, noise={'shape': shape, 'pose': pose})
    assert (out['shape_samples'] - ref['shape_samples']).abs().max() < 1e-6
    assert (out['pose_rotmats_samples'] - ref['pose_rotmats_samples']).abs().max() < 2e-5
    assert torch.equal(out['pose_rotmats_point_est'], ref['pose_rotmats_point_est'])
    # and against the oracle with the same draws
    o = omodel.forward(sd, oracle_cfg_from(cfg.MODEL), input_feats=feats.cpu(), num_samples=N, use_shape_mode_for_samples=use_mode,
                       noise={'shape': None if shape is None else shape.cpu(), 'pose': pose.cpu()})
    assert (out['pose_rotmats_samples'].cpu() - o['pose_rotmats_samples']).abs().max() < 1e-4


def test_variant_flow_draws_the_same_stream():
    cfg = H.get_humaniflow_cfg_defaults()
    cfg.MODEL.NUM_RESNET_LAYERS = 18
    m = init_model_for_tests(H.HumaniflowModel(torch.device('cpu'), cfg.MODEL, H.SMPL_PARENTS), seed=3).eval()
    m.force_generic_flow = True
    m = m.to('cuda:0')
    feats = (torch.randn(2, 512, generator=torch.Generator().manual_seed(5)) * 0.5).cuda()
    torch.manual_seed(3)
    state = torch.cuda.get_rng_state()
    with torch.no_grad():
        out = m(None, input_feats=feats, num_samples=6)
    torch.cuda.set_rng_state(state)
    shape, pose = _reference_draws(2, 6, 10)
    with torch.no_grad():
        ref = m(None, input_feats=feats, num_samples=6, noise={'shape': shape, 'pose': pose})
    assert (out['pose_rotmats_samples'] - ref['pose_rotmats_samples']).abs().max() < 2e-5


def test_captured_step_draws_fresh_noise_on_every_replay():
    model, sd, cfg = _model(18)
    feats = (torch.randn(4, 512, generator=torch.Generator().manual_seed(5)) * 0.5).cuda()
    torch.manual_seed(5)
    with torch.no_grad():
        model(None, input_feats=feats, num_samples=8)                      # eager pass: creates the device-resident generator state
    s = torch.cuda.Stream()
    with torch.cuda.stream(s), torch.no_grad():
        model(None, input_feats=feats, num_samples=8)                      # workspaces of this stream
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(g, stream=s, capture_error_mode='thread_local'):
        out = model(None, input_feats=feats, num_samples=8)
    seen = []
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        seen.append(out['pose_rotmats_samples'].clone())
        assert torch.isfinite(seen[-1]).all()
    assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2])
    with torch.no_grad():
        assert torch.equal(out['pose_rotmats_point_est'], model(None, input_feats=feats, num_samples=0)['pose_rotmats_point_est'])
    # replaying from a re-based state reproduces the eager pass from the same generator state
    torch.manual_seed(9)
    model.reseed_captured_noise('cuda:0')
    g.replay()
    torch.cuda.synchronize()
    replayed = out['pose_rotmats_samples'].clone()
    torch.manual_seed(9)
    with torch.no_grad():
        eager = model(None, input_feats=feats, num_samples=8)['pose_rotmats_samples']
    assert torch.equal(replayed, eager)


def test_pass_captured_after_eager_warmups_continues_the_stream_instead_of_repeating_it():
    """Advisor finding (round 4): the device-resident generator state captured passes draw from was written by the FIRST eager pass only, so
    a capture after k warm-up passes replayed the noise of warm-up passes 2..k.  Every eager pass now leaves the generator's state after
    itself on the device: 3 eager passes, capture, 2 replays = passes 4 and 5 of the reference's stream; and after
    sync_generator_with_captured_noise() torch's own generator continues with pass 6."""
    model, sd, cfg = _model(18)
    B, N, nb = 3, 7, 10
    same = lambda got, ref: (got - ref).abs().max().item() < 1e-6           # bit-equal when the device math library matches torch's, else to the last ulp
    torch.manual_seed(21)
    want = [_reference_draws(B, N, nb) for _ in range(6)]                  # the reference's 1 + 23 normal_() calls, six passes in a row
    torch.manual_seed(21)
    for k in range(3):
        es, ep = model.draw_noise('cuda:0', B, N)
        assert same(es, want[k][0]) and same(ep, want[k][1])
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s, capture_error_mode='thread_local'):
        es, ep = model.draw_noise('cuda:0', B, N)
    for k in (3, 4):
        g.replay()
        torch.cuda.synchronize()
        assert same(es, want[k][0]) and same(ep, want[k][1]), 'replay %d repeats or skips part of the stream' % (k - 2)
    model.sync_generator_with_captured_noise('cuda:0')
    es6, ep6 = _reference_draws(B, N, nb)                                  # torch.randn itself, continuing after the replays
    assert torch.equal(es6, want[5][0]) and torch.equal(ep6, want[5][1])   # torch.randn vs torch.randn: exact
    # ... and an eager pass after further replays + sync continues too
    g.replay()
    torch.cuda.synchronize()
    seen = ep.clone()
    model.sync_generator_with_captured_noise('cuda:0')
    _, ep8 = model.draw_noise('cuda:0', B, N)
    assert not same(ep8, seen)


def test_packed_smpl_launch_equals_the_two_reference_calls():
    model, sd, cfg = _model(18)
    smpl = H.SMPL(model_data=H.synthetic_smpl_model(seed=0), create_transl=False).to('cuda:0')
    B, N = 3, 21
    feats = (torch.randn(B, 512, generator=torch.Generator().manual_seed(2)) * 0.5).cuda()
    with torch.no_grad():
        pred = model(None, input_feats=feats, num_samples=N)
        pe, sa = smpl.forward_samples_and_point_est(pred)
        g = pred['glob_rotmat']
        pe2 = smpl(body_pose=pred['pose_rotmats_point_est'], global_orient=g[:, None], betas=pred['shape_mode'], pose2rot=False)
        sa2 = smpl(body_pose=pred['pose_rotmats_samples'].reshape(-1, 23, 3, 3), global_orient=g[:, None].expand(-1, N, -1, -1).reshape(-1, 1, 3, 3),
                   betas=pred['shape_samples'].reshape(-1, 10), pose2rot=False)
    for a, b in ((pe, pe2), (sa, sa2)):
        assert torch.equal(a.vertices, b.vertices) and torch.equal(a.joints, b.joints)
        assert torch.equal(a.full_pose, b.full_pose.reshape(a.full_pose.shape)) and torch.equal(a.global_orient, b.global_orient)
