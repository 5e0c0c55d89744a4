"""C-ABI boundary + host logic, no GPU: the library loads, exports every declared symbol, the operand packers
agree with a lane-level emulation of the MFMA kernels, the Python mirror keeps the reference's checkpoint layout."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

import humaniflow_amd as H
from humaniflow_amd import _lib
from helpers import ROOT, init_model_for_tests, oracle_cfg_from
import mfma_emulator as E


def _declared(header):
    return sorted(set(re.findall(r'\b(hmf_[a-zA-Z0-9_]+)\s*\(', open(os.path.join(ROOT, 'include', header)).read())))


def test_library_exports_every_declared_symbol(lib):
    declared = _declared('humaniflow_amd.h')                                     # the drop-in boundary
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), 'symbol %s declared in include/humaniflow_amd.h is not exported' % name
    assert sorted(declared) == sorted(_lib.EXPORTED_SYMBOLS)
    assert lib.hmf_abi_version() == 4


def test_product_library_has_no_timing_variants(lib):
    """The ablation / timing builds of the kernels (they compute wrong results) live in tools/libhumaniflow_amd_experiments.so
    (-DHMF_EXPERIMENTS); the product library neither exports their entry points nor accepts their selector values."""
    out = subprocess.check_output(['nm', '-D', '--defined-only', _lib.LIB_PATH]).decode()
    exported = set(re.findall(r' T (hmf_[a-zA-Z0-9_]+)', out))
    assert exported == set(_lib.EXPORTED_SYMBOLS), exported ^ set(_lib.EXPORTED_SYMBOLS)
    experiments = _declared('humaniflow_amd_experiments.h')
    assert experiments and sorted(experiments) == sorted(_lib._EXPERIMENT_SIGS)
    assert not exported & set(experiments)
    assert not any(s in n for n in exported for s in ('variant', 'debug', 'ablat'))
    buf = (C.c_float * 64)()                                                      # argument validation runs before any launch: no GPU needed
    p = C.cast(buf, C.c_void_p)
    p = C.c_void_p((p.value + 15) & ~15)
    for tile in (1064064, 2064064, 3064064, 4064064, 7064064, 8064064, 128064):
        assert lib.hmf_conv2d_nhwc(p, p, None, None, p, 1, 8, 8, 32, 64, 1, 1, 1, tile, None) == -1 and b'unknown tile' in lib.hmf_last_error()
    assert lib.hmf_conv_pair(p, None, p, p, None, p, p, p, p, 1, 8, 8, 64, 0, 0, 0, 1, 256, 64, 7, None) == -1 and b'variant' in lib.hmf_last_error()
    src = open(os.path.join(ROOT, 'humaniflow_amd', 'encoder.py')).read()
    fwd = src[src.rindex('    def forward(self, x):'):]
    assert 'forward_module(' not in fwd.split('\ndef ')[0].replace('`forward_module`', '')       # no ATen / vendor-library path inside forward


def test_error_reporting_without_device(lib):
    if lib.hmf_device_count() > 0:
        pytest.skip('GPU present')
    smpl = H.SMPL(model_data=H.synthetic_smpl_model())
    hs, keep = smpl.host_struct()
    h = C.c_void_p()
    rc = lib.hmf_smpl_create(C.byref(h), C.byref(hs))
    assert rc == -5 and b'no HIP device' in lib.hmf_last_error()         # HMF_ERR_NO_DEVICE, never a silent fallback
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        smpl(betas=torch.zeros(1, 10), body_pose=torch.zeros(1, 69), global_orient=torch.zeros(1, 3))
    model = H.HumaniflowModel(torch.device('cpu'), H.get_benchmark_cfg().MODEL, H.SMPL_PARENTS).eval()
    with torch.no_grad(), pytest.raises(RuntimeError, match='no CPU fallback'):
        model(torch.zeros(1, 18, 64, 64))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'humaniflow_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, re.M), f
    code = 'import sys; import humaniflow_amd; assert not any(m == "oracle" or m.startswith("oracle.") for m in sys.modules)'
    subprocess.check_call([sys.executable, '-c', code], cwd=ROOT)


def test_checkpoint_layout_matches_reference_contract():
    cfg = H.get_benchmark_cfg()
    m = H.HumaniflowModel(torch.device('cpu'), cfg.MODEL, H.SMPL_PARENTS)
    sd = m.state_dict()
    assert len(sd) == 744
    assert tuple(sd['init_glob'].shape) == (1, 6) and torch.equal(sd['init_glob'], torch.tensor([[1., 0, 0, 1, 0, 0]]))
    assert torch.equal(sd['init_cam'], torch.tensor([0.9, 0.0, 0.0]))
    assert tuple(sd['image_encoder.conv1.weight'].shape) == (64, 18, 7, 7)
    assert tuple(sd['fc_input_shape_glob_cam_feats.weight'].shape) == (256, 2048 + 10 + 9 + 3)
    anc = H.immediate_parent_to_all_ancestors(H.SMPL_PARENTS)
    assert anc[22] == [20, 18, 16, 13, 8, 5, 2] and anc[3] == [0] and anc[0] == []
    for j in range(23):
        assert tuple(sd['fc_flow_context.%d.weight' % j].shape) == (64, 256 + 9 * len(anc[j]))
    for k in range(46):
        shapes = [tuple(sd['pose_so3flow_transform_modules.%d.nn.layers.%d.weight' % (k, i)].shape) for i in range(4)]
        assert shapes == [(64, 65), (32, 64), (32, 32), (62, 32)]
    # strict round trip, as scripts/run_predict.py:75-76 does
    m2 = H.HumaniflowModel(torch.device('cpu'), cfg.MODEL, H.SMPL_PARENTS)
    m2.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    cfg18 = H.get_humaniflow_cfg_defaults()
    assert cfg18.MODEL.NUM_RESNET_LAYERS == 18
    m18 = H.HumaniflowModel(torch.device('cpu'), cfg18.MODEL, H.SMPL_PARENTS)
    assert tuple(m18.state_dict()['fc1.weight'].shape) == (512, 512)


def test_config_mirror():
    cfg = H.get_humaniflow_cfg_defaults()
    assert cfg.MODEL.NORM_FLOW.CONTEXT_DIM == 64 and cfg.MODEL.NORM_FLOW.TRANSFORM_NN_HIDDEN_DIMS == [64, 32, 32]
    assert abs(cfg.MODEL.NORM_FLOW.COMPACT_SUPPORT_RADIUS - 1.5 * np.pi) < 1e-12 and cfg.DATA.PROXY_REP_SIZE == 256
    c2 = cfg.clone()
    c2.merge_from_list(['MODEL.NUM_RESNET_LAYERS', '50', 'TRAIN.LR', '0.001'])
    assert c2.MODEL.NUM_RESNET_LAYERS == 50 and c2.TRAIN.LR == 0.001 and cfg.MODEL.NUM_RESNET_LAYERS == 18
    with pytest.raises(NotImplementedError):                       # pyro's stock transforms are not built
        c3 = cfg.clone()
        c3.MODEL.NORM_FLOW.TRANSFORM_TYPE = 'affine_coupling'
        H.HumaniflowModel(torch.device('cpu'), c3.MODEL, H.SMPL_PARENTS)


def test_flow_variant_checkpoint_layouts():
    """Non-default flows: the transform-module list interleaves [PLU module, coupling] per transform exactly as
    create_conditional_norm_flow builds it (pyro_conditional_norm_flow.py:60-115), so the reference's checkpoint keys line up."""
    cfg = H.get_humaniflow_cfg_defaults()
    c = cfg.clone()
    c.MODEL.NORM_FLOW.PERMUTE_TYPE = 'linear_plu'
    sd = H.HumaniflowModel(torch.device('cpu'), c.MODEL, H.SMPL_PARENTS).state_dict()
    assert tuple(sd['pose_so3flow_transform_modules.0.LU'].shape) == (3, 3) and tuple(sd['pose_so3flow_transform_modules.0.permutation'].shape) == (3, 3)
    assert torch.equal(sd['pose_so3flow_transform_modules.2.permutation'], torch.eye(3)[torch.tensor([1, 2, 0])])       # transform 1: shift by 1
    assert tuple(sd['pose_so3flow_transform_modules.1.nn.layers.0.weight'].shape) == (64, 65) and 'pose_so3flow_transform_modules.91.nn.layers.3.bias' in sd
    c = cfg.clone()
    c.MODEL.NORM_FLOW.PERMUTE_TYPE = 'conditional_linear_plu'
    c.MODEL.NORM_FLOW.PERMUTE_NN_HIDDEN_DIMS = [30, 30]
    c.MODEL.NORM_FLOW.TRANSFORM_TYPE = 'additive_coupling'
    c.MODEL.NORM_FLOW.TRANSFORM_NN_HIDDEN_DIMS = [40, 20]
    c.MODEL.NORM_FLOW.NUM_TRANSFORMS = 3
    sd = H.HumaniflowModel(torch.device('cpu'), c.MODEL, H.SMPL_PARENTS).state_dict()
    assert tuple(sd['pose_so3flow_transform_modules.0.cond_plu_nn.layers.0.weight'].shape) == (30, 64)
    assert tuple(sd['pose_so3flow_transform_modules.0.cond_plu_nn.layers.2.weight'].shape) == (9, 30)
    assert tuple(sd['pose_so3flow_transform_modules.1.nn.layers.0.weight'].shape) == (40, 65) and tuple(sd['pose_so3flow_transform_modules.1.nn.layers.2.weight'].shape) == (2, 20)
    assert 'pose_so3flow_transform_modules.%d.nn.layers.2.bias' % (23 * 6 - 1) in sd and 'pose_so3flow_transform_modules.%d.permutation' % (23 * 6 - 2) in sd


def test_smpl_blend_operand_pack_matches_mfma_lane_semantics(lib):
    smpl = H.SMPL(model_data=H.synthetic_smpl_model(seed=1))
    hs, keep = smpl.host_struct()
    n = lib.hmf_smpl_packed_blend_floats(10)
    packed = np.zeros(n, np.float32)
    assert lib.hmf_smpl_pack_blend_host(C.byref(hs), _lib.fptr(packed)) == 0
    md = smpl.model_dict()
    V = 6890
    Bsrc = np.zeros((224, V, 3), np.float32)
    Bsrc[:207] = md['posedirs'].numpy().reshape(207, V, 3)
    Bsrc[207:217] = md['shapedirs'].numpy().transpose(2, 0, 1)
    Bsrc[217] = md['v_template'].numpy()
    rng = np.random.RandomState(0)
    for (k, v, c) in [(0, 0, 0), (1, 31, 2), (206, 6889, 1), (217, 100, 0), (223, 6889, 2), (55, 3333, 1)]:
        assert packed[lib.hmf_smpl_packed_blend_index(k, v, c)] == Bsrc[k, v, c]
    # replay the k-loop of k_lbs_mfma for two vertex tiles with the emulated MFMA
    pf = rng.randn(32, 224).astype(np.float32)
    blk = packed.reshape(216, 28, 3, 64, 4)
    for vt in (0, 215, 107):
        acc = np.zeros((3, 16, 64), np.float32)
        for kg in range(28):
            bv = blk[vt, kg].transpose(1, 0, 2).reshape(64, 12)          # per lane: 3 float4 = 12 values
            for u in range(4):
                ks = kg * 4 + u
                a = pf[E.L31, 2 * ks + E.Q]
                for c in range(3):
                    E.mfma(a, bv[:, u * 3 + c], acc[c])
        ref = np.einsum('rk,kvc->rvc', pf.astype(np.float64), np.pad(Bsrc, ((0, 0), (0, 6912 - V), (0, 0)))[:, vt * 32:(vt + 1) * 32].astype(np.float64))
        got = np.zeros((32, 32, 3))
        for c in range(3):
            for reg in range(16):
                got[E.D_ROW[reg], E.L31, c] = acc[c, reg]
        assert np.abs(got - ref).max() < 1e-4


def _flow_pack(model, lib):
    hs, keep = model.flow_host_struct()
    n = lib.hmf_flow_packed_floats(C.byref(hs))
    assert n > 0, lib.hmf_last_error()
    buf = np.zeros(n, np.float32)
    assert lib.hmf_flow_pack_host(C.byref(hs), _lib.fptr(buf)) == 0
    return buf


@pytest.mark.parametrize('use_mode', [False, True])
def test_flow_operand_pack_and_kernel_dataflow_match_oracle(lib, use_mode):
    """Replays k_flow_ctx0 + k_flow_chain lane by lane on the C++-packed arena and compares with the oracle's
    reference-mode forward (Python loop over the 23 parts) on the same weights and injected noise."""
    from oracle import model as omodel
    cfg = H.get_humaniflow_cfg_defaults()          # ResNet-18 features (512) keep the CPU test light
    model = init_model_for_tests(H.HumaniflowModel(torch.device('cpu'), cfg.MODEL, H.SMPL_PARENTS), seed=3).eval()
    arena, parts, anc_off, glob = E.parse_flow_pack(_flow_pack(model, lib))
    B, N, nb = 2, 13, 10
    g = torch.Generator().manual_seed(0)
    feats = torch.randn(B, 512, generator=g)
    noise = {'shape': torch.randn(N, B, nb, generator=g), 'pose': torch.randn(23, B, N, 3, generator=g)}
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    ocfg = oracle_cfg_from(cfg.MODEL)
    ref = omodel.forward(sd, ocfg, input_feats=feats, num_samples=N, noise=noise, use_shape_mode_for_samples=use_mode)
    # image part G = W[:, feats|R|cam] . [f, vec(R), cam] + b (k_isgc), fp64 here
    W = sd['fc_input_shape_glob_cam_feats.weight'].double()
    bias = sd['fc_input_shape_glob_cam_feats.bias'].double()
    gc = torch.cat([ref['glob_rotmat'].reshape(B, 9), ref['cam_wp']], 1).double()
    G = (feats.double() @ W[:, :512].T + gc @ W[:, 522:].T + bias).float().numpy()
    assert np.array_equal(arena[glob['o_wisgc']:glob['o_wisgc'] + W.numel()].reshape(W.shape), W.float().numpy())
    # the packer's round schedule: <= 4 independent parts per round, ancestors strictly earlier, 8 rounds for SMPL
    levels = glob['rounds']
    anc = H.immediate_parent_to_all_ancestors(H.SMPL_PARENTS)
    seen = set()
    for rnd in levels:
        assert 1 <= len(rnd) <= 4 and all(set(anc[j]) <= seen for j in rnd)
        seen |= set(rnd)
    assert sorted(seen) == list(range(23)) and len(levels) == 8
    got = {}
    for tile in range((B * N + B + 31) // 32):
        got.update(E.emulate_flow_tile(arena, parts, anc_off, glob, levels, nb, G, ref['shape_mode'].numpy(),
                                       ref['shape_log_std'].numpy(), None if use_mode else noise['shape'].numpy(),
                                       noise['pose'].numpy(), B, N, True, tile, ocfg.radius, ocfg.base_std))
    assert sorted(got) == list(range(B * N + B))
    Rs = ref['pose_rotmats_samples'].reshape(B * N, 23, 9).numpy()
    err_s = max(np.abs(got[m][0] - Rs[m]).max() for m in range(B * N))
    err_p = max(np.abs(got[B * N + b][0] - ref['pose_rotmats_point_est'][b].reshape(23, 9).numpy()).max() for b in range(B))
    err_v = max(np.abs(got[B * N + b][1] - ref['pose_axisangle_point_est'][b].numpy()).max() for b in range(B))
    err_b = max(np.abs(got[m][2] - ref['shape_samples'].reshape(B * N, nb)[m].numpy()).max() for m in range(B * N))
    assert err_s < 2e-5 and err_p < 2e-5 and err_v < 2e-5 and err_b < 1e-6, (err_s, err_p, err_v, err_b)
    # the samples must actually exercise the spline (not all identity / out of support)
    assert np.abs(Rs - np.eye(3).reshape(9)).max() > 0.3


def test_flow_16_row_operand_pack_and_kernel_dataflow_match_oracle(lib):
    """The same replay for k_flow_chain16 (round 4: 16-row tiles on v_mfma_f32_16x16x4_f32): the `o16_*` operand blocks of the C++ packer, the ARL16
    register layout, the spline-parameter placement across lane groups, against the oracle's reference-mode forward."""
    from oracle import model as omodel
    cfg = H.get_humaniflow_cfg_defaults()
    model = init_model_for_tests(H.HumaniflowModel(torch.device('cpu'), cfg.MODEL, H.SMPL_PARENTS), seed=3).eval()
    arena, parts, anc_off, glob = E.parse_flow_pack(_flow_pack(model, lib))
    B, N, nb = 2, 7, 10                                        # 14 sample rows + 2 point-estimate rows = one 16-row tile
    g = torch.Generator().manual_seed(1)
    feats = torch.randn(B, 512, generator=g)
    noise = {'shape': torch.randn(N, B, nb, generator=g), 'pose': torch.randn(23, B, N, 3, generator=g)}
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    ocfg = oracle_cfg_from(cfg.MODEL)
    ref = omodel.forward(sd, ocfg, input_feats=feats, num_samples=N, noise=noise)
    # rows of the tile: samples b * N + n, then the point-estimate rows; C0 = the image/shape half of every context layer (k_flow_ctx0), dense here
    isgc_s = omodel.isgc_feats(sd, feats, ref['shape_samples'], ref['glob_rotmat'], ref['cam_wp']).reshape(B * N, 256)
    isgc_p = omodel.isgc_feats(sd, feats, ref['shape_mode'], ref['glob_rotmat'], ref['cam_wp'])
    hrows = torch.cat([isgc_s, isgc_p])                                                    # (16, 256)
    C0 = np.zeros((16, 23, 64), np.float32)
    for j in range(23):
        W, bias = sd['fc_flow_context.%d.weight' % j], sd['fc_flow_context.%d.bias' % j]
        C0[:, j] = (hrows @ W[:, :256].T + bias).numpy()
    z = np.zeros((23, 16, 3), np.float32)
    z[:, :B * N] = (noise['pose'].reshape(23, B * N, 3) * ocfg.base_std).numpy()
    is_point = np.arange(16) >= B * N
    R, v = E.emulate_flow_tile16(arena, parts, anc_off, glob['rounds'], C0, z, is_point, ocfg.radius)
    Rs = ref['pose_rotmats_samples'].reshape(B * N, 23, 9).numpy()
    assert np.abs(R[:B * N] - Rs).max() < 2e-5
    assert np.abs(R[B * N:] - ref['pose_rotmats_point_est'].reshape(B, 23, 9).numpy()).max() < 2e-5
    assert np.abs(v[B * N:] - ref['pose_axisangle_point_est'].numpy()).max() < 2e-5
    assert np.abs(Rs - np.eye(3).reshape(9)).max() > 0.3


def test_encoder_bn_fold_host_logic():
    """Host side of the encoder's inference plan (no GPU): eval-mode BatchNorm folded into (weight, bias) reproduces bn(conv(x)); the
    downsample branch concatenated along K reproduces conv3 + downsample; the plan is dropped when parameters can have changed."""
    import torch.nn.functional as F
    from humaniflow_amd.encoder import ResNetEncoder
    torch.manual_seed(0)
    enc = ResNetEncoder(50, 18).eval()
    for m in enc.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 2.0); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.2)
    blk = enc.layer2[0]                                              # has a stride-2 downsample branch
    x = torch.randn(2, 256, 12, 12)
    with torch.no_grad():
        w, b, s, p = enc._fold(blk.conv1, blk.bn1)
        assert torch.allclose(F.conv2d(x, w, b, s, p), blk.bn1(blk.conv1(x)), atol=1e-5)
        plan = enc._build_plan()
        stages, ds, dual = plan['blocks'][3]                         # layer2[0]
        assert dual is not None and dual[0].shape == (512, 128 + 256) and dual[2] == 2
        h = torch.randn(2, 128, 6, 6)
        ref = blk.bn3(blk.conv3(h)) + blk.downsample[1](blk.downsample[0](x))
        wcat, bias, stride = dual
        got = F.conv2d(h, wcat[:, :128, None, None]) + F.conv2d(x, wcat[:, 128:, None, None], None, stride) + bias[None, :, None, None]
        assert torch.allclose(got, ref, atol=1e-4)
    enc._plan = plan
    enc.train(); assert enc._plan is None
    enc._plan = plan
    enc.load_state_dict(enc.state_dict()); assert enc._plan is None
    assert len(enc.state_dict()) == 318 and 'layer1.0.downsample.0.weight' in enc.state_dict()     # torchvision names, no fc


def test_round4_entry_points_size_their_workspaces_and_validate_arguments_without_a_gpu(lib):
    """The workspace-size functions of the round-4 entries are pure host arithmetic: the small-launch split (fewer than 128 work-groups: the slab loop of a
    3x3 layer is split over up to 8 work-groups per tile, at most 512 in all), the polyphase stem's halo records, the training loss's partial sums; and
    the entries reject bad arguments before any launch."""
    out = lambda B, H, W, C: 4 * B * H * W * C
    # predict shape (one image): layer2 (2 slabs) / layer3 (4) / layer4 (8 slabs, two images per tile) are split; layer1 (one slab) and B = 32 are not
    assert lib.hmf_conv_winograd_workspace_bytes(1, 32, 32, 128, 128) == 2 * out(1, 32, 32, 128)
    assert lib.hmf_conv_winograd_workspace_bytes(1, 16, 16, 256, 256) == 4 * out(1, 16, 16, 256)
    assert lib.hmf_conv_winograd_workspace_bytes(1, 8, 8, 512, 512) == 8 * out(1, 8, 8, 512)
    assert lib.hmf_conv_winograd_workspace_bytes(1, 64, 64, 64, 64) == 0
    for hw, c in ((64, 64), (32, 128), (16, 256), (8, 512)):
        assert lib.hmf_conv_winograd_workspace_bytes(32, hw, hw, c, c) == 0
    assert lib.hmf_conv_winograd_workspace_bytes(1, 16, 16, 100, 256) == 0            # not a shape the kernel takes
    assert lib.hmf_conv_patch_workspace_bytes(1, 32, 32, 256, 256, 3, 2, 8210) == 4 * out(1, 16, 16, 256)
    assert lib.hmf_conv_patch_workspace_bytes(1, 16, 16, 512, 512, 3, 2, 8210) == 8 * out(1, 8, 8, 512)
    assert lib.hmf_conv_patch_workspace_bytes(32, 32, 32, 256, 256, 3, 2, 8210) == 0
    # polyphase stem: 24 rim positions x 64 channels per 8 x 16 output tile
    assert lib.hmf_stem_wino_pool_workspace_bytes(32, 256, 256) == 32 * 16 * 8 * 24 * 64 * 4
    assert lib.hmf_stem_wino_pool_workspace_bytes(1, 18, 22) == 1 * 2 * 1 * 24 * 64 * 4
    assert lib.hmf_stem_wino_packed_floats() == 4 * 9 * 13 * 64 * 4
    assert lib.hmf_train_loss_workspace_bytes(72) == (4 * 72 + 4) * 4
    buf = (C.c_float * 4096)()
    p = C.c_void_p((C.cast(buf, C.c_void_p).value + 15) & ~15)
    assert lib.hmf_stem_wino(p, p, None, p, 1, 17, 32, 32, 1, None) < 0 and b'18-channel' in lib.hmf_last_error()
    assert lib.hmf_stem_wino_pool(p, p, None, p, 1, 18, 32, 32, None, 0, None) < 0 and b'workspace' in lib.hmf_last_error()
    assert lib.hmf_stem_wino(p, C.c_void_p(p.value + 4), None, p, 1, 18, 32, 32, 1, None) < 0                       # unaligned packed filters
    assert lib.hmf_conv_winograd_ws(p, p, None, None, p, 1, 8, 8, 512, 512, 1, p, 16, None) < 0 and b'workspace' in lib.hmf_last_error()
    assert lib.hmf_train_loss(*([None] * 1), 23, *([None] * 3), 10, *([None] * 7), 2, 3, 256.0, 1.0, 1.0, 1.0, 1.0, 1, *([None] * 8), 0, None) < 0
