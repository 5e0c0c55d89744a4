"""The two drop-in FILE paths of the boundary that no synthetic-model test touched:

  * `SMPL(model_path=...)`: the smplx-layout pickle (README.md:56-59 of the reference: the official SMPL pkl converted to plain numpy / scipy
    objects) -> the device constants of the kernels.  A wrong reshape / transpose of `posedirs` would be invisible to every `model_data=` test.
  * the reference's checkpoint container (`{'best_model_state_dict': ..., ...}`, train/train_humaniflow.py:390-398; loaded strictly by
    scripts/run_predict.py:75-76, run_evaluate.py:64, run_optimise.py:53).

CPU tests for the host logic; one GPU test runs the loaded model through the kernel and compares with the `model_data=` path bit for bit."""
import os
import pickle

import numpy as np
import pytest
import scipy.sparse
import torch

import humaniflow_amd as H
from humaniflow_amd import smpl as hsmpl
from helpers import init_model_for_tests, smpl_oracle_model
from oracle import smpl as osmpl


def _write_pkl(path, syn, num_betas_in_file=10, with_faces=True):
    """A file with the converted SMPL pkl's keys, shapes and container types.  The arrays are filled from their SEMANTICS (not by inverting the
    loader): posedirs[v, c, k] is the displacement of coordinate c of vertex v per unit of pose feature k (SMPL paper eq. 9; smplx body_models.py
    reshapes it to (20670, 207) and transposes), shapedirs[v, c, l] the same per unit of beta l, weights[v, j] the skinning weight of joint j."""
    V = hsmpl.NUM_VERTS
    P = syn['posedirs']                                                  # (207, V*3): row k = pose feature k, column 3 v + c
    posedirs = np.zeros((V, 3, 207), np.float64)
    for c in range(3):
        posedirs[:, c, :] = P[:, c::3].T                                  # column 3 v + c -> [v, c, k]
    shapedirs = np.zeros((V, 3, num_betas_in_file), np.float64)
    shapedirs[:, :, :syn['shapedirs'].shape[2]] = syn['shapedirs']
    if num_betas_in_file > syn['shapedirs'].shape[2]:                      # the official file has 300 shape components; the model uses the first num_betas
        shapedirs[:, :, syn['shapedirs'].shape[2]:] = 7.0
    kintree = np.zeros((2, 24), np.uint32)
    kintree[0] = np.array([2 ** 32 - 1] + hsmpl.SMPL_PARENTS[1:], np.uint32)    # the root's parent is stored as uint32(-1)
    kintree[1] = np.arange(24)
    d = {'v_template': syn['v_template'].astype(np.float64), 'shapedirs': shapedirs, 'posedirs': posedirs,
         'J_regressor': scipy.sparse.csc_matrix(syn['J_regressor'].astype(np.float64)), 'weights': syn['lbs_weights'].astype(np.float64),
         'kintree_table': kintree, 'bs_type': 'lrotmin', 'bs_style': 'lbs', 'J': np.zeros((24, 3))}
    if with_faces:
        d['f'] = np.random.RandomState(0).randint(0, V, (13776, 3)).astype(np.uint32)
    with open(path, 'wb') as f:
        pickle.dump(d, f, protocol=2)


@pytest.fixture(scope='module')
def pkl_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp('smpl')
    syn = H.synthetic_smpl_model(seed=3)
    _write_pkl(str(d / 'SMPL_NEUTRAL.pkl'), syn)
    _write_pkl(str(d / 'SMPL_MALE.pkl'), H.synthetic_smpl_model(seed=4), num_betas_in_file=300, with_faces=False)
    return str(d), syn


def test_load_smpl_pkl_arrays_equal_the_model_data_path(pkl_dir):
    d, syn = pkl_dir
    got = hsmpl.load_smpl_pkl(d, 'neutral', 10)
    for k in ('v_template', 'shapedirs', 'posedirs', 'J_regressor', 'lbs_weights'):
        assert got[k].dtype == np.float32 and got[k].shape == syn[k].shape, k
        assert np.array_equal(got[k], syn[k]), k
    assert got['parents'].tolist() == hsmpl.SMPL_PARENTS and got['faces'].shape == (13776, 3) and got['faces'].dtype == np.int64
    # the file path itself is accepted as well as the directory (smplx: model_path may be either); gender picks the file
    assert np.array_equal(hsmpl.load_smpl_pkl(os.path.join(d, 'SMPL_NEUTRAL.pkl'))['posedirs'], syn['posedirs'])
    male = hsmpl.load_smpl_pkl(d, 'male', 10)
    assert male['shapedirs'].shape == (6890, 3, 10) and np.abs(male['shapedirs']).max() < 1.0 and male['faces'] is None      # first 10 of 300 components
    assert np.array_equal(male['posedirs'], H.synthetic_smpl_model(seed=4)['posedirs'])
    with pytest.raises(FileNotFoundError):
        hsmpl.load_smpl_pkl(d, 'female')


def test_smpl_module_from_pkl_has_the_same_buffers_and_the_pose_blend_semantics(pkl_dir):
    """SMPL(model_path=dir) == SMPL(model_data=...) buffer for buffer, and the loaded pose-blend basis means what the pkl says: the oracle's
    posed-vertex offsets computed from the module's buffers equal a direct contraction of the FILE's (V,3,207) array with the pose feature."""
    d, syn = pkl_dir
    a = H.SMPL(d, batch_size=1, gender='neutral', create_transl=False)
    b = H.SMPL(model_data=syn, create_transl=False)
    for (ka, va), (kb, vb) in zip(a.model_dict().items(), b.model_dict().items()):
        assert ka == kb and torch.equal(va, vb), ka
    assert a.parents.tolist() == hsmpl.SMPL_PARENTS and a.faces_tensor.shape == (13776, 3)
    with open(os.path.join(d, 'SMPL_NEUTRAL.pkl'), 'rb') as f:
        raw = pickle.load(f, encoding='latin1')
    g = torch.Generator().manual_seed(0)
    M = 3
    from oracle.so3 import so3_exp
    R = so3_exp(torch.randn(M * 24, 3, generator=g, dtype=torch.float64) * 0.4).float().view(M, 24, 3, 3)
    betas = torch.randn(M, 10, generator=g)
    om = smpl_oracle_model(a.model_dict())
    v, _ = osmpl.smpl_forward(om, betas, R[:, 1:], R[:, :1], pose2rot=False)
    v0, _ = osmpl.smpl_forward({**om, 'posedirs': torch.zeros_like(om['posedirs'])}, betas, R[:, 1:], R[:, :1], pose2rot=False)
    feat = (R[:, 1:] - torch.eye(3)).reshape(M, 207).double()
    off_file = torch.einsum('vck,mk->mvc', torch.from_numpy(raw['posedirs']), feat)                # rest-pose offsets straight from the file's layout
    # v - v0 = the skinning transforms applied to those offsets: compare in the rest frame by building them from the module's buffer instead
    off_mod = (feat.float() @ a.posedirs).view(M, -1, 3)
    assert (off_mod.double() - off_file).abs().max() < 1e-6
    assert (v - v0).abs().max() > 1e-3                                                           # the basis is not degenerate


@pytest.mark.gpu
def test_smpl_from_pkl_runs_the_kernel_bit_for_bit_like_model_data(pkl_dir):
    d, syn = pkl_dir
    dev = torch.device('cuda:0')
    a = H.SMPL(d, batch_size=4, gender='neutral', create_transl=False).to(dev)
    b = H.SMPL(model_data=syn, create_transl=False).to(dev)
    g = torch.Generator().manual_seed(1)
    M = 70
    from oracle.so3 import so3_exp
    R = so3_exp(torch.randn(M * 24, 3, generator=g, dtype=torch.float64) * 0.5).float().view(M, 24, 3, 3)
    betas = torch.randn(M, 10, generator=g)
    with torch.no_grad():
        oa = a(body_pose=R[:, 1:].to(dev), global_orient=R[:, :1].to(dev), betas=betas.to(dev), pose2rot=False)
        ob = b(body_pose=R[:, 1:].to(dev), global_orient=R[:, :1].to(dev), betas=betas.to(dev), pose2rot=False)
    assert torch.equal(oa.vertices, ob.vertices) and torch.equal(oa.joints, ob.joints) and oa.joints.shape == (M, 90, 3)
    v_ref, j_ref = osmpl.smpl_forward(smpl_oracle_model(a.model_dict()), betas, R[:, 1:], R[:, :1], pose2rot=False)
    assert (oa.vertices.cpu() - v_ref).abs().max() < 1e-4 and (oa.joints.cpu() - j_ref).abs().max() < 1e-4


def _cfg(layers, variant=None):
    cfg = H.get_humaniflow_cfg_defaults()
    cfg.MODEL.NUM_RESNET_LAYERS = layers
    if variant == 'conditional_linear_plu':
        cfg.MODEL.NORM_FLOW.PERMUTE_TYPE = 'conditional_linear_plu'
        cfg.MODEL.NORM_FLOW.PERMUTE_NN_HIDDEN_DIMS = [30, 30]
    return cfg


@pytest.mark.parametrize('layers,variant', [(18, None), (50, None), (18, 'conditional_linear_plu')])
def test_checkpoint_round_trip_in_the_reference_container(tmp_path, layers, variant):
    """torch.save({'epoch', 'best_epoch', 'model_state_dict', 'best_model_state_dict', 'optimiser_state_dict', ...}) as train/train_humaniflow.py:390-398
    writes it -> torch.load(map_location) -> load_state_dict(checkpoint['best_model_state_dict'], strict=True) as scripts/run_predict.py:75-76 reads it."""
    cfg = _cfg(layers, variant)
    m = init_model_for_tests(H.HumaniflowModel(torch.device('cpu'), cfg.MODEL, H.SMPL_PARENTS), seed=11)
    opt = torch.optim.Adam(m.parameters(), lr=cfg.TRAIN.LR)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    path = str(tmp_path / 'humaniflow_weights.tar')
    torch.save({'epoch': 3, 'best_epoch': 2, 'best_epoch_val_metrics': {'PVE-SC': 0.1}, 'model_state_dict': m.state_dict(),
                'best_model_state_dict': sd, 'optimiser_state_dict': opt.state_dict()}, path)
    ck = torch.load(path, map_location=torch.device('cpu'))
    m2 = H.HumaniflowModel(torch.device('cpu'), cfg.MODEL, H.SMPL_PARENTS)
    res = m2.load_state_dict(ck['best_model_state_dict'], strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    sd2 = m2.state_dict()
    assert list(sd2) == list(sd) and all(torch.equal(sd2[k], sd[k]) for k in sd)
    # resume path (scripts/run_train.py:104-107)
    m2.load_state_dict(ck['model_state_dict'])
    torch.optim.Adam(m2.parameters(), lr=cfg.TRAIN.LR).load_state_dict(ck['optimiser_state_dict'])
    # a checkpoint of another depth must be refused by strict loading, not silently half-loaded
    other = H.HumaniflowModel(torch.device('cpu'), _cfg(50 if layers == 18 else 18).MODEL, H.SMPL_PARENTS)
    with pytest.raises(RuntimeError):
        other.load_state_dict(ck['best_model_state_dict'], strict=True)
    assert ('pose_so3flow_transform_modules.0.cond_plu_nn.layers.0.weight' in sd) == (variant == 'conditional_linear_plu')


@pytest.mark.gpu
def test_loaded_checkpoint_drives_the_kernels(tmp_path):
    """load_state_dict AFTER the model has run on the device: the packed flow arena and the folded encoder follow the loaded weights."""
    dev = torch.device('cuda:0')
    cfg = _cfg(18)
    a = init_model_for_tests(H.HumaniflowModel(torch.device('cpu'), cfg.MODEL, H.SMPL_PARENTS), seed=21).eval().to(dev)
    b = init_model_for_tests(H.HumaniflowModel(torch.device('cpu'), cfg.MODEL, H.SMPL_PARENTS), seed=22).eval().to(dev)
    x = torch.rand(2, 18, 128, 128, generator=torch.Generator().manual_seed(0)).to(dev)
    g = torch.Generator().manual_seed(1)
    noise = {'shape': torch.randn(4, 2, 10, generator=g).to(dev), 'pose': torch.randn(23, 2, 4, 3, generator=g).to(dev)}
    with torch.no_grad():
        ra = a(x, num_samples=4, noise=noise)
        rb0 = b(x, num_samples=4, noise=noise)                           # b has packed ITS weights by now
        path = str(tmp_path / 'w.tar')
        torch.save({'best_model_state_dict': a.state_dict()}, path)
        b.load_state_dict(torch.load(path, map_location=dev)['best_model_state_dict'], strict=True)
        rb = b(x, num_samples=4, noise=noise)
    assert (ra['pose_rotmats_samples'] - rb0['pose_rotmats_samples']).abs().max() > 1e-3
    for k in ('pose_rotmats_samples', 'pose_rotmats_point_est', 'shape_mode', 'cam_wp', 'glob_rotmat'):
        assert torch.equal(ra[k], rb[k]), k
