"""SMPL body model with the reference's wrapper API, running on the HIP kernels.

Mirror of models/smpl.py:13-41 (class ``SMPL(smplx.SMPL)``: forces vertices, appends the three extra
joint regressors -> 90 joints) and of the smplx.SMPL pieces its callers rely on:

    smpl = SMPL(model_path, batch_size=B, gender='neutral', num_betas=10, create_transl=False).to(device)
    smpl.parents.tolist()                                   # scripts/run_predict.py:74
    out = smpl(body_pose=R (M,23,3,3), global_orient=R0 (M,1,3,3), betas=(M,10), pose2rot=False)
    out = smpl(body_pose=(M,69), global_orient=(M,3), betas=(M,10))          # axis-angle, pose2rot=True
    out.vertices (M,6890,3), out.joints (M,90,3), out.betas, out.global_orient, out.body_pose, out.full_pose

All arithmetic happens in libhumaniflow_amd (hmf_smpl_lbs / hmf_smpl_tpose / hmf_batch_rodrigues);
calling it with CPU tensors raises -- there is no CPU fallback.  The SMPL ``.pkl`` is licensed and not
shipped: give ``model_path`` (directory or file, smplx layout) or ``model_data`` (dict of arrays, e.g.
``synthetic_smpl_model()``).
"""
import ctypes as C
import os
import pickle
from collections import namedtuple

import numpy as np
import torch
from torch import nn

from . import _lib

NUM_VERTS = 6890
SMPL_PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]
# smplx VertexJointSelector order (face, feet, left-hand tips, right-hand tips), 'smplh' vertex ids
EXTRA_VERTEX_IDS = [332, 6260, 2800, 4071, 583, 3216, 3226, 3387, 6617, 6624, 6787,
                    2746, 2319, 2445, 2556, 2673, 6191, 5782, 5905, 6016, 6133]

SMPLOutput = namedtuple('SMPLOutput', ['vertices', 'joints', 'full_pose', 'betas', 'global_orient', 'body_pose'])
SMPLOutput.__new__.__defaults__ = (None,) * 6

_REG_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'model_files', 'extra_joint_regressors.npz')


def load_extra_regressors(path=_REG_FILE):
    """The reference's J_regressor_extra / cocoplus / h36m (models/smpl.py:17-19) as dense float32."""
    z = np.load(path)
    out = {}
    for key, name in (('extra', 'J_regressor_extra'), ('cocoplus', 'J_regressor_cocoplus'), ('h36m', 'J_regressor_h36m')):
        a = np.zeros(tuple(z[key + '_shape']), np.float64)
        a[z[key + '_rows'].astype(np.int64), z[key + '_cols'].astype(np.int64)] = z[key + '_vals']
        out[name] = a.astype(np.float32)
    return out


def synthetic_smpl_model(seed=0, num_betas=10, skin_nnz=4):
    """SMPL-shaped random body model: same array shapes, dtypes and sparsity as the real
    pkl (4 skinning weights per vertex, 8-vertex joint regressor rows), random content."""
    rng = np.random.RandomState(seed)
    V, J = NUM_VERTS, 24
    v_template = (rng.uniform(-1, 1, (V, 3)) * np.array([0.3, 0.9, 0.15])).astype(np.float32)
    shapedirs = (rng.randn(V, 3, num_betas) * 0.01).astype(np.float32)
    posedirs = (rng.randn(207, V * 3) * 0.002).astype(np.float32)
    J_regressor = np.zeros((J, V), np.float32)
    for j in range(J):
        idx = rng.choice(V, 8, replace=False)
        w = rng.dirichlet(np.ones(8))
        J_regressor[j, idx] = w
    lbs_weights = np.zeros((V, J), np.float32)
    for v in range(V):
        idx = rng.choice(J, skin_nnz, replace=False)
        lbs_weights[v, idx] = rng.dirichlet(np.ones(skin_nnz))
    return {'v_template': v_template, 'shapedirs': shapedirs, 'posedirs': posedirs, 'J_regressor': J_regressor,
            'lbs_weights': lbs_weights, 'parents': np.array(SMPL_PARENTS, np.int64)}


def load_smpl_pkl(model_path, gender='neutral', num_betas=10):
    """smplx-layout loader: `model_path` is the .pkl or a directory holding SMPL_{GENDER}.pkl."""
    if os.path.isdir(model_path):
        model_path = os.path.join(model_path, 'SMPL_%s.pkl' % gender.upper())
    if not os.path.exists(model_path):
        raise FileNotFoundError('SMPL model file %s not found (the SMPL pkl is licensed and not shipped; '
                                'pass model_data=synthetic_smpl_model() for benchmarks)' % model_path)
    with open(model_path, 'rb') as f:
        d = pickle.load(f, encoding='latin1')
    dense = lambda a: np.asarray(a.todense() if hasattr(a, 'todense') else a)
    shapedirs = np.asarray(d['shapedirs'])[:, :, :num_betas]
    posedirs = np.asarray(d['posedirs'])
    parents = np.asarray(d['kintree_table'])[0].astype(np.int64).copy()
    parents[0] = -1
    return {'v_template': np.asarray(d['v_template'], np.float32), 'shapedirs': shapedirs.astype(np.float32),
            'posedirs': posedirs.reshape(-1, posedirs.shape[-1]).T.astype(np.float32).copy(),
            'J_regressor': dense(d['J_regressor']).astype(np.float32), 'lbs_weights': np.asarray(d['weights'], np.float32),
            'parents': parents, 'faces': np.asarray(d['f']).astype(np.int64) if 'f' in d else None}


def _rodrigues_torch(rot_vecs):
    """smplx==0.1.26 lbs.batch_rodrigues in torch operators, quirk included (angle = |v + 1e-8|, added per component): the differentiable form of
    hmf_batch_rodrigues for the axis-angle inputs of a SMPL call under autograd.  (N, 3) -> (N, 3, 3)."""
    angle = torch.norm(rot_vecs + 1e-8, dim=1, keepdim=True)
    ax = rot_vecs / angle
    c, s_ = torch.cos(angle)[:, :, None], torch.sin(angle)[:, :, None]
    z = torch.zeros_like(angle[:, 0])
    K = torch.stack([z, -ax[:, 2], ax[:, 1], ax[:, 2], z, -ax[:, 0], -ax[:, 1], ax[:, 0], z], dim=1).view(-1, 3, 3)
    return torch.eye(3, dtype=rot_vecs.dtype, device=rot_vecs.device)[None] + s_ * K + (1.0 - c) * torch.bmm(K, K)


class SMPL(nn.Module):
    NUM_JOINTS = 23
    NUM_BODY_JOINTS = 23

    def __init__(self, model_path=None, batch_size=1, gender='neutral', num_betas=10, create_transl=True,
                 model_data=None, blend_dtype='fp32', **kwargs):
        super().__init__()
        if blend_dtype not in ('fp32', 'bf16', 'bf16_plain'):
            raise ValueError("blend_dtype must be 'fp32', 'bf16' (BASELINE configs[2]: bf16 MFMA blend-shape GEMMs, fp32 accumulate; operands as bf16 hi + lo "
                             "pairs, three MFMAs per product: vertices within 1e-4 m of fp32) or 'bf16_plain' (single bf16 operands: 1.5e-3 m)")
        self.blend_dtype = blend_dtype
        # 64: k_lbs_mfma (128 KB of LDS, one work-group per CU); 32: k_lbs_mfma32 (64 KB, two per CU); 0 (default): 32-row groups for calls of
        # <= 32 bodies (the predict shape, M = 11: a 64-row tile would spend half its MFMAs on dead rows), 64 beyond.  Bitwise the same results.
        self.lbs_rows = int(os.environ.get('HMF_LBS_ROWS', '0'))
        if model_data is None:
            model_data = load_smpl_pkl(model_path, gender, num_betas)
        self.gender = gender
        self.batch_size = batch_size
        self.num_betas = num_betas
        f32 = lambda k: torch.tensor(np.asarray(model_data[k]), dtype=torch.float32)
        self.register_buffer('v_template', f32('v_template'))
        self.register_buffer('shapedirs', f32('shapedirs')[:, :, :num_betas].contiguous())
        self.register_buffer('posedirs', f32('posedirs'))
        self.register_buffer('J_regressor', f32('J_regressor'))
        self.register_buffer('lbs_weights', f32('lbs_weights'))
        self.register_buffer('parents', torch.tensor(np.asarray(model_data['parents']), dtype=torch.long))
        if model_data.get('faces') is not None:
            self.register_buffer('faces_tensor', torch.tensor(np.asarray(model_data['faces']), dtype=torch.long))
        extra = model_data if 'J_regressor_extra' in model_data else load_extra_regressors()
        for k in ('J_regressor_extra', 'J_regressor_cocoplus', 'J_regressor_h36m'):      # models/smpl.py:20-25
            self.register_buffer(k, torch.tensor(np.asarray(extra[k]), dtype=torch.float32))
        # smplx default (learnable) parameters used when an argument is None
        self.betas = nn.Parameter(torch.zeros(batch_size, num_betas))
        self.global_orient = nn.Parameter(torch.zeros(batch_size, 3))
        self.body_pose = nn.Parameter(torch.zeros(batch_size, self.NUM_BODY_JOINTS * 3))
        if create_transl:
            self.transl = nn.Parameter(torch.zeros(batch_size, 3))
        self._handles = {}
        self._ws = {}

    # ---- device-side constants --------------------------------------------------------------------
    def host_struct(self):
        """(hmf_smpl_host_t, keep-alive list) built from the current buffers."""
        keep = []

        def arr(t, dt=np.float32):
            a = np.ascontiguousarray(t.detach().cpu().numpy().astype(dt))
            keep.append(a)
            return a
        more = torch.cat([self.J_regressor_extra, self.J_regressor_cocoplus, self.J_regressor_h36m], dim=0)
        hs = _lib.SmplHost()
        hs.num_betas = self.num_betas
        hs.v_template_host = _lib.fptr(arr(self.v_template))
        hs.shapedirs_host = _lib.fptr(arr(self.shapedirs))
        hs.posedirs_host = _lib.fptr(arr(self.posedirs))
        hs.J_regressor_host = _lib.fptr(arr(self.J_regressor))
        hs.lbs_weights_host = _lib.fptr(arr(self.lbs_weights))
        hs.parents_host = _lib.iptr(arr(self.parents, np.int32))
        ids = np.ascontiguousarray(np.array(EXTRA_VERTEX_IDS, np.int32))
        keep.append(ids)
        hs.extra_vertex_ids_host = _lib.iptr(ids)
        hs.J_regressor_more_host = _lib.fptr(arr(more))
        return hs, keep

    def _handle(self, device):
        key = (device.type, device.index)
        h = self._handles.get(key)
        if h is None:
            lib = _lib.load()
            hs, keep = self.host_struct()
            h = C.c_void_p()
            with torch.cuda.device(device):
                _lib.check(lib.hmf_smpl_create(C.byref(h), C.byref(hs)), 'hmf_smpl_create')
            self._handles[key] = h
        return h

    def _apply(self, fn, *a, **k):
        self._release()
        return super()._apply(fn, *a, **k)

    def _release(self):
        if getattr(self, '_handles', None):
            lib = _lib.load()
            for h in self._handles.values():
                lib.hmf_smpl_destroy(h)
        self._handles = {}
        self._ws = {}

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _impl(self, M=None):
        """hmf_smpl_lbs `impl` of this module's configuration for a call of M bodies."""
        if self.blend_dtype == 'fp32':
            return 4 if (self.lbs_rows == 32 or (self.lbs_rows == 0 and M is not None and M <= 32)) else 0
        return {'bf16': 3, 'bf16_plain': 2}[self.blend_dtype]

    def _workspace(self, device, M):
        # one scratch buffer per (device, stream): calls issued on different streams may be in flight together
        key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
        need = _lib.load().hmf_smpl_lbs_workspace_bytes(M)
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=device)
            self._ws[key] = ws
        return ws

    # ---- forward ------------------------------------------------------------------------------------
    @staticmethod
    def _f32c(t):
        return t.detach().to(torch.float32).contiguous()

    def _rodrigues(self, v, n):
        lib = _lib.load()
        v = self._f32c(v).reshape(n, 3)
        R = torch.empty(n, 9, dtype=torch.float32, device=v.device)
        _lib.check(lib.hmf_batch_rodrigues(_lib.dptr(v), n, _lib.dptr(R), _lib.current_stream()), 'hmf_batch_rodrigues')
        return R

    def _joints_kernel(self, g, b, be):
        """The 90 joints without the body (hmf_smpl_joints): g (M, 9), b (M, 207), be (M, nb) contiguous fp32 device tensors -> (M, 90, 3)."""
        lib = _lib.load()
        M = b.shape[0]
        joints = torch.empty(M, 90, 3, dtype=torch.float32, device=b.device)
        with torch.cuda.device(b.device):
            _lib.check(lib.hmf_smpl_joints(self._handle(b.device), _lib.dptr(g), _lib.dptr(b), _lib.dptr(be), M, _lib.dptr(joints), _lib.current_stream()),
                       'hmf_smpl_joints')
        return joints

    def _lbs_kernel(self, g, b, be, impl=None):
        """g (Mg, 9), b (M, 207), be (Mb, nb) contiguous fp32 device tensors (Mg, Mb in {1, M}) -> vertices (M,6890,3), joints (M,90,3)."""
        lib = _lib.load()
        dev = b.device
        M = b.shape[0]
        verts = torch.empty(M, NUM_VERTS, 3, dtype=torch.float32, device=dev)
        joints = torch.empty(M, 90, 3, dtype=torch.float32, device=dev)
        if M == 0:
            return verts, joints
        with torch.cuda.device(dev):
            ws = self._workspace(dev, M)
            _lib.check(lib.hmf_smpl_lbs(self._handle(dev), _lib.dptr(g), g.shape[0], _lib.dptr(b), _lib.dptr(be), be.shape[0], M,
                                        _lib.dptr(verts), _lib.dptr(joints), _lib.dptr(ws), ws.numel(),
                                        int(self._impl(M) if impl is None else impl),
                                        _lib.current_stream()), 'hmf_smpl_lbs')
        return verts, joints

    def forward(self, betas=None, body_pose=None, global_orient=None, transl=None, pose2rot=True, impl=None, joints_only=False, **kwargs):
        """models/smpl.py:27-41 over smplx SMPL.forward.  `get_skin`/`return_verts` kwargs are accepted and ignored
        (vertices are always produced, as the reference forces) -- except with the extension keyword joints_only=True under autograd (the training
        loss reads joints only: hmf_smpl_joints poses just the <= 320 vertices the joint tables touch; `.vertices` is then None).  Under autograd the JOINTS are differentiable w.r.t. global_orient / body_pose / betas (the training step's 2D reprojection loss; rotation matrices or,
        since round 6, axis-angle vectors with pose2rot=True).  The VERTICES are differentiable too (round 6): their gradient, when one arrives, is pulled back through a
        torch-operator restatement of the LBS (autograd._lbs_vertices_torch) -- the opt-in vertex loss of the reference, not a training-step path."""
        global_orient = self.global_orient if global_orient is None else global_orient
        body_pose = self.body_pose if body_pose is None else body_pose
        betas = self.betas if betas is None else betas
        if transl is None and hasattr(self, 'transl'):
            transl = self.transl
        _lib.require_cuda(betas, 'betas')
        _lib.require_cuda(body_pose, 'body_pose')
        _lib.require_cuda(global_orient, 'global_orient')
        dev = betas.device
        M = max(betas.shape[0], global_orient.shape[0], body_pose.shape[0])
        wants_grad = torch.is_grad_enabled() and (betas.requires_grad or body_pose.requires_grad or global_orient.requires_grad)
        with torch.cuda.device(dev):
            if wants_grad:
                if pose2rot:
                    # round 6 (the optimise driver's call, optimise/optimise_humaniflow.py:74-80): axis-angle inputs under autograd go through smplx'
                    # batch_rodrigues as torch operators (24 small rotations per body, differentiable), then through the rotation-matrix path below
                    global_orient = _rodrigues_torch(global_orient.reshape(-1, 3)).reshape(global_orient.shape[0], 1, 3, 3)
                    body_pose = _rodrigues_torch(body_pose.reshape(-1, 3)).reshape(body_pose.shape[0], 23, 3, 3)
                if global_orient.shape[0] != M or betas.shape[0] != M or body_pose.shape[0] != M:
                    raise ValueError('SMPL.forward under autograd: expand global_orient / betas to the pose batch (M = %d) first' % M)
                from .autograd import SmplJoints
                verts, joints = SmplJoints.apply(self, global_orient.to(torch.float32).reshape(M, 9).contiguous(),
                                                 body_pose.to(torch.float32).reshape(M, 23 * 9).contiguous(),
                                                 betas.to(torch.float32).contiguous(), -1 if joints_only else impl)
                if joints_only:
                    verts = None
            else:
                if pose2rot:
                    g = self._rodrigues(global_orient, global_orient.shape[0])
                    b = self._rodrigues(body_pose, body_pose.shape[0] * 23).reshape(body_pose.shape[0], 23 * 9)
                else:
                    g = self._f32c(global_orient).reshape(global_orient.shape[0], 9)
                    b = self._f32c(body_pose).reshape(body_pose.shape[0], 23 * 9)
                if b.shape[0] != M or g.shape[0] not in (1, M) or betas.shape[0] not in (1, M):
                    raise ValueError('SMPL.forward: batch sizes betas=%d global_orient=%d body_pose=%d are not broadcastable '
                                     '(smplx rule: 1 or M)' % (betas.shape[0], g.shape[0], b.shape[0]))
                verts, joints = self._lbs_kernel(g, b, self._f32c(betas), impl)
        if transl is not None:
            joints = joints + transl.unsqueeze(1)
            verts = None if verts is None else verts + transl.unsqueeze(1)
        full_pose = None
        if global_orient.dim() == body_pose.dim():
            full_pose = torch.cat([global_orient, body_pose], dim=1)
        return SMPLOutput(vertices=verts, joints=joints, full_pose=full_pose, betas=betas, global_orient=global_orient,
                          body_pose=body_pose)

    def forward_axis_angle(self, pose, betas, impl=None):
        """Bodies from whole axis-angle poses (M, 72) -- the evaluation targets of evaluate/evaluate_humaniflow.py:83-87 -- without slicing the pose
        tensor first: hmf_batch_rodrigues_pose writes the global-orientation and body-pose rotations as the two tensors the LBS launch takes (the two
        slices, their contiguous copies and smplx's `full_pose` cat were five small ATen launches per evaluation step).  Returns (vertices, joints)."""
        _lib.require_cuda(pose, 'pose')
        lib = _lib.load()
        p, be = self._f32c(pose).reshape(-1, 72), self._f32c(betas)
        M = p.shape[0]
        g = torch.empty(M, 9, dtype=torch.float32, device=p.device)
        b = torch.empty(M, 23 * 9, dtype=torch.float32, device=p.device)
        with torch.cuda.device(p.device):
            _lib.check(lib.hmf_batch_rodrigues_pose(_lib.dptr(p), M, _lib.dptr(g), _lib.dptr(b), _lib.current_stream()), 'hmf_batch_rodrigues_pose')
            return self._lbs_kernel(g, b, be, impl)

    def forward_samples_and_point_est(self, pred, return_packed=False, joints_only=False):
        """One launch for what the reference does with two SMPL calls (evaluate/evaluate_humaniflow.py:124-127,168-172,
        predict/predict_humaniflow.py:128-131,160-163): the B point-estimate bodies and the B*N sampled bodies of a
        HumaniflowModel.forward(num_samples=N) result, M = B*N + B rows -- the M = B launch alone fills 1/8 of the GPU for 1 % of the work.
        Returns (point_est_output, samples_output) whose tensors are views of the shared result."""
        pk = pred['smpl_packed_inputs']
        g = pred['glob_rotmat']
        B, N = g.shape[0], pk['num_samples']
        BN, M = B * N, B * N + B
        body, betas = pk['body_pose'], pk['betas']
        wants_grad = torch.is_grad_enabled() and (body.requires_grad or betas.requires_grad or g.requires_grad)
        if wants_grad or N == 0:
            glob_all = torch.cat([g[:, None].expand(B, N, 3, 3).reshape(BN, 1, 3, 3), g[:, None]], dim=0)
            out = self(body_pose=body, global_orient=glob_all, betas=betas, pose2rot=False, joints_only=joints_only and wants_grad)
            verts, joints, full_pose = out.vertices, out.joints, out.full_pose
        else:
            # hmf_smpl_lbs_packed indexes image b's rotation for its N sample rows and its point-estimate row inside the kernel (the reference
            # expands and concatenates a (B*N + B, 1, 3, 3) tensor first) and writes smplx's `full_pose` from the same launch
            _lib.require_cuda(betas, 'betas')
            _lib.require_cuda(body, 'body_pose')
            _lib.require_cuda(g, 'global_orient')
            lib = _lib.load()
            dev = betas.device
            gc, bc, bec = self._f32c(g).reshape(B, 9), self._f32c(body).reshape(M, 23 * 9), self._f32c(betas)
            verts = torch.empty(M, NUM_VERTS, 3, dtype=torch.float32, device=dev)
            joints = torch.empty(M, 90, 3, dtype=torch.float32, device=dev)
            full_pose = torch.empty(M, 24, 3, 3, dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                ws = self._workspace(dev, M)
                _lib.check(lib.hmf_smpl_lbs_packed(self._handle(dev), _lib.dptr(gc), B, N, _lib.dptr(bc), _lib.dptr(bec), _lib.dptr(verts), _lib.dptr(joints),
                                                   _lib.dptr(full_pose), _lib.dptr(ws), ws.numel(),
                                                   int(self._impl(M)), _lib.current_stream()),
                           'hmf_smpl_lbs_packed')
            if hasattr(self, 'transl') and self.transl is not None:
                joints = joints + self.transl.unsqueeze(1)
                verts = verts + self.transl.unsqueeze(1)
        cut = lambda sl: SMPLOutput(vertices=None if verts is None else verts[sl], joints=joints[sl], full_pose=None if full_pose is None else full_pose[sl], betas=betas[sl],
                                    global_orient=full_pose[sl, :1] if full_pose is not None else None, body_pose=body[sl])
        if return_packed:
            return cut(slice(BN, BN + B)), cut(slice(0, BN)), joints
        return cut(slice(BN, BN + B)), cut(slice(0, BN))

    def tpose(self, betas):
        """T-pose vertices `v_template + shapedirs . beta` for (M, num_betas) -- what the reference obtains with a
        zero-pose LBS call (evaluate/evaluate_humaniflow.py:131-133,186-192; predict_humaniflow.py:147)."""
        _lib.require_cuda(betas, 'betas')
        lib = _lib.load()
        be = self._f32c(betas)
        M = be.shape[0]
        verts = torch.empty(M, NUM_VERTS, 3, dtype=torch.float32, device=be.device)
        with torch.cuda.device(be.device):
            _lib.check(lib.hmf_smpl_tpose(self._handle(be.device), _lib.dptr(be), M, _lib.dptr(verts), None,
                                          _lib.current_stream()), 'hmf_smpl_tpose')
        return verts

    def tpose_errors(self, betas_pred, betas_target):
        """PVE-T / PVE-T-SC of T-pose bodies WITHOUT building them (hmf_smpl_tpose_errors): betas_pred (Bt, nb) or (Bt, N, nb) against
        betas_target (Bt, nb) -> (..., 3) = [mean |P - T|, mean |sc(P) - T|, 0], the layout of the point-set kernels' output.  The reference
        (evaluate/evaluate_humaniflow.py:131-133,186-192 -> metrics/eval_metrics_tracker.py:200-222) materialises both vertex sets first."""
        _lib.require_cuda(betas_pred, 'betas_pred')
        lib = _lib.load()
        bp, bt = self._f32c(betas_pred), self._f32c(betas_target)
        lead = bp.shape[:-1]
        npairs = 1
        for d_ in lead:
            npairs *= d_
        out = torch.empty(npairs, 3, dtype=torch.float32, device=bp.device)
        ws = torch.empty(int(lib.hmf_smpl_tpose_errors_workspace_bytes(npairs)), dtype=torch.uint8, device=bp.device)
        with torch.cuda.device(bp.device):
            _lib.check(lib.hmf_smpl_tpose_errors(self._handle(bp.device), _lib.dptr(bp), _lib.dptr(bt), npairs, npairs // max(bt.shape[0], 1), _lib.dptr(out),
                                                 ws.data_ptr(), ws.numel(), _lib.current_stream()), 'hmf_smpl_tpose_errors')
        return out.reshape(*lead, 3)

    def tpose_errors_packed(self, betas_packed, betas_target, B, N):
        """The same for the packed rows of an evaluation step (sample row b N + n, then point-estimate row B N + b) in ONE launch
        (hmf_smpl_tpose_errors_packed): (B N + B, nb) against (B, nb) -> (B N + B, 3)."""
        _lib.require_cuda(betas_packed, 'betas_packed')
        lib = _lib.load()
        bp, bt = self._f32c(betas_packed), self._f32c(betas_target)
        M = B * N + B
        assert bp.shape[0] == M and bt.shape[0] == B
        out = torch.empty(M, 3, dtype=torch.float32, device=bp.device)
        ws = torch.empty(int(lib.hmf_smpl_tpose_errors_workspace_bytes(M)), dtype=torch.uint8, device=bp.device)
        with torch.cuda.device(bp.device):
            _lib.check(lib.hmf_smpl_tpose_errors_packed(self._handle(bp.device), _lib.dptr(bp), _lib.dptr(bt), B, N, _lib.dptr(out), ws.data_ptr(), ws.numel(),
                                                        _lib.current_stream()), 'hmf_smpl_tpose_errors_packed')
        return out

    def model_dict(self):
        """CPU copies of the model arrays (for the oracle in tests)."""
        keys = ['v_template', 'shapedirs', 'posedirs', 'J_regressor', 'lbs_weights', 'parents',
                'J_regressor_extra', 'J_regressor_cocoplus', 'J_regressor_h36m']
        return {k: getattr(self, k).detach().cpu() for k in keys}


def compute_vertex_variance_from_samples(vertices_samples):
    """utils/sampling_utils.py:22-33 on the device: (N, V, 3) -> ((V,), (V, 3))."""
    _lib.require_cuda(vertices_samples, 'vertices_samples')
    lib = _lib.load()
    v = vertices_samples.detach().to(torch.float32).contiguous()
    N, V = v.shape[0], v.shape[1]
    avg = torch.empty(V, dtype=torch.float32, device=v.device)
    std = torch.empty(V, 3, dtype=torch.float32, device=v.device)
    with torch.cuda.device(v.device):
        _lib.check(lib.hmf_vertex_variance(_lib.dptr(v), N, V, _lib.dptr(avg), _lib.dptr(std), _lib.current_stream()),
                   'hmf_vertex_variance')
    return avg, std
