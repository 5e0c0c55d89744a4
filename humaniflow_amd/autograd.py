"""torch.autograd glue for the training step (BASELINE configs[4]).

What the reference differentiates in train/train_humaniflow.py:248-330 and where its backward runs here:

  teacher-forced pose NLL   conditioned_pose_SO3flow_dists_for_loglik[j].log_prob(R_j)     FlowLogProbSO3   (k_flow_logprob_bwd)
  teacher-forced contexts   compute_flow_context on the target rotations                     FlowContext      (dense reductions)
  (all flow parameters enter the graph once, through FlowParamAnchor: one gradient per tensor per pass)
  reparameterised samples   pose_SO3_samples, ancestors feeding descendants (:302-311)       FlowSampleChain  (k_flow_chain_bwd, 8 rounds reversed)
  2D reprojection           SMPL joints of the samples / point estimate                      SmplJoints       (k_smpl_joints_bwd)
  6D -> R of the global rotation                                                              Rot6dToRotmat

The HIP kernels do the structured part (spline / hyper-network / kinematic-chain / skinning reverse sweeps) and emit per-row deltas
and activations; the weight gradients are dense reductions over the row axis (dW = delta^T . a, db = column sums of delta): the library's
batched A^T B kernel (hmf_rows_atb_batched, csrc/flow_grad.hip); the one dense pull-back to the shared input features (delta . W over all
23 context layers) runs on the 1x1-convolution GEMM (hmf_conv1x1_gemm).  No vendor GEMM is called from the flow's backward.
Forward values always come from the same inference kernels as under no_grad.
"""
import ctypes as C

import math

import torch
import torch.nn.functional as F

from . import _lib


def _atb(A, B, batch, rows, m, n, lda, ldb, sA, sB, want_sum):
    """C (batch, m, n) = A^T B over `rows` (+ column sums of A (batch, m)); A / B are fp32 device tensors addressed by pitch and batch stride."""
    lib = _lib.load()
    Cm = torch.empty(batch, m, n, dtype=torch.float32, device=A.device)
    sm = torch.empty(batch, m, dtype=torch.float32, device=A.device) if want_sum else None
    with torch.cuda.device(A.device):
        _lib.check(lib.hmf_rows_atb_batched(A.data_ptr(), B.data_ptr(), Cm.data_ptr(), None if sm is None else sm.data_ptr(), batch, rows, m, n, lda, ldb,
                                            sA, sB, _lib.current_stream()), 'hmf_rows_atb_batched')
    return Cm, sm


# ---- records of the flow backward kernels ---------------------------------------------------------------------------------
class _Records:
    """Device buffers for hmf_flow_grad_records_t: part-major [P][2][rows][width]."""

    def __init__(self, P, rows, device, with_dctx):
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=device)
        self.P, self.rows = P, rows
        self.D1, self.D4, self.A1 = f(P, 2, rows, 64), f(P, 2, rows, 64), f(P, 2, rows, 64)
        self.D2, self.D3, self.A2, self.A3 = f(P, 2, rows, 32), f(P, 2, rows, 32), f(P, 2, rows, 32), f(P, 2, rows, 32)
        self.XIN = f(P, 2, rows)
        self.ACTX = f(P, rows, 64)
        self.DCTX = f(P, rows, 64) if with_dctx else None

    def struct(self):
        s = _lib.FlowGradRecords()
        for k in ('D1', 'D2', 'D3', 'D4', 'A1', 'A2', 'A3', 'XIN', 'ACTX'):
            setattr(s, k, getattr(self, k).data_ptr())
        s.DCTX = self.DCTX.data_ptr() if self.DCTX is not None else None
        s.rows = self.rows
        return s

    def hypernet_grads(self):
        """[(dW, db) for layer 1..4], each (P, 2, out, in) / (P, 2, out): dW_l = D_l^T . A_{l-1} over the row axis."""
        P, R = self.P, self.rows
        a0 = torch.cat([self.ACTX[:, None].expand(-1, 2, -1, -1), self.XIN[..., None]], dim=-1).contiguous()   # (P,2,rows,65) = [context | x1]

        def tm(d, m, a, n):                                       # (P,2,rows,wd)^T . (P,2,rows,wa) -> (P,2,m,n), (P,2,m)
            wd, wa = d.shape[-1], a.shape[-1]
            Cm, sm = _atb(d, a, 2 * P, R, m, n, wd, wa, R * wd, R * wa, True)
            return Cm.view(P, 2, m, n), sm.view(P, 2, m)
        return [tm(self.D1, 64, a0, 65), tm(self.D2, 32, self.A1, 64), tm(self.D3, 32, self.A2, 32), tm(self.D4, 62, self.A3, 32)]


def _hypernet_params(model, part0, P):
    """Flat list [part][coupling][layer](weight, bias) of parts part0 .. part0+P-1: 16 tensors per part (2 couplings x 4 layers)."""
    T = model.model_cfg.NORM_FLOW.NUM_TRANSFORMS
    out = []
    for j in range(part0, part0 + P):
        for t in range(T):
            for lin in model.pose_so3flow_transform_modules[T * j + t].nn.layers:
                out += [lin.weight, lin.bias]
    return out


def _is_variant(model):
    """The handle runs the generic kernels of csrc/flow_generic.hip (any non-default configuration, or the default one forced through them)."""
    return (not model.default_flow) or bool(model.force_generic_flow)


def _variant_dims(model):
    """(T, L, LP, K, N, KP, NP): transforms per part, layers of a transform / PLU hyper-network and their (in, out) widths."""
    nf = model.model_cfg.NORM_FLOW
    hd = list(nf.TRANSFORM_NN_HIDDEN_DIMS)
    odim = 62 if nf.TRANSFORM_TYPE == 'spline_coupling' else 2
    K, N = [nf.CONTEXT_DIM + 1] + hd, hd + [odim]
    KP = NP = []
    if nf.PERMUTE_TYPE == 'conditional_linear_plu':
        ph = list(nf.PERMUTE_NN_HIDDEN_DIMS)
        KP, NP = [nf.CONTEXT_DIM] + ph, ph + [9]
    return nf.NUM_TRANSFORMS, len(N), len(NP), K, N, KP, NP


def _variant_modules(model, j, t):
    """(permute module or None, coupling module) of transform t of part j in pose_so3flow_transform_modules' order."""
    per = 1 if model.model_cfg.NORM_FLOW.PERMUTE_TYPE == 'permute' else 2
    base = (j * model.model_cfg.NORM_FLOW.NUM_TRANSFORMS + t) * per
    mods = model.pose_so3flow_transform_modules
    return (mods[base] if per == 2 else None), mods[base + per - 1]


def _variant_params(model, part0, P):
    """Flat list [part][transform]([LU] | [PLU layers (weight, bias)]) + [coupling layers (weight, bias)] -- the order VariantRecords.grads() is unpacked in."""
    T = model.model_cfg.NORM_FLOW.NUM_TRANSFORMS
    out = []
    for j in range(part0, part0 + P):
        for t in range(T):
            perm, cpl = _variant_modules(model, j, t)
            if perm is not None:
                if hasattr(perm, 'LU'):
                    out.append(perm.LU)
                else:
                    for lin in perm.cond_plu_nn.layers:
                        out += [lin.weight, lin.bias]
            for lin in cpl.nn.layers:
                out += [lin.weight, lin.bias]
    return out


class _VariantRecords:
    """Device buffers for hmf_flow_generic_grad_records_t, zero-filled (the kernels skip masked rows and channels beyond a layer's width)."""

    def __init__(self, model, P, rows, device, with_dctx):
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=device)
        self.T, self.L, self.LP, self.K, self.N, self.KP, self.NP = _variant_dims(model)
        self.P, self.rows = P, rows
        T, L, LP = self.T, self.L, self.LP
        self.D, self.A, self.XIN = z(P, T, L, rows, 64), z(P, T, L, rows, 64), z(P, T, rows)
        self.DP = z(P, T, LP, rows, 64) if LP else None
        self.AP = z(P, T, LP, rows, 64) if LP else None
        self.GLU = z(P, T, rows, 9) if model.model_cfg.NORM_FLOW.PERMUTE_TYPE == 'linear_plu' else None
        self.ACTX = z(P, rows, 64)
        self.DCTX = z(P, rows, 64) if with_dctx else None

    def struct(self):
        s = _lib.FlowVariantGradRecords()
        for k in ('D', 'A', 'XIN', 'DP', 'AP', 'GLU', 'ACTX', 'DCTX'):
            t = getattr(self, k)
            setattr(s, k, None if t is None else t.data_ptr())
        s.rows, s.L, s.LP = self.rows, self.L, self.LP
        return s

    def grads(self):
        """{'tr': [(dW (P,T,N_l,K_l), db (P,T,N_l)) per layer], 'plu': [...] | None, 'lu': (P,T,3,3) | None}: dW_l = D_l^T . input_l over the row axis."""
        P, T, R = self.P, self.T, self.rows

        def net(D, A, L, K, N, a0):
            out = []
            for l in range(L):
                if l == 0:
                    Cm, sm = _atb(D[:, :, 0], a0, P * T, R, N[0], K[0], 64, a0.shape[-1], L * R * 64, R * a0.shape[-1], True)
                else:
                    Cm, sm = _atb(D[:, :, l], A[:, :, l], P * T, R, N[l], K[l], 64, 64, L * R * 64, L * R * 64, True)
                out.append((Cm.view(P, T, N[l], K[l]), sm.view(P, T, N[l])))
            return out
        ctx_t = self.ACTX[:, None].expand(-1, T, -1, -1)
        res = {'tr': net(self.D, self.A, self.L, self.K, self.N, torch.cat([ctx_t, self.XIN[..., None]], dim=-1).contiguous()), 'plu': None, 'lu': None}
        if self.LP:
            res['plu'] = net(self.DP, self.AP, self.LP, self.KP, self.NP, ctx_t.contiguous())
        if self.GLU is not None:
            res['lu'] = self.GLU.sum(dim=2).view(P, T, 3, 3)
        return res


def _variant_sum(a, b):
    if a is None:
        return b
    out = {'tr': [(x[0] + y[0], x[1] + y[1]) for x, y in zip(a['tr'], b['tr'])], 'plu': None, 'lu': None}
    if a['plu'] is not None:
        out['plu'] = [(x[0] + y[0], x[1] + y[1]) for x, y in zip(a['plu'], b['plu'])]
    if a['lu'] is not None:
        out['lu'] = a['lu'] + b['lu']
    return out


def _variant_embed(model, g, part0, n):
    """Gradients of parts part0 .. part0+n-1 embedded into the full stack of P parts."""
    P = model.num_bodyparts

    def emb(t):
        full = t.new_zeros((P,) + t.shape[1:])
        full[part0:part0 + n] = t
        return full
    return {'tr': [(emb(w), emb(b)) for w, b in g['tr']], 'plu': None if g['plu'] is None else [(emb(w), emb(b)) for w, b in g['plu']],
            'lu': None if g['lu'] is None else emb(g['lu'])}


def _variant_unpack(model, g):
    """Per-parameter gradients in _variant_params(model, 0, P) order."""
    P, T = model.num_bodyparts, model.model_cfg.NORM_FLOW.NUM_TRANSFORMS
    out = []
    for j in range(P):
        for t in range(T):
            if g['lu'] is not None:
                out.append(g['lu'][j, t])
            if g['plu'] is not None:
                for dW, db in g['plu']:
                    out += [dW[j, t], db[j, t]]
            for dW, db in g['tr']:
                out += [dW[j, t], db[j, t]]
    return out


# ---- context layers: shared dense pull-back ---------------------------------------------------------------------------------
def _anc_index(model, device):
    """Per part: column indices of its ancestors' flattened rotations inside a (rows, 23*9) rotation matrix block."""
    key = ('anc_idx', str(device))
    cache = model.__dict__.setdefault('_grad_cache', {})
    if key not in cache:
        cache[key] = [torch.tensor([a * 9 + e for a in model.ancestors_dict[j] for e in range(9)], dtype=torch.long, device=device)
                      for j in range(model.num_bodyparts)]
    return cache[key]


def _context_pullback(model, delta, isgc_rows, rot_rows, want_weight_grads=True):
    """delta (P, rows, 64) = d L / d (pre-activation of fc_flow_context[j]); isgc_rows (rows, 256); rot_rows (rows, 23, 9).
    Returns (g_full (P, 64, 256 + 207) = delta^T . [isgc | all rotations], db (P, 64), g_isgc (rows, 256)); the per-part weight gradient is a
    column selection of g_full (`_context_weight_grads`), done once per backward pass in FlowParamAnchor.  want_weight_grads=False (the flow is
    frozen: the consumer's anchor token needs no gradient) skips the weight-gradient GEMM and returns (None, None, g_isgc)."""
    lib = _lib.load()
    P, R = delta.shape[0], delta.shape[1]
    delta = delta.float().contiguous()
    g_full = db = None
    if want_weight_grads:
        x_full = torch.cat([isgc_rows, rot_rows.reshape(rot_rows.shape[0], -1)], dim=1).float().contiguous()   # (rows, 256 + 207)
        g_full, db = _atb(delta, x_full, P, R, 64, x_full.shape[1], 64, x_full.shape[1], R * 64, 0, True)      # (P, 64, 463): one shared B operand
    nin = isgc_rows.shape[1]
    # g_isgc[r] = sum_p delta[p, r] . W_p[:, :nin]: ONE GEMM (rows, P * 64) x (P * 64, nin) on the 1x1-convolution kernel (K = 1472, N = 256)
    d_rows = delta.permute(1, 0, 2).reshape(R, P * 64).contiguous()
    w0t = torch.cat([m.weight.detach()[:, :nin].float() for m in model.fc_flow_context], dim=0).t().contiguous()   # (nin, P * 64) = [N][K]
    g_isgc = torch.empty(R, nin, dtype=torch.float32, device=delta.device)
    with torch.cuda.device(delta.device):
        _lib.check(lib.hmf_conv1x1_gemm(d_rows.data_ptr(), None, w0t.data_ptr(), None, None, g_isgc.data_ptr(), 1, R, 1, P * 64, 0, 0, 0, 1, nin, 0, 0,
                                        _lib.current_stream()), 'hmf_conv1x1_gemm')
    return g_full, db, g_isgc


def _context_weight_grads(model, g_full, nin=256):
    """[dW_j (64, nin + 9 a_j)] from g_full (P, 64, nin + 207): the columns of part j's ancestors.  ONE gather for all parts through a cached flat
    index (round 2: index arithmetic + index_select + cat per part = 69 launches per backward pass); the results are views of its output."""
    P, rows, width = g_full.shape
    key = ('ctx_gather', str(g_full.device), nin, rows, width)
    cache = model.__dict__.setdefault('_grad_cache', {})
    if key not in cache:
        idx = _anc_index(model, g_full.device)
        flat, widths = [], []
        for j in range(P):
            cols = torch.cat([torch.arange(nin, device=g_full.device), idx[j] + nin])
            r = torch.arange(rows, device=g_full.device)
            flat.append(((j * rows + r)[:, None] * width + cols[None, :]).reshape(-1))
            widths.append(int(cols.numel()))
        cache[key] = (torch.cat(flat), widths)
    flat_idx, widths = cache[key]
    out = g_full.reshape(-1).index_select(0, flat_idx)
    res, o = [], 0
    for w in widths:
        res.append(out[o:o + rows * w].view(rows, w))
        o += rows * w
    return res


def _ctx_params(model):
    out = []
    for m in model.fc_flow_context:
        out += [m.weight, m.bias]
    return out


class FlowParamAnchor(torch.autograd.Function):
    """One gradient per flow parameter per backward pass.  The three consumers below (context layers on the targets, SO(3) log-density, sample
    chain) all differentiate w.r.t. the same ~420 parameter tensors; as independent autograd nodes each would hand autograd its own per-tensor
    gradients and autograd would add them tensor by tensor (~830 tiny launches per step).  Instead the parameters enter the graph ONCE, here;
    the node's output is a scalar token the consumers take as an input.  In backward a consumer leaves its STACKED gradients (all parts in
    one tensor) in `stash` and returns a zero for the token; this node runs after all of them (autograd's dependency order), sums the stacked
    tensors (a dozen launches) and returns per-parameter views.

    `stash` is a dict keyed by (consumer node, kind): a consumer that runs again before this node has run (torch.autograd.grad restricted to
    inputs that do not reach the parameters, then a second backward over the retained graph) REPLACES its entry instead of adding a stale one.
    A consumer whose token input needs no gradient (flow parameters frozen) leaves nothing here and skips its weight-gradient products."""

    @staticmethod
    def forward(ctx, model, stash, *params):
        ctx.model, ctx.stash = model, stash
        return torch.zeros((), dtype=torch.float32, device=params[0].device)

    @staticmethod
    def backward(ctx, g):
        model, entries = ctx.model, list(ctx.stash.values())
        ctx.stash.clear()
        P = model.num_bodyparts
        needs = ctx.needs_input_grad[2:]
        g_full, g_db, hyper, hyperv = None, None, None, None
        for e in entries:
            if e[0] == 'ctx':
                g_full = e[1] if g_full is None else g_full + e[1]
                g_db = e[2] if g_db is None else g_db + e[2]
            elif e[0] == 'hyperv':                               # variants: stacked dictionaries (_VariantRecords.grads)
                _, part0, n, grads = e
                hyperv = _variant_sum(hyperv, grads if n == P else _variant_embed(model, grads, part0, n))
            else:
                _, part0, n, grads = e
                if n != P:                                       # a sub-range of parts (per-part log_prob calls): embed into the full stack
                    full = []
                    for dW, db in grads:
                        fW, fb = dW.new_zeros((P,) + dW.shape[1:]), db.new_zeros((P,) + db.shape[1:])
                        fW[part0:part0 + n], fb[part0:part0 + n] = dW, db
                        full.append((fW, fb))
                    grads = full
                hyper = list(grads) if hyper is None else [(a[0] + b[0], a[1] + b[1]) for a, b in zip(hyper, grads)]
        out = []
        if g_full is not None:
            dW = _context_weight_grads(model, g_full)
            for j in range(P):
                out += [dW[j], g_db[j]]
        else:
            out += [None] * (2 * P)
        if hyperv is not None:
            out += _variant_unpack(model, hyperv)
        elif hyper is not None:
            for p in range(P):
                for c in range(2):
                    for l in range(4):
                        out += [hyper[l][0][p, c], hyper[l][1][p, c]]
        else:
            out += [None] * (len(needs) - len(out))
        return (None, None) + tuple(gr if n else None for gr, n in zip(out, needs))


def new_flow_anchor(model):
    """(token, stash) for one forward pass: pass both to every consumer of that pass."""
    stash = {}
    hyper = _variant_params(model, 0, model.num_bodyparts) if _is_variant(model) else _hypernet_params(model, 0, model.num_bodyparts)
    token = FlowParamAnchor.apply(model, stash, *(_ctx_params(model) + hyper))
    return token, stash


class FlowContext(torch.autograd.Function):
    """compute_flow_context for all 23 parts from GIVEN (teacher-forced) rotations: (B,256), (B,23,3,3) -> (B,23,64)."""

    @staticmethod
    def forward(ctx, model, isgc, pose_R, token, stash):
        out = model._flow_context_kernel(isgc.detach(), pose_R.detach())
        ctx.model, ctx.stash = model, stash
        ctx.save_for_backward(isgc.detach(), pose_R.detach().float().reshape(pose_R.shape[0], 23, 9), out)
        ctx.r_needs_grad = pose_R.requires_grad
        ctx.r_dtype = pose_R.dtype
        return out

    @staticmethod
    def backward(ctx, g):
        isgc, rot, out = ctx.saved_tensors
        delta = (g * torch.where(out > 0, torch.ones_like(out), out + 1.0)).transpose(0, 1).contiguous()   # ELU', (P, B, 64)
        g_full, db, g_isgc = _context_pullback(ctx.model, delta, isgc, rot, want_weight_grads=ctx.needs_input_grad[3])
        if ctx.needs_input_grad[3]:                                   # the token: some flow parameter wants its gradient
            ctx.stash[(id(ctx), 'ctx')] = ('ctx', g_full, db)
        g_rot = None
        if ctx.r_needs_grad:
            # round 6: d L / d pose_R_for_loglik through the context layers (models/humaniflow_model.py:152-186: part j reads the rotations of its ancestors):
            # g_R[b, a] += W_j[:, nin + 9 pos_j(a) : + 9]^T delta_j[b] for every descendant j of a.  The flow-as-prior call pattern only
            # (optimise/optimise_humaniflow.py:96-109): 23 small products on torch operators, not a training-step path.
            model, nin = ctx.model, isgc.shape[1]
            g_flat = torch.zeros(rot.shape[0], model.num_bodyparts * 9, dtype=torch.float32, device=rot.device)
            idx = _anc_index(model, rot.device)
            for j, lin in enumerate(model.fc_flow_context):
                if idx[j].numel():
                    g_flat.index_add_(1, idx[j], delta[j].float() @ lin.weight.detach()[:, nin:].float())
            g_rot = g_flat.view(rot.shape[0], model.num_bodyparts, 3, 3).to(ctx.r_dtype)
        return None, g_isgc, g_rot, g.new_zeros(()), None


class FlowLogProbSO3(torch.autograd.Function):
    """log p(R_j | ctx_j) for parts part0 .. part0+P-1: ctx (B,P,64) fp32, R (B,P,3,3) fp64 -> (B,P) fp32."""

    @staticmethod
    def forward(ctx, model, part0, P, cvec, R64, token, stash):
        out = model._log_prob_kernel(cvec.detach(), R64, part0, P, group=True)
        ctx.model, ctx.part0, ctx.P, ctx.stash = model, part0, P, stash
        ctx.save_for_backward(cvec.detach().contiguous(), R64)
        return out

    @staticmethod
    def backward(ctx, g):
        cvec, R64 = ctx.saved_tensors
        model, P, part0 = ctx.model, ctx.P, ctx.part0
        B = cvec.shape[0]
        lib = _lib.load()
        variant = _is_variant(model)
        rec = _VariantRecords(model, P, 3 * B, cvec.device, with_dctx=False) if variant else _Records(P, 3 * B, cvec.device, with_dctx=False)
        g_ctx = torch.empty_like(cvec)
        gc = g.detach().to(torch.float32).contiguous()
        ids = (C.c_int32 * P)(*range(part0, part0 + P)) if P != model.num_bodyparts else None
        st = rec.struct()
        fn, name = ((lib.hmf_flow_generic_log_prob_SO3_backward, 'hmf_flow_generic_log_prob_SO3_backward') if variant else
                    (lib.hmf_flow_log_prob_SO3_backward, 'hmf_flow_log_prob_SO3_backward'))
        with torch.cuda.device(cvec.device):
            _lib.check(fn(model._flow_handle(cvec.device), _lib.dptr(cvec), ids, P, _lib.dptr(R64), B, _lib.dptr(gc), _lib.dptr(g_ctx), C.byref(st),
                          _lib.current_stream()), name)
        if ctx.needs_input_grad[5]:                                   # the token
            ctx.stash[(id(ctx), 'hyper')] = ('hyperv', part0, P, rec.grads()) if variant else ('hyper', part0, P, rec.hypernet_grads())
        return None, None, None, g_ctx if ctx.needs_input_grad[3] else None, None, g.new_zeros(()), None


class FlowLogProbAlgebra(torch.autograd.Function):
    """log p(v | ctx) on the algebra for parts part0 .. part0+P-1, differentiable w.r.t. its ARGUMENT v (B,P,3) and the context (B,P,64); the flow's weights
    are constants here (hmf_flow_log_prob_so3_vjp writes no weight-gradient records).  Round 6, the flow-as-prior call pattern."""

    @staticmethod
    def forward(ctx, model, part0, P, cvec, v):
        out = model._log_prob_kernel(cvec.detach(), v.detach(), part0, P, group=False)
        ctx.model, ctx.part0, ctx.P = model, part0, P
        ctx.save_for_backward(cvec.detach().to(torch.float32).contiguous(), v.detach().to(torch.float32).contiguous())
        return out

    @staticmethod
    def backward(ctx, g):
        cvec, v = ctx.saved_tensors
        model, P, part0 = ctx.model, ctx.P, ctx.part0
        B = cvec.shape[0]
        lib = _lib.load()
        g_ctx, g_v = torch.empty_like(cvec), torch.empty_like(v)
        gc = g.detach().to(torch.float32).contiguous()
        ids = (C.c_int32 * P)(*range(part0, part0 + P)) if P != model.num_bodyparts else None
        with torch.cuda.device(cvec.device):
            _lib.check(lib.hmf_flow_log_prob_so3_vjp(model._flow_handle(cvec.device), _lib.dptr(cvec), ids, P, _lib.dptr(v), B, _lib.dptr(gc), _lib.dptr(g_ctx),
                                                     _lib.dptr(g_v), _lib.current_stream()), 'hmf_flow_log_prob_so3_vjp')
        return None, None, None, g_ctx if ctx.needs_input_grad[3] else None, g_v if ctx.needs_input_grad[4] else None


def _so3_log_vector64(R):
    """Differentiable fp64 logarithm map R (..., 3, 3) -> v (..., 3) with the branch structure of the reference's so3_log (utils/rigid_transform_utils.py:204-279):
    theta = acos(clamp((tr R - 1) / 2)), v = theta / sin(theta) vee((R - R^T) / 2) (series below 1e-20), and within 1e-2 of pi the components up to sign from the
    symmetric part, sqrt(clamp(., 1e-8) / 2), with the sign triple whose exponential is closest to R (chosen without gradient, :246-262).  torch operators: this
    runs only where a density is differentiated w.r.t. the rotation it is evaluated at, never in inference or training."""
    from .rigid_transform_utils import so3_exp
    Rt = R.transpose(-1, -2)
    anti = 0.5 * (R - Rt)
    w = torch.stack((anti[..., 2, 1], anti[..., 0, 2], anti[..., 1, 0]), dim=-1)
    cos_t = (0.5 * (R[..., 0, 0] + R[..., 1, 1] + R[..., 2, 2] - 1.0)).clamp(-1.0, 1.0)
    theta = torch.acos(cos_t)
    ratio = torch.where(theta < 1e-20, 1.0 + theta ** 2 / 6.0, theta / torch.sin(theta))
    v = ratio[..., None] * w
    near_pi = (math.pi - theta).abs() < 1e-2
    if bool(near_pi.any()):
        z = (theta ** 2 / (1.0 - torch.cos(theta)))[..., None, None] * (0.5 * (R + Rt) - torch.eye(3, dtype=R.dtype, device=R.device))
        q = torch.stack((z[..., 0, 0], z[..., 1, 1], z[..., 2, 2]), dim=-1)
        mag = torch.sqrt((2.0 * q - q.sum(dim=-1, keepdim=True)).clamp(min=1e-8) / 2.0)                 # q_i - q_j - q_k
        with torch.no_grad():
            signs = torch.tensor([[a, b, c] for a in (-1.0, 1.0) for b in (-1.0, 1.0) for c in (-1.0, 1.0)], dtype=R.dtype, device=R.device)
            cand = signs.view(8, *([1] * (mag.dim() - 1)), 3) * mag.detach()[None]
            err = (so3_exp(cand.contiguous()) - R.detach()[None]).pow(2).sum(dim=(-1, -2))
            pick = signs[torch.argmin(err, dim=0)]
        v = torch.where(near_pi[..., None], pick * mag, v)
    return v


def _so3_exp_logdet64(x):
    """log |det J_exp(x)| = log((2 - 2 cos |x|) / |x|^2) in fp64 (utils/rigid_transform_utils.py:298-314; 1 - |x|^2 / 12 below 1e-10), differentiable."""
    n = x.norm(dim=-1)
    big = n > 1e-10
    ns = torch.where(big, n, torch.ones_like(n))
    return torch.log(torch.where(big, (2.0 - 2.0 * torch.cos(ns)) / ns ** 2, 1.0 - ns ** 2 / 12.0))


def log_prob_SO3_wrt_value(model, cvec, R, part0, P):
    """log p(R_j | ctx_j) differentiable w.r.t. R (B,P,3,3) -- the rotation the density is evaluated AT -- and w.r.t. the context: the composition of
    local_diffeo_transformed_distribution.py:84-142 (x = log R, its two 2 pi k pre-images inside the support, -log|det J_exp| + the algebra density of each,
    log-sum-exp) with the algebra density on the library (FlowLogProbAlgebra).  The forward equals hmf_flow_log_prob_SO3 (same pre-images, fp64 exp-map terms,
    fp32 flow) to rounding; flow weights are constants on this path.  The gradient is undefined at R = I exactly, as in the reference (x / |x|)."""
    R64 = R.to(torch.float64)
    x = _so3_log_vector64(R64)
    n = x.norm(dim=-1, keepdim=True)
    radius = float(model.model_cfg.NORM_FLOW.COMPACT_SUPPORT_RADIUS)
    terms = []
    for k in (0, -1, 1):
        if k == 0:
            xk, ok = x, None
        else:
            xk = x / n * (n + 2.0 * math.pi * k)
            ok = xk.norm(dim=-1) < radius                                   # NaN at x = 0 fails the comparison, as in the reference
            xk = torch.where(ok[..., None], xk, torch.zeros_like(xk))       # masked_fill_(0) of so3_exp_transform.py:36-41
        t = FlowLogProbAlgebra.apply(model, part0, P, cvec, xk.to(torch.float32)) - _so3_exp_logdet64(xk).to(torch.float32)
        terms.append(t if ok is None else torch.where(ok, t, torch.full_like(t, float('-inf'))))
    return torch.logsumexp(torch.stack(terms), dim=0)


class FlowSampleChain(torch.autograd.Function):
    """Attaches the reparameterised-sample graph to the rotations hmf_flow_forward produced: inputs are the per-row isgc features
    H (B*N, 256) (differentiable), the forward's rotations and base noise; output = the same rotations, now differentiable w.r.t.
    H, the 23 context layers and the 46 hyper-networks (ancestor recursion included)."""

    @staticmethod
    def forward(ctx, model, H, rot_samples, eps_pose, token, stash):
        B, N = rot_samples.shape[:2]
        ctx.model, ctx.B, ctx.N, ctx.stash = model, B, N, stash
        ctx.save_for_backward(H.detach().contiguous(), rot_samples.detach().contiguous(), eps_pose.detach().contiguous())
        return rot_samples.detach().clone()

    @staticmethod
    def backward(ctx, g):
        H, rot, eps = ctx.saved_tensors
        model, B, N = ctx.model, ctx.B, ctx.N
        P = model.num_bodyparts
        lib = _lib.load()
        variant = _is_variant(model)
        rec = _VariantRecords(model, P, B * N, H.device, with_dctx=True) if variant else _Records(P, B * N, H.device, with_dctx=True)
        g_rot = g.detach().to(torch.float32).contiguous().clone()                                       # in/out: descendants add to ancestors
        st = rec.struct()
        fn, name = ((lib.hmf_flow_generic_sample_backward, 'hmf_flow_generic_sample_backward') if variant else
                    (lib.hmf_flow_sample_backward, 'hmf_flow_sample_backward'))
        with torch.cuda.device(H.device):
            _lib.check(fn(model._flow_handle(H.device), _lib.dptr(H), _lib.dptr(rot), _lib.dptr(eps), B, N, _lib.dptr(g_rot), C.byref(st),
                          _lib.current_stream()), name)
        g_full, db, g_H = _context_pullback(model, rec.DCTX, H, rot.reshape(B * N, P, 9), want_weight_grads=ctx.needs_input_grad[4])
        if ctx.needs_input_grad[4]:                                   # the token
            ctx.stash[(id(ctx), 'ctx')] = ('ctx', g_full, db)
            ctx.stash[(id(ctx), 'hyper')] = ('hyperv', 0, P, rec.grads()) if variant else ('hyper', 0, P, rec.hypernet_grads())
        return None, g_H, None, None, g_rot.new_zeros(()), None


def _lbs_vertices_torch(smpl, glob9, body207, betas):
    """The posed vertices as a differentiable composition of torch operators on the module's buffers (smplx lbs: shape blend, rest joints, pose blend, the 24-joint
    chain, 4-sparse skinning written densely).  Used ONLY to pull a VERTEX gradient back (round 6: LOSS.APPLY_POINT_EST_LOSS, vertex terms of an optimisation) -- the
    forward vertices always come from hmf_smpl_lbs; this restatement never produces an output, it is differentiated at the saved inputs."""
    M = betas.shape[0]
    R = torch.cat([glob9.view(M, 1, 3, 3), body207.view(M, 23, 3, 3)], dim=1)
    v_shaped = smpl.v_template[None] + torch.einsum('vck,mk->mvc', smpl.shapedirs, betas)
    J = torch.einsum('jv,mvc->mjc', smpl.J_regressor, v_shaped)
    feat = (R[:, 1:] - torch.eye(3, dtype=R.dtype, device=R.device)).reshape(M, 207)
    v_posed = v_shaped + (feat @ smpl.posedirs).view(M, -1, 3)
    parents = smpl.parents.tolist()
    GR, Gt = [R[:, 0]], [J[:, 0]]
    for i in range(1, 24):
        p = parents[i]
        GR.append(GR[p] @ R[:, i])
        Gt.append((GR[p] @ (J[:, i] - J[:, p])[..., None])[..., 0] + Gt[p])
    GR, Gt = torch.stack(GR, dim=1), torch.stack(Gt, dim=1)                              # (M,24,3,3), (M,24,3)
    At = Gt - (GR @ J[..., None])[..., 0]                                                # A_i = [G_R | G_t - G_R J_i]
    TR = torch.einsum('vj,mjab->mvab', smpl.lbs_weights, GR)
    Tt = torch.einsum('vj,mja->mva', smpl.lbs_weights, At)
    return (TR @ v_posed[..., None])[..., 0] + Tt


class SmplJoints(torch.autograd.Function):
    """hmf_smpl_lbs under autograd.  JOINTS: the backward is hmf_smpl_joints_backward (re-derives the joints from the fp32 blend operand whatever `impl` the
    forward ran with: with impl = 2 (bf16 blend-shape operands, BASELINE configs[2]) the gradient is that of the fp32 function at the same inputs -- the two forwards
    differ by <= 1.5e-3 m, the Jacobians to the same relative order; training (configs[4]) runs impl = 0).  VERTICES (round 6): when a vertex gradient arrives --
    the opt-in LOSS.APPLY_POINT_EST_LOSS of the reference, vertex terms of an optimisation; never the default training step -- it is pulled back through
    `_lbs_vertices_torch` on torch operators and added to the joints' gradients."""

    @staticmethod
    def forward(ctx, smpl, glob9, body207, betas, impl):
        if impl == -1:                                                # joints only (hmf_smpl_joints): the training loss never reads the vertices
            joints = smpl._joints_kernel(glob9.detach(), body207.detach(), betas.detach())
            verts = joints.new_empty(0)
            ctx.mark_non_differentiable(verts)
        else:
            verts, joints = smpl._lbs_kernel(glob9.detach(), body207.detach(), betas.detach(), impl)
        ctx.smpl = smpl
        ctx.save_for_backward(glob9.detach(), body207.detach(), betas.detach())
        ctx.set_materialize_grads(False)
        return verts, joints

    @staticmethod
    def backward(ctx, g_verts, g_joints):
        if g_verts is None and g_joints is None:
            return None, None, None, None, None
        glob9, body207, betas = ctx.saved_tensors
        M = body207.shape[0]
        gg = gb = gbe = None
        if g_joints is not None:
            lib = _lib.load()
            gj = g_joints.detach().to(torch.float32).contiguous()
            gg, gb, gbe = torch.empty_like(glob9), torch.empty_like(body207), torch.empty_like(betas)
            with torch.cuda.device(gj.device):
                _lib.check(lib.hmf_smpl_joints_backward(ctx.smpl._handle(gj.device), _lib.dptr(glob9), _lib.dptr(body207), _lib.dptr(betas), M,
                                                        _lib.dptr(gj), _lib.dptr(gg), _lib.dptr(gb), _lib.dptr(gbe), _lib.current_stream()),
                           'hmf_smpl_joints_backward')
        if g_verts is not None:
            with torch.enable_grad():
                ins = [t.detach().clone().requires_grad_(True) for t in (glob9, body207, betas)]
                v = _lbs_vertices_torch(ctx.smpl, *ins)
                vg = torch.autograd.grad(v, ins, g_verts.detach().to(v.dtype))
            gg, gb, gbe = (a if b is None else b + a for a, b in zip(vg, (gg, gb, gbe)))
        return None, gg, gb, gbe, None


def _rot6d_torch(x):
    """utils/rigid_transform_utils.py:86-100 in torch ops (used for the pull-back only)."""
    x = x.reshape(-1, 3, 2)
    a1, a2 = x[:, :, 0], x[:, :, 1]
    b1 = F.normalize(a1)
    b2 = F.normalize(a2 - torch.einsum('bi,bi->b', b1, a2).unsqueeze(-1) * b1)
    b3 = torch.linalg.cross(b1, b2, dim=-1)
    return torch.stack((b1, b2, b3), dim=-1)


class Rot6dToRotmat(torch.autograd.Function):
    """hmf_rot6d_to_rotmat forward; hmf_rot6d_to_rotmat_backward = the pull-back of the 6 -> 9 Gram-Schmidt map in one launch
    (`_rot6d_torch` above restates the map in torch ops: the tests differentiate it as the reference)."""

    @staticmethod
    def forward(ctx, x):
        from .rigid_transform_utils import rot6d_to_rotmat_kernel
        ctx.save_for_backward(x.detach())
        return rot6d_to_rotmat_kernel(x.detach())

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        lib = _lib.load()
        x6 = x.reshape(-1, 6).float().contiguous()
        gR = g.reshape(-1, 9).float().contiguous()
        gx = torch.empty_like(x6)
        with torch.cuda.device(x.device):
            _lib.check(lib.hmf_rot6d_to_rotmat_backward(x6.data_ptr(), gR.data_ptr(), x6.shape[0], gx.data_ptr(), _lib.current_stream()),
                       'hmf_rot6d_to_rotmat_backward')
        return gx.reshape(x.shape).to(x.dtype)


class IsgcTrain(torch.autograd.Function):
    """compute_input_shape_glob_cam_feats (models/humaniflow_model.py:116-150) under autograd for BOTH uses of a training pass -- the per-sample rows of the
    reparameterised samples (:302-311, predicted global rotation) and the teacher-forced rows (:279-283, target global rotation, target shape) -- on the
    library: forward hmf_flow_isgc_train (three launches), backward hmf_flow_isgc_train_backward (two) + hmf_rows_atb_batched for every dense product.
    Round 4 ran it as F.linear x 5 + cat + ELU on torch operators: the one vendor GEMM kernel left in a training step and ~45 small launches.
    Reads the layer's weights from the flow handle's arena (kept in step with the parameters by HumaniflowModel._flow_handle: re-packed whenever a
    parameter's version counter or storage address changes), returns the gradient w.r.t. the parameter tensors.  The forward therefore runs on the
    MODULE's weights: a `W` that is another tensor than the module's parameter (torch.func.functional_call with substituted parameters) would give a
    forward and a backward on different weights, so it is refused."""

    @staticmethod
    def forward(ctx, model, feats, W, bias, glob_R, cam, shape_samples, glob_R_ll, shape_ll):
        lib = _lib.load()
        own_W, own_b = model._flow_params()[0], model._flow_params()[1]
        if (W is not own_W and W.data_ptr() != own_W.data_ptr()) or (bias is not own_b and bias.data_ptr() != own_b.data_ptr()):
            raise RuntimeError('IsgcTrain: the weight/bias passed in are not fc_input_shape_glob_cam_feats\' own parameters; the kernel reads the packed '
                               'arena of the module parameters (substituted parameters are not supported: set HMF_ISGC_TRAIN=0 for the torch-operator form)')
        f = lambda t: t.detach().to(torch.float32).contiguous()
        feats_, R_, cam_, ss_, Rl_, sl_ = f(feats), f(glob_R).reshape(-1, 9), f(cam), f(shape_samples), f(glob_R_ll).reshape(-1, 9), f(shape_ll)
        B, N, nb = ss_.shape
        dev = feats_.device
        pre = torch.empty(2, B, 256, dtype=torch.float32, device=dev)
        H = torch.empty(B * N + B, 256, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.hmf_flow_isgc_train(model._flow_handle(dev), _lib.dptr(feats_), _lib.dptr(R_), _lib.dptr(Rl_), _lib.dptr(cam_), _lib.dptr(ss_),
                                               _lib.dptr(sl_), B, N, _lib.dptr(pre), _lib.dptr(H), _lib.current_stream()), 'hmf_flow_isgc_train')
        rc = torch.cat([torch.cat([Rl_, cam_], dim=1), torch.cat([R_, cam_], dim=1)], dim=0)          # (2B, 12): teacher-forced rows, then sample-path rows
        shapes = torch.cat([ss_.reshape(B * N, nb), sl_], dim=0)
        ctx.save_for_backward(feats_, W.detach(), H, rc, shapes)
        ctx.model, ctx.dims = model, (B, N, nb)
        ctx.needs = (feats.requires_grad, glob_R.requires_grad, cam.requires_grad, shape_samples.requires_grad)
        return H[:B * N].view(B, N, 256), H[B * N:]

    @staticmethod
    def backward(ctx, g_rows, g_ll):
        lib = _lib.load()
        feats, W, H, rc, shapes = ctx.saved_tensors
        B, N, nb = ctx.dims
        Fd, in_dim = feats.shape[1], W.shape[1]
        dev = feats.device
        f = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()
        g_rows, g_ll = f(g_rows), f(g_ll)
        new = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)
        delta, g_pre_f, g_pre_f_t, g_pre_s_t, g_shape = new(B * N + 2 * B, 256), new(B, 256), new(256, B), new(256, B), new(B * N + B, nb)
        with torch.cuda.device(dev):
            _lib.check(lib.hmf_flow_isgc_train_backward(ctx.model._flow_handle(dev), _lib.dptr(H), _lib.dptr(g_rows), _lib.dptr(g_ll), B, N, _lib.dptr(delta),
                                                        _lib.dptr(g_pre_f), _lib.dptr(g_pre_f_t), _lib.dptr(g_pre_s_t), _lib.dptr(g_shape),
                                                        _lib.current_stream()), 'hmf_flow_isgc_train_backward')
        Wc = W.to(torch.float32)
        assert Wc.is_contiguous()
        gW = new(256, in_dim)
        gWf, gb = _atb(g_pre_f, feats, 1, B, 256, Fd, 256, Fd, 0, 0, True)                       # d W[:, :F], d bias
        gWs = _atb(delta, shapes, 1, B * N + B, 256, nb, 256, nb, 0, 0, False)[0]                # d W[:, F:F+nb]
        gWrc = _atb(delta[B * N:], rc, 1, 2 * B, 256, 12, 256, 12, 0, 0, False)[0]               # d W[:, F+nb:]: teacher-forced rows + g_pre_s rows
        gW[:, :Fd] = gWf[0]
        gW[:, Fd:Fd + nb] = gWs[0]
        gW[:, Fd + nb:] = gWrc[0]
        need_f, need_R, need_c, need_s = ctx.needs
        g_feats = _atb(g_pre_f_t, Wc, 1, 256, B, Fd, B, in_dim, 0, 0, False)[0][0] if need_f else None
        g_R = _atb(g_pre_s_t, Wc[:, Fd + nb:], 1, 256, B, 9, B, in_dim, 0, 0, False)[0][0].view(B, 3, 3) if need_R else None
        g_cam = _atb(g_pre_f_t, Wc[:, Fd + nb + 9:], 1, 256, B, 3, B, in_dim, 0, 0, False)[0][0] if need_c else None
        g_ss = g_shape[:B * N].view(B, N, nb) if need_s else None
        return None, g_feats, gW, gb[0], g_R, g_cam, g_ss, None, None


class HeadsTrain(torch.autograd.Function):
    """models/humaniflow_model.py:232-253 under autograd on the library: forward = hmf_heads_train (fc1 + ELU, fc_cam / fc_glob / fc_shape, init adds,
    6-D -> rotation: two launches), backward = hmf_heads_backward (everything below fc1's pre-activation, one launch) + two hmf_rows_atb_batched
    calls (d fc1.weight / d fc1.bias, and the pull-back to the image features).  Replaces four nn.Linear (one vendor GEMM kernel per step in
    profiles/r04_train_kernel_stats.md) and ~40 small ATen launches per training step."""

    @staticmethod
    def forward(ctx, feats, w1, b1, wc, bc, wg, bg, ws, bs, init_cam, init_glob):
        lib = _lib.load()
        f = lambda t: t.detach().to(torch.float32).contiguous()
        feats_, w1_, b1_, wc_, bc_, wg_, bg_, ws_, bs_ = map(f, (feats, w1, b1, wc, bc, wg, bg, ws, bs))
        B, Fd = feats_.shape
        N1, nb = w1_.shape[0], ws_.shape[0] // 2
        dev = feats_.device
        new = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)
        cam, glob_R, mode, lstd, std, x1, glob6 = new(B, 3), new(B, 3, 3), new(B, nb), new(B, nb), new(B, nb), new(B, N1), new(B, 6)
        ws_buf = torch.empty(int(lib.hmf_heads_workspace_bytes(B, Fd, N1)), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.hmf_heads_train(_lib.dptr(feats_), B, Fd, _lib.dptr(w1_), _lib.dptr(b1_), N1, _lib.dptr(wc_), _lib.dptr(bc_), _lib.dptr(f(init_cam)),
                                           _lib.dptr(wg_), _lib.dptr(bg_), _lib.dptr(f(init_glob)), _lib.dptr(ws_), _lib.dptr(bs_), nb, _lib.dptr(cam),
                                           _lib.dptr(glob_R), _lib.dptr(mode), _lib.dptr(lstd), _lib.dptr(std), _lib.dptr(x1), _lib.dptr(glob6), ws_buf.data_ptr(),
                                           ws_buf.numel(), _lib.current_stream()), 'hmf_heads_train')
        ctx.save_for_backward(feats_, w1_, wc_, wg_, ws_, x1, glob6)
        ctx.need_feats = feats.requires_grad
        return cam, glob_R, mode, lstd

    @staticmethod
    def backward(ctx, g_cam, g_R, g_mode, g_lstd):
        lib = _lib.load()
        feats, w1, wc, wg, ws, x1, glob6 = ctx.saved_tensors
        B, Fd = feats.shape
        N1, nb = w1.shape[0], ws.shape[0] // 2
        dev = feats.device
        f = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()
        g_cam, g_R, g_mode, g_lstd = f(g_cam), f(g_R), f(g_mode), f(g_lstd)
        new = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)
        g_pre, g_pre_t = new(B, N1), new(N1, B)
        gwc, gbc, gwg, gbg, gws, gbs = new(3, N1), new(3), new(6, N1), new(6), new(2 * nb, N1), new(2 * nb)
        with torch.cuda.device(dev):
            _lib.check(lib.hmf_heads_backward(_lib.dptr(x1), _lib.dptr(glob6), B, N1, nb, _lib.dptr(wc), _lib.dptr(wg), _lib.dptr(ws), _lib.dptr(g_cam),
                                              None if g_R is None else g_R.data_ptr(), _lib.dptr(g_mode), _lib.dptr(g_lstd), _lib.dptr(g_pre), _lib.dptr(g_pre_t),
                                              _lib.dptr(gwc), _lib.dptr(gbc), _lib.dptr(gwg), _lib.dptr(gbg), _lib.dptr(gws), _lib.dptr(gbs), _lib.current_stream()),
                       'hmf_heads_backward')
        gw1, gb1 = _atb(g_pre, feats, 1, B, N1, Fd, N1, Fd, 0, 0, True)                   # d fc1.weight = g_pre^T feats, d fc1.bias = column sums of g_pre
        g_feats = _atb(g_pre_t, w1, 1, N1, B, Fd, B, Fd, 0, 0, False)[0][0] if ctx.need_feats else None     # g_pre W1 as (g_pre^T)^T W1 over the N1 rows
        return g_feats, gw1[0], gb1[0], gwc, gbc, gwg, gbg, gws, gbs, None, None


class FusedTrainLoss(torch.autograd.Function):
    """HumaniflowLoss on the training step's own quantities (losses/humaniflow_loss.py:25-79, J2D_LOSS_ON = 'point_est+samples', no 3D point-estimate
    terms) with the joint selection / flip / projection of train/train_humaniflow.py:283-318 folded in: hmf_train_loss computes the scalar AND its
    gradients in two launches; backward scales the stored gradients by the incoming one (one multi-tensor launch)."""

    @staticmethod
    def forward(ctx, lp, shape_mode, shape_log_std, joints, cam, glob_R, shape_t, j2d_t, vis, glob_t, N, coco, img_wh, weights, mean):
        lib = _lib.load()
        dev = lp.device
        f = lambda t: t.detach().to(torch.float32).contiguous()
        lp_, mode_, ls_, j_, cam_, g_ = f(lp), f(shape_mode), f(shape_log_std), f(joints), f(cam), f(glob_R).reshape(-1, 9)
        st_, jt_, gt_ = f(shape_t), f(j2d_t), f(glob_t).reshape(-1, 9)
        vis_ = vis.detach().to(torch.uint8).contiguous()
        B, P, nb = lp_.shape[0], lp_.shape[1], mode_.shape[1]
        assert j_.shape == (B * N + B, 90, 3) and cam_.shape == (B, 3) and jt_.shape == (B, 17, 2) and vis_.shape == (B, 17)
        new = lambda t: torch.empty_like(t)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        grads = [new(lp_), new(mode_), new(ls_), new(j_), new(cam_), new(g_)]
        import ctypes as C
        ws = torch.empty(int(lib.hmf_train_loss_workspace_bytes(B)), dtype=torch.uint8, device=dev)
        idx = (C.c_int * 17)(*[int(i) for i in coco])
        with torch.cuda.device(dev):
            _lib.check(lib.hmf_train_loss(_lib.dptr(lp_), P, _lib.dptr(mode_), _lib.dptr(ls_), _lib.dptr(st_), nb, _lib.dptr(j_), _lib.dptr(cam_), idx, _lib.dptr(jt_),
                                          vis_.data_ptr(), _lib.dptr(g_), _lib.dptr(gt_), B, N, float(img_wh), float(weights[0]), float(weights[1]), float(weights[2]),
                                          float(weights[3]), int(mean), _lib.dptr(loss), *[_lib.dptr(t) for t in grads], ws.data_ptr(), ws.numel(),
                                          _lib.current_stream()), 'hmf_train_loss')
        ctx.grads = grads
        ctx.glob_shape = glob_R.shape
        return loss

    @staticmethod
    def backward(ctx, g):
        if ctx.grads is None:
            raise RuntimeError('humaniflow_amd: FusedTrainLoss gradients were freed')
        if g.requires_grad:
            raise RuntimeError('humaniflow_amd: FusedTrainLoss has no double-backward (create_graph=True); use HumaniflowLoss on torch operators for that')
        grads = list(torch._foreach_mul(ctx.grads, g.to(torch.float32)))          # the stored gradients stay: backward may run again (retain_graph=True)
        grads[5] = grads[5].reshape(ctx.glob_shape)
        return tuple(grads) + (None,) * 9
