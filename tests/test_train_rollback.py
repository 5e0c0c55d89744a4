"""Host logic of the reference's NaN roll-back (train/train_humaniflow.py:261-278 reload, :348-349 deep copies) as train.LastBatchState keeps it:
in-place restore of parameters, buffers and optimiser state must give the trajectory the reference's deepcopy + load_state_dict gives."""
import copy

import pytest
import torch

from humaniflow_amd import train as T


def _toy(seed):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.BatchNorm1d(8), torch.nn.ReLU(), torch.nn.Linear(8, 3))


def _batches(n):
    g = torch.Generator().manual_seed(11)
    return [(torch.randn(16, 6, generator=g), torch.randn(16, 3, generator=g)) for _ in range(n)]


@pytest.mark.parametrize('make_opt', [lambda ps: torch.optim.Adam(ps, lr=1e-2), lambda ps: torch.optim.SGD(ps, lr=1e-2, momentum=0.9)])
@pytest.mark.parametrize('poison_after', [0, 2])
def test_in_place_restore_follows_the_reference_reload_trajectory(make_opt, poison_after):
    batches = _batches(5)

    def run(reference_style):
        m = _toy(3).train()
        opt = make_opt(list(m.parameters()))
        keep = T.LastBatchState(m, opt)
        last_m = last_o = None
        for i, (x, y) in enumerate(batches):
            out = m(x)
            if bool(out.isnan().any()):
                if reference_style:
                    m.load_state_dict(last_m)
                    opt.load_state_dict(last_o)
                else:
                    keep.restore()
                out = m(x)
                assert not bool(out.isnan().any())
            opt.zero_grad()
            loss = ((out - y) ** 2).mean()
            if reference_style:
                last_m, last_o = copy.deepcopy(m.state_dict()), copy.deepcopy(opt.state_dict())
            else:
                keep.save()
            loss.backward()
            opt.step()
            if i == poison_after:                                   # an update that leaves a NaN weight behind, found by the next forward
                with torch.no_grad():
                    m[0].weight[0, 0] = float('nan')
        return m, opt

    (ma, oa), (mb, ob) = run(False), run(True)
    for (ka, va), (kb, vb) in zip(ma.state_dict().items(), mb.state_dict().items()):
        assert ka == kb and torch.equal(va, vb), ka
    sa, sb = oa.state_dict()['state'], ob.state_dict()['state']
    assert sa.keys() == sb.keys()
    for k in sa:
        assert sa[k].keys() == sb[k].keys()
        for kk in sa[k]:
            va, vb = sa[k][kk], sb[k][kk]
            assert torch.equal(torch.as_tensor(va), torch.as_tensor(vb)), (k, kk)


def test_restore_keeps_parameter_and_moment_storage():
    m = _toy(0).train()
    opt = torch.optim.Adam(list(m.parameters()), lr=1e-2)
    keep = T.LastBatchState(m, opt)
    with pytest.raises(RuntimeError):
        keep.restore()
    (x, y), (x2, y2) = _batches(2)
    for xb, yb in ((x, y), (x2, y2)):
        opt.zero_grad()
        loss = ((m(xb) - yb) ** 2).mean()
        keep.save()
        loss.backward()
        opt.step()
    ptr_p = [p.data_ptr() for p in m.parameters()]
    ptr_s = [opt.state[p]['exp_avg'].data_ptr() for p in m.parameters()]
    w_now = [p.detach().clone() for p in m.parameters()]
    keep.restore()
    assert ptr_p == [p.data_ptr() for p in m.parameters()]
    assert ptr_s == [opt.state[p]['exp_avg'].data_ptr() for p in m.parameters()]
    assert not all(torch.equal(a, b.detach()) for a, b in zip(w_now, m.parameters()))
    assert all(float(opt.state[p]['step']) == 1.0 for p in m.parameters())


def test_nans_in_output_reads_every_floating_tensor_of_the_dict():
    ok = {'a': torch.zeros(3), 'b': torch.ones(2, 2, dtype=torch.float64), 'idx': torch.arange(3), 'dists': [1, 2]}
    assert not bool(T.nans_in_output(ok))
    ok['b'][1, 0] = float('nan')
    assert bool(T.nans_in_output(ok))
    assert not bool(T.nans_in_output({'n': 3}))


def test_train_step_rejects_unknown_policy_and_missing_state():
    with pytest.raises(ValueError):
        T.train_step(None, None, None, None, None, None, None, 2, nan_policy='reload')
    with pytest.raises(ValueError):
        T.train_step(None, None, None, None, None, None, None, 2, nan_policy='rollback')


@pytest.mark.parametrize('permute,transform,T,hidden,plu_hidden', [('linear_plu', 'spline_coupling', 2, (64, 32, 32), None),
                                                                    ('conditional_linear_plu', 'additive_coupling', 3, (40, 20), (16,)),
                                                                    ('permute', 'spline_coupling', 3, (48, 48, 24, 24), None)])
def test_variant_gradient_bookkeeping_matches_the_module_parameter_order(permute, transform, T, hidden, plu_hidden):
    """Host logic of the variants' training path (autograd._variant_params / _variant_unpack / _variant_embed / _variant_sum): the stacked per-layer
    gradient tensors the reverse sweeps produce are unpacked into exactly the parameters FlowParamAnchor registered, in the same order and shapes."""
    import humaniflow_amd as H
    from humaniflow_amd import autograd as AG
    cfg = H.get_humaniflow_cfg_defaults()
    nf = cfg.MODEL.NORM_FLOW
    nf.PERMUTE_TYPE, nf.TRANSFORM_TYPE, nf.NUM_TRANSFORMS = permute, transform, T
    nf.TRANSFORM_NN_HIDDEN_DIMS = list(hidden)
    nf.PERMUTE_NN_HIDDEN_DIMS = None if plu_hidden is None else list(plu_hidden)
    cfg.MODEL.NUM_RESNET_LAYERS = 18
    model = H.HumaniflowModel(torch.device('cpu'), cfg.MODEL, H.SMPL_PARENTS)
    assert AG._is_variant(model)
    P = model.num_bodyparts
    params = AG._variant_params(model, 0, P)
    # every parameter of the transform modules exactly once, in module order
    want = [p for m in model.pose_so3flow_transform_modules for p in m.parameters()]
    assert len(params) == len(want) and all(a is b for a, b in zip(params, want))
    Tn, L, LP, K, N, KP, NP = AG._variant_dims(model)
    assert (Tn, L) == (T, len(hidden) + 1) and K[0] == 65 and N[-1] == (62 if transform == 'spline_coupling' else 2)
    g = torch.Generator().manual_seed(0)
    mk = lambda *s: torch.randn(*s, generator=g)
    grads = {'tr': [(mk(P, T, N[l], K[l]), mk(P, T, N[l])) for l in range(L)],
             'plu': [(mk(P, T, NP[l], KP[l]), mk(P, T, NP[l])) for l in range(LP)] if LP else None,
             'lu': mk(P, T, 3, 3) if permute == 'linear_plu' else None}
    flat = AG._variant_unpack(model, grads)
    assert len(flat) == len(params) and all(tuple(a.shape) == tuple(b.shape) for a, b in zip(flat, params))
    # a sub-range of parts embeds into the full stack with zeros elsewhere, and sums add tensor by tensor
    sub = {'tr': [(w[3:5], b[3:5]) for w, b in grads['tr']], 'plu': None if grads['plu'] is None else [(w[3:5], b[3:5]) for w, b in grads['plu']],
           'lu': None if grads['lu'] is None else grads['lu'][3:5]}
    emb = AG._variant_embed(model, sub, 3, 2)
    assert torch.equal(emb['tr'][0][0][3:5], grads['tr'][0][0][3:5]) and float(emb['tr'][0][0][:3].abs().sum()) == 0.0 and float(emb['tr'][0][0][5:].abs().sum()) == 0.0
    tot = AG._variant_sum(AG._variant_sum(None, grads), emb)
    assert torch.equal(tot['tr'][-1][1][4], grads['tr'][-1][1][4] * 2) and torch.equal(tot['tr'][-1][1][0], grads['tr'][-1][1][0])
    if grads['lu'] is not None:
        assert torch.equal(tot['lu'][3], grads['lu'][3] * 2)
