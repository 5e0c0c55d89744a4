"""Oracle: the reference's evaluation metric functions and tracker update rules on CPU tensors.  Test infrastructure only.

Restates utils/eval_utils.py:62-125 (procrustes_analysis_batch, scale_and_translation_transform_batch) and the update rules of
metrics/eval_metrics_tracker.py:95-433 op for op in torch; PINNED by tests/golden/metrics_golden.npz and tracker_golden.npz,
both generated from the reference's own numpy code (tests/golden/make_golden.py).  The product's device epilogue
(humaniflow_amd/csrc/eval_kernels.hip) is checked against these functions; the product never imports this module.
"""
import torch


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


def _pa_many(P, T):
    sh = P.shape
    return procrustes_analysis_batch(P.reshape(-1, *sh[-2:]), T.expand_as(P).reshape(-1, *sh[-2:])).reshape(sh)



def pointset_errors(P, T, want_pa=True):
    """CPU counterpart of hmf_pointset_errors: P (..., npts, 3) vs T (Bt, npts, 3), P's leading dims (Bt,) or (Bt, N)
    -> (..., 3) mean errors [plain, scale+translation corrected, Procrustes aligned]."""
    Tb = T if P.dim() == T.dim() else T[:, None].expand_as(P)
    nrm = lambda a, b: torch.linalg.norm(a - b, dim=-1).mean(-1)
    pa = nrm(_pa_many(P, Tb) if P.dim() > 3 else procrustes_analysis_batch(P, Tb), Tb) if want_pa else torch.zeros(P.shape[:-2], dtype=P.dtype)
    return torch.stack([nrm(P, Tb), nrm(scale_and_translation_transform_batch(P, Tb), Tb), pa], dim=-1)


def sample_diversity(S):
    """CPU counterpart of hmf_sample_diversity: S (B, N, npts, 3) -> (B,) sum over samples and points of |S - mean_n S|."""
    return torch.linalg.norm(S - S.mean(dim=1, keepdim=True), dim=-1).double().sum(dim=(1, 2))
