"""GPU parity: hmf_proxy_rep (fused Canny + heat-maps) vs the oracle and the reference-generated golden vectors."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN
from oracle import proxy as oproxy

pytestmark = pytest.mark.gpu


def _frac_diff(a, b, tol=1e-5):
    return (np.abs(a - b) > tol).mean()


def test_canny_vs_golden_and_oracle():
    from humaniflow_amd.proxy import CannyEdgeDetector, convert_2Djoints_to_gaussian_heatmaps_torch
    z = np.load(os.path.join(GOLDEN, 'proxy_golden.npz'))
    img = torch.from_numpy(z['img']).cuda()
    for nms, thr in ((True, 0.0), (True, 0.2), (False, 0.1)):
        out = CannyEdgeDetector(non_max_suppression=nms, threshold=thr)(img)
        for k, v in out.items():
            ref = z['canny_%d_%g_%s' % (int(nms), thr, k)]
            if k == 'grad_magnitude':
                assert np.abs(v.cpu().numpy() - ref).max() < 1e-5
            else:   # orientation bins / NMS are discontinuous in the gradients: isolated flips from summation order are allowed
                assert _frac_diff(v.cpu().numpy(), ref) < 2e-3, (k, _frac_diff(v.cpu().numpy(), ref))
    hm = convert_2Djoints_to_gaussian_heatmaps_torch(torch.from_numpy(z['joints2D']).cuda(), 48, 4.0)
    assert np.abs(hm.cpu().numpy() - z['heatmaps']).max() < 1e-6


@pytest.mark.parametrize('B,S', [(1, 31), (3, 256), (2, 100)])
def test_proxy_representation_matches_oracle(B, S):
    from humaniflow_amd.proxy import proxy_representation
    import humaniflow_amd as H
    g = torch.Generator().manual_seed(S)
    img = torch.nn.functional.interpolate(torch.rand(B, 3, 9, 9, generator=g), size=(S, S), mode='bicubic', align_corners=False).contiguous()
    j2d = torch.rand(B, 17, 2, generator=g) * S
    vis = (torch.rand(B, 17, generator=g) > 0.2).float()
    out = proxy_representation(img.cuda(), j2d.cuda(), vis.cuda(), H.get_humaniflow_cfg_defaults().DATA).cpu()
    ref = oproxy.proxy_representation(img, j2d, vis)
    assert out.shape == (B, 18, S, S)
    assert (out[:, 1:] - ref[:, 1:]).abs().max() < 1e-6
    assert _frac_diff(out[:, 0].numpy(), ref[:, 0].numpy()) < 2e-3
    assert (out[:, 0] > 0).float().mean() > 0.01                      # edges exist and are thin
    assert (out[:, 0] > 0).float().mean() < 0.6
