"""Pin the proxy front-end oracle (Canny + heat-maps) on vectors produced by the reference's own modules."""
import os

import numpy as np
import torch

from helpers import GOLDEN
from oracle import proxy


def test_canny_and_heatmaps_match_reference_modules():
    z = np.load(os.path.join(GOLDEN, 'proxy_golden.npz'))
    img = torch.from_numpy(z['img'])
    for nms, thr in ((True, 0.0), (True, 0.2), (False, 0.1)):
        out = proxy.canny(img, nms, 1.0, 5, thr)
        for k, v in out.items():
            ref = z['canny_%d_%g_%s' % (int(nms), thr, k)]
            diff = np.abs(v.numpy() - ref)
            if k in ('thin_edges', 'thresholded_thin_edges', 'grad_orientation', 'thresholded_grad_magnitude'):
                # discontinuous in the gradients (orientation bins, > 0 tests): allow isolated flips from summation order
                assert (diff > 1e-5).mean() < 2e-3, (k, (diff > 1e-5).mean())
            else:
                assert diff.max() < 1e-5, (k, diff.max())
    hm = proxy.gaussian_heatmaps(torch.from_numpy(z['joints2D']), 48, 4.0)
    assert np.abs(hm.numpy() - z['heatmaps']).max() < 1e-6
    pr = proxy.proxy_representation(img, torch.from_numpy(z['joints2D']))
    assert pr.shape == (2, 18, 48, 48)
    assert np.abs(proxy.gaussian_taps().numpy().sum() - 1) < 1e-6
