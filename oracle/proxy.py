"""Oracle: proxy-representation front end (Canny edge image + 2D-joint Gaussian heat-maps). Test infrastructure only.

Follows models/canny_edge_detector.py:11-166 (Gaussian 5-tap blur as two zero-padded 1-D cross-correlations per channel,
Sobel gradients summed over channels and averaged, magnitude, orientation binned to 45 degrees, 8-direction non-maximum
suppression, threshold) and utils/label_conversions.py:106-125 (heat-maps; meshgrid in 'ij' order).  Written with explicit
shifted-slice sums instead of nn.Conv2d so it is independent of the reference's modules.
PINNED by tests/golden/proxy_golden.npz (the reference's own modules run on seeded inputs).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def gaussian_taps(size=5, std=1.0):
    """scipy.signal.windows.gaussian(size, std) normalised to sum 1, rounded to fp32 (canny_edge_detector.py:23-24,31)."""
    n = np.arange(size, dtype=np.float64) - (size - 1) / 2.0
    w = np.exp(-0.5 * (n / std) ** 2)
    return torch.from_numpy(w / w.sum()).float()


def _shift(x, dy, dx):
    """x[..., y + dy, x + dx] with zeros outside (what zero-padded cross-correlation reads)."""
    H, W = x.shape[-2:]
    p = F.pad(x, (2, 2, 2, 2))
    return p[..., 2 + dy:2 + dy + H, 2 + dx:2 + dx + W]


def canny(img, non_max_suppression=True, gaussian_filter_std=1.0, gaussian_filter_size=5, threshold=0.2):
    assert gaussian_filter_size == 5
    g = gaussian_taps(5, gaussian_filter_std)
    C = img.shape[1]
    hb = sum(g[t] * _shift(img, 0, t - 2) for t in range(5))
    blurred = sum(g[t] * _shift(hb, t - 2, 0) for t in range(5))
    gx = torch.zeros_like(img[:, :1])
    gy = torch.zeros_like(img[:, :1])
    for c in range(C):
        b = blurred[:, c:c + 1]
        gx = gx + ((_shift(b, -1, -1) + 2 * _shift(b, 0, -1) + _shift(b, 1, -1)) - (_shift(b, -1, 1) + 2 * _shift(b, 0, 1) + _shift(b, 1, 1)))
        gy = gy + ((_shift(b, -1, -1) + 2 * _shift(b, -1, 0) + _shift(b, -1, 1)) - (_shift(b, 1, -1) + 2 * _shift(b, 1, 0) + _shift(b, 1, 1)))
    gx, gy = gx / C, gy / C
    mag = (gx ** 2 + gy ** 2) ** 0.5
    ori = torch.atan2(gy, gx) * (180.0 / np.pi) + 180.0
    ori = torch.round(ori / 45.0) * 45.0
    out = {'blurred_img': blurred, 'grad_magnitude': mag, 'grad_orientation': ori,
           'thresholded_grad_magnitude': torch.where(mag < threshold, torch.zeros_like(mag), mag)}
    if non_max_suppression:
        nb = [(0, 1), (1, 1), (1, 0), (1, -1), (0, -1), (-1, -1), (-1, 0), (-1, 1)]          # filter_0 .. filter_315: centre - neighbour
        d = [mag - _shift(mag, dy, dx) for dy, dx in nb]
        idx = (ori / 45) % 8
        thin = mag.clone()
        for p in range(4):
            oriented = (idx == p) | (idx == p + 4)
            is_max = torch.minimum(d[p], d[p + 4]) > 0.0
            thin = torch.where(oriented & ~is_max, torch.zeros_like(thin), thin)
        out['thin_edges'] = thin
        out['thresholded_thin_edges'] = torch.where(thin < threshold, torch.zeros_like(thin), thin)
    return out


def gaussian_heatmaps(joints2D, img_wh, std=4):
    """utils/label_conversions.py:106-125: (B, N, 2) -> (B, N, img_wh, img_wh); rows compare with joints2D[...,1], columns with [...,0]."""
    r = torch.arange(img_wh).float()
    u = joints2D[:, :, 0, None, None]
    v = joints2D[:, :, 1, None, None]
    rows = r[None, None, :, None]
    cols = r[None, None, None, :]
    return torch.exp(-(((rows - v) / std) ** 2) / 2 - (((cols - u) / std) ** 2) / 2)


def proxy_representation(img, joints2D, visib=None, edge_nms=True, edge_threshold=0.0, edge_std=1.0, heatmap_std=4.0):
    """predict/predict_humaniflow.py:101-109: cat([edges, heatmaps * visibility]) -> (B, 18, S, S)."""
    e = canny(img, edge_nms, edge_std, 5, edge_threshold)
    edge = e['thresholded_thin_edges'] if edge_nms else e['thresholded_grad_magnitude']
    hm = gaussian_heatmaps(joints2D, img.shape[-1], heatmap_std)
    if visib is not None:
        hm = hm * visib[:, :, None, None]
    return torch.cat([edge, hm], dim=1).float()
