"""Proxy-representation front end on the device: Canny edge image + 2D-joint Gaussian heat-maps.

Mirrors models/canny_edge_detector.py (class ``CannyEdgeDetector``: same constructor arguments, ``forward(img)`` returning the
same dict keys except ``blurred_img``) and ``convert_2Djoints_to_gaussian_heatmaps_torch`` of utils/label_conversions.py:106-125;
``proxy_representation`` is the composition at predict/predict_humaniflow.py:101-109 / evaluate/evaluate_humaniflow.py:71-73
producing the (B, 18, S, S) model input in one launch (hmf_proxy_rep).
"""
import torch
from torch import nn

from . import _lib


def _run(img, joints2D, visib, nms, thr, gstd, gsize, hstd, want_proxy, want_extra):
    _lib.require_cuda(img, 'img')
    lib = _lib.load()
    img = img.detach().float().contiguous()
    B, C, S, S2 = img.shape
    assert C == 3 and S == S2, 'expects (B, 3, S, S) images'
    J = 0 if joints2D is None else joints2D.shape[1]
    j = None if joints2D is None else joints2D.detach().float().contiguous()
    v = None if visib is None else visib.detach().float().contiguous()
    proxy = torch.empty(B, 1 + J, S, S, device=img.device) if want_proxy else None
    extra = [torch.empty(B, 1, S, S, device=img.device) for _ in range(3)] if want_extra else [None, None, None]
    with torch.cuda.device(img.device):
        _lib.check(lib.hmf_proxy_rep(_lib.dptr(img), _lib.dptr(j), _lib.dptr(v), B, S, J, int(nms), float(thr), float(gstd), int(gsize),
                                     float(hstd), _lib.dptr(proxy), _lib.dptr(extra[0]), _lib.dptr(extra[1]), _lib.dptr(extra[2]),
                                     _lib.current_stream()), 'hmf_proxy_rep')
    return proxy, extra


class CannyEdgeDetector(nn.Module):
    def __init__(self, non_max_suppression=True, gaussian_filter_std=1.0, gaussian_filter_size=5, threshold=0.2):
        super().__init__()
        self.threshold = threshold
        self.non_max_suppression = non_max_suppression
        self.gaussian_filter_std = gaussian_filter_std
        self.gaussian_filter_size = gaussian_filter_size

    def forward(self, img):
        proxy, (mag, ori, thin) = _run(img, None, None, self.non_max_suppression, self.threshold, self.gaussian_filter_std,
                                       self.gaussian_filter_size, 4.0, True, True)
        out = {'grad_magnitude': mag, 'grad_orientation': ori,
               'thresholded_grad_magnitude': torch.where(mag < self.threshold, torch.zeros_like(mag), mag)}
        if self.non_max_suppression:
            out['thin_edges'] = thin
            out['thresholded_thin_edges'] = proxy
        return out


def convert_2Djoints_to_gaussian_heatmaps_torch(joints2D, img_wh, std=4):
    dummy = torch.zeros(joints2D.shape[0], 3, img_wh, img_wh, device=joints2D.device)
    proxy, _ = _run(dummy, joints2D, None, True, 0.0, 1.0, 5, std, True, False)
    return proxy[:, 1:]


def proxy_representation(img, joints2D, joints2D_visib=None, data_cfg=None):
    """(B,3,S,S) cropped RGB + (B,17,2) 2D joints (+ (B,17) visibility) -> (B,18,S,S), configured by cfg.DATA
    (EDGE_NMS, EDGE_THRESHOLD, EDGE_GAUSSIAN_STD, EDGE_GAUSSIAN_SIZE, HEATMAP_GAUSSIAN_STD; configs/humaniflow_config.py:28-35)."""
    nms, thr, gstd, gsize, hstd = True, 0.0, 1.0, 5, 4.0
    if data_cfg is not None:
        nms, thr, gstd, gsize, hstd = (data_cfg.EDGE_NMS, data_cfg.EDGE_THRESHOLD, data_cfg.EDGE_GAUSSIAN_STD,
                                       data_cfg.EDGE_GAUSSIAN_SIZE, data_cfg.HEATMAP_GAUSSIAN_STD)
    proxy, _ = _run(img, joints2D, joints2D_visib, nms, thr, gstd, gsize, hstd, True, False)
    return proxy
