"""Regenerate humaniflow_amd/model_files/extra_joint_regressors.npz from a reference checkout.

    python tools/make_model_files.py /root/reference
"""
import sys
import numpy as np

ref = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
out = {}
for key, f in [('extra', 'J_regressor_extra'), ('cocoplus', 'cocoplus_regressor'), ('h36m', 'J_regressor_h36m')]:
    a = np.load('%s/model_files/%s.npy' % (ref, f))
    r, c = np.nonzero(a)
    out[key + '_rows'] = r.astype(np.int16)
    out[key + '_cols'] = c.astype(np.int16)
    out[key + '_vals'] = a[r, c].astype(np.float64)
    out[key + '_shape'] = np.array(a.shape, np.int32)
np.savez_compressed('humaniflow_amd/model_files/extra_joint_regressors.npz', **out)
