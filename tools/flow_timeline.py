"""Phase timeline of k_flow_chain16 (tile 0; `--chain32`: the 32-row kernel of rounds 1-3) from in-kernel shader-clock stamps: where do the 8
dependency levels spend time?"""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import humaniflow_amd as H
from humaniflow_amd import _lib
from humaniflow_amd.synthetic import init_benchmark_weights
lib = _lib.use_experiments_library()      # tools/libhumaniflow_amd_experiments.so: product entry points + timing variants
dev = torch.device('cuda:0')
model = init_benchmark_weights(H.HumaniflowModel(torch.device('cpu'), H.get_benchmark_cfg().MODEL, H.SMPL_PARENTS)).eval().to(dev)
feats = torch.randn(32, 2048, device=dev) * 0.3
model.flow_chain32 = '--chain32' in sys.argv        # False: force the 16-row kernel at this size
tl = torch.zeros(23, 8, dtype=torch.int64, device=dev)
with torch.no_grad():
    for _ in range(3):
        model(None, input_feats=feats, num_samples=100)
    lib.hmf_debug_set_timeline(tl.data_ptr())
    model(None, input_feats=feats, num_samples=100)
    torch.cuda.synchronize()
    lib.hmf_debug_set_timeline(None)
t = tl.cpu().double()
t0 = t[:, 0].min()
names = ['issue loads', 'anc mfma', 'elu', 'L1 mfma', 'coupling0', 'coupling1', 'tanh+exp+store']
anc = H.immediate_parent_to_all_ancestors(H.SMPL_PARENTS)
print('part depth start(cyc)  ' + '  '.join('%12s' % n for n in names) + '   total')
for j in sorted(range(23), key=lambda j: (len(anc[j]), j)):
    d = (t[j, 1:] - t[j, :-1]).tolist()
    print('%4d %5d %10.0f  ' % (j, len(anc[j]), t[j, 0] - t0) + '  '.join('%12.0f' % x for x in d) + '   %8.0f' % (t[j, 7] - t[j, 0]))
print('kernel span (tile 0): %.0f cycles' % (t[:, 7].max() - t0))
