"""Round 6: what-if (timing-only, WRONG results) builds of the 1x1 family and of the kinematic loop inside the THREE-STREAM step (DESIGN section 3, fact 6's
method): what would each never-built item buy the headline AT MOST?

    python tools/whatif_1x1.py [reps]          (GPU box; uses tools/libhumaniflow_amd_experiments.so)

variants:  base        the product kernels
           pair_nostream   layer1's two seam launches (k_conv_pair) with W3 / W1 loaded once (no L2 -> VGPR operand stream): ceiling of "seams on the LDS-DMA ring"
           pair_pinned     the same two launches with their operand requests pinned by scheduling barriers (CORRECT results; k_lbs_mfma16's round-6 lever)
           gemm_noxpose    every k_conv_gemm epilogue without its transposition through LDS (same stores, same residual loads): ceiling of a cheaper epilogue
           gemm_a_once     K <= 256 k_conv_gemm launches with several n-tiles: the activation panel requested for the prologue chunks only: ceiling of "panel-resident"
           gemm_both       the two above together
           no_chain        the step without k_flow_chain: ceiling of any faster kinematic-loop kernel for the headline
           chain16         k_flow_chain16 (product kernel, 16-row tiles: 117 us alone instead of 158) forced at 3,232 rows
Each variant: `reps` interleaved runs of bench.run_inference (same process, same streams), ms per step of the timed region; also the one-stream eager kernel sum
of the 1x1 family from the measurement pass."""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from humaniflow_amd import _lib
lib = _lib.use_experiments_library()
import bench

VARIANTS = [('base', 0, 0, 0), ('pair_nostream', 0, 32, 0), ('pair_pinned', 0, 64, 0), ('gemm_noxpose', 1, 0, 0), ('gemm_a_once', 2, 0, 0), ('gemm_both', 3, 0, 0), ('no_chain', 0, 0, 1),
            ('chain16', 0, 0, 16)]       # chain16: the PRODUCT 16-row kinematic loop forced at full size (correct results)
if os.environ.get('WHATIF_ONLY'):
    VARIANTS = [v for v in VARIANTS if v[0] in os.environ['WHATIF_ONLY'].split(',')]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    sys.argv = [sys.argv[0]]
    args = bench.parse()
    args.steps, args.warmup, args.profile_steps, args.no_cpu_baseline = 60, 6, 3, True
    res = {v[0]: [] for v in VARIANTS}
    fam = {v[0]: [] for v in VARIANTS}
    for r in range(reps):
        for name, g, p, f in VARIANTS:
            lib.hmf_debug_set_gemm_whatif(g)
            lib.hmf_debug_set_pair_whatif(p)
            lib.hmf_debug_set_flow_whatif(f & 1)
            import humaniflow_amd as H
            H.HumaniflowModel.flow_chain32 = False if (f & 16) else None
            out, _ = bench.run_inference(copy.copy(args), dev, None, 1, 0)
            res[name].append(out['ms_per_step'])
            ck = out['conv_kernels'] or {}
            fam[name].append(sum(v['us_per_step'] for k, v in ck.items() if '1x1' in k))
            torch.cuda.synchronize()
    lib.hmf_debug_set_gemm_whatif(0); lib.hmf_debug_set_pair_whatif(0); lib.hmf_debug_set_flow_whatif(0)
    base = float(np.median(res['base']))
    print('what-if builds inside the three-stream step (B = 32, N = 100, 60 timed steps, %d interleaved repetitions; median [all])' % reps)
    print('%-16s %10s %10s   %s   %s' % ('variant', 'ms/step', 'delta us', 'all runs', '1x1 family one-stream us (eager, HIP events)'))
    for name, *_ in VARIANTS:
        m = float(np.median(res[name]))
        print('%-16s %10.3f %+10.1f   %s   %.1f' % (name, m, (m - base) * 1e3, ' '.join('%.3f' % v for v in res[name]), float(np.median(fam[name]))))


if __name__ == '__main__':
    main()
