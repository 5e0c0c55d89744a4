"""Consolidate profiles/<round>_bench.json, <round>_pmc.json and <round>_mfma_busy.txt into one per-kernel roofline table.
    python tools/roofline_table.py r01 > profiles/r01_roofline.md"""
import json
import os
import re
import sys

R = sys.argv[1] if len(sys.argv) > 1 else 'r01'
P = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'profiles')
b = json.load(open(os.path.join(P, R + '_bench.json')))
pmc = json.load(open(os.path.join(P, R + '_pmc.json')))
busy = {}
for line in open(os.path.join(P, R + '_mfma_busy.txt')):
    m = re.match(r'(\S.*?)\s+grid\s+(\d+).*mfma_busy=([0-9.]+)%', line)
    if m:
        busy.setdefault(m.group(1).strip(), []).append(float(m.group(3)))
PEAK_F, PEAK_B = 157.3, 8.0


def traffic(key):
    keys_ = key if isinstance(key, tuple) else (key,)
    c = [v for k, v in pmc.get('kernels', pmc).items() if any(kk in k for kk in keys_)]
    return sum(v['hbm_bytes_per_launch'] * v['launches'] for v in c) / max(1, sum(v['launches'] for v in c)) if c else None


def mb(key):
    keys_ = key if isinstance(key, tuple) else (key,)
    v = [x for k, l in busy.items() if any(kk in k for kk in keys_) for x in l]
    return '%.0f-%.0f %%' % (min(v), max(v)) if v else '-'


print('# Per-kernel roofline, %s (1x MI355X, B=32, N=100, fp32; one stream unless noted)\n' % R)
print('Peaks: fp32 MFMA %.1f TFLOP/s, HBM %.0f TB/s.  Durations: HIP events on the launch stream in the one-stream measurement pass of bench.py (`conv_kernels` / `kernels_us`).' % (PEAK_F, PEAK_B))
print('MfmaUtil: SQ_VALU_MFMA_BUSY_CYCLES (%s_mfma_busy.txt).  HBM bytes: FETCH_SIZE x2 + WRITE_SIZE (%s_pmc.json).\n' % (R, R))
print('| kernel | launches/step | us/step | algorithmic work/step | achieved | bound | frac of peak | MfmaUtil | HBM MB/launch (PMC) |')
print('|---|---|---|---|---|---|---|---|---|')
keys = {'k_stem_wino | k_stem_conv (7x7/2)': ('k_stem_wino', 'k_stem_conv'), 'k_stem_conv (7x7/2)': 'k_stem_conv', 'k_conv_igemm (1x1)': 'k_conv_igemm', 'k_conv_igemm + k_conv_pair (1x1)': ('k_conv_igemm', 'k_conv_pair'), 'k_conv_gemm + k_conv_pair (1x1)': ('k_conv_gemm', 'k_conv_igemm', 'k_conv_pair'), 'k_conv_winograd (3x3)': 'k_conv_winograd', 'k_conv_patch (3x3)': 'k_conv_patch'}
tot_us = 0.0
for name, d in b['conv_kernels'].items():
    t = traffic(keys[name])
    tot_us += d['us_per_step']
    print('| `%s` | %d | %.1f | %.1f GFLOP | %.1f TFLOP/s | mfma | %.2f | %s | %s |' %
          (name, d['launches_per_step'], d['us_per_step'], d['gflop_per_step'], d['tflops'], d['tflops'] / PEAK_F, mb(keys[name]), '%.1f' % (t / 1e6) if t else '-'))
B, N = b['config']['global_batch'], b['config']['samples_per_image']
k = b['kernels_us']
lbs = b['roofline_lbs']
print('| `%s` | 1 | %.1f | %.2f GFLOP | %.1f TFLOP/s | mfma | %.2f | %s | %.1f |' %
      (lbs['kernel'].split(' (')[0] + ' (M=%d)' % (B * N + B), lbs['avg_launch_us'], lbs['algorithmic_flop_per_launch'] / 1e9, lbs['achieved'], lbs['frac'], mb('k_lbs_mfma'), (lbs['traffic'] or 0) / 1e6))
for nm, note in (('k_flow_chain', 'latency (8 dependent rounds of small GEMMs)'), ('k_flow_ctx0', 'latency / L2 weight stream'), ('k_isgc', 'latency'),
                 ('k_rigid_chain', 'latency'), ('k_joints', 'latency / gather'), ('k_tpose', 'HBM write')):
    v = [x for kk, x in k.items() if kk.startswith(nm)]
    if v:
        print('| `%s` | %d | %.1f | - | - | %s | - | %s | - |' % (nm, sum(x['launches_per_step'] for x in v), sum(x['us'] * x['launches_per_step'] for x in v), note, mb(nm)))
print('\nWhole step (%d streams, %s): %.3f ms, %.0f images*samples/s; all convolution + blend-GEMM FLOPs / step time = %.1f TFLOP/s (%.2f of the fp32 MFMA peak).' %
      (b['config']['streams_per_gpu'], b['config']['issue'], b['ms_per_step'], b['value'], b['roofline']['step_aggregate']['achieved'], b['roofline']['step_aggregate']['frac']))
print('Dominant family: %s -- %.1f TFLOP/s of kernel execution time (frac %.2f).' %
      (b['roofline']['kernel'].split(' convolutions')[0], b['roofline']['achieved'], b['roofline']['frac']))
cb = b.get('cpu_baseline')
if cb:
    print('CPU baseline (%s, %d threads): %.0f images*samples/s -> %.0fx.' % (cb['kind'], cb['cores'], cb['value'], b['value'] / cb['value']))
