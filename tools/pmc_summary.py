"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate passes as the MI355X guide prescribes:
TCC slots do not fit both).  Units: the counters are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane)
coalesced reads -> doubled (guide, HBM section); WRITE_SIZE calibrated on k_tpose (writes exactly M*82680 B): ratio printed.

    python tools/pmc_summary.py gpurun_out/pmc_fetch/f_results.db gpurun_out/pmc_write/w_results.db > profiles/r02_pmc.json

The output is stamped with the library source hash (humaniflow_amd.build.source_hash) it was collected at: bench.py only quotes
`traffic` from a file whose stamp matches the library it is benchmarking.
"""
import json
import os
import sqlite3
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from humaniflow_amd.build import source_hash


def load(db):
    con = sqlite3.connect(db)
    rows = con.execute("select kernel_name, grid_size, avg(value), count(*) from counters_collection where kernel_name like '%hmf::%' "
                       "group by kernel_name, grid_size").fetchall()
    return {(r[0].split('(')[0].replace('void ', ''), r[1]): (r[2], r[3]) for r in rows}


# FETCH_SIZE correction: x2 for kernels whose streaming reads are 16 B/lane (the guide's calibrated case).  k_stem_conv fetches its input patch with
# 4 B/lane loads: its raw FETCH_SIZE (218 MB at B=32) already matches the expected patch bytes (151 MB input x 1.52 halo = 230 MB), so factor 1.
# (Round 2's table keyed this dict by the full name; the kernel is a template -- `hmf::k_stem_conv<true>` -- so the entry never matched and the stem was
# doubled like everything else: 470 MB reported, 252 MB = 218 + 33.5 MB of pooled output is the corrected figure.  Keys are name PREFIXES now.)
# Every other kernel of the step reads with 16 B/lane global / buffer loads or LDS-DMA (`buffer_load_dwordx4 ... lds`): factor 2.
FETCH_FACTOR = {'hmf::k_stem_conv': 1.0}


def fetch_factor(name):
    for k, v in FETCH_FACTOR.items():
        if name.startswith(k):
            return v
    return 2.0
f, w = load(sys.argv[1]), load(sys.argv[2])
out = {}
for key in sorted(set(f) | set(w)):
    name, grid = key
    fk = f.get(key, (0, 0))[0]
    wk = w.get(key, (0, 0))[0]
    ff = fetch_factor(name)
    out['%s[grid=%d]' % (name, grid)] = {'FETCH_SIZE_KiB': round(fk, 1), 'WRITE_SIZE_KiB': round(wk, 1), 'fetch_correction': ff,
                                         'hbm_read_bytes_corrected': int(ff * fk * 1024), 'hbm_write_bytes': int(wk * 1024),
                                         'hbm_bytes_per_launch': int((ff * fk + wk) * 1024), 'launches': f.get(key, (0, 0))[1]}
try:
    head = subprocess.run(['git', 'rev-parse', 'HEAD'], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, cwd=os.path.dirname(os.path.abspath(__file__))).stdout.decode().strip() or None
except Exception:
    head = None
json.dump({'lib_source_hash': source_hash(), 'git_head': head, 'workload': os.environ.get('HMF_PMC_WORKLOAD', 'B=32,N=100,fp32'),
           'how': 'two separate rocprofv3 --kernel-trace --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --no-cpu-baseline`; KiB units; FETCH x2 on gfx950 for 16 B/lane reads '
                  '(every kernel but k_stem_conv, whose 4 B/lane NCHW row loads are counted in full: per-kernel factor in `fetch_correction`)',
           'kernels': out}, sys.stdout, indent=1)
