"""Per-kernel issue-pipe table from two rocprofv3 --pmc passes of the one-stream step (tools/pipe_profile.sh).
Units (MI355X guide): GRBM_GUI_ACTIVE = shader cycles summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES = cycles summed over the 1,024 SIMDs;
SQ_ACTIVE_INST_* / SQ_WAIT_* / SQ_WAVE_CYCLES = quad-cycles (x4) summed over waves.  Derived per launch:
  mfma %  = MFMA busy cycles / (kernel cycles x 1024 SIMDs)         valu % = VALU-active quad-cycles x 4 / (kernel cycles x 1024)
  (on gfx950 the fp32 MFMA occupies the vector datapath: the two do not overlap on a SIMD, DESIGN.md 3 fact 2, so mfma % + valu % <= 100)
  VALU per MFMA = vector instructions other than MFMAs issued per MFMA."""
import sqlite3
import sys


def load(db):
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(counters_collection)").fetchall()]
    cname = 'counter_name' if 'counter_name' in cols else 'name'
    rows = con.execute("select kernel_name, grid_size, %s, avg(value), count(*) from counters_collection where kernel_name like '%%hmf::%%' "
                       "group by kernel_name, grid_size, %s" % (cname, cname)).fetchall()
    out = {}
    for k, g, c, v, n in rows:
        d = out.setdefault((k.split('(')[0].replace('void ', ''), g), {'launches': n})
        d[c] = v
    return out


a, b = load(sys.argv[1]), load(sys.argv[2])
print('| kernel [grid] | launches | kernel kcyc | mfma % | valu % | wait-any % of wave cycles | LDS-active % | VMEM-active % | VALU instr per MFMA | MFMAs per SIMD |')
print('|---|---|---|---|---|---|---|---|---|---|')
tot = {'cyc': 0.0, 'mfma': 0.0, 'valu': 0.0}
for key in sorted(a, key=lambda k: -a[k].get('GRBM_GUI_ACTIVE', 0) * a[k]['launches']):
    d, e = a[key], b.get(key, {})
    cyc = d.get('GRBM_GUI_ACTIVE', 0) / 8.0
    if cyc <= 0:
        continue
    simd = cyc * 1024.0
    mf = d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / simd * 100
    va = d.get('SQ_ACTIVE_INST_VALU', 0) * 4 / simd * 100
    wc = d.get('SQ_WAVE_CYCLES', 0)
    cycb = e.get('GRBM_GUI_ACTIVE', 0) / 8.0 * 1024.0
    wait = e.get('SQ_WAIT_INST_ANY', 0) / wc * 100 if wc else float('nan')
    lds = e.get('SQ_ACTIVE_INST_LDS', 0) * 4 / cycb * 100 if cycb else float('nan')
    vm = e.get('SQ_ACTIVE_INST_VMEM', 0) * 4 / cycb * 100 if cycb else float('nan')
    nm, nv = d.get('SQ_INSTS_MFMA', 0), d.get('SQ_INSTS_VALU', 0)
    n = d['launches']
    tot['cyc'] += cyc * n; tot['mfma'] += d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) * n; tot['valu'] += d.get('SQ_ACTIVE_INST_VALU', 0) * 4 * n
    print('| `%s` [%d] | %d | %.0f | %.1f | %.1f | %.1f | %.1f | %.1f | %s | %.0f |' % (key[0][:64], key[1], n, cyc / 1e3, mf, va, wait, lds, vm,
                                                                            ('%.2f' % ((nv - nm) / nm)) if nm else '-', nm / 1024.0))
if tot['cyc']:
    print('\nall launches: MFMA busy %.1f %%, VALU active (incl. MFMA issue) %.1f %% of SIMD cycles' % (tot['mfma'] / (tot['cyc'] * 1024) * 100, tot['valu'] / (tot['cyc'] * 1024) * 100))
