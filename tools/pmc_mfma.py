"""Summarise a rocprofv3 --pmc pass (rocpd sqlite) per kernel: every collected counter averaged over the launches, plus, when present,
MFMA busy % = SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs) / (GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 * 1024),
the implied shader clock GRBM_GUI_ACTIVE / 8 / duration is ~2.37 GHz on these kernels (checked against the kernel trace), and the
LDS bank-conflict share SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.
    python tools/pmc_mfma.py <results.db> [substring-filter]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else 'hmf::'
cols = [r[1] for r in con.execute("pragma table_info(counters_collection)").fetchall()]
cname = 'counter_name' if 'counter_name' in cols else 'name'
rows = con.execute("select kernel_name, grid_size, %s, avg(value), count(*) from counters_collection where kernel_name like ? "
                   "group by kernel_name, grid_size, %s" % (cname, cname), ('%' + flt + '%',)).fetchall()
out = {}
for k, g, c, v, n in rows:
    out.setdefault((k.split('(')[0].replace('void ', ''), g), {})[c] = v
for (k, g), d in sorted(out.items()):
    extra = ''
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in d and 'GRBM_GUI_ACTIVE' in d and d['GRBM_GUI_ACTIVE']:
        extra += ' mfma_busy=%.1f%%' % (100.0 * d['SQ_VALU_MFMA_BUSY_CYCLES'] / (d['GRBM_GUI_ACTIVE'] / 8 * 1024))
    if 'SQ_LDS_BANK_CONFLICT' in d and d.get('SQ_LDS_IDX_ACTIVE'):
        extra += ' lds_conflict=%.1f%%' % (100.0 * d['SQ_LDS_BANK_CONFLICT'] / d['SQ_LDS_IDX_ACTIVE'])
    print('%-60s grid %8d %s%s' % (k[:60], g, ' '.join('%s=%.4g' % kv for kv in sorted(d.items())), extra))
