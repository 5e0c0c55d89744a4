"""Average every collected counter per (kernel, grid) from a rocprofv3 --pmc rocpd database.   python tools/pmc_generic.py <db> [filter]"""
import sqlite3
import sys
con = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else 'hmf::'
cols = [r[1] for r in con.execute("pragma table_info(counters_collection)").fetchall()]
cname = 'counter_name' if 'counter_name' in cols else 'name'
rows = con.execute("select kernel_name, grid_size, %s, avg(value) from counters_collection where kernel_name like ? group by kernel_name, grid_size, %s" % (cname, cname),
                   ('%' + flt + '%',)).fetchall()
out = {}
for k, g, c, v in rows:
    out.setdefault((k.split('(')[0].replace('void ', ''), g), {})[c] = v
for (k, g), d in sorted(out.items()):
    print('%-44s grid %8d ' % (k[:44], g) + ' '.join('%s=%.4g' % kv for kv in sorted(d.items())))
