"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into the per-kernel stats table committed under profiles/.

    python tools/rocpd_stats.py gpurun_out/prof_r01/r01_results.db > profiles/r01_kernel_stats.md
"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name || case when name like '%hmf::%' then ' [grid ' || grid_x || ']' else '' end as nm, count(*), sum(duration), "
                   "avg(duration), min(duration), max(duration), max(vgpr_count), max(accum_vgpr_count), max(lds_size), max(scratch_size), "
                   "max(grid_x), max(workgroup_x) from kernels group by nm order by sum(duration) desc").fetchall()
total = sum(r[2] for r in rows)
print('| kernel | calls | total ms | avg us | min us | max us | % | vgpr | agpr | lds B | scratch B | grid | wg |')
print('|---|---|---|---|---|---|---|---|---|---|---|---|---|')
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    name = r[0] if len(r[0]) < 110 else r[0][:70] + '...' + r[0][-36:]
    print('| `%s` | %d | %.3f | %.2f | %.2f | %.2f | %.1f | %d | %d | %d | %d | %d | %d |'
          % (name, r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / total, r[6], r[7], r[8], r[9], r[10], r[11]))
print('\ntotal kernel time: %.3f ms over %d dispatches' % (total / 1e6, sum(r[1] for r in rows)))
