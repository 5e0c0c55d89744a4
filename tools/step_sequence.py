"""Ordered kernel list of the LAST step in a rocprofv3 (rocpd sqlite) kernel trace: every dispatch between the last two launches of a marker kernel.
    python tools/step_sequence.py <results.db> [marker substring, default k_stem_conv] [min us to list individually, default 0]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else 'k_stem_conv'
floor = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
marks = [r[0] for r in con.execute("select start from kernels where name like ? order by start", ('%' + marker + '%',))]
t0, t1 = marks[-2], marks[-1]
rows = con.execute("select name, start, end, grid_x, workgroup_x from kernels where start >= ? and start < ? order by start", (t0, t1)).fetchall()
print('%d dispatches in the last step, %.3f ms from first start to the next step\'s first start; kernel time %.3f ms'
      % (len(rows), (t1 - t0) / 1e6, sum(r[2] - r[1] for r in rows) / 1e6))
prev_end, small, small_t = t0, 0, 0.0
for name, s, e, g, w in rows:
    us = (e - s) / 1e3
    short = name.replace('void ', '').replace('at::native::', '').replace('(anonymous namespace)::', '')
    short = short[:110]
    if us < floor:
        small += 1
        small_t += us
        prev_end = max(prev_end, e)
        continue
    if small:
        print('      ... %d launches under %.0f us: %.1f us' % (small, floor, small_t))
        small, small_t = 0, 0.0
    print('%9.1f us  gap %6.1f  grid %9d x %3d  %s' % (us, (s - prev_end) / 1e3, g, w, short))
    prev_end = max(prev_end, e)
if small:
    print('      ... %d launches under %.0f us: %.1f us' % (small, floor, small_t))
