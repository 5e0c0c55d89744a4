"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into the per-kernel stats table committed under profiles/.

    python tools/rocpd_stats.py <results.db> [rows] [--steps K]

--steps K restricts the table to the bench's TIMED region: one hmf::k_isgc is launched per step, so the window opens at the end of
the k_isgc that precedes the last K of them (= the last warm-up step's; with 2 streams the first timed step's encoder may already
be running, i.e. the window holds K steps' worth of every kernel, shifted by at most one encoder) and closes at the end of the last hmf:: kernel
(what follows is the bench's checksum of the outputs).
Run the bench with --profile-steps 0 --no-cpu-baseline so that the timed region is the end of the process.
"""
import sqlite3
import sys

args = sys.argv[1:]
steps = None
if '--steps' in args:
    i = args.index('--steps')
    steps = int(args[i + 1])
    del args[i:i + 2]
con = sqlite3.connect(args[0])
where, note = '', 'whole process'
if steps:
    marks = con.execute("select end from kernels where name like '%hmf::k_isgc%' order by start").fetchall()
    if len(marks) > steps:
        t0 = marks[len(marks) - steps - 1][0]
        t1 = con.execute("select max(end) from kernels where name like '%hmf::%'").fetchone()[0]      # the last step's last kernel
        where = ' where start >= %d and start <= %d' % (t0, t1)
        after = con.execute("select count(*), coalesce(sum(duration), 0) from kernels where start > %d" % t1).fetchone()
        note = ('timed region: the last %d steps of the process (run bench.py with --profile-steps 0 so that nothing follows the timed region; window opens at the '
                'end of the last warm-up step\'s k_isgc and closes with the last step\'s last kernel; the %d dispatches after it, %.3f ms, are bench.py\'s checksum '
                'of the outputs -- abs / mean / stack on torch -- not part of a step)' % (steps, after[0], after[1] / 1e6))
rows = con.execute("select name || case when name like '%hmf::%' then ' [grid ' || grid_x || ']' else '' end as nm, count(*), sum(duration), "
                   "avg(duration), min(duration), max(duration), max(vgpr_count), max(accum_vgpr_count), max(lds_size), max(scratch_size), "
                   "max(grid_x), max(workgroup_x) from kernels" + where + " group by nm order by sum(duration) desc").fetchall()
total = sum(r[2] for r in rows)
span = con.execute("select max(end) - min(start) from kernels" + where).fetchone()[0]
print('scope: %s\n' % note)
print('| kernel | calls | total ms | avg us | min us | max us | % | vgpr | agpr | lds B | scratch B | grid | wg |')
print('|---|---|---|---|---|---|---|---|---|---|---|---|---|')
for r in rows[:int(args[1]) if len(args) > 1 else 40]:
    name = r[0] if len(r[0]) < 110 else r[0][:70] + '...' + r[0][-36:]
    print('| `%s` | %d | %.3f | %.2f | %.2f | %.2f | %.1f | %d | %d | %d | %d | %d | %d |'
          % (name, r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / total, r[6], r[7], r[8], r[9], r[10], r[11]))
print('\ntotal kernel time: %.3f ms over %d dispatches; wall span of the window %.3f ms' % (total / 1e6, sum(r[1] for r in rows), span / 1e6))
if steps:
    print('per step: %.3f ms of kernel time, %.3f ms of wall span' % (total / 1e6 / steps, span / 1e6 / steps))
    groups = [('encoder convolutions (hmf::k_stem_wino + k_stem_pool_rim | k_stem_conv, k_conv_winograd + k_wino_splitk_finish, k_conv_patch, k_conv_gemm, k_conv_igemm, k_conv_pair)', "(name like '%hmf::k_stem_%' or name like '%hmf::k_conv_%' or name like '%hmf::k_wino_%')"),
              ('encoder max-pool (hmf::k_maxpool_nhwc)', "name like '%k_maxpool_nhwc%'"),
              ('flow (hmf::k_isgc, k_flow_*)', "(name like '%hmf::k_isgc%' or name like '%hmf::k_flow%')"),
              ('SMPL (hmf::k_lbs_mfma, k_rigid_chain, k_joints, k_tpose)', "(name like '%hmf::k_lbs%' or name like '%hmf::k_rigid%' or name like '%hmf::k_joints%' or name like '%hmf::k_tpose%')")]
    print('\n| stage | ms / step | % of kernel time |\n|---|---|---|')
    acc = 0
    for label, cond in groups:
        v = con.execute("select coalesce(sum(duration), 0) from kernels" + where + " and " + cond).fetchone()[0]
        acc += v
        print('| %s | %.3f | %.1f |' % (label, v / 1e6 / steps, 100.0 * v / total))
    print('| everything else (hmf::k_heads_fc1, k_heads_out, k_zero; any torch / vendor kernel inside a step would be counted here) | %.3f | %.1f |' % ((total - acc) / 1e6 / steps, 100.0 * (total - acc) / total))
