#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: regenerates every file under profiles/ that bench.py's numbers are
# checked against.  Usage: tools/profile_round.sh r02     -> writes gpurun_out/profiles/r02_*; copy those into profiles/.
# Counter passes are separate runs with --kernel-trace only (no sys/hip/hsa traces), one counter group per pass.
set -u
R=${1:-r02}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/profiles
mkdir -p "$OUT"
STEPS=30; WARM=6
# the profiled command = the default bench without its trailing measurement pass / CPU baseline, so that the timed region ends the process
BENCH="python bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --profile-steps 0"
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- $BENCH > "$OUT/${R}_bench_under_rocprof.json" 2>/tmp/kt.log
DB=$(find /tmp/prof_kt -name '*.db' | head -1)
python tools/rocpd_stats.py "$DB" 48 --steps $STEPS > "$OUT/${R}_kernel_stats.md"
# the same steps on ONE stream, eagerly: kernel execution times without co-running kernels = what bench.py's measurement pass (roofline.frac) reports
rocprofv3 --kernel-trace --stats -d /tmp/prof_k1 -o k1 -- $BENCH --streams 1 --no-graph > /dev/null 2>/tmp/k1.log
python tools/rocpd_stats.py "$(find /tmp/prof_k1 -name '*.db' | head -1)" 48 --steps $STEPS > "$OUT/${R}_kernel_stats_1stream.md"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_f -o f -- $BENCH >/dev/null 2>/tmp/f.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_w -o w -- $BENCH >/dev/null 2>/tmp/w.log
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/prof_m -o m -- $BENCH --streams 1 --no-graph >/dev/null 2>/tmp/m.log
python tools/pmc_mfma.py "$(find /tmp/prof_m -name '*.db' | head -1)" > "$OUT/${R}_mfma_busy.txt"
python tools/pmc_summary.py "$(find /tmp/prof_f -name '*.db' | head -1)" "$(find /tmp/prof_w -name '*.db' | head -1)" > "$OUT/${R}_pmc.json"
cp "$OUT"/${R}_* profiles/
# the headline line LAST: it quotes `traffic` from the PMC summary just written (same library source hash)
python bench.py > "$OUT/${R}_bench.json" 2> "$OUT/${R}_bench.log"
cp "$OUT"/${R}_* profiles/ && python tools/roofline_table.py $R > "$OUT/${R}_roofline.md"
ls -la "$OUT"
