#!/bin/bash
# Run ON THE GPU BOX: one-stream kernel table + headline bench line into gpurun_out/ (a short version of tools/profile_round.sh).
# Usage: tools/quick_profile.sh <tag>
set -u
T=${1:-q}
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf /tmp/prof_k1
rocprofv3 --kernel-trace --stats -d /tmp/prof_k1 -o k1 -- python bench.py --steps 30 --warmup 6 --no-cpu-baseline --profile-steps 0 --streams 1 --no-graph > /dev/null 2>/tmp/k1.log
python tools/rocpd_stats.py "$(find /tmp/prof_k1 -name '*.db' | head -1)" 48 --steps 30 > gpurun_out/${T}_kernel_stats_1stream.md
python bench.py --no-cpu-baseline > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.log
tail -12 gpurun_out/${T}_kernel_stats_1stream.md
cut -c1-260 gpurun_out/${T}_bench.json
