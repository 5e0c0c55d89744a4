#!/bin/bash
# Run ON THE GPU BOX: per-kernel issue-pipe counters of the one-stream step (two separate --pmc passes, kernel trace only):
# how much of a kernel's time its SIMDs spend issuing MFMAs, issuing other VALU work, and waiting.   tools/pipe_profile.sh <tag>
set -u
T=${1:-r04}
export TMPDIR=/tmp
mkdir -p gpurun_out/profiles
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --profile-steps 0 --streams 1 --no-graph"
rm -rf /tmp/pp_a /tmp/pp_b
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES -d /tmp/pp_a -o a -- $B >/dev/null 2>/tmp/pp_a.log
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d /tmp/pp_b -o b -- $B >/dev/null 2>/tmp/pp_b.log
python tools/pipe_summary.py "$(find /tmp/pp_a -name '*.db' | head -1)" "$(find /tmp/pp_b -name '*.db' | head -1)" > gpurun_out/profiles/${T}_pipe_counters.md
cat gpurun_out/profiles/${T}_pipe_counters.md
tail -n 2 /tmp/pp_a.log; tail -n 2 /tmp/pp_b.log
