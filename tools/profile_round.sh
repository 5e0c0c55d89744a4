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
BENCH="python bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-secondary --profile-steps 0"
# counter passes serialise every dispatch: a few steps carry every kernel of the step often enough
PMCBENCH="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary --profile-steps 0"
stamp() { echo "[$(date +%H:%M:%S)] $*"; }
PART=${HMF_PROFILE_PART:-all}          # all | main | flavours (two gpurun calls when one would exceed its time limit)
if [ "$PART" != "flavours" ]; then
stamp "kernel trace, timed region"
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- $BENCH > "$OUT/${R}_bench_under_rocprof.json" 2>/tmp/kt.log
DB=$(find /tmp/prof_kt -name '*.db' | head -1)
python tools/rocpd_stats.py "$DB" 48 --steps $STEPS > "$OUT/${R}_kernel_stats.md"
# the same steps on ONE stream, eagerly: kernel execution times without co-running kernels = what bench.py's measurement pass (roofline.frac) reports
rocprofv3 --kernel-trace --stats -d /tmp/prof_k1 -o k1 -- $BENCH --streams 1 --no-graph > /dev/null 2>/tmp/k1.log
python tools/rocpd_stats.py "$(find /tmp/prof_k1 -name '*.db' | head -1)" 48 --steps $STEPS > "$OUT/${R}_kernel_stats_1stream.md"
stamp "PMC FETCH_SIZE / WRITE_SIZE / MFMA busy"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_f -o f -- $PMCBENCH >/dev/null 2>/tmp/f.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_w -o w -- $PMCBENCH >/dev/null 2>/tmp/w.log
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/prof_m -o m -- $PMCBENCH --streams 1 --no-graph >/dev/null 2>/tmp/m.log
python tools/pmc_mfma.py "$(find /tmp/prof_m -name '*.db' | head -1)" > "$OUT/${R}_mfma_busy.txt"
python tools/pmc_summary.py "$(find /tmp/prof_f -name '*.db' | head -1)" "$(find /tmp/prof_w -name '*.db' | head -1)" > "$OUT/${R}_pmc.json"
cp "$OUT"/${R}_* profiles/
stamp "headline bench line (quotes the PMC summary just written)"
python bench.py > "$OUT/${R}_bench.json" 2> "$OUT/${R}_bench.log"
cp "$OUT"/${R}_* profiles/ && python tools/roofline_table.py $R > "$OUT/${R}_roofline.md"
stamp "predict-shape latency (B = 1, N = 10), flow kernel timeline"
python tools/predict_latency.py 1 10 > "$OUT/${R}_predict_latency.txt" 2>/dev/null
python tools/flow_time.py > "$OUT/${R}_flow_time.txt" 2>/dev/null
stamp "issue-pipe counters of the one-stream step (MFMA busy / VALU active / waits per kernel)"
tools/pipe_profile.sh $R > /dev/null 2>&1
fi
# ---- the other BASELINE configurations: eval step (configs[3] per GPU), bf16 blend operands at B = 64 (configs[2]), training step (configs[4]):
# two PMC passes each (FETCH_SIZE, WRITE_SIZE), then the bench line that quotes them
pmc_pair() {   # $1 = tag, $2 = workload label, rest = bench flags
    tag=$1; wl=$2; shift 2
    rm -rf /tmp/pf_$tag /tmp/pw_$tag
    stamp "PMC pair: $tag"
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf_$tag -o f -- python bench.py --no-cpu-baseline --profile-steps 0 --steps 6 --warmup 2 "$@" >/dev/null 2>/tmp/pf_$tag.log
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw_$tag -o w -- python bench.py --no-cpu-baseline --profile-steps 0 --steps 6 --warmup 2 "$@" >/dev/null 2>/tmp/pw_$tag.log
    HMF_PMC_WORKLOAD="$wl" python tools/pmc_summary.py "$(find /tmp/pf_$tag -name '*.db' | head -1)" "$(find /tmp/pw_$tag -name '*.db' | head -1)" > "$OUT/${R}_${tag}_pmc.json"
    cp "$OUT/${R}_${tag}_pmc.json" profiles/
}
if [ "${HMF_PROFILE_FLAVOURS:-1}" = "1" ] && [ "$PART" != "main" ]; then
    pmc_pair eval "B=32,N=100,fp32,eval" --eval
    python bench.py --eval --no-cpu-baseline > "$OUT/${R}_bench_eval_configs3.json" 2> "$OUT/${R}_bench_eval.log"
    # the eval step on ONE stream, eagerly: which kernels the evaluation epilogue adds to the predict step (no at::native row should be left inside a step)
    rocprofv3 --kernel-trace --stats -d /tmp/prof_e1 -o e1 -- python bench.py --eval --steps 20 --warmup 4 --no-cpu-baseline --profile-steps 0 --streams 1 --no-graph > /dev/null 2>/tmp/e1.log
    python tools/rocpd_stats.py "$(find /tmp/prof_e1 -name '*.db' | head -1)" 70 --steps 20 > "$OUT/${R}_eval_kernel_stats_1stream.md"
    pmc_pair bf16 "B=64,N=100,bf16" --batch 64 --blend-dtype bf16
    python bench.py --batch 64 --blend-dtype bf16 --no-cpu-baseline > "$OUT/${R}_bench_configs2_bf16_B64.json" 2> "$OUT/${R}_bench_bf16.log"
    pmc_pair train "train,B=72,N=8" --train
    python bench.py --train --no-cpu-baseline > "$OUT/${R}_bench_train_configs4.json" 2> "$OUT/${R}_bench_train.log"
    rocprofv3 --kernel-trace --stats -d /tmp/prof_tr -o tr -- python bench.py --train --no-cpu-baseline --steps 12 --warmup 4 --profile-steps 0 > /dev/null 2>/tmp/tr.log
    python tools/rocpd_stats.py "$(find /tmp/prof_tr -name '*.db' | head -1)" 60 > "$OUT/${R}_train_kernel_stats.md"
    python tools/step_sequence.py "$(find /tmp/prof_tr -name '*.db' | head -1)" k_stem_wino_pack 8 > "$OUT/${R}_train_step_sequence.txt"
    # opt-in flavours of the headline workload (never the headline line): 1x1 GEMMs on the bf16 pipe with split fp32 operands
    python bench.py --gemm-terms 6 --no-cpu-baseline > "$OUT/${R}_bench_optin_gemm_terms6.json" 2> "$OUT/${R}_bench_terms6.log"
    python bench.py --gemm-terms 3 --no-cpu-baseline > "$OUT/${R}_bench_optin_gemm_terms3.json" 2> "$OUT/${R}_bench_terms3.log"
fi
stamp done
ls -la "$OUT"
