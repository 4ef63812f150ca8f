#!/bin/bash
# The round's final evidence set in ONE gpurun session (run from the repo root on the GPU box):
#   tools/refresh_profiles.sh r03      ->  gpurun_out/profiles/r03_*   (copy the files to profiles/ afterwards)
# PMC passes first (own rocprofv3 runs, --kernel-trace only), then the bench line, the headline's kernel stats, the training trace.
set -u
R=${1:-rNN}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles
mkdir -p "$OUT"
export TMPDIR=/tmp
bash tools/collect_pmc.sh > "$OUT/${R}_pmc.log" 2>&1
cp gpurun_out/pmc_summary.json "$OUT/${R}_pmc_summary.json"
mkdir -p profiles && cp "$OUT/${R}_pmc_summary.json" profiles/${R}_pmc_summary.json      # the bench line below reads it
bash tools/collect_pmc_train.sh > "$OUT/${R}_pmc_train.log" 2>&1
cp gpurun_out/pmc_train_summary.json "$OUT/${R}_pmc_train_summary.json"
cp "$OUT/${R}_pmc_train_summary.json" profiles/${R}_pmc_train_summary.json
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/${R}_bench.json" 2> "$OUT/${R}_bench.err"
cd /tmp
rm -rf /tmp/hl /tmp/tr
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hl -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline \
    --no-train-step --no-split-precision --no-distinct-frames --no-reference-graph --no-minecraft --no-shard-balance \
    > "$OUT/${R}_headline_bench.json" 2> "$OUT/${R}_headline.err"
cp /tmp/hl/*/*_kernel_stats.csv "$OUT/${R}_headline_kernel_stats.csv"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -- python $ROOT/tools/perf/perf_train_leg.py 6 3 \
    > "$OUT/${R}_train_leg.json" 2> "$OUT/${R}_train_leg.err"
cp /tmp/tr/*/*_kernel_stats.csv "$OUT/${R}_train_step_kernel_stats.csv"
cd "$ROOT"
python tools/summarise_train_trace.py /tmp/tr/*/*_kernel_trace.csv > "$OUT/${R}_train_step_trace_summary.json"
python tools/trace_timeline.py /tmp/tr/*/*_kernel_trace.csv --all > "$OUT/${R}_train_step_timeline.txt"
ls -la "$OUT"
