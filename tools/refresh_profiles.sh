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
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 --full-json "$OUT/${R}_bench_full.json" > "$OUT/${R}_bench.json" 2> "$OUT/${R}_bench.err"
cd /tmp
rm -rf /tmp/hl /tmp/tr
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hl -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline \
    --no-train-step --no-split-precision --no-distinct-frames --no-reference-graph --no-minecraft --no-shard-balance --no-native-frame --full-json /tmp/headline_full.json \
    > "$OUT/${R}_headline_bench.json" 2> "$OUT/${R}_headline.err"
cp /tmp/hl/*/*_kernel_stats.csv "$OUT/${R}_headline_kernel_stats.csv"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -- python $ROOT/tools/perf/perf_train_leg.py 6 3 \
    > "$OUT/${R}_train_leg.json" 2> "$OUT/${R}_train_leg.err"
cp /tmp/tr/*/*_kernel_stats.csv "$OUT/${R}_train_step_kernel_stats.csv"
cd "$ROOT"
python tools/summarise_train_trace.py /tmp/tr/*/*_kernel_trace.csv > "$OUT/${R}_train_step_trace_summary.json"
python tools/trace_timeline.py /tmp/tr/*/*_kernel_trace.csv --all > "$OUT/${R}_train_step_timeline.txt"
# the split-precision (fp16-pair) training step: its own trace
cd /tmp; rm -rf /tmp/tr3
PR_PERF_PRECISION=f16x3 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr3 -- python $ROOT/tools/perf/perf_train_leg.py 6 3 \
    > "$OUT/${R}_train_f16x3_leg.json" 2> "$OUT/${R}_train_f16x3_leg.err"
cp /tmp/tr3/*/*_kernel_stats.csv "$OUT/${R}_train_f16x3_kernel_stats.csv"
cd "$ROOT"
python tools/summarise_train_trace.py /tmp/tr3/*/*_kernel_trace.csv > "$OUT/${R}_train_f16x3_trace_summary.json"
# matrix-pipe busy fractions of the training kernels, HBM traffic + kernel stats of the native evaluation frame
mkdir -p gpurun_out/r4
bash tools/pmc_train_busy.sh fp32 > /dev/null 2>&1;  cp gpurun_out/r4/pmc_busy_fp32.txt "$OUT/${R}_pmc_train_busy_fp32.txt"
bash tools/pmc_train_busy.sh f16x3 > /dev/null 2>&1; cp gpurun_out/r4/pmc_busy_f16x3.txt "$OUT/${R}_pmc_train_busy_f16x3.txt"
for W in tennis minecraft; do
  bash tools/pmc_native_frame.sh $W > /dev/null 2>&1; cp gpurun_out/r4/pmc_native_$W.txt "$OUT/${R}_pmc_native_frame_$W.txt"
  cd /tmp; rm -rf /tmp/nf_$W
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/nf_$W -- python $ROOT/tools/perf/perf_native_frame.py $W fp32 \
      > "$OUT/${R}_native_frame_$W.log" 2>&1
  cp /tmp/nf_$W/*/*_kernel_stats.csv "$OUT/${R}_native_frame_${W}_kernel_stats.csv"
  cd "$ROOT"
done
ls -la "$OUT"
