#!/bin/bash
# round 4: fp16 tier measurements, then the headline PMC passes + the full bench line + the headline's kernel stats
R=r04
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles
mkdir -p "$OUT" gpurun_out/r4
export TMPDIR=/tmp
bash tools/run_r4i.sh > gpurun_out/r4/r4i.log 2>&1
bash tools/collect_pmc.sh > "$OUT/${R}_pmc.log" 2>&1
cp gpurun_out/pmc_summary.json "$OUT/${R}_pmc_summary.json"
grep -q k_mlp_mfma gpurun_out/pmc_summary.json && cp "$OUT/${R}_pmc_summary.json" profiles/${R}_pmc_summary.json
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/${R}_bench.json" 2> "$OUT/${R}_bench.err"
cd /tmp
rm -rf /tmp/hl
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hl -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline \
    --no-train-step --no-split-precision --no-distinct-frames --no-reference-graph --no-minecraft --no-shard-balance --no-native-frame \
    > "$OUT/${R}_headline_bench.json" 2> "$OUT/${R}_headline.err"
cp /tmp/hl/*/*_kernel_stats.csv "$OUT/${R}_headline_kernel_stats.csv"
cd $ROOT
cat gpurun_out/r4/r4i.log
tail -3 "$OUT/${R}_pmc.log"
head -c 600 "$OUT/${R}_bench.json"
