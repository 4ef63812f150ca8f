#!/bin/bash
# HBM traffic and kernel times of the SHIPPED renderers' frames (tennis 4+4+32+32, minecraft 16+1+32+32), both precisions:
#   tools/collect_pmc_shipped.sh  ->  gpurun_out/pmc_shipped/<world>/{stats,fetch,write}  +  gpurun_out/pmc_shipped_summary.json
# (28 renders per precision and run: 3 warm-up + 20 timed + 5 with the HIP-event profile)
set -u
ROOT=$(pwd)
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmc_shipped
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
for world in tennis minecraft; do
  CMD="python $ROOT/tools/perf/perf_minecraft_eval.py $world"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$world/stats" -- $CMD > "$OUT/$world.stats.log" 2>&1
  for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
    name=${pass%%:*}; counters=${pass#*:}
    timeout 600 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d "$OUT/$world/$name" -- $CMD > "$OUT/$world.$name.log" 2>&1
    echo "$world pass $name rc=$?"
  done
done
cd "$ROOT"
python tools/summarise_pmc_shipped.py "$OUT" 28 > gpurun_out/pmc_shipped_summary.json
cat gpurun_out/pmc_shipped_summary.json
