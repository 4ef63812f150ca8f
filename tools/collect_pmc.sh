#!/bin/bash
# Collects the PMC evidence behind bench.py's roofline object on the GPU box (run through gpurun from the repo root):
#   tools/collect_pmc.sh            -> gpurun_out/pmc/{fetch,write,busy}/...csv  +  gpurun_out/pmc_summary.json
# Each counter set is its own rocprofv3 pass with --kernel-trace only (FETCH_SIZE and WRITE_SIZE do not fit one pass;
# PMC passes must not be combined with the API / memory-copy trace domains on this pool).
set -u
ROOT=$(pwd)
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmc
rm -rf "$OUT"; mkdir -p "$OUT"
CMD="python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-train-step --no-split-precision --no-distinct-frames --no-minecraft --no-reference-graph --no-shard-balance --no-native-frame"
# (--no-native-frame: that leg replays captured HIP graphs, whose packets the counter-collection service cannot instrument - 
#  "HSA_STATUS_ERROR_INVALID_PACKET_FORMAT" and a hung pass, measured in round 4)
cd /tmp
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "busy:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES"; do
  name=${pass%%:*}; counters=${pass#*:}
  timeout 900 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d "$OUT/$name" -- $CMD > "$OUT/$name.log" 2>&1
  echo "pass $name rc=$?"
done
cd "$ROOT"
python tools/summarise_pmc.py "$OUT" > gpurun_out/pmc_summary.json
tail -c 2500 gpurun_out/pmc_summary.json
