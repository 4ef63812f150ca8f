#!/bin/bash
# HBM traffic of the TRAINING step's kernels (run through gpurun from the repo root):
#   tools/collect_pmc_train.sh  ->  gpurun_out/pmc_train/{fetch,write}/...csv  +  gpurun_out/pmc_train_summary.json
# Same rules as collect_pmc.sh: one counter set per rocprofv3 pass, --kernel-trace only.
set -u
ROOT=$(pwd)
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmc_train
rm -rf "$OUT"; mkdir -p "$OUT"
STEPS=4; WARM=2
CMD="python $ROOT/tools/perf/perf_train_leg.py $STEPS $WARM"
cd /tmp
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  name=${pass%%:*}; counters=${pass#*:}
  timeout 900 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d "$OUT/$name" -- $CMD > "$OUT/$name.log" 2>&1
  echo "pass $name rc=$?"
done
cd "$ROOT"
python tools/summarise_pmc_train.py "$OUT" $((2 * STEPS + WARM)) > gpurun_out/pmc_train_summary.json
tail -c 3000 gpurun_out/pmc_train_summary.json
