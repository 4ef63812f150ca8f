#!/bin/bash
# HBM traffic of the native evaluation frame (288x512, strides [4, 8]): FETCH_SIZE / WRITE_SIZE passes over tools/perf/perf_native_frame.py
#   tools/pmc_native_frame.sh <world> [extra perf_native_frame args]   -> gpurun_out/r4/pmc_native_<world>.txt
set -u
ROOT=$(pwd)
export TMPDIR=/tmp
W=${1:-tennis}; shift
OUT=/tmp/pmc_native_$W
rm -rf $OUT; mkdir -p $OUT $ROOT/gpurun_out/r4
cd /tmp
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  name=${pass%%:*}; counters=${pass#*:}
  timeout 600 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d "$OUT/$name" -- python $ROOT/tools/perf/perf_native_frame.py $W fp32 "$@" > "$OUT/$name.log" 2>&1
  echo "pass $name rc=$?"
done
cd $ROOT
python - "$OUT" <<'PY' | tee $ROOT/gpurun_out/r4/pmc_native_$W.txt
import csv, glob, os, sys
from collections import defaultdict
per = defaultdict(lambda: defaultdict(float)); n = defaultdict(set)
for path in glob.glob(os.path.join(sys.argv[1], "*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "")
        per[name][row["Counter_Name"]] += float(row["Counter_Value"]); n[(name, row["Counter_Name"])].add(row["Dispatch_Id"])
for name, c in sorted(per.items(), key=lambda kv: -sum(kv[1].values())):
    if not name.startswith("pr::"): continue
    calls = max(len(n[(name, k)]) for k in c)
    print(f"{name:40s} calls {calls:5d}  fetch x2 {2 * c.get('FETCH_SIZE', 0) * 1024 / calls / 1e6:9.2f} MB/launch  write {c.get('WRITE_SIZE', 0) * 1024 / calls / 1e6:9.2f} MB/launch")
PY
