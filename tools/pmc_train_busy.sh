#!/bin/bash
# matrix-pipe busy fraction of the training leg's kernels: tools/pmc_train_busy.sh [fp32|f16x3]
set -u
ROOT=$(pwd); export TMPDIR=/tmp
P=${1:-fp32}; OUT=/tmp/pmc_busy_$P; rm -rf $OUT; mkdir -p $OUT $ROOT/gpurun_out/r4
cd /tmp
PR_PERF_PRECISION=$P timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/busy -- python $ROOT/tools/perf/perf_train_leg.py 6 3 > $OUT/busy.log 2>&1
cd $ROOT
python - "$OUT" <<'PY' | tee $ROOT/gpurun_out/r4/pmc_busy_$P.txt
import csv, glob, os, sys
from collections import defaultdict
per = defaultdict(lambda: defaultdict(float)); n = defaultdict(set)
for path in glob.glob(os.path.join(sys.argv[1], "*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "")
        per[name][row["Counter_Name"]] += float(row["Counter_Value"]); n[name].add(row["Dispatch_Id"])
for name, c in sorted(per.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    if not name.startswith("pr::") or not c.get("GRBM_GUI_ACTIVE"): continue
    busy = (c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024.0) / (c["GRBM_GUI_ACTIVE"] / 8.0)
    print(f"{name:36s} launches {len(n[name]):4d}  gui_active/launch {c['GRBM_GUI_ACTIVE'] / 8 / len(n[name]) / 1e3:9.1f} kcycles  mfma busy {busy:6.3f}")
PY
