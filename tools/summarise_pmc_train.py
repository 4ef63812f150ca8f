"""Per-kernel HBM traffic of bench.py's training leg from the rocprofv3 counter CSVs of tools/collect_pmc_train.sh.

    python tools/summarise_pmc_train.py <dir> <training steps in the run>

FETCH_SIZE / WRITE_SIZE are reported in KB; FETCH_SIZE is doubled for the 16 B/lane streaming reads of these kernels as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (both figures are printed).  perf_train_leg.py runs the leg's timed
steps twice (the timed pass and the per-kernel HIP-event pass), plus the warm-up steps of each: the step count on the command line
is the total."""
import csv
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict

root, steps = sys.argv[1], int(sys.argv[2])
per = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for path in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"].split("(")[0]
            if name.startswith("void "):
                name = name[5:]
            if not name.startswith("pr::"):
                continue
            per[name][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[(name, row["Counter_Name"])].add(row["Dispatch_Id"])
out = {"source": "tools/collect_pmc_train.sh: rocprofv3 --kernel-trace --pmc <FETCH_SIZE | WRITE_SIZE> -- python tools/perf/perf_train_leg.py",
       "training_steps_in_run": steps, "per_step": {}}
lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "playableenvironments_amd", "libplayrender.so")
if os.path.exists(lib):
    with open(lib, "rb") as f:
        out["library_sha256"] = hashlib.sha256(f.read()).hexdigest()
total = 0.0
for name, counters in sorted(per.items(), key=lambda kv: -(kv[1].get("FETCH_SIZE", 0) * 2 + kv[1].get("WRITE_SIZE", 0))):
    fetch = counters.get("FETCH_SIZE", 0.0) * 1024.0 / steps
    write = counters.get("WRITE_SIZE", 0.0) * 1024.0 / steps
    launches = max(len(calls[(name, k)]) for k in counters) / steps
    out["per_step"][name] = {"launches": round(launches, 2), "fetch_MB_raw": round(fetch / 1e6, 1), "fetch_MB_corrected_x2": round(2 * fetch / 1e6, 1),
                             "write_MB": round(write / 1e6, 1), "hbm_MB": round((2 * fetch + write) / 1e6, 1)}
    total += 2 * fetch + write
out["hbm_MB_per_step_all_library_kernels"] = round(total / 1e6, 1)
print(json.dumps(out, indent=1))
