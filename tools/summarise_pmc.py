"""Sums the rocprofv3 counter_collection CSVs of tools/collect_pmc.sh per kernel and prints the JSON kept under profiles/.

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB; FETCH_SIZE is doubled for the wide (16 B/lane) streaming reads
of these kernels as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950; WRITE_SIZE is left as is
and cross-checked against the algorithmic bytes written (feature rows x row bytes)."""
import csv
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
per = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for path in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"].split("(")[0]
            if name.startswith("void "):          # template instantiations are printed with their return type
                name = name[5:]
            per[name][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[(name, row["Counter_Name"])].add(row["Dispatch_Id"])
out = {"source": "tools/collect_pmc.sh: rocprofv3 --kernel-trace --pmc <one set per pass> -- python bench.py --steps 1 --warmup 0 "
                 "--no-cpu-baseline --no-train-step --no-split-precision (2 renders: the timed step + the sample-count pass)",
       "renders": 2, "kernels": {}}
# the library the counters were collected from: bench.py compares it with the library it runs (a stale traffic figure shows)
lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "playableenvironments_amd", "libplayrender.so")
if os.path.exists(lib):
    with open(lib, "rb") as f:
        out["library_sha256"] = hashlib.sha256(f.read()).hexdigest()
for name, counters in sorted(per.items()):
    if not name.startswith("pr::"):
        continue
    entry = {k: v for k, v in counters.items()}
    entry["dispatches"] = max(len(calls[(name, k)]) for k in counters)
    out["kernels"][name] = entry
# the evaluation MLP runs as k_mlp_mfma_group (every object of a model type in one launch; one object: k_mlp_mfma)
for kernel in ("pr::k_mlp_mfma_group", "pr::k_mlp_mfma", "pr::k_composite<4>"):
    k = out["kernels"].get(kernel)
    if not k or "FETCH_SIZE" not in k or "WRITE_SIZE" not in k:
        continue
    if kernel == "pr::k_mlp_mfma" and "k_mlp_mfma" in out:
        continue
    launches = k["dispatches"]
    fetch = k["FETCH_SIZE"] * 1024.0
    write = k["WRITE_SIZE"] * 1024.0
    key = "k_mlp_mfma" if "k_mlp_mfma" in kernel else kernel.split("::")[1].split("<")[0]
    out[key] = {
        "kernel": kernel,
        "launches_per_render": launches / out["renders"],
        "fetch_bytes_per_render_raw": fetch / out["renders"],
        "fetch_bytes_per_render_corrected_x2": 2 * fetch / out["renders"],
        "write_bytes_per_render": write / out["renders"],
        "hbm_bytes_per_launch_avg": (2 * fetch + write) / launches,
    }
    if "SQ_VALU_MFMA_BUSY_CYCLES" in k and k.get("GRBM_GUI_ACTIVE"):
        # the SQ counter is summed over the 1024 SIMDs (4 per CU x 256 CUs), GRBM_GUI_ACTIVE over the 8 XCDs
        out[key]["mfma_busy_fraction"] = (k["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (k["GRBM_GUI_ACTIVE"] / 8.0)
print(json.dumps(out, indent=1))
