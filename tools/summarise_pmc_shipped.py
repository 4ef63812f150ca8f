"""Per-kernel time and HBM traffic per frame of the shipped renderers from tools/collect_pmc_shipped.sh:
    python tools/summarise_pmc_shipped.py <dir> <renders per precision>
FETCH_SIZE / WRITE_SIZE in KB; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 (raw figure kept)."""
import csv
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict

root, renders = sys.argv[1], int(sys.argv[2])
out = {"source": "tools/collect_pmc_shipped.sh (tools/perf/perf_minecraft_eval.py <world>: fp32 then f16x3, 28 renders each)", "renders_per_precision": renders}
lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "playableenvironments_amd", "libplayrender.so")
if os.path.exists(lib):
    with open(lib, "rb") as f:
        out["library_sha256"] = hashlib.sha256(f.read()).hexdigest()


def short(name):
    name = name.split("(")[0]
    return name[5:] if name.startswith("void ") else name


for world in ("tennis", "minecraft"):
    per = defaultdict(lambda: defaultdict(float))
    for counter in ("fetch", "write"):
        for path in glob.glob(os.path.join(root, world, counter, "**", "*counter_collection.csv"), recursive=True):
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    name = short(row["Kernel_Name"])
                    if name.startswith("pr::"):
                        per[name][row["Counter_Name"]] += float(row["Counter_Value"])
    times = defaultdict(lambda: [0.0, 0])
    for path in glob.glob(os.path.join(root, world, "stats", "**", "*kernel_trace.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                name = short(row["Kernel_Name"])
                if name.startswith("pr::"):
                    times[name][0] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
                    times[name][1] += 1
    kernels = {}
    for name in sorted(times, key=lambda n: -times[n][0]):
        c = per.get(name, {})
        # a kernel belongs to one precision (k_mlp_mfma_group: fp32, k_mlp_split_group: f16x3) or runs in both (everything else)
        both = not ("k_mlp_mfma" in name or "k_mlp_split" in name)
        n = renders * (2 if both else 1)
        kernels[name] = {"ms_per_frame": round(times[name][0] / n, 4), "launches_per_frame": round(times[name][1] / n, 2),
                         "fetch_MB_corrected_x2_per_frame": round(2 * c.get("FETCH_SIZE", 0.0) * 1024 / n / 1e6, 1),
                         "write_MB_per_frame": round(c.get("WRITE_SIZE", 0.0) * 1024 / n / 1e6, 1)}
    out[world] = kernels
print(json.dumps(out, indent=1))
