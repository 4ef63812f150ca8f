"""Summarises ONE training step out of a rocprofv3 kernel trace of bench.py's train_step leg (tools/perf/perf_train_leg.py):

    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python tools/perf/perf_train_leg.py 6 3
    python tools/summarise_train_trace.py <dir>/*/*_kernel_trace.csv > profiles/rNN_train_step_trace_summary.json

A step starts at a coarse-placement launch (pr::k_place_coarse[_group] is the first kernel of a renderer call; one call per
step) and ends before the next one - the optimiser, the loss and the next call's scene set-up launches included; the LAST
complete step (one with a backward pass) of the trace is reported."""
import csv
import json
import re
import sys
from collections import OrderedDict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    if name.startswith("at::native") or name.startswith("at::"):
        return "at::native (torch: optimiser, loss, zero fills)"
    return name


def main(path):
    rows = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if "k_place_coarse" in r[2]]
    # objects are placed back to back at the head of a call: keep the first launch of every run
    heads = [i for n, i in enumerate(starts) if n == 0 or rows[i][0] - rows[starts[n - 1]][0] > 2_000_000]
    if len(heads) < 3:
        raise SystemExit("need at least three renderer calls in the trace")
    # the harness ends with forward-only calls: take the last interval that contains a backward pass
    for n in range(len(heads) - 1, 0, -1):
        a, b = heads[n - 1], heads[n]
        if any("k_composite_bwd" in r[2] for r in rows[a:b]):
            break
    else:
        raise SystemExit("no training step (k_composite_bwd) in the trace")
    step = rows[a:b]
    kernels = OrderedDict()
    for s, e, name in step:
        k = kernels.setdefault(short(name), {"ms": 0.0, "calls": 0})
        k["ms"] += (e - s) / 1e6
        k["calls"] += 1
    kernels = OrderedDict(sorted(((k, {"ms": round(v["ms"], 4), "calls": v["calls"]}) for k, v in kernels.items()),
                                 key=lambda kv: -kv[1]["ms"]))
    busy = sum(v["ms"] for v in kernels.values())
    wall = (rows[b][0] - rows[a][0]) / 1e6
    print(json.dumps({"trace": path.split("/")[-1], "step_wall_ms_call_to_call": round(wall, 3), "launches": len(step),
                      "sum_of_kernel_durations_ms": round(busy, 3), "kernel_time_over_wall": round(busy / wall, 3),
                      "gpu_idle_ms": round(max(0.0, wall - busy), 3),
                      "note": "one stream: every object of a model type in one grouped launch per phase (round 3); wall = from this step's "
                              "first launch to the next step's, under the profiler",
                      "kernels": kernels}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
