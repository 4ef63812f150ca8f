"""Timeline of ONE training step out of a rocprofv3 kernel trace (the same step tools/summarise_train_trace.py picks):

    python tools/trace_timeline.py <dir>/*/*_kernel_trace.csv [--all]

Prints every launch of the step - start (us from the step's first launch), duration, gap to the previous launch on the same
queue, queue, workgroups, kernel - then how much of the step's wall time had 0 / 1 / 2+ kernels in flight.  Development tool:
it shows where the GPU idles between launches and which launches overlap."""
import csv
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"<.*$", "", name)
    return name[-48:]


def main(path, show_all):
    rows = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
            grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Queue_Id"]), grid // max(1, wg)))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if "k_place" in r[2]]
    heads = [i for n, i in enumerate(starts) if n == 0 or rows[i][0] - rows[starts[n - 1]][0] > 2_000_000]
    for n in range(len(heads) - 1, 0, -1):
        a, b = heads[n - 1], heads[n]
        if any("k_composite_bwd" in r[2] for r in rows[a:b]):
            break
    else:
        raise SystemExit("no training step in the trace")
    # the optimiser / loss kernels that precede the placement launch belong to the previous step: start at the placement
    step = rows[a:b]
    t0 = step[0][0]
    wall = (rows[b][0] - t0) / 1e3
    last_end = {}
    print(f"step wall {wall:.1f} us, {len(step)} launches")
    if show_all:
        for s, e, name, q, wgs in step:
            gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
            last_end[q] = e
            print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} gap {gap:7.1f} q{q} wg {wgs:6d} {short(name)}")
    # occupancy of the timeline
    events = []
    for s, e, *_ in step:
        events.append((s, 1))
        events.append((e, -1))
    events.sort()
    depth, prev, hist = 0, t0, {}
    for t, d in events:
        hist[min(depth, 2)] = hist.get(min(depth, 2), 0) + (t - prev)
        depth += d
        prev = t
    hist[0] = hist.get(0, 0) + max(0, rows[b][0] - prev)
    print("in flight: " + ", ".join(f"{k}{'+' if k == 2 else ''}: {v / 1e3:.0f} us" for k, v in sorted(hist.items())))
    per = {}
    for s, e, name, q, wgs in step:
        k = per.setdefault(short(name), [0.0, 0])
        k[0] += (e - s) / 1e3
        k[1] += 1
    for name, (us, calls) in sorted(per.items(), key=lambda kv: -kv[1][0]):
        print(f"{us:9.1f} us {calls:4d} x {name}")


if __name__ == "__main__":
    main(sys.argv[1], "--all" in sys.argv)
