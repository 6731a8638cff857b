#!/usr/bin/env python3
"""Two tables from a rocprofv3 --kernel-trace CSV (g_kernel_trace.csv):
  --steps N    the last N steps of the run, kernel by kernel (start/end in microseconds from the step's first kernel)
  --regimes    per kernel AND grid size: launches, mean / min / max duration -- a kernel that runs on a full grid, on
               a shared grid and on a handful of blocks in one run is three regimes, not one average
usage: trace_tables.py <g_kernel_trace.csv> [--steps N] [--regimes] [--first k_lsi]"""
import argparse, collections, csv, re, sys
ap = argparse.ArgumentParser()
ap.add_argument("csv"); ap.add_argument("--steps", type=int, default=0); ap.add_argument("--regimes", action="store_true")
ap.add_argument("--first", default="k_lsi", help="the kernel that opens a step")
ap.add_argument("--skip", type=int, default=0, help="ignore the last K step openers (bench.py ends with 4 solo queries per kind)")
a = ap.parse_args()
rows = []
for r in csv.DictReader(open(a.csv)):
    m = re.search(r"rj::(?:\(anonymous namespace\)::)?(k_[a-z_0-9]+)", r["Kernel_Name"])
    name = m.group(1) if m else r["Kernel_Name"].split("(")[0].split("<")[0][-34:]
    wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 256)) or 256)
    grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, grid // max(1, wg)))
rows.sort()
if a.regimes:
    acc = collections.defaultdict(list)
    for s, e, n, b in rows:
        if n.startswith("k_"):
            acc[(n, b)].append((e - s) / 1e3)
    print("kernel,blocks,launches,mean_us,min_us,max_us")
    for (n, b), v in sorted(acc.items()):
        print("%s,%d,%d,%.1f,%.1f,%.1f" % (n, b, len(v), sum(v) / len(v), min(v), max(v)))
if a.steps:
    starts = [i for i, r in enumerate(rows) if r[2] in (a.first, a.first + "2", a.first + "x", a.first + "2x")]  # (k_lsi / k_lsi2 / their x-order-only forms, whichever the run used)
    if a.skip:
        starts = starts[:-a.skip]
    starts = starts[-a.steps - 1:] if len(starts) > a.steps else starts
    t0 = rows[starts[0]][0]
    print("%-34s %10s %10s %9s %7s" % ("kernel", "start_us", "end_us", "dur_us", "blocks"))
    for s, e, n, b in rows[starts[0]:starts[-1]]:
        print("%-34s %10.1f %10.1f %9.1f %7d" % (n, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, b))
