#!/usr/bin/env python3
"""Every kernel of a run in launch order with durations and the idle time before it (rocprofv3 --kernel-trace CSV of
tools/build_probe.py --reps 2 --maps <one map>): where an index build's milliseconds go, host round trips included.
usage: build_trace.py <g_kernel_trace.csv>"""
import csv, re, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"rj::(?:\(anonymous namespace\)::)?(k_[a-z_0-9]+)", r["Kernel_Name"])
    name = m.group(1) if m else re.sub(r"^void ", "", r["Kernel_Name"])[:90]
    wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 256)) or 256)
    grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, grid // max(1, wg)))
rows.sort()
t0, prev = rows[0][0], rows[0][0]
print("%10s %8s %8s %8s  %s" % ("start_us", "idle_us", "dur_us", "blocks", "kernel"))
for s, e, n, g in rows:
    print("%10.1f %8.1f %8.1f %8d  %s" % ((s - t0) / 1e3, max(0, s - prev) / 1e3, (e - s) / 1e3, g, n))
    prev = max(prev, e)
