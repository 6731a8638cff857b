#!/usr/bin/env python3
"""Kernel by kernel, a few steps from the MIDDLE of bench.py's pipelined loop (rocprofv3 --kernel-trace CSV).
usage: pipelined_trace.py <g_kernel_trace.csv> <first opener index> <steps>   (openers: k_lsi / k_lsi2 launches in order)"""
import csv, re, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"rj::(?:\(anonymous namespace\)::)?(k_[a-z_0-9]+)", r["Kernel_Name"])
    name = m.group(1) if m else r["Kernel_Name"].split("(")[0].split("<")[0][-34:]
    wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 256)) or 256)
    grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, grid // max(1, wg), r.get("Queue_Id", "")))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2] in ("k_lsi", "k_lsi2")]
a, n = int(sys.argv[2]), int(sys.argv[3])
t0 = rows[starts[a]][0]
print("%-22s %10s %10s %9s %7s %6s" % ("kernel", "start_us", "end_us", "dur_us", "blocks", "queue"))
for s, e, nm, b, q in rows[starts[a]:starts[a + n]]:
    print("%-22s %10.1f %10.1f %9.1f %7d %6s" % (nm, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, b, q))
