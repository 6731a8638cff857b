#!/usr/bin/env python3
"""The table of tools/fetch_calibrate.hip: FETCH_SIZE per read of a known size, per access pattern.

usage (on the GPU box, cd /tmp first):
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -o g -- <repo>/tools/fetch_calibrate > <dir>/run.json
    python3 tools/fetch_calibrate.py <dir> [<out.json>]
FETCH_SIZE is in KB (tools/pmc_summary.py).  The program runs every pattern three times; the counter is averaged over them."""
import collections
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
run = json.loads(open(os.path.join(d, "run.json")).read().strip().splitlines()[-1])
n, table = run["reads"], run["table_bytes"]
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]) * 1024.0)


def of(fragment):
    v = [x for k, xs in acc.items() if fragment in k for x in xs]
    return sum(v) / len(v) if v else None


# pattern -> (kernel-name fragment, reads per lane, bytes asked for per lane, 128-byte lines touched per lane, 64-byte halves touched per lane)
P = collections.OrderedDict([
    ("stream", ("stream", None, None, None, None)),
    ("rand16", ("scattered<16, -1, 128>", 1, 16, 1.0, 1.0)),
    ("rand16_same64", ("scattered<16, 16, 128>", 2, 32, 1.0, 1.0)),
    ("rand16_other64", ("scattered<16, 64, 128>", 2, 32, 1.0, 2.0)),
    ("rand_pair", ("scattered<16, 16, 16>", 2, 32, 1.0 + 1.0 / 8.0, 1.0 + 1.0 / 4.0)),   # entries j, j + 1 at a random 16-byte-aligned j: straddle a line 1 time in 8, a half 1 in 4
    ("rand8", ("scattered<8, -1, 8>", 1, 8, 1.0, 1.0)),
    ("rand4", ("scattered<4, -1, 4>", 1, 4, 1.0, 1.0))])
out = {"table_bytes": table, "lane_reads_per_pattern": n, "patterns": {}}
print("%-16s %14s %12s %12s %12s %10s %14s" % ("pattern", "FETCH_SIZE B", "B asked", "per lane", "per line", "ms", "G lane-reads/s"))
for name, (frag, reads, asked, lines, halves) in P.items():
    fs = of(frag)
    ms = run.get(name + "_ms")
    if fs is None:
        continue
    if name == "stream":
        row = {"fetch_size_bytes": fs, "bytes_read": table, "counter_over_bytes": fs / table, "ms": ms, "GBps": table / ms / 1e6}
        print("%-16s %14.0f %12d %12s %12s %10.3f   (counter / bytes = %.3f, %.0f GB/s)" % (name, fs, table, "-", "-", ms, fs / table, table / ms / 1e6))
    else:
        row = {"fetch_size_bytes": fs, "bytes_asked": asked * n, "counter_bytes_per_lane": fs / n, "counter_bytes_per_128B_line": fs / n / lines,
               "counter_bytes_per_64B_half": fs / n / halves, "ms": ms, "G_lane_reads_per_s": reads * n / ms / 1e6,
               "TBps_if_128B_lines": lines * n * 128 / ms / 1e9, "TBps_if_64B_halves": halves * n * 64 / ms / 1e9}
        print("%-16s %14.0f %12d %12.1f %12.1f %10.3f %14.2f" % (name, fs, asked * n, fs / n, fs / n / lines, ms, reads * n / ms / 1e6))
    out["patterns"][name] = row
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
