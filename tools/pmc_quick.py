#!/usr/bin/env python3
"""Per-kernel averages of every counter in a rocprofv3 --pmc output directory (quick look, no files written).
usage: pmc_quick.py <dir> [kernel-substring ...]"""
import collections, csv, glob, os, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"rj::(k_[a-z_0-9]+)(<[a-z0-9]+>)?", r["Kernel_Name"])
        name = (m.group(1) + (m.group(2) or "")) if m else r["Kernel_Name"].split("(")[0][:40]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
want = sys.argv[2:]
for k in sorted(acc):
    if want and not any(w in k for w in want):
        continue
    print(k, {c: (len(v), round(sum(v) / len(v))) for c, v in sorted(acc[k].items())})
