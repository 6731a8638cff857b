#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs (one directory per counter pass) into
profiles/<tag>_pmc_summary.csv and profiles/traffic.json (which records the hash of the kernel
sources it was measured on: bench.py quotes it only while that hash matches the tree).

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KB, and on gfx950
FETCH_SIZE reports half of the bytes of 16-B-per-lane reads (MI355X_MICROARCH.md, HBM section;
calibrated on k_build_segs in profiles/README.md) -- and, since round 6, calibrated on SCATTERED reads
too (tools/fetch_calibrate.hip, profiles/r06_fetch_calibration.txt): a 4-, 8- or 16-byte read of a line
nobody else reads moves the whole 128-byte line and is tallied as 64 bytes, two reads in the two halves
of one line are ONE request: the factor 2 holds for k_pip_strip's access pattern as for a stream.

usage: pmc_summary.py <tag> <fetch_dir> <write_dir> [<sq_dir> ...]
(the optional SQ_* passes of tools/profile_run.sh go to profiles/<tag>_sq_counters.csv, k_lsi / k_pip only)
"""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    if "onesweep" in name or "radix" in name:
        return "rocprim_radix_sort"
    m = re.search(r"rj::(?:\(anonymous namespace\)::)?(k_[a-z_0-9]+)", name)
    if m:
        return m.group(1)
    return name.split("(")[0][:40]


KERNELS = ("k_lsi", "k_lsi2", "k_lsix", "k_lsi2x", "k_pip_walk", "k_pip_walk2", "k_pip_walk4", "k_pip_strip", "k_pip_exact", "k_pip", "k_lsi_points", "k_lsi_points_gcd")


def full_grid(vals):
    """k_pip runs twice per PIP query of a profiled run: never on the whole query map (it only takes the walk's
    overflowed lists).  Counter values of launches that did next to nothing would halve an average: keep the
    launches within 4x of the largest."""
    if not vals:
        return vals
    top = max(vals)
    return [v for v in vals if v * 4 >= top]


def collect(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Counter_Name"]][short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc


def summarise(fetch_dir, write_dir, sq_dirs):
    """-> (rows of the FETCH/WRITE table, traffic per kernel, SQ counters per kernel) of one set of counter passes"""
    acc = collect(fetch_dir)
    for k, v in collect(write_dir).items():
        acc[k].update(v)
    rows = []
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        for kern, vals in sorted(acc.get(counter, {}).items()):
            vals = full_grid(vals) if kern in KERNELS else vals
            rows.append((counter, kern, len(vals), sum(vals) / len(vals)))
    avg = {(c, k): v for c, k, _, v in rows}
    traffic = {}
    for kern in KERNELS:
        if ("FETCH_SIZE", kern) in avg and ("WRITE_SIZE", kern) in avg:
            traffic[kern] = int((2 * avg[("FETCH_SIZE", kern)] + avg[("WRITE_SIZE", kern)]) * 1024)
    # instruction-issue evidence for the bench line's "limiter" (SQ passes, optional)
    sq, sq_rows = {}, []
    for d in sq_dirs:
        for counter, per_kernel in sorted(collect(d).items()):
            for kern in KERNELS:
                v = full_grid(per_kernel.get(kern))
                if v:
                    sq.setdefault(kern, {})[counter] = sum(v) / len(v)
                    sq_rows.append((counter, kern, len(v), sum(v) / len(v)))
    return rows, traffic, sq, sq_rows


def main():
    """pmc_summary.py <tag> <fetch_dir> <write_dir> [<sq_dir> ...] [--section <name> <fetch_dir> <write_dir> <sq_dir> <sq_dir>] ...
    The first set = the headline pair, every kernel alone on its full grid (bench.py --serial-kernels); a --section is
    another regime or pair (tools/regime_probe.py): "shared" = the headline's kernels on the grids of the shared schedule,
    "<base>_<query>" = another pair.  All of it goes to profiles/traffic.json; bench.py quotes a section by its name."""
    argv = sys.argv[1:]
    sections = []
    while "--section" in argv:
        i = argv.index("--section")
        sections.append(argv[i + 1:i + 6])
        argv = argv[:i] + argv[i + 6:]
    tag, fetch_dir, write_dir = argv[:3]
    rows, traffic, sq, sq_rows = summarise(fetch_dir, write_dir, argv[3:])
    if sq_rows:
        with open(os.path.join(ROOT, "profiles", "%s_sq_counters.csv" % tag), "w") as f:
            f.write("counter,kernel,dispatches,avg_value\n")
            for r in sq_rows:
                f.write("%s,%s,%d,%.0f\n" % r)
    out = os.path.join(ROOT, "profiles", "%s_pmc_summary.csv" % tag)
    with open(out, "w") as f:
        f.write("counter,kernel,dispatches,avg_value_KB\n")
        for r in rows:
            f.write("%s,%s,%d,%.1f\n" % r)
    sys.path.insert(0, ROOT)
    from rayjoin_amd._capi import kernel_source_hash
    doc = {"tag": tag, "kernel_source_hash": kernel_source_hash(), "traffic": traffic, "sq": sq, "sections": {},
           # (tools/fetch_calibrate.hip: what the factor 2 on FETCH_SIZE rests on for scattered reads)
           "fetch_calibration": "x2: FETCH_SIZE tallies a 128-byte line as 64 bytes for streams AND for scattered 4/8/16-byte reads (one request per line touched, "
                                "both halves of a line = one request): profiles/r06_fetch_calibration.txt"}
    if sections:
        with open(os.path.join(ROOT, "profiles", "%s_pmc_sections.csv" % tag), "w") as f:
            f.write("section,counter,kernel,dispatches,avg_value\n")
            for name, fd, wd, s1, s2 in sections:
                r2, t2, q2, sr2 = summarise(fd, wd, [s1, s2])
                doc["sections"][name] = {"traffic": t2, "sq": q2}
                for r in r2:
                    f.write("%s,%s_KB,%s,%d,%.1f\n" % ((name, r[0]) + r[1:]))
                for r in sr2:
                    f.write("%s,%s,%s,%d,%.0f\n" % ((name,) + r))
    json.dump(doc, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    print(out, json.dumps(doc)[:600])


if __name__ == "__main__":
    main()
