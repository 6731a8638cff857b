#!/usr/bin/env python3
"""How long the scans of k_pip_strip are (GPU only): the instrumented kernel's counters for a ring-shaped pair -- trips of a
wave per 128-position group (a wave goes on until its slowest point is done), entries read per point, how many of them end
below the point (read because ONE tall box of the strip lowers every point's start), the longest scan, points by length.
usage: strip_stats_probe.py [--base WaterBodiesLike --query BlockGroup]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayjoin_amd import _capi, maps, synth
ap = argparse.ArgumentParser()
ap.add_argument("--base", default="WaterBodiesLike"); ap.add_argument("--query", default="BlockGroup"); ap.add_argument("--solo", default="0")
a = ap.parse_args()
ctx = maps.Context([synth.standin(a.base), synth.standin(a.query)]).load()
b, q = ctx.maps
h = _capi.Handle(0)
h.upload_map(0, b.pts, b.row_index, b.left, b.right); h.upload_map(1, q.pts, q.row_index, q.left, q.right)
h.build_lbvh(0)
n = q.n_points
closest = h.alloc(4 * n); faces = h.alloc(4 * n)
h.set_option("pip_walk", 2)
for solo in [int(v) for v in a.solo.split(",")]:
    h.set_debug_option("walk_stack", solo)
    ms = []
    for _ in range(3):
        h.pip_query(0, 1, None, 0, n, closest, faces)
        ms.append(h.last_ms(_capi.RJ_T_PIP_WALK))
    if len(a.solo.split(",")) > 1:
        print(json.dumps({"pair": "%s x %s" % (a.base, a.query), "solo_entries": solo or "default", "first_pass_ms": round(min(ms), 4)}), flush=True)
h.set_option("stats", 1)
h.pip_query(0, 1, None, 0, n, closest, faces)
st = h.last_stats_raw()
bins = ("0", "<=2", "<=4", "<=8", "<=16", "<=32", "<=64", ">64")
print(json.dumps({"pair": "%s x %s" % (a.base, a.query), "points": n, "first_pass_ms": round(min(ms), 4), "column_shift": h.get_option("pip_column_shift0"),
                  "entries": h.get_option("pip_column_entries0"), "groups": st[0], "solo_trips_of_lane0_per_group": round(st[1] / st[0], 2), "cooperative_trips_per_group": round(st[13] / st[0], 2),
                  "entries_read_solo_per_point": round(st[2] / n, 3), "ending_below_the_point_per_point": round(st[3] / n, 3), "longest_scan": st[4],
                  "points_by_scan_length": {k: round(st[5 + i] / n, 5) for i, k in enumerate(bins)}}))
