#!/usr/bin/env python3
"""What a 128-position group of k_pip_walk2 does (GPU only): the instrumented kernel's event counts per group -- pops
(stale ones among them), node expansions, leaf blocks opened, (leaf, set) visits, scan steps, candidate hits, sweeps --
for one map pair; the kernel's time without the instrumentation beside them.  DESIGN.md section 4 prices the
events with the instruction counts of the kernel's ISA.
usage: walk_stats_probe.py [--base USCounty --query BlockGroup --scale 1.0]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayjoin_amd import _capi, maps, synth
ap = argparse.ArgumentParser()
ap.add_argument("--base", default="USCounty"); ap.add_argument("--query", default="BlockGroup")
ap.add_argument("--scale", type=float, default=1.0); ap.add_argument("--walk-points", type=int, default=2)
ap.add_argument("--cycles", action="store_true", help="cycle stamps of the phases instead of the event counts"); ap.add_argument("--max-blocks", type=int, default=0)
a = ap.parse_args()
ctx = maps.Context([synth.standin(a.base, a.scale), synth.standin(a.query, a.scale)]).load()
b, q = ctx.maps
h = _capi.Handle(0)
h.upload_map(0, b.pts, b.row_index, b.left, b.right); h.upload_map(1, q.pts, q.row_index, q.left, q.right)
h.build_lbvh(0)
n = q.n_points
closest = h.alloc(4 * n); faces = h.alloc(4 * n)
h.set_option("pip_walk", 2)
h.set_option("pip_walk_points", a.walk_points)
ms = []
for _ in range(4):
    h.pip_query(0, 1, None, 0, n, closest, faces)
    ms.append((h.last_ms(_capi.RJ_T_PIP_WALK), h.last_ms(_capi.RJ_T_PIP_KERNEL)))
if a.max_blocks:
    h.set_debug_option("max_blocks", a.max_blocks)
h.set_option("stats", 1)
if a.cycles:
    h.set_debug_option("stack_cap", 7777)  # kWalkCycleStamps
    h.pip_query(0, 1, None, 0, n, closest, faces)
    st = h.last_stats_raw()
    names = ["total", "scheduler", "first_loads_wait", "bound_and_root", None, "pops", "node_wait", "node", "leaf_wait", "leaf_scan", "bound_and_sweeps", "hand_over"]
    g, waves = st[4], st[12]
    print(json.dumps({"pair": "%s x %s" % (a.base, a.query), "blocks": a.max_blocks or "full grid", "waves": waves, "groups": g,
                      "cycles_per_group_of_a_wave": {k: round(st[i] / g) for i, k in enumerate(names) if k},
                      "share": {k: round(st[i] / st[0], 3) for i, k in enumerate(names) if k and k != "total"}}))
    sys.exit(0)
h.pip_query(0, 1, None, 0, n, closest, faces)
st = h.last_stats_raw() if hasattr(h, "last_stats_raw") else None
names = {0: "leaf_blocks", 1: "rest_points", 2: "nodes_expanded", 3: "scan_steps", 4: "groups", 5: "leaf_set_visits", 6: "hit_bodies",
         7: "sweeps", 8: "swept_entries", 10: "pops", 11: "pushed", 13: "stale_pops", 15: "wanting_lanes"}
g = st[4]
print(json.dumps({"pair": "%s x %s" % (a.base, a.query), "points": n, "walk_points": h.get_option("pip_last_walk_points"),
                  "walk_ms": round(min(m[0] for m in ms), 4), "pip_query_ms": round(min(m[1] for m in ms), 4), "groups": g,
                  "per_group": {v: round(st[k] / g, 3) for k, v in names.items() if k != 4}}))
