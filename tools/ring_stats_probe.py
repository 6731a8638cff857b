#!/usr/bin/env python3
"""Where a PIP query over a ring-shaped base map spends its leaf visits (GPU only): the instrumented kernel's counters
per query point for several group sizes (queries per wave), and the distribution of answers (miss / hit)."""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayjoin_amd import _capi, maps, synth
ap = argparse.ArgumentParser()
ap.add_argument("--base", default="WaterBodiesLike"); ap.add_argument("--query", default="BlockGroup")
ap.add_argument("--scale", type=float, default=1.0)
a = ap.parse_args()
ctx = maps.Context([synth.standin(a.base, a.scale), synth.standin(a.query, a.scale)]).load()
b, q = ctx.maps
h = _capi.Handle(0)
h.upload_map(0, b.pts, b.row_index, b.left, b.right); h.upload_map(1, q.pts, q.row_index, q.left, q.right)
h.build_lbvh(0)
n = q.n_points
closest = h.alloc(4 * n)
h.pip_query(0, 1, None, 0, n, closest, None)
e = closest.to_host(np.uint32)
print(json.dumps({"points": n, "miss_frac": float((e == 0xFFFFFFFF).mean()), "skyline": h.get_option("skyline_used0"), "slots_per_segment": h.get_option("leaf_slots0") / b.n_edges}))
h.set_option("pip_walk", 0)
h.set_option("stats", 1)
for gl in (64, 16, 4):
    h.set_debug_option("group_lanes", gl)
    h.pip_query(0, 1, None, 0, n, closest, None)
    st = h.last_stats()
    print(json.dumps({"group_lanes": gl, "ms": round(h.last_ms(_capi.RJ_T_PIP_KERNEL), 2), "per_point": {k: round(st[k] / n, 3) for k in ("leaf_blocks", "nodes_expanded", "leaf_box_tests", "stale_pops", "leaf_interested_lanes", "exact_tests")}}))
