#!/usr/bin/env python3
"""Hilbert-neighbour leaves vs chain-run leaves ("leaf_order" 0 / 1) on a stand-in pair (GPU only): index build time
and size, kernel times alone on the chip, visit counters of the instrumented kernels, equality of the results."""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayjoin_amd import _capi, maps, synth
ap = argparse.ArgumentParser()
ap.add_argument("--base", default="USCounty"); ap.add_argument("--query", default="BlockGroup")
ap.add_argument("--scale", type=float, default=1.0); ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--variants", default="0:0,1:0", help="leaf_order:pack_solo[:run_cap[:pack_spread[:pip_columns]]] , ...  (0 = the default of each; pip_columns: 2 = auto, 0 never, 1 always)")
a = ap.parse_args()
ctx = maps.Context([synth.standin(a.base, a.scale), synth.standin(a.query, a.scale)]).load()
b, q = ctx.maps
h = _capi.Handle(0)
h.upload_map(0, b.pts, b.row_index, b.left, b.right); h.upload_map(1, q.pts, q.row_index, q.left, q.right)
cap = int(0.25 * (b.n_edges + q.n_edges)) + 1024
pairs = h.alloc(8 * cap); closest = h.alloc(4 * q.n_points); face = h.alloc(4 * q.n_points)
ng_p, ng_s = (q.n_points + 63) // 64, (q.n_edges + 63) // 64
res = {}
first = None
for var in a.variants.split(","):
    f = [int(x) for x in var.split(":")]
    order, solo, rcap, spread = f[0], (f[1] if len(f) > 1 else 0), (f[2] if len(f) > 2 else 0), (f[3] if len(f) > 3 else 0)
    h.set_option("pip_columns", -1 if len(f) <= 4 or f[4] == 2 else f[4])
    if rcap != h.get_debug_option("run_cap"):  # (the runs are cut once per uploaded map: cut them again)
        h.upload_map(0, b.pts, b.row_index, b.left, b.right)
    h.set_debug_option("run_cap", rcap)
    h.set_option("leaf_order", order)
    h.set_debug_option("pack_solo", solo)
    h.set_debug_option("pack_spread", spread)
    h.build_lbvh(0); first_ms = h.last_ms(_capi.RJ_T_BUILD); h.build_lbvh(0)
    order = var
    out = {"pair": "%s x %s" % (a.base, a.query), "variant": var, "first_build_ms": round(first_ms, 3), "build_ms": round(h.last_ms(_capi.RJ_T_BUILD), 3),
           "slots_per_segment": round(h.get_option("leaf_slots0") / b.n_edges, 3), "used": h.get_option("leaf_order_used0"),
           "skyline": h.get_option("skyline_used0"), "columns": h.get_option("pip_columns_used0")}
    l, p, w = [], [], []
    for _ in range(a.reps):
        n = h.lsi_query(0, 1, 0, q.n_edges, cap, pairs); l.append(h.last_ms(_capi.RJ_T_LSI_KERNEL))
        h.pip_query(0, 1, None, 0, q.n_points, closest, face); p.append(h.last_ms(_capi.RJ_T_PIP_KERNEL)); w.append(h.last_ms(_capi.RJ_T_PIP_WALK))
    out.update(lsi_ms=round(float(np.median(l)), 4), pip_query_ms=round(float(np.median(p)), 4), pip_walk_ms=round(float(np.median(w)), 4), xsects=n)
    h.sort_pairs(pairs, n)
    res[order] = (pairs.to_host(np.uint32, 2 * n).copy(), closest.to_host(np.uint32).copy(), face.to_host(np.int32).copy())
    first = first or order
    h.set_option("pip_walk", 0)
    for _ in range(2):
        h.pip_query(0, 1, None, 0, q.n_points, closest, face)
    out["pip_single_kernel_ms"] = round(h.last_ms(_capi.RJ_T_PIP_KERNEL), 4)
    h.set_option("stats", 1)
    h.pip_query(0, 1, None, 0, q.n_points, closest, face); st = h.last_stats()
    out["pip_per_group"] = {k: round(st[k] / ng_p, 2) for k in ("leaf_blocks", "nodes_expanded", "leaf_box_tests", "stale_pops", "leaf_interested_lanes")}
    h.lsi_query(0, 1, 0, q.n_edges, cap, pairs); st = h.last_stats()
    out["lsi_per_group"] = {k: round(st[k] / ng_s, 2) for k in ("leaf_blocks", "nodes_expanded", "leaf_box_tests", "exact_tests")}
    h.set_option("stats", 0); h.set_option("pip_walk", 1)
    print(json.dumps(out), flush=True)
print(json.dumps({"pair": "%s x %s" % (a.base, a.query), "same_pairs": all(np.array_equal(res[first][0], r[0]) for r in res.values()),
                  "same_eids": all(np.array_equal(res[first][1], r[1]) for r in res.values()),
                  "same_faces": all(np.array_equal(res[first][2], r[2]) for r in res.values())}))
