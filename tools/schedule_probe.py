#!/usr/bin/env python3
"""What "pip_concurrent" 2 measures and decides on a stand-in pair: the span it saw per schedule, the grids of the
shared schedule, and the wall time of the settled steps (GPU only)."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayjoin_amd import _capi, maps, synth
ap = argparse.ArgumentParser()
ap.add_argument("--base", default="USCounty"); ap.add_argument("--query", default="BlockGroup")
ap.add_argument("--scale", type=float, default=1.0); ap.add_argument("--steps", type=int, default=12)
a = ap.parse_args()
ctx = maps.Context([synth.standin(a.base, a.scale), synth.standin(a.query, a.scale)]).load()
b, q = ctx.maps
h = _capi.Handle(0)
h.upload_map(0, b.pts, b.row_index, b.left, b.right); h.upload_map(1, q.pts, q.row_index, q.left, q.right)
h.build_lbvh(0)
cap = int(0.1 * (b.n_edges + q.n_edges)) + 1024
pairs = h.alloc(8 * cap); xs = h.alloc(48 * cap); closest = h.alloc(4 * q.n_points); faces = h.alloc(4 * q.n_points)
h.lsi_query(0, 1, 0, q.n_edges, cap, pairs); h.pip_query(0, 1, None, 0, q.n_points, closest, faces)  # (setup, as in bench.py)
h.set_option("pip_concurrent", 2)
for i in range(a.steps):
    t0 = time.perf_counter()
    h.lsi_query_async(0, 1, 0, q.n_edges, cap, pairs)
    early = h.get_option("pip_schedule") in (1, 2)
    if early:
        h.pip_query(0, 1, None, 0, q.n_points, closest, faces, sync=False)
    h.lsi_points_async(pairs, cap, xs)
    if not early:
        h.pip_query(0, 1, None, 0, q.n_points, closest, faces, sync=False)
    h.lsi_query_finish(cap); h.sync()
    dt = time.perf_counter() - t0
    print(json.dumps({"step": i, "wall_ms": round(dt * 1e3, 4), "schedule": h.get_option("pip_schedule"), "trials": h.get_option("pip_schedule_trials"),
                      "span_us": [h.get_option("pip_schedule_us%d" % m) for m in range(3)],
                      "share": [h.get_option("lsi_share_blocks"), h.get_option("pip_share_blocks")],
                      "lsi_k": round(h.last_ms(_capi.RJ_T_LSI_KERNEL), 4), "pip_q": round(h.last_ms(_capi.RJ_T_PIP_KERNEL), 4)}), flush=True)
