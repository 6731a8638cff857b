#!/usr/bin/env python3
"""Would a PIP query issued in K parts, each part's exact kernel on its own stream beside the next part's walk
("pip_exact_stream" 1), shorten the step of a pair whose exact kernel is long?  The synchronised step of bench.py with
the PIP side as K calls over K point ranges.  usage: parts_probe.py [--base USCounty --query NestedBlockGroup --parts 1,2,4]"""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayjoin_amd import _capi, maps, synth
ap = argparse.ArgumentParser()
ap.add_argument("--base", default="USCounty"); ap.add_argument("--query", default="NestedBlockGroup")
ap.add_argument("--parts", default="1,2,4"); ap.add_argument("--steps", type=int, default=30)
a = ap.parse_args()
ctx = maps.Context([synth.standin(a.base), synth.standin(a.query)]).load()
b, q = ctx.maps
h = _capi.Handle(0)
h.upload_map(0, b.pts, b.row_index, b.left, b.right); h.upload_map(1, q.pts, q.row_index, q.left, q.right)
h.build_lbvh(0)
cap = int(0.12 * (b.n_edges + q.n_edges))
pairs = h.alloc(8 * cap); xs = h.alloc(48 * cap); closest = h.alloc(4 * q.n_points); faces = h.alloc(4 * q.n_points)
ref = None
import time
for K in [int(v) for v in a.parts.split(",")]:
    h.set_option("pip_concurrent", 0); h.set_option("pip_concurrent", 2)   # (the schedule is decided again)
    h.set_option("pip_exact_stream", 1 if K > 1 else 0)
    n = q.n_points
    cuts = [(n * k // K) // 256 * 256 for k in range(K)] + [n]
    def step():
        early = h.get_option("pip_schedule") in (1, 2)
        h.lsi_query_async(0, 1, 0, q.n_edges, cap, pairs)
        def pip():
            for k in range(K):
                h.pip_query(0, 1, None, cuts[k], cuts[k + 1] - cuts[k], closest.data_ptr() + 4 * cuts[k], faces.data_ptr() + 4 * cuts[k], sync=False)
        if early: pip()
        h.lsi_points_async(pairs, cap, xs)
        if not early: pip()
        m = h.lsi_query_finish(cap); h.sync(); return m
    for _ in range(10): step()
    t = []
    for _ in range(a.steps):
        t0 = time.perf_counter(); step(); t.append((time.perf_counter() - t0) * 1e3)
    e = closest.to_host(np.uint32)[:n].copy()
    if ref is None: ref = e
    print(json.dumps({"pair": a.base + " x " + a.query, "parts": K, "step_ms_median": round(float(np.median(t)), 4), "schedule": h.get_option("pip_schedule"),
                      "equal_to_one_part": bool(np.array_equal(e, ref))}), flush=True)
