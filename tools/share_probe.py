#!/usr/bin/env python3
"""Whole step (LSI + records + PIP) with k_lsi and the PIP kernels SHARING the chip ("pip_concurrent" 1) over a grid
of (k_lsi blocks, k_pip_walk blocks), beside the two other schedules: where do the two sides end together?"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayjoin_amd import _capi, maps, synth
ap = argparse.ArgumentParser()
ap.add_argument("--base", default="USCounty"); ap.add_argument("--query", default="BlockGroup")
ap.add_argument("--scale", type=float, default=1.0); ap.add_argument("--reps", type=int, default=12)
ap.add_argument("--lsi", default="192,256,320,448"); ap.add_argument("--pip", default="1280,1536,1792,2048")
a = ap.parse_args()
ctx = maps.Context([synth.standin(a.base, a.scale), synth.standin(a.query, a.scale)]).load()
b, q = ctx.maps
h = _capi.Handle(0)
h.upload_map(0, b.pts, b.row_index, b.left, b.right); h.upload_map(1, q.pts, q.row_index, q.left, q.right)
h.build_lbvh(0)
cap = int(0.1 * (b.n_edges + q.n_edges)) + 1024
pairs = h.alloc(8 * cap); xs = h.alloc(48 * cap); closest = h.alloc(4 * q.n_points); faces = h.alloc(4 * q.n_points)
def run(early):
    ts = []
    for r in range(a.reps + 4):
        t0 = time.perf_counter()
        h.lsi_query_async(0, 1, 0, q.n_edges, cap, pairs)
        if early:
            h.pip_query(0, 1, None, 0, q.n_points, closest, faces, sync=False)
        h.lsi_points_async(pairs, cap, xs)
        if not early:
            h.pip_query(0, 1, None, 0, q.n_points, closest, faces, sync=False)
        h.lsi_query_finish(cap)
        h.sync()
        ts.append(time.perf_counter() - t0)
    return {"step_ms": round(float(np.median(ts[4:])) * 1e3, 4), "lsi_k": round(h.last_ms(_capi.RJ_T_LSI_KERNEL), 4),
            "pts_k": round(h.last_ms(_capi.RJ_T_LSI_POINTS), 4), "pip_k": round(h.last_ms(_capi.RJ_T_PIP_KERNEL), 4)}
h.set_option("pip_concurrent", 0)
print(json.dumps({"schedule": "turns", **run(False)}), flush=True)
h.set_option("pip_concurrent", 1)
for lb in [int(v) for v in a.lsi.split(",")]:
    for pb in [int(v) for v in a.pip.split(",")]:
        h.set_debug_option("lsi_share_blocks", lb); h.set_debug_option("pip_share_blocks", pb)
        print(json.dumps({"schedule": "shared", "lsi_blocks": lb, "pip_blocks": pb, **run(True)}), flush=True)
