#!/usr/bin/env python3
"""Sweep a handle option on the full-size workload (GPU only): LSI / PIP kernel ms per value."""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayjoin_amd import _capi, maps, synth
ap = argparse.ArgumentParser()
ap.add_argument("--base", default="USCounty"); ap.add_argument("--query", default="BlockGroup")
ap.add_argument("--scale", type=float, default=1.0)
ap.add_argument("--opt", default="chunk_groups"); ap.add_argument("--values", default="1,2,4,8,16,32,64")
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
ctx = maps.Context([synth.standin(a.base, a.scale), synth.standin(a.query, a.scale)]).load()
b, q = ctx.maps
h = _capi.Handle(0)
h.upload_map(0, b.pts, b.row_index, b.left, b.right); h.upload_map(1, q.pts, q.row_index, q.left, q.right)
h.build_lbvh(0)
cap = int(0.1 * (b.n_edges + q.n_edges)) + 1024
pairs = h.alloc(8 * cap); closest = h.alloc(4 * q.n_points)
for v in a.values.split(","):
    try:
        h.set_option(a.opt, int(v))
    except Exception:
        h.set_debug_option(a.opt, int(v))  # (grids, chunk sizes, run lengths: the experiment knobs)
    l, p = [], []
    for _ in range(a.reps):
        n = h.lsi_query(0, 1, 0, q.n_edges, cap, pairs); l.append(h.last_ms(_capi.RJ_T_LSI_KERNEL))
        h.pip_query(0, 1, None, 0, q.n_points, closest, None); p.append(h.last_ms(_capi.RJ_T_PIP_KERNEL))
    print(json.dumps({a.opt: int(v), "lsi_ms": round(float(np.median(l)), 4), "pip_ms": round(float(np.median(p)), 4), "n": n}))
