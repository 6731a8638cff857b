#!/usr/bin/env python3
"""-mode=grid vs -mode=lbvh on the device at full size: build / LSI / PIP times and result equality."""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayjoin_amd import _capi, maps, synth
ap = argparse.ArgumentParser()
ap.add_argument("--base", default="USCounty"); ap.add_argument("--query", default="BlockGroup")
ap.add_argument("--scale", type=float, default=1.0); ap.add_argument("--grid-sizes", default="2048,8192")
a = ap.parse_args()
ctx = maps.Context([synth.standin(a.base, a.scale), synth.standin(a.query, a.scale)]).load()
b, q = ctx.maps
h = _capi.Handle(0)
h.upload_map(0, b.pts, b.row_index, b.left, b.right); h.upload_map(1, q.pts, q.row_index, q.left, q.right)
h.build_lbvh(0)
cap = int(0.1 * (b.n_edges + q.n_edges)) + 1024
pairs = h.alloc(8 * cap); closest = h.alloc(4 * q.n_points)
n = h.lsi_query(0, 1, 0, q.n_edges, cap, pairs); lsi_lbvh = h.last_ms(_capi.RJ_T_LSI_KERNEL)
h.sort_pairs(pairs, n); want_pairs = pairs.to_host(np.uint32, 2 * n)
h.pip_query(0, 1, None, 0, q.n_points, closest, None); pip_lbvh = h.last_ms(_capi.RJ_T_PIP_KERNEL)
want_eids = closest.to_host(np.uint32)
out = {"base": a.base, "query": a.query, "intersections": n, "lbvh": {"lsi_ms": round(lsi_lbvh, 3), "pip_ms": round(pip_lbvh, 3)}}
for g in [int(v) for v in a.grid_sizes.split(",")]:
    h.build_grid(0, g); b0 = h.last_ms(_capi.RJ_T_BUILD)
    h.build_grid(1, g); b1 = h.last_ms(_capi.RJ_T_BUILD)
    ng = h.lsi_query_grid(cap, pairs); lsi_ms = h.last_ms(_capi.RJ_T_LSI_KERNEL)
    h.sort_pairs(pairs, ng)
    same_pairs = bool(ng == n and np.array_equal(pairs.to_host(np.uint32, 2 * ng), want_pairs))
    h.pip_query_grid(0, 1, None, 0, q.n_points, closest, None); pip_ms = h.last_ms(_capi.RJ_T_PIP_KERNEL)
    same_eids = bool(np.array_equal(closest.to_host(np.uint32), want_eids))
    out["grid_%d" % g] = {"build_ms": [round(b0, 2), round(b1, 2)], "lsi_ms": round(lsi_ms, 3), "pip_ms": round(pip_ms, 3),
                          "pairs_equal_lbvh": same_pairs, "eids_equal_lbvh": same_eids}
print(json.dumps(out))
