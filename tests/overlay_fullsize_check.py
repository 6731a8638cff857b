#!/usr/bin/env python3
"""Overlay stages through the C ABI at FULL size, checked bit for bit against the oracle pipeline
(test infrastructure: run as a child process by tests/test_gpu_fullsize.py so that its 30 M-segment
maps are freed before the next test)."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayjoin_amd import _capi, maps, synth
ap = argparse.ArgumentParser()
ap.add_argument("--m0", default="USCounty"); ap.add_argument("--m1", default="Zipcode")
ap.add_argument("--scale", type=float, default=1.0)
a = ap.parse_args()
ctx = maps.Context([synth.standin(a.m0, a.scale), synth.standin(a.m1, a.scale)]).load()
m = ctx.maps
h = _capi.Handle(0)
t = {}
def timed(name, fn, reps=3):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); h.sync(); best = min(best, time.perf_counter() - t0)
    t[name] = round(best * 1e3, 3)
    return r
for im in range(2):
    h.upload_map(im, m[im].pts, m[im].row_index, m[im].left, m[im].right)
timed("build_index_both", lambda: (h.build_lbvh(0), h.build_lbvh(1)))
cap = int(0.2 * (m[0].n_edges + m[1].n_edges))
pairs = h.alloc(8 * cap)
n = timed("intersect_edges(0)", lambda: h.lsi_query(1, 0, 0, m[0].n_edges, cap, pairs))
cl = [h.alloc(4 * m[i].n_points) for i in range(2)]; fc = [h.alloc(4 * m[i].n_points) for i in range(2)]
for im in range(2):
    timed("locate_vertices(%d)" % im, lambda im=im: h.pip_query(1 - im, im, None, 0, m[im].n_points, cl[im], fc[im]))
xs = [h.alloc(48 * max(1, n)) for _ in range(2)]
timed("compute_output_polygons", lambda: [h.overlay_edge_xsects(im, pairs, n, xs[im]) for im in range(2)])
out = {"map0_edges": m[0].n_edges, "map1_edges": m[1].n_edges, "intersections": n, "ms": t,
       "total_query_ms": round(sum(v for k, v in t.items() if k != "build_index_both"), 3)}
if True:
    from oracle import rjoracle as O
    O.lib().rjo_set_num_threads(16)
    om = [O.Map(m[i].pts, m[i].row_index, m[i].left, m[i].right) for i in range(2)]
    h.sort_pairs(pairs, n)
    got_pairs = pairs.to_host(np.uint32, 2 * n).reshape(-1, 2)
    want_pairs = O.lsi_grid(om[0], om[1], 2048)["eid"]
    ok = np.array_equal(got_pairs, want_pairs)
    for im in range(2):
        want = O.overlay_edge_xsects(om[0], om[1], im, want_pairs, 2048)
        h.overlay_edge_xsects(im, pairs, n, xs[im])
        got = xs[im].to_host(_capi.XSECT_DTYPE, n)
        ok = ok and all(np.array_equal(got[f], want[f]) for f in ("x_num", "y_num", "eid", "mid_point_polygon_id"))
        we = O.pip_grid(om[1 - im], 1 - im, m[im].pts, 2048)
        ok = ok and np.array_equal(fc[im].to_host(np.int32), om[1 - im].face_ids(we))
    out["bit_exact_vs_oracle"] = bool(ok)
print(json.dumps(out))
