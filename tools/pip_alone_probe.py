#!/usr/bin/env python3
"""The PIP query of a pair alone on the chip, synchronous, a few times: medians of the first-pass timer (RJ_T_PIP_WALK)
and of the whole query (RJ_T_PIP_KERNEL).  Runs with any revision's library (RAYJOIN_AMD_LIB): for same-box A/B of
first-pass variants.  usage: pip_alone_probe.py [--base WaterBodiesLike --query BlockGroup --reps 9]"""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayjoin_amd import _capi, maps, synth
ap = argparse.ArgumentParser()
ap.add_argument("--base", default="WaterBodiesLike"); ap.add_argument("--query", default="BlockGroup"); ap.add_argument("--reps", type=int, default=9)
ap.add_argument("--columns", type=int, default=None, help="rj_set_option pip_columns before the build (1: force the column index)")
ap.add_argument("--strip-shift", type=int, default=0, help="debug option strip_shift for the build (0: by the map)")
ap.add_argument("--query-order", type=int, default=None, help="rj_set_option query_order (2: always through the Morton permutation)")
ap.add_argument("--strip-major", type=int, default=0, help="debug option query_key_strips: a re-ordered set over a column index is sorted strip-major")
a = ap.parse_args()
ctx = maps.Context([synth.standin(a.base), synth.standin(a.query)]).load()
b, q = ctx.maps
h = _capi.Handle(0)
h.upload_map(0, b.pts, b.row_index, b.left, b.right); h.upload_map(1, q.pts, q.row_index, q.left, q.right)
if a.columns is not None:
    h.set_option("pip_columns", a.columns)
if a.strip_shift:
    h.set_debug_option("strip_shift", a.strip_shift)
h.build_lbvh(0)
if a.query_order is not None:
    h.set_option("query_order", a.query_order)
if a.strip_major:
    h.set_debug_option("query_key_strips", 1)
closest = h.alloc(4 * q.n_points); faces = h.alloc(4 * q.n_points)
w, k = [], []
for _ in range(a.reps):
    h.pip_query(0, 1, None, 0, q.n_points, closest, faces)
    w.append(h.last_ms(_capi.RJ_T_PIP_WALK)); k.append(h.last_ms(_capi.RJ_T_PIP_KERNEL))
e = closest.to_host(np.uint32)[:q.n_points]
order_ms = h.last_ms(_capi.RJ_T_ORDER) if a.query_order == 2 else None
print(json.dumps({"strip_major": a.strip_major, "query_order": a.query_order, "strip_shift": h.get_option("pip_column_shift0"), "entries": h.get_option("pip_column_entries0"), "order_ms_last": order_ms, "ordered": h.get_option("query_last_ordered"), "lib": os.environ.get("RAYJOIN_AMD_LIB", "tree"), "pair": a.base + " x " + a.query, "first_pass_ms": round(float(np.median(w[2:])), 4),
                  "query_ms": round(float(np.median(k[2:])), 4), "hits": int((e != 0xFFFFFFFF).sum()), "eid_sum": int(e[e != 0xFFFFFFFF].astype(np.uint64).sum())}))
