#!/usr/bin/env python3
"""Two-pass PIP (k_pip_walk + k_pip over the rest) vs k_pip alone on a stand-in pair (GPU only):
kernel ms of each pass, how many points the walk left over, equality of the results."""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayjoin_amd import _capi, maps, synth
ap = argparse.ArgumentParser()
ap.add_argument("--base", default="USCounty"); ap.add_argument("--query", default="BlockGroup")
ap.add_argument("--scale", type=float, default=1.0); ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--max-blocks", default=""); ap.add_argument("--no-stats", action="store_true")
a = ap.parse_args()
ctx = maps.Context([synth.standin(a.base, a.scale), synth.standin(a.query, a.scale)]).load()
b, q = ctx.maps
h = _capi.Handle(0)
h.upload_map(0, b.pts, b.row_index, b.left, b.right); h.upload_map(1, q.pts, q.row_index, q.left, q.right)
h.build_lbvh(0)
out = {}
res = {}
for mode in (0, 2):
    h.set_option("pip_walk", mode)
    closest = h.alloc(4 * q.n_points); face = h.alloc(4 * q.n_points)
    for mb in ([int(v) for v in a.max_blocks.split(",")] if a.max_blocks else [1 << 20]):
        h.set_debug_option("max_blocks", mb)
        tot, walk = [], []
        for _ in range(a.reps):
            h.pip_query(0, 1, None, 0, q.n_points, closest, face)
            tot.append(h.last_ms(_capi.RJ_T_PIP_KERNEL))
            if mode:
                walk.append(h.last_ms(_capi.RJ_T_PIP_WALK))
        print(json.dumps({"pip_walk": mode, "max_blocks": mb, "pip_ms": round(float(np.median(tot)), 4),
                          "walk_ms": round(float(np.median(walk)), 4) if walk else None,
                          "rest": h.get_option("pip_rest") if mode else None, "points": q.n_points}), flush=True)
    h.set_debug_option("max_blocks", 1 << 20)
    res[mode] = (closest.to_host(np.uint32), face.to_host(np.int32))
print(json.dumps({"equal_eids": bool(np.array_equal(res[0][0], res[2][0])), "equal_faces": bool(np.array_equal(res[0][1], res[2][1]))}))
# instrumented kernels: visit counts and cycle stamps of k_pip and of k_pip_walk (sums over waves)
ngroups = (q.n_points + 63) // 64
for mb in (() if a.no_stats else (256, 1 << 20)):
    h.set_debug_option("max_blocks", mb)
    for mode in (0, 2):
        h.set_option("pip_walk", mode); h.set_option("stats", 1)
        h.pip_query(0, 1, None, 0, q.n_points, closest, face)
        st = h.last_stats()
        h.set_option("stats", 0)
        print(json.dumps({"max_blocks": mb, "pip_walk": mode, "per_group": {k: round(v / ngroups, 2) for k, v in st.items()}}), flush=True)
