#!/usr/bin/env python3
"""1/N shard step (LSI + records + PIP) wall time on one GPU for combinations of "pip_concurrent" and
"max_blocks": do the two kernels of a small shard overlap better when neither fills the chip alone?"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayjoin_amd import _capi, maps, synth, dist as rjd
ap = argparse.ArgumentParser()
ap.add_argument("--base", default="USCounty"); ap.add_argument("--query", default="BlockGroup")
ap.add_argument("--shards", type=int, default=8); ap.add_argument("--reps", type=int, default=30)
a = ap.parse_args()
ctx = maps.Context([synth.standin(a.base), synth.standin(a.query)]).load()
b, q = ctx.maps
h = _capi.Handle(0)
h.upload_map(0, b.pts, b.row_index, b.left, b.right); h.upload_map(1, q.pts, q.row_index, q.left, q.right)
h.build_lbvh(0)
sh = rjd.shard_of(q, a.shards, 0)
(e0, e1), (p0, p1) = sh["eids"], sh["points"]
cap = int(0.1 * (b.n_edges + q.n_edges)) + 1024
pairs = h.alloc(8 * cap); xs = h.alloc(48 * cap); closest = h.alloc(4 * q.n_points); faces = h.alloc(4 * q.n_points)
for conc in (0, 1):
    for mb in (1 << 20, 1024, 768, 512):
        h.set_option("pip_concurrent", conc); h.set_debug_option("max_blocks", mb)
        ts = []
        for r in range(a.reps + 5):
            t0 = time.perf_counter()
            h.lsi_query_async(0, 1, e0, e1, cap, pairs)
            h.lsi_points_async(pairs, cap, xs)
            h.pip_query(0, 1, None, p0, p1 - p0, closest, faces, sync=False)
            n = h.lsi_query_finish(cap)
            h.sync()
            ts.append(time.perf_counter() - t0)
        print(json.dumps({"shards": a.shards, "pip_concurrent": conc, "max_blocks": mb, "step_ms": round(float(np.median(ts[5:])) * 1e3, 4),
                          "lsi_k": round(h.last_ms(_capi.RJ_T_LSI_KERNEL), 4), "pip_k": round(h.last_ms(_capi.RJ_T_PIP_KERNEL), 4)}), flush=True)
