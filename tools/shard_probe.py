#!/usr/bin/env python3
"""1/N shard of the query map on one GPU: LSI / PIP kernel ms for chunk_groups x group_lanes
settings (the strong-scaling floor of a small shard is scheduling granularity, not throughput)."""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayjoin_amd import _capi, maps, synth, dist as rjd
ap = argparse.ArgumentParser()
ap.add_argument("--base", default="USCounty"); ap.add_argument("--query", default="BlockGroup")
ap.add_argument("--shards", type=int, default=8); ap.add_argument("--reps", type=int, default=7)
ap.add_argument("--chunks", default="1,2,4"); ap.add_argument("--lanes", default="0,64,32,16")
a = ap.parse_args()
ctx = maps.Context([synth.standin(a.base), synth.standin(a.query)]).load()
b, q = ctx.maps
h = _capi.Handle(0)
h.upload_map(0, b.pts, b.row_index, b.left, b.right); h.upload_map(1, q.pts, q.row_index, q.left, q.right)
h.build_lbvh(0)
sh = rjd.shard_of(q, a.shards, 0)
(e0, e1), (p0, p1) = sh["eids"], sh["points"]
cap = int(0.1 * (b.n_edges + q.n_edges)) + 1024
pairs = h.alloc(8 * cap); closest = h.alloc(4 * q.n_points); faces = h.alloc(4 * q.n_points)
for c in a.chunks.split(","):
    for gl in a.lanes.split(","):
        h.set_debug_option("chunk_groups", int(c)); h.set_debug_option("group_lanes", int(gl))
        l, p = [], []
        for _ in range(a.reps):
            n = h.lsi_query(0, 1, e0, e1, cap, pairs); l.append(h.last_ms(_capi.RJ_T_LSI_KERNEL))
            h.pip_query(0, 1, None, p0, p1 - p0, closest, faces); p.append(h.last_ms(_capi.RJ_T_PIP_KERNEL))
        print(json.dumps({"shards": a.shards, "chunk_groups": int(c), "group_lanes": int(gl), "lsi_ms": round(float(np.median(l)), 4),
                          "pip_ms": round(float(np.median(p)), 4), "segs": e1 - e0, "points": p1 - p0, "n": n}), flush=True)
