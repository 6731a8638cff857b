#!/usr/bin/env python3
"""A step of a join launched kernel by kernel vs replayed as one captured hipGraph (rj_graph_*): wall time per step,
equality of the results (GPU only)."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayjoin_amd import _capi, maps, synth, dist as rjd
ap = argparse.ArgumentParser()
ap.add_argument("--base", default="USCounty"); ap.add_argument("--query", default="BlockGroup")
ap.add_argument("--scale", type=float, default=1.0); ap.add_argument("--steps", type=int, default=30); ap.add_argument("--shards", type=int, default=1)
a = ap.parse_args()
ctx = maps.Context([synth.standin(a.base, a.scale), synth.standin(a.query, a.scale)]).load()
b, q = ctx.maps
h = _capi.Handle(0)
h.upload_map(0, b.pts, b.row_index, b.left, b.right); h.upload_map(1, q.pts, q.row_index, q.left, q.right)
h.build_lbvh(0)
sh = rjd.shard_of(q, a.shards, 0)
(e0, e1), (p0, p1) = sh["eids"], sh["points"]
cap = int(0.1 * (b.n_edges + q.n_edges)) + 1024
pairs = h.alloc(8 * cap); xs = h.alloc(48 * cap); closest = h.alloc(4 * q.n_points); faces = h.alloc(4 * q.n_points)
h.lsi_query(0, 1, e0, e1, cap, pairs); h.pip_query(0, 1, None, p0, p1 - p0, closest, faces)
h.set_option("pip_concurrent", 2)
def enqueue():
    h.lsi_query_async(0, 1, e0, e1, cap, pairs)
    early = h.get_option("pip_schedule") in (1, 2)
    if early:
        h.pip_query(0, 1, None, p0, p1 - p0, closest, faces, sync=False)
    h.lsi_points_async(pairs, cap, xs)
    if not early:
        h.pip_query(0, 1, None, p0, p1 - p0, closest, faces, sync=False)
def plain():
    enqueue(); n = h.lsi_query_finish(cap); h.sync(); return n
for _ in range(6):
    n0 = plain()
ts = []
for _ in range(a.steps):
    t0 = time.perf_counter(); plain(); ts.append(time.perf_counter() - t0)
ref = (pairs.to_host(np.uint32, 2 * n0).copy(), closest.to_host(np.uint32).copy(), faces.to_host(np.int32).copy(), xs.to_host(np.int64, 6 * n0).copy())
h.graph_begin(0); enqueue(); h.graph_end()
def replay():
    h.graph_launch(0); n = h.graph_lsi_count(cap); h.sync(); return n
for _ in range(3):
    n1 = replay()
tg = []
for _ in range(a.steps):
    t0 = time.perf_counter(); replay(); tg.append(time.perf_counter() - t0)
got = (pairs.to_host(np.uint32, 2 * n1).copy(), closest.to_host(np.uint32).copy(), faces.to_host(np.int32).copy(), xs.to_host(np.int64, 6 * n1).copy())
def canon(p, x):
    o = np.lexsort((p.reshape(-1, 2)[:, 1], p.reshape(-1, 2)[:, 0])); return p.reshape(-1, 2)[o], x.reshape(-1, 6)[o]
same = n0 == n1 and all(np.array_equal(u, v) for u, v in zip(canon(ref[0], ref[3]), canon(got[0], got[3]))) and np.array_equal(ref[1], got[1]) and np.array_equal(ref[2], got[2])
print(json.dumps({"pair": "%s x %s" % (a.base, a.query), "shards": a.shards, "schedule": h.get_option("pip_schedule"), "step_ms_launches": round(float(np.median(ts)) * 1e3, 4),
                  "step_ms_graph": round(float(np.median(tg)) * 1e3, 4), "same_results": bool(same), "intersections": int(n1),
                  "kernel_ms_in_graph": {"lsi": round(h.last_ms(_capi.RJ_T_LSI_KERNEL), 4), "pip": round(h.last_ms(_capi.RJ_T_PIP_KERNEL), 4)}}))
