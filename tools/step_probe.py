#!/usr/bin/env python3
"""bench.py's synchronised step with nothing but the round-1 entry points (so that ANY revision's library runs it:
RAYJOIN_AMD_LIB=rayjoin_amd/variants/librj_<name>.so): "pip_concurrent" 2, eight warm-up pairs, then the median and mean
of `--steps` steps and the stage timers of the last one.  usage: step_probe.py [--base USCounty --query BlockGroup --steps 40]"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayjoin_amd import _capi, maps, synth
ap = argparse.ArgumentParser()
ap.add_argument("--base", default="USCounty"); ap.add_argument("--query", default="BlockGroup"); ap.add_argument("--steps", type=int, default=40)
a = ap.parse_args()
ctx = maps.Context([synth.standin(a.base), synth.standin(a.query)]).load()
b, q = ctx.maps
h = _capi.Handle(0)
h.upload_map(0, b.pts, b.row_index, b.left, b.right); h.upload_map(1, q.pts, q.row_index, q.left, q.right)
h.build_lbvh(0)
cap = int(0.1 * (b.n_edges + q.n_edges))
pairs = h.alloc(8 * cap); xs = h.alloc(48 * cap); closest = h.alloc(4 * q.n_points); faces = h.alloc(4 * q.n_points)
h.set_option("pip_concurrent", 2)
def step():
    early = h.get_option("pip_schedule") in (1, 2)
    h.lsi_query_async(0, 1, 0, q.n_edges, cap, pairs)
    if early:
        h.pip_query(0, 1, None, 0, q.n_points, closest, faces, sync=False)
    h.lsi_points_async(pairs, cap, xs)
    if not early:
        h.pip_query(0, 1, None, 0, q.n_points, closest, faces, sync=False)
    n = h.lsi_query_finish(cap)
    h.sync()
    return n
for _ in range(8):
    n = step()
ts = []
for _ in range(a.steps):
    t0 = time.perf_counter(); step(); ts.append(time.perf_counter() - t0)
print(json.dumps({"lib": os.environ.get("RAYJOIN_AMD_LIB", "tree"), "pair": "%s x %s" % (a.base, a.query), "intersections": n,
                  "step_ms_median": round(float(np.median(ts)) * 1e3, 4), "step_ms_mean": round(float(np.mean(ts)) * 1e3, 4),
                  "schedule": h.get_option("pip_schedule"), "lsi_k": round(h.last_ms(_capi.RJ_T_LSI_KERNEL), 4), "points_k": round(h.last_ms(_capi.RJ_T_LSI_POINTS), 4),
                  "walk_k": round(h.last_ms(_capi.RJ_T_PIP_WALK), 4), "pip_k": round(h.last_ms(_capi.RJ_T_PIP_KERNEL), 4)}))
