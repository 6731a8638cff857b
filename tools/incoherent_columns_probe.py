#!/usr/bin/env python3
"""A spatially incoherent point set (the reference's GeneratePIPQueries, run_query.cu:147-167: uniform random points; or a
shuffled vertex set) over a lattice base map: the tree walk through the Morton permutation against the column index
("pip_columns" 1).  Per setting: the PIP query's time once the permutation is cached, the one-time sort, the index build.
usage: incoherent_columns_probe.py [--base USCounty --n 4194304]"""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayjoin_amd import _capi, maps, synth
ap = argparse.ArgumentParser()
ap.add_argument("--base", default="USCounty"); ap.add_argument("--n", type=int, default=1 << 22)
a = ap.parse_args()
ctx = maps.Context([synth.standin(a.base), synth.lattice_map(8, 8, 3)]).load()
b, q = ctx.maps
rng = np.random.default_rng(1)
sets = {"uniform_random": synth.generate_pip_queries(ctx.bb, ctx.scaling, a.n, 3)}
import time
for columns in (0, 1, -1):   # (-1: auto -- since round 6 the first incoherent query of >= 2^22 points builds the index)
    h = _capi.Handle(0)
    h.upload_map(0, b.pts, b.row_index, b.left, b.right); h.upload_map(1, q.pts, q.row_index, q.left, q.right)
    h.set_option("pip_columns", columns)
    h.build_lbvh(0)
    build = h.last_ms(_capi.RJ_T_BUILD)
    for tag, pts in sets.items():
        d = h.alloc(16 * len(pts)).from_host(pts); c = h.alloc(4 * len(pts)); f = h.alloc(4 * len(pts))
        ms, wall = [], []
        for _ in range(6):
            t0 = time.perf_counter()
            h.pip_query(0, 1, d, 0, len(pts), c, f)
            wall.append((time.perf_counter() - t0) * 1e3)
            ms.append(h.last_ms(_capi.RJ_T_PIP_KERNEL))
        e = c.to_host(np.uint32)
        print(json.dumps({"base": a.base, "points": len(pts), "set": tag, "pip_columns": columns, "first_build_ms": round(build, 3),
                          "first_query_wall_ms": round(wall[0], 3), "query_wall_ms": round(min(wall[2:]), 3), "first_query_ms": round(ms[0], 3), "columns_why": h.get_plan()["index"][0]["columns_why"][:60], "query_ms": round(min(ms[2:]), 4), "ordered": h.get_option("query_last_ordered"),
                          "first_pass": h.get_plan()["pip"]["first_pass"]["kernel"], "eid_sum": int(e[e != 0xFFFFFFFF].astype(np.uint64).sum())}))
        d.free(); c.free(); f.free()
    h.close()
