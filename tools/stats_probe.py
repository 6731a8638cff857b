#!/usr/bin/env python3
"""Traversal statistics of one LSI and one PIP query (diagnostic; GPU only)."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayjoin_amd import _capi, maps, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--base", default="USCounty")
ap.add_argument("--query", default="BlockGroup")
ap.add_argument("--scale", type=float, default=1.0)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
ctx = maps.Context([synth.standin(a.base, a.scale), synth.standin(a.query, a.scale)]).load()
b, q = ctx.maps
h = _capi.Handle(0)
h.upload_map(0, b.pts, b.row_index, b.left, b.right)
h.upload_map(1, q.pts, q.row_index, q.left, q.right)
h.build_lbvh(0)
cap = int(0.1 * (b.n_edges + q.n_edges)) + 1024
pairs = h.alloc(8 * cap)
closest = h.alloc(4 * q.n_points)
out = {"n_r": b.n_edges, "n_s": q.n_edges, "n_p": q.n_points, "build_ms": h.last_ms(_capi.RJ_T_BUILD)}
for stats in (0, 1):
    h.set_option("stats", stats)
    for _ in range(a.reps):
        n = h.lsi_query(0, 1, 0, q.n_edges, cap, pairs)
        lsi_ms = h.last_ms(_capi.RJ_T_LSI_KERNEL)
    if stats:
        out["lsi_stats"] = h.last_stats()
    for _ in range(a.reps):
        h.pip_query(0, 1, None, 0, q.n_points, closest, None)
        pip_ms = h.last_ms(_capi.RJ_T_PIP_KERNEL)
    if stats:
        out["pip_stats"] = h.last_stats()
    out["lsi_ms_stats%d" % stats] = lsi_ms
    out["pip_ms_stats%d" % stats] = pip_ms
# per-chunk PIP time: is the cost uniform or concentrated in a few point ranges?
h.set_option("stats", 0)
nchunk = 32
per = q.n_points // nchunk
chunk_ms = []
for i in range(nchunk):
    h.pip_query(0, 1, None, i * per, per, closest, None)
    chunk_ms.append(round(h.last_ms(_capi.RJ_T_PIP_KERNEL), 3))
out["pip_chunk_ms"] = chunk_ms
out["xsects"] = n
c = closest.to_host(np.uint32)
out["pip_miss"] = int((c == 0xFFFFFFFF).sum())
print(json.dumps(out, indent=1))
