#!/usr/bin/env python3
"""Kernel time vs query-set size (first n edges / points of the query map): exposes fixed costs
per launch that limit strong scaling (GPU only)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayjoin_amd import _capi, maps, synth  # noqa: E402

ctx = maps.Context([synth.standin("USCounty"), synth.standin("BlockGroup")]).load()
b, q = ctx.maps
h = _capi.Handle(0)
h.upload_map(0, b.pts, b.row_index, b.left, b.right)
h.upload_map(1, q.pts, q.row_index, q.left, q.right)
h.build_lbvh(0)
cap = int(0.1 * (b.n_edges + q.n_edges))
pairs = h.alloc(8 * cap)
closest = h.alloc(4 * q.n_points)
out = []
for n in (64, 4096, 65536, 1 << 20, 3600000, 14400000, q.n_edges):
    ls, ps = [], []
    for _ in range(7):
        h.lsi_query(0, 1, 0, n, cap, pairs)
        ls.append(h.last_ms(_capi.RJ_T_LSI_KERNEL))
        h.pip_query(0, 1, None, 0, n, closest, None)
        ps.append(h.last_ms(_capi.RJ_T_PIP_KERNEL))
    out.append({"n": n, "lsi_ms": round(min(ls), 4), "pip_ms": round(min(ps), 4)})
print(json.dumps(out))
