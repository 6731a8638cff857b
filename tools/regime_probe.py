#!/usr/bin/env python3
"""One map pair's query kernels, each ALONE on a grid of a given size (GPU only) -- what a rocprofv3 counter pass can see
of a schedule in which two kernels share the chip: the profiler serialises kernels, so `k_lsi2` on its 512-block share
and `k_pip_walk2` on its 1 536 are profiled one after the other on exactly those grids ("max_blocks" caps a launch).
Without --lsi-blocks / --pip-blocks: the full grids (the ring-shaped pairs, whose counters bench.py quotes per pair).
usage: regime_probe.py [--base USCounty --query BlockGroup] [--lsi-blocks 512 --pip-blocks 1536] [--reps 4]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayjoin_amd import _capi, maps, synth
ap = argparse.ArgumentParser()
ap.add_argument("--base", default="USCounty"); ap.add_argument("--query", default="BlockGroup")
ap.add_argument("--lsi-blocks", type=int, default=0); ap.add_argument("--pip-blocks", type=int, default=0)
ap.add_argument("--reps", type=int, default=4)
a = ap.parse_args()
ctx = maps.Context([synth.standin(a.base), synth.standin(a.query)]).load()
b, q = ctx.maps
h = _capi.Handle(0)
h.upload_map(0, b.pts, b.row_index, b.left, b.right); h.upload_map(1, q.pts, q.row_index, q.left, q.right)
h.build_lbvh(0)
cap = int(0.1 * (b.n_edges + q.n_edges)) + 1024
pairs = h.alloc(8 * cap); xs = h.alloc(48 * cap); closest = h.alloc(4 * q.n_points); faces = h.alloc(4 * q.n_points)
big = 1 << 20
ms = {"lsi": [], "records": [], "pip_first_pass": [], "pip": []}
for r in range(a.reps + 1):
    h.set_debug_option("max_blocks", a.lsi_blocks or big)
    n = h.lsi_query(0, 1, 0, q.n_edges, cap, pairs)
    h.set_debug_option("max_blocks", big)
    h.lsi_points(pairs, n, xs)
    h.set_debug_option("max_blocks", a.pip_blocks or big)
    h.pip_query(0, 1, None, 0, q.n_points, closest, faces)
    if r:  # (the first round pays first touches)
        ms["lsi"].append(h.last_ms(_capi.RJ_T_LSI_KERNEL)); ms["records"].append(h.last_ms(_capi.RJ_T_LSI_POINTS))
        ms["pip"].append(h.last_ms(_capi.RJ_T_PIP_KERNEL))
        if h.get_option("pip_last_passes") == 3:
            ms["pip_first_pass"].append(h.last_ms(_capi.RJ_T_PIP_WALK))
plan = h.get_plan()
print(json.dumps({"pair": "%s x %s" % (a.base, a.query), "intersections": n, "lsi": plan["lsi"], "pip_first_pass": plan["pip"]["first_pass"],
                  "ms_min": {k: round(min(v), 4) for k, v in ms.items() if v}}))
