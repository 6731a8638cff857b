#!/usr/bin/env python3
"""Index build time (rj_build_lbvh, device timer) per stand-in map, best of --reps (GPU only)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayjoin_amd import _capi, maps, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--maps", default="USCounty,BlockGroup,WaterBodies,LakesNA")
ap.add_argument("--scale", type=float, default=1.0)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
out = {}
for name in a.maps.split(","):
    g = synth.standin(name, a.scale)
    ctx = maps.Context([g, None]).load()
    m = ctx.maps[0]
    h = _capi.Handle(0)
    h.upload_map(0, m.pts, m.row_index, m.left, m.right)
    import time
    out[name] = {"segments": int(m.n_edges)}
    for order in (1, 0):  # polyline-run leaves (the default; the runs are cut on the device by the first build) and Hilbert leaves
        h.set_option("leaf_order", order)
        ms, wall, stages = [], [], []
        for _ in range(a.reps):
            t0 = time.perf_counter()
            h.build_lbvh(0)
            wall.append((time.perf_counter() - t0) * 1e3)
            ms.append(h.last_ms(_capi.RJ_T_BUILD))
            stages.append(h.last_ms_all())
        names = {"runs": _capi.RJ_T_BUILD_RUNS, "keys": _capi.RJ_T_BUILD_KEYS, "sort": _capi.RJ_T_BUILD_SORT, "leaves": _capi.RJ_T_BUILD_LEAVES, "levels": _capi.RJ_T_BUILD_LEVELS}
        out[name]["leaf_order_%d" % order] = {"used": h.get_option("leaf_order_used0"), "first_build_ms": round(ms[0], 3), "rebuild_ms": round(min(ms[1:] or ms), 3),
                                              "first_wall_ms": round(wall[0], 3), "rebuild_wall_ms": round(min(wall[1:] or wall), 3),
                                              "first_stages_ms": {k: round(stages[0][v], 3) for k, v in names.items()},
                                              "stitch_rounds": h.get_option("stitch_rounds"), "stitch_loop_ends": h.get_option("stitch_loop_ends"),
                                              "runs": h.get_option("leaf_runs0"), "Msegs_per_s_first_build": round(m.n_edges / ms[0] / 1e3, 1),
                                              "slots_per_segment": round(h.get_option("leaf_slots0") / m.n_edges, 3)}
    h.close()
print(json.dumps(out))
