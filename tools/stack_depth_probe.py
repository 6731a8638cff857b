#!/usr/bin/env python3
"""How deep does the walk's stack get?  (GPU only.)  The instrumented one-point walk (k_pip_walk<STATS>, "pip_walk" 2 +
"stats" 1) reports the deepest stack any 64-point group reached on each stand-in pair: what kWalkStack
(rj_kernels.hip) is sized by -- a group that would need more leaves the walk for k_pip's worst-case stack."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayjoin_amd import _capi, maps, synth
PAIRS = [("USCounty", "BlockGroup"), ("USCounty", "NestedBlockGroup"), ("WaterBodies", "BlockGroup"), ("LakesNA", "ParksNA"),
         ("Gaussian5M", "Gaussian1M"), ("WaterBodiesLike", "BlockGroup"), ("LakesLike", "ParksLike")]
for base, query in PAIRS:
    ctx = maps.Context([synth.standin(base), synth.standin(query)]).load()
    b, q = ctx.maps
    h = _capi.Handle(0)
    h.upload_map(0, b.pts, b.row_index, b.left, b.right); h.upload_map(1, q.pts, q.row_index, q.left, q.right)
    h.set_option("pip_columns", 0)  # (the tree walk on the ring maps too)
    h.build_lbvh(0)
    closest = h.alloc(4 * q.n_points)
    h.set_option("pip_walk", 2); h.set_option("stats", 1)
    h.pip_query(0, 1, None, 0, q.n_points, closest, None); st = h.last_stats()
    h.set_option("stats", 0)
    h.pip_query(0, 1, None, 0, q.n_points, closest, None)
    h.pip_query(0, 1, None, 0, q.n_points, closest, None)
    print(json.dumps({"pair": "%s x %s" % (base, query), "groups": (q.n_points + 63) // 64,
                      "deepest_stack": st["walk_stack_max"], "left_the_walk_or_overflowed_lists": h.get_option("pip_rest"),
                      "walk_points": h.get_option("pip_last_walk_points")}), flush=True)
    h.close()
