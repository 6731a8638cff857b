#!/usr/bin/env python3
"""More seeds of tests/test_gpu_fuzz.py's generator, by hand (GPU box):  python tests/fuzz_more.py [first_seed [count]]
Beside that test's loop it draws the round-6 knobs at random -- the second order of steep leaf blocks on / off, one or two
query segments per lane, the column index auto / off / forced, and a small "lazy_columns_min" so that an incoherent vertex
set builds the index at its first query -- and checks LSI pairs, closest edges and face ids against the brute-force oracle.
Test infrastructure (it imports oracle/): not collected by pytest, not part of the product."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import rjoracle as oracle  # noqa: E402
from rayjoin_amd import _capi, maps  # noqa: E402
from test_gpu_fuzz import _maps  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
oracle.lib().rjo_set_num_threads(16)
done = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    for _ in range(3):
        ctx = maps.Context([_maps(rng), _maps(rng)]).load()
        m = ctx.maps
        om = [oracle.Map(x.pts, x.row_index, x.left, x.right) for x in m]
        want_pairs = oracle.lsi_brute(om[0], om[1])
        h = _capi.Handle(0)
        try:
            for i in (0, 1):
                h.upload_map(i, m[i].pts, m[i].row_index, m[i].left, m[i].right)
            cap = max(1024, 2 * len(want_pairs))
            pairs = h.alloc(8 * cap)
            for base in (0, 1):
                q = m[1 - base]
                want = oracle.pip_brute(om[base], 1 - base, q.pts)
                closest, faces = h.alloc(4 * max(1, q.n_points)), h.alloc(4 * max(1, q.n_points))
                knobs = {"leaf_ysort": int(rng.integers(0, 2)), "lsi_segments": int(rng.integers(1, 3)), "pip_columns": int(rng.integers(-1, 2)),
                         "pip_walk_points": int(rng.integers(1, 3))}
                for k, v in knobs.items():
                    h.set_option(k, v)
                h.set_debug_option("lazy_columns_min", int(rng.integers(0, 2)) * 64)
                h.build_lbvh(base)
                what = (seed, base, knobs, m[0].n_edges, m[1].n_edges)
                n = h.lsi_query(base, 1 - base, 0, q.n_edges, cap, pairs)
                h.sort_pairs(pairs, n)
                got = pairs.to_host(np.uint32, 2 * n).reshape(-1, 2)
                assert np.array_equal(got, want_pairs), what
                for rep in range(2):   # (the second query runs on whatever the first one built)
                    h.pip_query(base, 1 - base, None, 0, q.n_points, closest, faces)
                    assert np.array_equal(closest.to_host(np.uint32)[:q.n_points], want), (what, rep)
                    assert np.array_equal(faces.to_host(np.int32)[:q.n_points], om[base].face_ids(want)), (what, rep)
                done += 1
        finally:
            h.close()
    print("seed %d ok (%d index / query rounds so far)" % (seed, done), flush=True)
print("all %d seeds ok" % count)
