"""oracle/pip_double_vs_exact.py (the report profiles/r03_pip_double_vs_exact.txt comes from): its `double` leg must be
the oracle's rule -- the same winners as oracle/rjoracle.c::pip_brute -- and on coordinates small enough for every
product to be exact in double the exact-rational leg must agree with it everywhere."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import pip_double_vs_exact as R  # noqa: E402
from oracle import rjoracle as O  # noqa: E402
from rayjoin_amd import maps, synth  # noqa: E402


def _winners(points, segs, q, exact):
    out = []
    for px, py in points:
        best = None
        for eid, seg in enumerate(segs):
            r = R.evaluate(int(px), int(py), seg, q)
            if r is None:
                continue
            acc, y, s = r[1] if exact else r[0]
            if acc and (best is None or R._better(y, s, eid, best[0], best[1], best[2], q)):
                best = (y, s, eid)
        out.append(best[2] if best else 0xFFFFFFFF)
    return np.array(out, dtype=np.uint32)


def test_small_lattice_is_exact_in_double():
    segs = synth.adversarial_segments(60, 5, 7).reshape(-1, 4)
    pts = synth.adversarial_segments(40, 5, 8).reshape(-1, 2)
    for q in (0, 1):
        c = R.compare(pts, segs, q)
        assert c["pairs_in_x_range"] > 500 and c["decisions_differ"] == 0 and c["winners_differ"] == 0


def test_double_leg_is_the_oracles_rule():
    """chains of a map (edge ids of the map, so the oracle's winners are comparable), at map-like magnitudes"""
    ctx = maps.Context([synth.lattice_map(4, 9, 5), synth.lattice_map(7, 5, 6)]).load()
    base, query = ctx.maps
    m0 = O.Map(base.pts, base.row_index, base.left, base.right)
    segs = base.segments() if hasattr(base, "segments") else None
    if segs is None:
        ri = base.row_index.astype(np.int64)
        segs = np.concatenate([np.concatenate([base.pts[ri[c]:ri[c + 1] - 1], base.pts[ri[c] + 1:ri[c + 1]]], 1)
                               for c in range(len(ri) - 1)])
    pts = np.concatenate([query.pts[:120], base.pts[:60]])          # generic points and points ON base vertices
    want = O.pip_brute(m0, 1, pts)
    assert np.array_equal(_winners(pts, segs, 1, exact=False), want)
