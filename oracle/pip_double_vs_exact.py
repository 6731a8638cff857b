#!/usr/bin/env python3
"""How often does the reference's PIP rule -- which evaluates the ray/edge crossing in `double`
(src/algo/pip.h:53-61, == src/app/pip_lbvh.h:76-88: xsect_y = (double)(-a px - c) / (double) b, then the sign of
(double) py - xsect_y with the simulation-of-simplicity substitutes when it is 0) -- differ from the SAME rule in exact
rational arithmetic?  A report, not an oracle: it documents the semantics the oracle and the HIP kernels restate (and
must reproduce bit for bit, rounding included), on the adversarial fixtures of the parity tests and on map-like
magnitudes.  Pure Python integers and fractions; no reference code is imported, compiled or stood in for.

Per (point, edge) pair that passes the x-range rule (pip.h:44-46) it compares
  * the accept / reject decision (diff_y <= 0 after the substitutes), and
per point the edge that wins (smallest xsect_y, ties by the slope rule pip.h:73-95 restated as in rj_predicates.h).
usage: python oracle/pip_double_vs_exact.py            (test infrastructure: lives beside the oracle)
"""
import os
import sys
from fractions import Fraction

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _edge_terms(x1, y1, x2, y2):
    """a, b of the edge equation a x + b y + c = 0 as map.h:216-226 forms it (b >= 0 after normalisation)"""
    a, b = y1 - y2, x2 - x1
    return (a, b), ((-a, -b) if b < 0 else (a, b))


def _decide(diff, an, bn, q):
    """pip.h:62-71: diff_y == 0 -> substitute -a / a, then -b / b by query map id; accept iff diff_y <= 0"""
    if diff == 0:
        diff = -an if q == 0 else an
        if diff == 0:
            diff = -bn if q == 0 else bn
    return diff <= 0


def evaluate(px, py, seg, q):
    """-> None (x-range rule rejects) or ((accept_double, y_double, slope_double), (accept_exact, y_exact, slope_exact))"""
    x1, y1, x2, y2 = (int(v) for v in seg)
    xmin, xmax = min(x1, x2), max(x1, x2)
    if px < xmin or px > xmax or px == (xmin if q == 0 else xmax):
        return None
    (a, b), (an, bn) = _edge_terms(x1, y1, x2, y2)
    num = a * (x1 - px) + b * y1                 # == -a px - c, exactly (Python integers)
    y_d = float(num) / float(b)                  # int -> float rounds to nearest even, like (double) __int128; IEEE division
    acc_d = _decide(float(py) - y_d, float(an), float(bn), q)
    y_e = Fraction(num, b)
    acc_e = _decide(Fraction(py) - y_e, an, bn, q)
    return (acc_d, y_d, float(an) / float(bn)), (acc_e, y_e, Fraction(an, bn))


def _better(y, slope, eid, by, bslope, beid, q):
    """the total order of rj_predicates.h::pip_better (pip.h:73-95), on floats or on fractions alike"""
    if y != by:
        return y < by
    if q:
        return slope > bslope if slope != bslope else eid < beid
    return slope < bslope if slope != bslope else eid > beid


def compare(points, segs, q):
    """-> dict of counts over all (point, edge) pairs"""
    out = dict(pairs_in_x_range=0, decisions_differ=0, accepted_double=0, points=len(points), winners_differ=0,
               ties_in_double_not_in_exact=0, ties_in_exact=0)
    for px, py in points:
        best_d = best_e = None
        for eid, seg in enumerate(segs):
            r = evaluate(int(px), int(py), seg, q)
            if r is None:
                continue
            (acc_d, y_d, s_d), (acc_e, y_e, s_e) = r
            out["pairs_in_x_range"] += 1
            out["accepted_double"] += acc_d
            out["decisions_differ"] += acc_d != acc_e
            if acc_d:
                if best_d is not None and y_d == best_d[0]:
                    out["ties_in_double_not_in_exact"] += best_e is None or not acc_e or y_e != best_e[0]
                if best_d is None or _better(y_d, s_d, eid, best_d[0], best_d[1], best_d[2], q):
                    best_d = (y_d, s_d, eid)
            if acc_e:
                if best_e is not None and y_e == best_e[0]:
                    out["ties_in_exact"] += 1
                if best_e is None or _better(y_e, s_e, eid, best_e[0], best_e[1], best_e[2], q):
                    best_e = (y_e, s_e, eid)
        out["winners_differ"] += (best_d[2] if best_d else -1) != (best_e[2] if best_e else -1)
    return out


def workloads():
    from rayjoin_amd import maps, synth
    rng = np.random.default_rng(3)
    # 1. the tiny integer lattice of tests/test_gpu_parity.py::test_adversarial_integer_lattice: every value is exact in double
    a = synth.adversarial_segments(160, 6, 91).reshape(-1, 4)
    b = synth.adversarial_segments(160, 6, 141).reshape(-1, 2)
    yield "adversarial lattice, |coord| <= 6 (everything exact in double)", b[:200], a
    # 2. the same lattice in a +-2^46 corner of the scaled range
    a = synth.adversarial_segments(160, 6, 91, True).reshape(-1, 4)
    b = synth.adversarial_segments(160, 6, 141, True).reshape(-1, 2)
    yield "adversarial lattice at the +-2^46 corners", b[:200], a
    # 3. map-like magnitudes: a nested pair -- query vertices ON base vertices and on base edges, coordinates ~2^44
    ctx = maps.Context([synth.standin("USCounty", 0.03), synth.standin("NestedBlockGroup", 0.03)]).load()
    base, query = ctx.maps
    ri = base.row_index.astype(np.int64)
    segs = np.concatenate([np.concatenate([base.pts[ri[c]:ri[c + 1] - 1], base.pts[ri[c] + 1:ri[c + 1]]], 1)
                           for c in range(min(8, len(ri) - 1))])
    on_vertex = segs[rng.integers(0, len(segs), 150), :2]                       # exactly on base vertices
    t = rng.integers(1, 1 << 20, 150)
    s = segs[rng.integers(0, len(segs), 150)]
    near = np.stack([s[:, 0] + (s[:, 2] - s[:, 0]) * t // (1 << 20), s[:, 1] + (s[:, 3] - s[:, 1]) * t // (1 << 20)], 1)  # within a unit of an edge
    tag = "nested pair at 0.03 of full resolution (coordinates ~2^44, -a px - c ~2^75 > 2^53): "
    yield tag + "points exactly ON base vertices", on_vertex, segs
    yield tag + "points within one unit of a base edge", np.concatenate([near, near + [0, 1], near - [0, 1]]), segs
    lo, hi = segs[:, :2].min(0), segs[:, :2].max(0)
    yield tag + "uniform random points of the same box", rng.integers(lo, hi, (400, 2)), segs


def main():
    for name, pts, segs in workloads():
        for q in (0, 1):
            c = compare(pts, segs, q)
            print("%s, query map %d: %d points x %d edges -> %d pairs in x-range, %d accepted (double); accept/reject differs on %d; "
                  "the winning edge differs for %d points; %d ties in double that are not ties exactly, %d exact ties"
                  % (name, q, c["points"], len(segs), c["pairs_in_x_range"], c["accepted_double"], c["decisions_differ"],
                     c["winners_differ"], c["ties_in_double_not_in_exact"], c["ties_in_exact"]))


if __name__ == "__main__":
    main()
