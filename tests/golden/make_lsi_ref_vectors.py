#!/usr/bin/env python3
"""Generate tests/golden/lsi_ref_vectors.json from the REFERENCE's own predicate headers.

Runs only in the authoring container (needs /root/reference): `make -C oracle ref` compiles
src/algo/lsi.h + src/util/rational.h + src/grid/cell.h in place into oracle/_ref/liblsi_ref.so
(see oracle/ref/lsi_ref_driver.cc), and this script evaluates it over a seeded set of segment
pairs.  The JSON holds inputs and expected outputs only (data, not source).

    python tests/golden/make_lsi_ref_vectors.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import rjoracle as O  # noqa: E402

GSIZE = 2048
BIG = (1 << 46) - 1


def cases():
    rng = np.random.default_rng(20241024)
    out = []
    # known answers recorded in SURVEY.md 8c
    out += [([0, 0, 10, 10], [0, 10, 10, 0]), ([0, 0, 10, 7], [0, 10, 10, 0]),
            ([0, 0, 10, 0], [5, 0, 5, 7]), ([5, 0, 5, 7], [0, 0, 10, 0])]
    # tiny lattice: shared endpoints, T-junctions, collinear overlap, duplicates, axis-parallel
    for span in (2, 4, 9):
        p = rng.integers(-span, span + 1, size=(140, 8))
        for r in p:
            a, b = r[:4].tolist(), r[4:].tolist()
            if a[0:2] == a[2:4] or b[0:2] == b[2:4]:
                continue
            out.append((a, b))
    # identical / reversed / sub-segment / touching-at-endpoint families
    for _ in range(40):
        x1, y1, x2, y2 = rng.integers(-50, 51, 4).tolist()
        if (x1, y1) == (x2, y2):
            continue
        out.append(([x1, y1, x2, y2], [x1, y1, x2, y2]))
        out.append(([x1, y1, x2, y2], [x2, y2, x1, y1]))
        out.append(([x1, y1, x2, y2], [x2, y2, x2 + 3, y2 - 7]))
        out.append(([x1, y1, x2, y2], [x1, y1, 2 * x2 - x1, 2 * y2 - y1]))
        mx, my = x1 + x2, y1 + y2  # midpoint of the doubled segment
        out.append(([2 * x1, 2 * y1, 2 * x2, 2 * y2], [mx, my, mx + 5, my + 11]))
        out.append(([mx, my, mx + 5, my + 11], [2 * x1, 2 * y1, 2 * x2, 2 * y2]))
    # general position, map-like magnitudes (edge extent <= 2^39: no int128 wrap)
    for _ in range(150):
        c = rng.integers(-BIG + (1 << 40), BIG - (1 << 40), 2)
        d = rng.integers(-(1 << 38), 1 << 38, 8)
        a = [int(c[0] + d[0]), int(c[1] + d[1]), int(c[0] + d[2]), int(c[1] + d[3])]
        b = [int(c[0] + d[4]), int(c[1] + d[5]), int(c[0] + d[6]), int(c[1] + d[7])]
        out.append((a, b))
    # extremes of the internal range (+-2^46 corners), short edges
    for _ in range(60):
        sx, sy = rng.choice([-1, 1], 2)
        d = rng.integers(0, 1000, 8)
        a = [int(sx * (BIG - d[0])), int(sy * (BIG - d[1])), int(sx * (BIG - d[2])), int(sy * (BIG - d[3]))]
        b = [int(sx * (BIG - d[4])), int(sy * (BIG - d[5])), int(sx * (BIG - d[6])), int(sy * (BIG - d[7]))]
        if a[0:2] == a[2:4] or b[0:2] == b[2:4]:
            continue
        out.append((a, b))
    # long edges: the reference's numx/numy products wrap in int128 here; the wrapped values are
    # still what it stores, so they are golden too
    for _ in range(40):
        v = rng.integers(-BIG, BIG, 8).tolist()
        out.append((v[:4], v[4:]))
    return out


def main():
    R = O.ref_lib()
    if R is None:
        sys.exit("oracle/_ref/liblsi_ref.so is not available (needs /root/reference)")
    vec = []
    for a, b in cases():
        rec = {"s1": a, "s2": b, "hit": O.intersect_test(a, b, "ref")}
        r = O.intersect_point(a, b, GSIZE, "ref")
        assert (r is not None) == bool(rec["hit"])
        if r:
            rec.update(x=[str(r["x"][0]), str(r["x"][1])], y=[str(r["y"][0]), str(r["y"][1])],
                       stored=list(r["stored"]), cell=list(r["cell"]))
        vec.append(rec)
    cells_int = [int(v) for v in np.random.default_rng(7).integers(-BIG - 1, BIG + 1, 64)] + [-BIG - 1, BIG, 0, -1, 1]
    cells = [{"v": v, "g": g, "cell": int(R.ref_cell_of_int(g, v))} for v in cells_int for g in (1, 64, 2048, 15000)]
    cells_f = [float(v) for v in np.random.default_rng(8).uniform(-BIG, BIG, 64)]
    cellsf = [{"v": v, "g": g, "cell": int(R.ref_cell_of_double(g, v))} for v in cells_f for g in (64, 2048, 15000)]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lsi_ref_vectors.json")
    with open(path, "w") as f:
        json.dump({"source": "reference src/algo/lsi.h + src/util/rational.h + src/grid/cell.h compiled on the host "
                             "(oracle/ref/lsi_ref_driver.cc); generator tests/golden/make_lsi_ref_vectors.py",
                   "gsize": GSIZE, "pairs": vec, "cell_of_int": cells, "cell_of_double": cellsf}, f, separators=(",", ":"))
    print("wrote", path, len(vec), "pairs,", sum(v["hit"] for v in vec), "hits")


if __name__ == "__main__":
    main()
