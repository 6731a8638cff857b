#!/usr/bin/env python3
"""Generate tests/golden/scaling_ref_vectors.json from the REFERENCE's own Scaling class
(src/map/scaling.h + src/map/bounding_box.h + src/util/type_traits.h compiled in place by
`make -C oracle ref`, host build, -ffp-contract=off; see oracle/ref/lsi_ref_driver.cc).

Runs only in the authoring container (needs /root/reference).  The JSON holds inputs and expected
outputs only: doubles as C99 hex strings (exact bit patterns), scaled coordinates as integers.

    python tests/golden/make_scaling_ref_vectors.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import rjoracle as O  # noqa: E402

BBOXES = [
    (-179.149, -14.5487, 179.778, 71.3905),      # US-wide, expr/draw/query_lsi/*_lbvh.log:4
    (-179.149, 18.9107, -66.9496, 71.3652),      # CONUS-like
    (-117.958, 35.1402, -69.3684, 64.8991),      # synthetic gaussian runs, expr/draw/scal_lsi_synthetic
    (0.0, 0.0, 1.0, 1.0),
    (-1e-3, -2e-3, 3e-3, 1e-3),                  # tiny extent: the margin dominates
    (1.0e6, 2.0e6, 1.0e6 + 5000.0, 2.0e6 + 7000.0),  # projected coordinates far from the origin
    (-73.98123456789, 40.70123456789, -73.93987654321, 40.80987654321),
]


def main():
    R = O.ref_lib()
    if R is None or not hasattr(R, "ref_scale_points"):
        sys.exit("oracle/_ref/liblsi_ref.so (with Scaling) is not available: needs /root/reference")
    rng = np.random.default_rng(20241024)
    consts = np.zeros(4, dtype=np.int64)
    R.ref_scaling_consts(consts)
    sets = []
    for bb in BBOXES:
        b = np.array(bb, dtype=np.float64)
        n = 150
        xy = np.stack([rng.uniform(bb[0], bb[2], n), rng.uniform(bb[1], bb[3], n)], 1)
        # corners, edges, centre, and neighbours of representable doubles at the corners
        special = [(bb[0], bb[1]), (bb[2], bb[3]), (bb[0], bb[3]), (bb[2], bb[1]),
                   (0.5 * (bb[0] + bb[2]), 0.5 * (bb[1] + bb[3])),
                   (np.nextafter(bb[0], np.inf), np.nextafter(bb[1], np.inf)),
                   (np.nextafter(bb[2], -np.inf), np.nextafter(bb[3], -np.inf))]
        xy = np.ascontiguousarray(np.concatenate([xy, np.array(special)]))
        out = np.zeros(xy.shape, dtype=np.int64)
        R.ref_scale_points(b, xy, len(xy), out)
        # unscale: the scaled points themselves and arbitrary internal coordinates
        ints = np.ascontiguousarray(np.concatenate([out, rng.integers(consts[0], consts[1], (40, 2))]))
        back = np.zeros(ints.shape, dtype=np.float64)
        R.ref_unscale_points(b, ints, len(ints), back)
        sets.append({"bb": [float(v).hex() for v in bb],
                     "xy": [[float(p[0]).hex(), float(p[1]).hex()] for p in xy],
                     "scaled": out.tolist(),
                     "ints": ints.tolist(),
                     "unscaled": [[float(p[0]).hex(), float(p[1]).hex()] for p in back]})
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scaling_ref_vectors.json")
    with open(path, "w") as f:
        json.dump({"source": "reference src/map/scaling.h (Scaling<double,int64_t,17>) compiled on the host, "
                             "-ffp-contract=off (oracle/ref/lsi_ref_driver.cc); generator "
                             "tests/golden/make_scaling_ref_vectors.py",
                   "internal_min": int(consts[0]), "internal_max": int(consts[1]), "internal_range": int(consts[2]),
                   "sizeof_scaling": int(consts[3]), "sets": sets}, f, separators=(",", ":"))
    print("wrote", path, sum(len(s["xy"]) for s in sets), "points in", len(sets), "bounding boxes")


if __name__ == "__main__":
    main()
