#!/usr/bin/env python3
"""Generate the committed sample CDB pair and its answers (BASELINE.json configs[0]: the
reference's own test/dataset sample pair is missing from the snapshot, so a synthetic pair takes
its place, in the same role as test/test_overlay.sh's fixture + answer file).

Answers come from the CPU oracle's -mode=grid restatement (oracle/), cross-checked against brute
force before they are written.     python tests/golden/make_sample_pair.py

ALL THREE ANSWER FILES ARE SELF-GENERATED: lsi_answer.txt, pip_answer.txt and overlay_answer.txt are
what THIS repository's oracle pipeline (oracle/ + tests/overlay_ref.py) says, not output of the
reference.  Byte-identity of the HIP path / the C++ writer with them is self-consistency of two
independent implementations here; it is NOT parity with the reference's own
test/dataset/br_countyXbr_soil_answer.txt, which is absent from the snapshot (.MISSING_LARGE_BLOBS).
Only the LSI predicate, the rational store, calculate_cell and Scaling are pinned by the reference
itself (tests/golden/lsi_ref_vectors.json, scaling_ref_vectors.json: DESIGN.md section 2).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import rjoracle as O  # noqa: E402
from rayjoin_amd import maps, synth  # noqa: E402

D = os.path.join(ROOT, "tests", "golden", "sample_pair")
bbox = (-74.3, 40.4, -73.6, 41.0)
ga = synth.lattice_map(5, 64, 101, bbox)
gb = synth.lattice_map(9, 40, 102, bbox)
maps.write_cdb(os.path.join(D, "map0.cdb"), ga, "%.9f")
maps.write_cdb(os.path.join(D, "map1.cdb"), gb, "%.9f")
# answers are defined on what a loader reads back from the text files
ctx = maps.Context([maps.read_cdb(os.path.join(D, "map0.cdb")), maps.read_cdb(os.path.join(D, "map1.cdb"))]).load()
m0 = O.Map(ctx.maps[0].pts, ctx.maps[0].row_index, ctx.maps[0].left, ctx.maps[0].right)
m1 = O.Map(ctx.maps[1].pts, ctx.maps[1].row_index, ctx.maps[1].left, ctx.maps[1].right)
xs = O.lsi_grid(m0, m1, 2048)
assert np.array_equal(xs["eid"], O.lsi_brute(m0, m1))
with open(os.path.join(D, "lsi_answer.txt"), "w") as f:  # eid0 eid1 x y  (query_exec -output format)
    for r in xs:
        f.write("%d %d %d %d\n" % (r["eid"][0], r["eid"][1], r["x_num"], r["y_num"]))
eids = O.pip_grid(m0, 0, ctx.maps[1].pts, 2048)
assert np.array_equal(eids, O.pip_brute(m0, 1, ctx.maps[1].pts))
faces = m0.face_ids(eids)
with open(os.path.join(D, "pip_answer.txt"), "w") as f:  # closest_eid face_id per vertex of map 1
    for e, fc in zip(eids, faces):
        f.write("%d %d\n" % (e, fc))
# overlay answer (polyover_exec -output), produced by the oracle pipeline + tests/overlay_ref.py
sys.path.insert(0, os.path.join(ROOT, "tests"))
import overlay_ref  # noqa: E402
(nch, nfc), _, _ = overlay_ref.oracle_overlay(O, ctx, os.path.join(D, "overlay_answer.txt"))
print("overlay: %d chains, %d faces" % (nch, nfc))
print("map0: %d edges, map1: %d edges, %d intersections, %d points (%d misses)"
      % (m0.ne, m1.ne, len(xs), len(eids), int((eids == O.MISS).sum())))
