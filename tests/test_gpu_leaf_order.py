"""Polyline-run leaves vs Hilbert-neighbour leaves (rj_set_option "leaf_order", SURVEY 8f-3: the reference's RT grouping,
src/rt/primitive.h:120-260): the index is a different tree, every result is the same -- against the oracle, for both
map roles, on chains longer than a leaf (cut into pieces), short chains that continue each other through junctions
(stitched into polylines, in file order or not) and isolated polygons (the fall-back to Hilbert leaves)."""
import numpy as np
import pytest

from rayjoin_amd import _capi, maps, synth

pytestmark = pytest.mark.gpu


def _omap(oracle, m):
    return oracle.Map(m.pts, m.row_index, m.left, m.right)


def _cut_chains(pg, piece):
    """every chain of `pg` cut into chains of <= `piece` edges that follow each other in the file, each starting on
    the point the one before ended on (what a map digitised junction to junction looks like)"""
    pts, rows, chains = [], [0], []
    for c in range(pg.n_chains):
        a, b = int(pg.row_index[c]), int(pg.row_index[c + 1])
        P = pg.points[a:b]
        for s0 in range(0, len(P) - 1, piece):
            part = P[s0:s0 + piece + 1]
            pts.append(part)
            rows.append(rows[-1] + len(part))
            chains.append((len(chains), rows[-2], rows[-1] - 1, pg.chains[c, 3], pg.chains[c, 4]))
    return maps.PlanarGraph(np.array(chains, dtype=np.int64), np.array(rows, dtype=np.uint32), np.concatenate(pts))


def _run(h, base, q, cap):
    h.build_lbvh(base)
    pairs = h.alloc(8 * cap)
    n = h.lsi_query(base, 1 - base, 0, q.n_edges, cap, pairs)
    h.sort_pairs(pairs, n)
    closest = h.alloc(4 * q.n_points)
    face = h.alloc(4 * q.n_points)
    h.pip_query(base, 1 - base, None, 0, q.n_points, closest, face)
    return pairs.to_host(np.uint32, 2 * n).reshape(-1, 2).copy(), closest.to_host(np.uint32).copy(), face.to_host(np.int32).copy()


@pytest.mark.parametrize("shape", ["long", "short_rows", "mixed"])
def test_both_leaf_orders_equal_the_oracle(oracle, shape):
    if shape == "long":      # 150- and 70-edge chains: pieces of 50 and 35
        g = [synth.lattice_map(7, 150, 21), synth.lattice_map(16, 70, 22)]
    elif shape == "short_rows":  # 15- and 7-edge chains that continue each other: packed 4 / 9 to a leaf
        g = [_cut_chains(synth.lattice_map(7, 150, 23), 15), _cut_chains(synth.lattice_map(14, 70, 24), 7)]
    else:                    # long chains in one map; 7-edge chains in lattice file order (horizontal / vertical alternating) in the other:
        g = [synth.lattice_map(5, 200, 25), synth.lattice_map(30, 7, 26)]  # stitched through the lattice nodes into rows and columns
    ctx = maps.Context(g).load()
    m = ctx.maps
    om = [_omap(oracle, m[0]), _omap(oracle, m[1])]
    want_pairs = oracle.lsi_grid(om[0], om[1], 256)["eid"]
    h = _capi.Handle(0)
    try:
        for i in (0, 1):
            h.upload_map(i, m[i].pts, m[i].row_index, m[i].left, m[i].right)
        used = {}
        for order in (1, 0):
            h.set_option("leaf_order", order)
            for base in (0, 1):
                q = m[1 - base]
                pairs, closest, face = _run(h, base, q, 8 * len(want_pairs) + 1024)
                used[(order, base)] = h.get_option("leaf_order_used%d" % base)
                assert np.array_equal(pairs, want_pairs), (shape, order, base)
                want_e = oracle.pip_grid(om[base], base, q.pts, 256)
                assert np.array_equal(closest, want_e), (shape, order, base)
                assert np.array_equal(face, om[base].face_ids(want_e)), (shape, order, base)
        assert all(used[(0, b)] == 0 for b in (0, 1))
        assert used[(1, 0)] == 1  # (the chain-run trees really were built)
        # every one of these maps stitches into long polylines: leaves nearly full, whatever the chain length
        assert used[(1, 1)] == 1
        for i in (0, 1):
            assert 1.0 <= h.get_option("leaf_slots%d" % i) / m[i].n_edges <= 1.35, (shape, i)
    finally:
        h.close()


def test_short_rings_share_leaves(oracle):
    """Polygons of a few edges in random file order (the reference's gaussian workload) and lake-shaped rings: a leaf per
    ring would be mostly padding (rounds 2-3 fell back to Hilbert leaves there), a leaf of FILE neighbours would span the
    map -- neighbouring rings of the Hilbert-sorted order share a leaf instead (k_pack_runs): polyline-run leaves, nearly
    full, every result the oracle's; with sharing forbidden ("debug_pack_solo" 1) the build still falls back."""
    for g, what in ((synth.gaussian_polygons(4000, 5), "gaussian"), (synth.ring_map(3000, 33000, 7), "rings")):
        ctx = maps.Context([g, synth.lattice_map(8, 20, 27)]).load()
        m = ctx.maps
        om = [_omap(oracle, m[0]), _omap(oracle, m[1])]
        want_pairs = oracle.lsi_grid(om[0], om[1], 128)["eid"]
        want_e = oracle.pip_grid(om[0], 0, m[1].pts, 128)
        h = _capi.Handle(0)
        try:
            for i in (0, 1):
                h.upload_map(i, m[i].pts, m[i].row_index, m[i].left, m[i].right)
            # (sharing whatever the gaps, runs of up to 48 edges -- nearly full leaves; the same with the default run rule;
            #  the default rule -- a shared leaf at most 4 x as large as
            #  what it holds: full where the rings are dense, one ring per leaf or the Hilbert fall-back where they are
            #  far apart; sharing forbidden -- the fall-back)
            for solo, spread, used, bound in ((48, 1000, 1, 1.3), (0, 1000, 1, 1.7), (0, 0, None, None), (1, 0, 0, None)):
                h.set_option("leaf_order", 1)
                h.set_debug_option("pack_solo", solo)
                h.set_debug_option("pack_spread", spread)
                pairs, closest, face = _run(h, 0, m[1], 8 * len(want_pairs) + 1024)
                if used is not None:
                    assert h.get_option("leaf_order_used0") == used, (what, solo, spread)
                if bound:   # (the default keeps a run of more than 3/4 of the cap alone: the heavy tail of the rings' sizes costs slots)
                    assert h.get_option("leaf_slots0") <= bound * m[0].n_edges + 64, (what, solo, h.get_option("leaf_slots0"), m[0].n_edges)
                assert np.array_equal(pairs, want_pairs), (what, solo, spread)
                assert np.array_equal(closest, want_e), (what, solo, spread)
                assert np.array_equal(face, om[0].face_ids(want_e)), (what, solo, spread)
        finally:
            h.close()


@pytest.mark.parametrize("shape", ["long", "short_rows", "rings"])
def test_the_second_order_of_steep_blocks_changes_no_result(oracle, shape):
    """Round 6, "leaf_ysort": a leaf block taller than wide gets a SECOND order, by y0 with a bucket table on y -- what a query
    SEGMENT's scan wants of a steep run -- beside the x order the upward rays use.  On (the default) and off: LSI through both
    kernels (one and two segments per lane) against the oracle, both map roles; the PIP kernels, which never see the second
    order, likewise; and the plan says what the build did."""
    if shape == "long":
        g = [synth.lattice_map(7, 150, 31), synth.lattice_map(16, 70, 32)]
    elif shape == "short_rows":  # 9- and 7-edge chains stitched into rows and columns: the steep ones fold back and forth in x
        g = [synth.lattice_map(34, 9, 33), synth.lattice_map(30, 7, 34)]
    else:
        g = [synth.ring_map(2500, 30000, 35), synth.lattice_map(40, 12, 36)]
    ctx = maps.Context(g).load()
    m = ctx.maps
    om = [_omap(oracle, m[0]), _omap(oracle, m[1])]
    want_pairs = oracle.lsi_grid(om[0], om[1], 256)["eid"]
    h = _capi.Handle(0)
    try:
        assert h.get_option("leaf_ysort") == 1
        for i in (0, 1):
            h.upload_map(i, m[i].pts, m[i].row_index, m[i].left, m[i].right)
        for ysort in (1, 0):
            h.set_option("leaf_ysort", ysort)
            for base in (0, 1):
                q = m[1 - base]
                want_e = oracle.pip_grid(om[base], base, q.pts, 256)
                for lsi_segments, walk_points, pip_walk in ((2, 2, 1), (1, 1, 1), (2, 2, 0)):
                    h.set_option("lsi_segments", lsi_segments)
                    h.set_option("pip_walk_points", walk_points)
                    h.set_option("pip_walk", pip_walk)
                    pairs, closest, face = _run(h, base, q, 8 * len(want_pairs) + 1024)
                    used = ysort if not (shape == "rings" and base == 0) else 0   # (not on a map of closed rings: packed rings gain nothing)
                    assert h.get_option("leaf_ysort_used%d" % base) == used
                    assert h.get_plan()["index"][base]["steep_blocks_sorted_by_y"] is bool(used)
                    assert np.array_equal(pairs, want_pairs), (shape, ysort, base, lsi_segments)
                    assert np.array_equal(closest, want_e), (shape, ysort, base, walk_points, pip_walk)
                    assert np.array_equal(face, om[base].face_ids(want_e)), (shape, ysort, base, walk_points, pip_walk)
    finally:
        h.close()
