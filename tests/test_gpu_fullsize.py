"""BASELINE.json configs[1] at FULL size (USCounty x BlockGroup stand-ins, 7.1 M x 28.8 M
segments): bit-exact parity with the CPU grid oracle (it finishes in seconds on the GPU box's
cores) plus the size-independent properties the domain offers."""
import numpy as np
import pytest
import torch

from rayjoin_amd import _capi, maps, synth

pytestmark = pytest.mark.gpu


def test_uscounty_blockgroup_full_size(oracle):
    oracle.lib().rjo_set_num_threads(16)  # the box's CPU share for one GPU
    ctx = maps.Context([synth.standin("USCounty"), synth.standin("BlockGroup")]).load()
    base, query = ctx.maps
    h = _capi.Handle(0)
    h.upload_map(0, base.pts, base.row_index, base.left, base.right)
    h.upload_map(1, query.pts, query.row_index, query.left, query.right)
    h.build_lbvh(0)
    cap = int(0.1 * (base.n_edges + query.n_edges))
    pairs = h.alloc(8 * cap)
    n = h.lsi_query(0, 1, 0, query.n_edges, cap, pairs)
    h.sort_pairs(pairs, n)
    got = pairs.to_host(np.uint32, 2 * n).reshape(-1, 2)
    key = (got[:, 0].astype(np.uint64) << np.uint64(32)) | got[:, 1]
    assert (key[1:] > key[:-1]).all()  # sorted, no duplicates
    xs_dev = h.alloc(48 * n)
    h.lsi_points(pairs, n, xs_dev)
    xs = xs_dev.to_host(_capi.XSECT_DTYPE, n)
    closest = h.alloc(4 * query.n_points)
    faces = h.alloc(4 * query.n_points)
    h.pip_query(0, 1, None, 0, query.n_points, closest, faces)
    eids = closest.to_host(np.uint32)
    fids = faces.to_host(np.int32)

    # --- bit-exact against the oracle's -mode=grid at full size
    m0 = oracle.Map(base.pts, base.row_index, base.left, base.right)
    m1 = oracle.Map(query.pts, query.row_index, query.left, query.right)
    want = oracle.lsi_grid(m0, m1, 2048, cap=cap)
    assert len(want) == n > 100000
    assert np.array_equal(want["eid"], got)
    assert np.array_equal(want["x_num"], xs["x_num"]) and np.array_equal(want["y_num"], xs["y_num"])
    we = oracle.pip_grid(m0, 0, query.pts, 2048)
    assert np.array_equal(we, eids)
    assert np.array_equal(m0.face_ids(we), fids)
    del m1, want

    # --- role symmetry: index the other map; the predicate order is fixed, so the set is identical
    h.build_lbvh(1)
    p2 = h.alloc(8 * cap)
    n2 = h.lsi_query(1, 0, 0, base.n_edges, cap, p2)
    h.sort_pairs(p2, n2)
    assert n2 == n and np.array_equal(p2.to_host(np.uint32, 2 * n2).reshape(-1, 2), got)
    # --- shard additivity over 8 chain-range shards
    parts = []
    for c0, c1 in query.shard_chain_ranges(8):
        e0, e1 = query.chain_range_to_eids(c0, c1)
        k = h.lsi_query(0, 1, e0, e1, cap, p2)
        parts.append(p2.to_host(np.uint32, 2 * k).reshape(-1, 2))
    allp = oracle.sort_pairs(np.concatenate(parts))
    assert np.array_equal(allp, got)
    # --- PIP permutation invariance (also exercises the Morton re-ordering of a shuffled set)
    m = 1 << 21
    perm = np.random.default_rng(3).permutation(m)
    d = h.alloc(16 * m).from_host(query.pts[:m][perm])
    c = h.alloc(4 * m)
    h.pip_query(0, 1, d, 0, m, c, None)
    assert np.array_equal(c.to_host(np.uint32), eids[:m][perm])
    assert h.last_ms(_capi.RJ_T_ORDER) > 0
    h.close()


def test_four_level_tree_parity(oracle):
    """A base map big enough for a 4-level tree (24.4 M segments -> 382 k leaf blocks > 64^3):
    WaterBodies stand-in as base, a 2 M-segment / 2 M-point slice of BlockGroup as queries,
    bit-exact against the oracle's grid."""
    oracle.lib().rjo_set_num_threads(16)
    ctx = maps.Context([synth.standin("WaterBodies"), synth.standin("BlockGroup", 0.27)]).load()
    base, query = ctx.maps
    assert base.n_edges > 64 ** 3 * 64
    h = _capi.Handle(0)
    h.upload_map(0, base.pts, base.row_index, base.left, base.right)
    h.upload_map(1, query.pts, query.row_index, query.left, query.right)
    h.build_lbvh(0)
    cap = int(0.2 * (base.n_edges + query.n_edges))
    pairs = h.alloc(8 * cap)
    n = h.lsi_query(0, 1, 0, query.n_edges, cap, pairs)
    h.sort_pairs(pairs, n)
    got = pairs.to_host(np.uint32, 2 * n).reshape(-1, 2)
    closest = h.alloc(4 * query.n_points)
    faces = h.alloc(4 * query.n_points)
    h.pip_query(0, 1, None, 0, query.n_points, closest, faces)
    m0 = oracle.Map(base.pts, base.row_index, base.left, base.right)
    m1 = oracle.Map(query.pts, query.row_index, query.left, query.right)
    want = oracle.lsi_grid(m0, m1, 4096, cap=cap)
    assert len(want) == n > 10000 and np.array_equal(want["eid"], got)
    we = oracle.pip_grid(m0, 0, query.pts, 4096)
    assert np.array_equal(we, closest.to_host(np.uint32))
    assert np.array_equal(m0.face_ids(we), faces.to_host(np.int32))
    h.close()


def test_lakes_parks_pip_full_size(oracle):
    """BASELINE.json configs[2] (Lakes x Parks, -query=pip) at FULL size: the 66.9 M-segment base map
    (1.05 M leaf blocks, 4 levels) and EVERY vertex of the Parks stand-in (27.8 M points) against the
    oracle's grid PIP, bit-exact eids and face ids.  (With less host memory than the oracle's grid
    needs: every 13th vertex.)"""
    import psutil
    avail = psutil.virtual_memory().available
    if avail < 40 << 30:
        pytest.skip("needs ~30 GB of host memory for the 66.9 M-segment map and the oracle's grid")
    oracle.lib().rjo_set_num_threads(16)
    ctx = maps.Context([synth.standin("LakesNA"), synth.standin("ParksNA")]).load()
    base, query = ctx.maps
    assert base.n_edges > 66_000_000
    pts = np.ascontiguousarray(query.pts if avail > (60 << 30) else query.pts[::13])
    h = _capi.Handle(0)
    h.upload_map(0, base.pts, base.row_index, base.left, base.right)
    h.build_lbvh(0)
    d = h.alloc(16 * len(pts)).from_host(pts)
    closest = h.alloc(4 * len(pts))
    faces = h.alloc(4 * len(pts))
    h.pip_query(0, 1, d, 0, len(pts), closest, faces)
    m0 = oracle.Map(base.pts, base.row_index, base.left, base.right)
    we = oracle.pip_grid(m0, 0, pts, 4096)
    eids = closest.to_host(np.uint32)
    assert (eids != 0xFFFFFFFF).sum() > len(pts) // 2
    assert np.array_equal(we, eids)
    assert np.array_equal(m0.face_ids(we), faces.to_host(np.int32))
    h.close()


def test_waterbodies_blockgroup_lsi_full_size_and_8_shards(oracle):
    """BASELINE.json configs[4] (WaterBodies x BlockGroup LSI, query map sharded 8-way) on one GPU:
    the whole join (24.4 M x 28.8 M segments, 4-level tree) bit-exact against the oracle's grid --
    pairs and stored points -- and the 8 chain-range shards of the query map, queried one after the
    other, reproduce it exactly (what the 8 ranks of the real run compute, minus the exchange)."""
    import psutil
    if psutil.virtual_memory().available < 40 << 30:
        pytest.skip("needs ~25 GB of host memory for the two maps and the oracle's grid")
    oracle.lib().rjo_set_num_threads(16)
    ctx = maps.Context([synth.standin("WaterBodies"), synth.standin("BlockGroup")]).load()
    base, query = ctx.maps
    h = _capi.Handle(0)
    h.upload_map(0, base.pts, base.row_index, base.left, base.right)
    h.upload_map(1, query.pts, query.row_index, query.left, query.right)
    h.build_lbvh(0)
    cap = int(0.1 * (base.n_edges + query.n_edges))
    pairs = h.alloc(8 * cap)
    n = h.lsi_query(0, 1, 0, query.n_edges, cap, pairs)
    h.sort_pairs(pairs, n)
    got = pairs.to_host(np.uint32, 2 * n).reshape(-1, 2)
    xs_dev = h.alloc(48 * n)
    h.lsi_points(pairs, n, xs_dev)
    xs = xs_dev.to_host(_capi.XSECT_DTYPE, n)
    m0 = oracle.Map(base.pts, base.row_index, base.left, base.right)
    m1 = oracle.Map(query.pts, query.row_index, query.left, query.right)
    want = oracle.lsi_grid(m0, m1, 4096, cap=cap)
    assert len(want) == n > 1_000_000
    assert np.array_equal(want["eid"], got)
    assert np.array_equal(want["x_num"], xs["x_num"]) and np.array_equal(want["y_num"], xs["y_num"])
    del m1, want
    parts = []
    for c0, c1 in query.shard_chain_ranges(8):
        e0, e1 = query.chain_range_to_eids(c0, c1)
        k = h.lsi_query(0, 1, e0, e1, cap, pairs)
        parts.append(pairs.to_host(np.uint32, 2 * k).reshape(-1, 2))
    assert np.array_equal(oracle.sort_pairs(np.concatenate(parts)), got)
    # Round 6: the PIP query of this pair on the path the handle CHOOSES for a map of short chains -- the column index
    # (rj_get_plan says on what grounds), the steep blocks of the tree in their second order for the LSI query above -- every
    # vertex of the query map against the oracle's grid, eids and face ids; then the same on the tree walk ("pip_columns" 0).
    closest = h.alloc(4 * query.n_points)
    faces = h.alloc(4 * query.n_points)
    h.pip_query(0, 1, None, 0, query.n_points, closest, faces)
    plan = h.get_plan()
    assert plan["index"][0]["columns"] and "short chains" in plan["index"][0]["columns_why"] and plan["index"][0]["steep_blocks_sorted_by_y"]
    assert plan["pip"]["first_pass"]["kernel"] == "k_pip_strip"
    we = oracle.pip_grid(m0, 0, query.pts, 4096)
    assert (we != 0xFFFFFFFF).sum() > query.n_points // 2
    assert np.array_equal(we, closest.to_host(np.uint32))
    assert np.array_equal(m0.face_ids(we), faces.to_host(np.int32))
    h.set_option("pip_columns", 0)
    h.build_lbvh(0)
    h.pip_query(0, 1, None, 0, query.n_points, closest, faces)
    assert h.get_plan()["pip"]["first_pass"]["kernel"] == "k_pip_walk2" and not h.get_plan()["index"][0]["columns"]
    assert np.array_equal(we, closest.to_host(np.uint32))
    assert np.array_equal(m0.face_ids(we), faces.to_host(np.int32))
    h.close()


def test_uscounty_zipcode_overlay_stages_full_size():
    """BASELINE.json configs[3] (USCounty x Zipcode overlay) at FULL size: IntersectEdge,
    LocateVerticesInOtherMap (both maps) and the per-map ComputeOutputPolygons records, bit-exact
    against the oracle's -mode=grid pipeline (tests/overlay_fullsize_check.py does the comparison;
    run as a child process so its 30 M-segment maps are freed before the next test)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "overlay_fullsize_check.py")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["bit_exact_vs_oracle"] is True and out["intersections"] > 10000
    assert out["map0_edges"] > 7_000_000 and out["map1_edges"] > 23_000_000


def test_walk_two_points_per_lane_equals_one_and_the_single_kernel():
    """k_pip_walk2 (128 positions per wave, "pip_walk_points" 2, the default where it applies) against k_pip_walk and
    against k_pip alone on a nested pair large enough for four 128-position groups per resident wave (4.6 M query
    vertices, a quarter of them on base vertices): closest eids and face ids of every vertex, bit for bit; and it is what ran."""
    ctx = maps.Context([synth.standin("USCounty", 0.4), synth.standin("NestedBlockGroup", 0.4)]).load()
    base, query = ctx.maps
    h = _capi.Handle(0)
    h.upload_map(0, base.pts, base.row_index, base.left, base.right)
    h.upload_map(1, query.pts, query.row_index, query.left, query.right)
    h.build_lbvh(0)
    n = query.n_points
    assert n > 1 << 22
    out = {}
    for name, walk, pts in (("two", 2, 2), ("one", 2, 1), ("single", 0, 1)):
        h.set_option("pip_walk", walk)
        h.set_option("pip_walk_points", pts)
        c, f = h.alloc(4 * n), h.alloc(4 * n)
        h.pip_query(0, 1, None, 0, n, c, f)
        if walk:
            assert h.get_option("pip_last_walk_points") == pts
        out[name] = (c.to_host(np.uint32).copy(), f.to_host(np.int32).copy())
    for other in ("one", "single"):
        assert np.array_equal(out["two"][0], out[other][0]) and np.array_equal(out["two"][1], out[other][1]), other
    assert (out["two"][0] != 0xFFFFFFFF).mean() > 0.5
    # a sub-range whose length is not a multiple of 128 (the last wave's second point set is partly / wholly empty)
    for m in (n - 37, (1 << 20) + 64 + 5):
        c2 = h.alloc(4 * m)
        h.set_option("pip_walk", 2); h.set_option("pip_walk_points", 2)
        h.pip_query(0, 1, None, 0, m, c2, None)
        assert np.array_equal(c2.to_host(np.uint32), out["single"][0][:m]), m
    h.set_option("pip_walk", 1)
    h.close()
