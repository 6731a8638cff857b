"""The column index ("pip_columns": per vertical strip the slots whose box touches it, sorted by y0 -- rj_strip.hip) as
the first pass of a PIP query against the oracle and against the tree walk: ring-shaped base maps (where it is built by
default), a lattice base map (forced), adversarial integer lattices dense in ties (forced), caller-owned shuffled
point arrays, both query-map ids, and the pair with an LSI query in flight."""
import numpy as np
import pytest

from rayjoin_amd import _capi, maps, synth

pytestmark = pytest.mark.gpu


def _omap(oracle, m):
    return oracle.Map(m.pts, m.row_index, m.left, m.right)


def _pip(h, base, qpts_dev, n, closest, faces):
    h.pip_query(base, 1 - base, qpts_dev, 0, n, closest, faces)
    return closest.to_host(np.uint32)[:n].copy(), faces.to_host(np.int32)[:n].copy()


@pytest.mark.parametrize("what", ["rings", "gaussian", "lattice", "short_chains"])
def test_columns_equal_the_walk_and_the_oracle(oracle, what):
    g0 = {"rings": lambda: synth.ring_map(6000, 70000, 3), "gaussian": lambda: synth.gaussian_polygons(20000, 4, polysize=0.01),
          "lattice": lambda: synth.lattice_map(12, 60, 5),
          "short_chains": lambda: synth.lattice_map(55, 9, 7)}[what]()   # (round 6: a lattice of 9-edge chains gets the column index too)
    ctx = maps.Context([g0, synth.lattice_map(40, 25, 6)]).load()
    m = ctx.maps
    om = [_omap(oracle, m[0]), _omap(oracle, m[1])]
    rng = np.random.default_rng(1)
    h = _capi.Handle(0)
    try:
        for i in (0, 1):
            h.upload_map(i, m[i].pts, m[i].row_index, m[i].left, m[i].right)
        for base in (0, 1):
            q = m[1 - base]
            want = oracle.pip_grid(om[base], base, q.pts, 256)
            shuffled = q.pts[rng.permutation(q.n_points)]
            want_sh = oracle.pip_grid(om[base], base, np.ascontiguousarray(shuffled), 256)
            closest, faces = h.alloc(4 * q.n_points), h.alloc(4 * q.n_points)
            dsh = h.alloc(16 * q.n_points).from_host(shuffled)
            got = {}
            for columns in (1, 0, -1):
                h.set_option("pip_columns", columns)
                h.build_lbvh(base)
                used = h.get_option("pip_columns_used%d" % base)
                assert used == (1 if columns == 1 else 0 if columns == 0 else used)
                if columns == -1 and base == 0:
                    assert used == (0 if what == "lattice" else 1), what   # built where most chains are closed rings, or short (mean < 16 edges)
                if what == "rings" and base == 0:
                    # the skyline table: filled by the column index's own pass or, without one, from the leaves' boxes
                    assert h.get_option("skyline_used0") == 1, (what, columns)
                e, f = _pip(h, base, None, q.n_points, closest, faces)
                assert h.get_option("pip_last_columns") == used
                assert np.array_equal(e, want), (what, base, columns)
                assert np.array_equal(f, om[base].face_ids(want)), (what, base, columns)
                e2, _ = _pip(h, base, dsh, q.n_points, closest, faces)
                assert np.array_equal(e2, want_sh), (what, base, columns, "shuffled caller array")
                got[columns] = e
            assert np.array_equal(got[1], got[0])
    finally:
        h.close()


@pytest.mark.parametrize("seed,span,extreme", [(1, 4, False), (3, 6, True)])
def test_columns_on_adversarial_integer_lattices(oracle, seed, span, extreme):
    a = synth.adversarial_segments(700, span, seed, extreme)
    b = synth.adversarial_segments(900, span, seed + 50, extreme)
    ma, mb = maps.ScaledMap.from_segments(0, a), maps.ScaledMap.from_segments(1, b)
    o = [oracle.Map(a), oracle.Map(b)]
    rng = np.random.default_rng(seed)
    pts = rng.integers(-span - 1, span + 2, size=(3000, 2))
    if extreme:
        pts = np.clip(a[rng.integers(0, len(a), 3000)] + rng.integers(-3, 4, size=(3000, 2)), -(1 << 46), (1 << 46) - 1)
    pts = np.ascontiguousarray(pts, dtype=np.int64)
    h = _capi.Handle(0)
    try:
        h.upload_map(0, ma.pts, ma.row_index, ma.left, ma.right)
        h.upload_map(1, mb.pts, mb.row_index, mb.left, mb.right)
        h.set_option("pip_columns", 1)
        d = h.alloc(16 * len(pts)).from_host(pts)
        closest = h.alloc(4 * len(pts))
        for base in (0, 1):
            h.build_lbvh(base)
            assert h.get_option("pip_columns_used%d" % base) == 1
            h.pip_query(base, 1 - base, d, 0, len(pts), closest, None)
            assert np.array_equal(closest.to_host(np.uint32), oracle.pip_brute(o[base], 1 - base, pts)), (seed, base)
    finally:
        h.close()


def test_columns_beside_an_lsi_query(oracle):
    ctx = maps.Context([synth.ring_map(5000, 60000, 8), synth.lattice_map(40, 25, 9)]).load()
    b, q = ctx.maps
    m0, m1 = _omap(oracle, b), _omap(oracle, q)
    want_pairs = oracle.lsi_grid(m0, m1, 256)["eid"]
    want_e = oracle.pip_grid(m0, 0, q.pts, 256)
    h = _capi.Handle(0)
    try:
        h.upload_map(0, b.pts, b.row_index, b.left, b.right)
        h.upload_map(1, q.pts, q.row_index, q.left, q.right)
        h.build_lbvh(0)
        assert h.get_option("pip_columns_used0") == 1
        cap = 4 * len(want_pairs) + 64
        pairs, xs = h.alloc(8 * cap), h.alloc(48 * cap)
        closest, faces = h.alloc(4 * q.n_points), h.alloc(4 * q.n_points)
        for conc in (1, 2, 0):
            h.set_option("pip_concurrent", conc)
            for rep in range(6):
                closest.from_host(np.full(q.n_points, 0xDEADBEEF, dtype=np.uint32))
                h.lsi_query_async(0, 1, 0, q.n_edges, cap, pairs)
                h.pip_query(0, 1, None, 0, q.n_points, closest, faces, sync=False)
                h.lsi_points_async(pairs, cap, xs)
                n = h.lsi_query_finish(cap)
                h.sync()
                assert n == len(want_pairs)
                assert np.array_equal(oracle.sort_pairs(pairs.to_host(np.uint32, 2 * n).reshape(-1, 2).copy()), want_pairs)
                assert np.array_equal(closest.to_host(np.uint32), want_e), (conc, rep)
                assert np.array_equal(faces.to_host(np.int32), m0.face_ids(want_e)), (conc, rep)
    finally:
        h.close()


def test_the_strip_width_follows_the_map_and_never_changes_an_answer(oracle):
    """The strips are as wide as the map's segments ask (rj_device.h DeviceStrips: widest power of two below 2.3 x their
    mean x-extent, within 2^15..2^17 quanta); "strip_shift" (an experiment knob) forces a width.  Every width answers
    alike -- narrower strips register a segment more often, wider ones scan more entries beside the point -- and a
    rebuild on narrower strips than the tables were made for allocates them again."""
    ctx = maps.Context([synth.ring_map(6000, 70000, 3), synth.lattice_map(40, 25, 6)]).load()
    b, q = ctx.maps
    m0 = _omap(oracle, b)
    want = oracle.pip_grid(m0, 0, q.pts, 256)
    h = _capi.Handle(0)
    try:
        h.upload_map(0, b.pts, b.row_index, b.left, b.right)
        h.upload_map(1, q.pts, q.row_index, q.left, q.right)
        closest, faces = h.alloc(4 * q.n_points), h.alloc(4 * q.n_points)
        h.build_lbvh(0)
        assert h.get_option("pip_columns_used0") == 1
        auto = h.get_option("pip_column_shift0")
        assert 15 <= auto <= 17       # (6000 rings over the whole domain: long segments, the widest strips)
        entries = {}
        with pytest.raises(_capi.RayJoinError):   # (the 32-bit sort key holds 16 bits of strip: no strips narrower than 2^15)
            h.set_debug_option("strip_shift", 14)
        for shift in (17, 20, 15, 16, 0):
            h.set_debug_option("strip_shift", shift)
            h.build_lbvh(0)
            assert h.get_option("pip_columns_used0") == 1
            assert h.get_option("pip_column_shift0") == (shift or auto)
            entries[shift] = h.get_option("pip_column_entries0")
            e, f = _pip(h, 0, None, q.n_points, closest, faces)
            assert h.get_option("pip_last_columns") == 1
            assert np.array_equal(e, want), shift
            assert np.array_equal(f, m0.face_ids(want)), shift
        assert entries[15] > entries[16] > entries[17] > entries[20]
        assert entries[0] == entries[auto]
    finally:
        h.close()


def test_an_incoherent_point_set_makes_the_column_index_at_its_first_query(oracle):
    """Round 6: the reference's GeneratePIPQueries workload (run_query.cu:147-167: uniform random points) over a lattice of LONG
    chains -- a map that gets no column index at its build.  The first PIP query whose point set turns out spatially incoherent
    builds the index instead of sorting the points ("pip_columns" auto, enough points: "lazy_columns_min"), says so in the plan,
    and every later query runs on it; results are the oracle's either way, for a caller-owned array and for the map's own
    vertices; with "pip_columns" 0 nothing is built and the points go through the Morton permutation as before."""
    ctx = maps.Context([synth.lattice_map(14, 60, 51), synth.lattice_map(30, 25, 52)]).load()
    b, q = ctx.maps
    ob = _omap(oracle, b)
    rnd = np.ascontiguousarray(synth.generate_pip_queries(ctx.bb, ctx.scaling, 150000, 7))
    want_rnd = oracle.pip_grid(ob, 0, rnd, 256)
    want_own = oracle.pip_grid(ob, 0, q.pts, 256)
    for columns in (-1, 0):
        h = _capi.Handle(0)
        try:
            h.upload_map(0, b.pts, b.row_index, b.left, b.right)
            h.upload_map(1, q.pts, q.row_index, q.left, q.right)
            h.set_option("pip_columns", columns)
            h.set_debug_option("lazy_columns_min", 100000)   # (the default, 2^22 points, scaled to this test: between the two query sets)
            h.build_lbvh(0)
            assert h.get_option("pip_columns_used0") == 0   # (long open chains: the tree walk at build time)
            closest, faces = h.alloc(4 * len(rnd)), h.alloc(4 * len(rnd))
            # the map's own vertices first: fewer points than the rule asks for, nothing changes
            e, f = _pip(h, 0, None, q.n_points, h.alloc(4 * q.n_points), h.alloc(4 * q.n_points))
            assert np.array_equal(e, want_own) and h.get_option("pip_columns_used0") == 0
            d = h.alloc(16 * len(rnd)).from_host(rnd)
            for rep in range(3):
                e, f = _pip(h, 0, d, len(rnd), closest, faces)
                assert np.array_equal(e, want_rnd), (columns, rep)
                assert np.array_equal(f, ob.face_ids(want_rnd)), (columns, rep)
                ix = h.get_plan()["index"][0]
                if columns == -1:
                    assert ix["columns"] and "incoherent" in ix["columns_why"] and h.get_option("pip_last_columns") == 1
                    assert h.get_option("query_last_ordered") == 0
                else:
                    assert not ix["columns"] and h.get_option("pip_last_columns") == 0 and h.get_option("query_last_ordered") == 1
            e, f = _pip(h, 0, None, q.n_points, h.alloc(4 * q.n_points), h.alloc(4 * q.n_points))
            assert np.array_equal(e, want_own) and np.array_equal(f, ob.face_ids(want_own))
        finally:
            h.close()
