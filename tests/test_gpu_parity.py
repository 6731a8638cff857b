"""Parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on the
same seeded inputs (bit-exact: integer eid pairs, integer face ids, truncated int64 points)."""
import numpy as np
import pytest

from rayjoin_amd import _capi, maps, ops, synth

pytestmark = pytest.mark.gpu


def _omap(oracle, m):
    return oracle.Map(m.pts, m.row_index, m.left, m.right)


@pytest.fixture(scope="module")
def lattice_pair():
    ctx = maps.Context([synth.lattice_map(9, 120, 11), synth.lattice_map(21, 50, 12)]).load()
    dctx = ops.DeviceContext(ctx).LoadToDevice()
    dctx.BuildIndex(0)
    dctx.BuildIndex(1)
    yield ctx, dctx
    dctx.close()


def _lsi(dctx, query_map_id, cap, eid_range=None):
    lsi = ops.LSILBVH(dctx)
    lsi.Init(cap)
    lsi.Query(query_map_id, eid_range)
    return lsi


def test_lsi_equals_grid_and_brute(oracle, lattice_pair):
    ctx, dctx = lattice_pair
    m0, m1 = _omap(oracle, ctx.maps[0]), _omap(oracle, ctx.maps[1])
    grid = oracle.lsi_grid(m0, m1, 512)
    brute = oracle.lsi_brute(m0, m1)
    assert len(grid) > 300 and np.array_equal(grid["eid"], brute)
    for qm in (1, 0):  # query_exec direction (base 0) and the overlay's (base 1)
        lsi = _lsi(dctx, qm, 4 * len(grid))
        assert lsi.n_xsects == len(grid)
        assert np.array_equal(lsi.get_pairs(), grid["eid"])
        xs = lsi.get_xsects()
        for f in ("x_num", "x_den", "y_num", "y_den", "eid", "mid_point_polygon_id"):
            assert np.array_equal(xs[f], grid[f]), f


def test_lsi_shards_union_equals_whole(oracle, lattice_pair):
    ctx, dctx = lattice_pair
    whole = _lsi(dctx, 1, 100000).get_pairs()
    q = ctx.maps[1]
    for nshard in (2, 3, 8):
        parts = []
        for c0, c1 in q.shard_chain_ranges(nshard):
            parts.append(_lsi(dctx, 1, 100000, q.chain_range_to_eids(c0, c1)).get_pairs(sort=False))
        got = oracle.sort_pairs(np.concatenate(parts))
        assert np.array_equal(got, whole)


def test_lsi_queue_overflow_reports_true_count(lattice_pair):
    ctx, dctx = lattice_pair
    n = _lsi(dctx, 1, 100000).n_xsects
    lsi = ops.LSILBVH(dctx)
    lsi.Init(7)
    with pytest.raises(_capi.QueueOverflow) as ei:
        lsi.Query(1)
    assert ei.value.n_found == n and ei.value.code == _capi.RJ_E_OVERFLOW


def test_pip_equals_grid_and_brute(oracle, lattice_pair):
    ctx, dctx = lattice_pair
    om = [_omap(oracle, ctx.maps[0]), _omap(oracle, ctx.maps[1])]
    for qm in (1, 0):
        base = 1 - qm
        pts = ctx.maps[qm].pts
        want = oracle.pip_grid(om[base], base, pts, 512)
        assert np.array_equal(want, oracle.pip_brute(om[base], qm, pts))
        pip = ops.PIPLBVH(dctx)
        pip.Init(len(pts))
        pip.Query(qm)  # the query map's own vertices, run_query.cu:346
        got = pip.get_closest_eids()
        assert (got != _capi.MISS_EID).sum() > 1000 and (got == _capi.MISS_EID).sum() > 0
        assert np.array_equal(got, want)
        assert np.array_equal(pip.get_face_ids(), om[base].face_ids(want))
        # explicit point array + a sub-range (a shard)
        pip.Query(qm, query_points=pts[100:1357])
        assert np.array_equal(pip.get_closest_eids(), want[100:1357])
        pip.Query(qm, point_range=(64, 1000))
        assert np.array_equal(pip.get_closest_eids(), want[64:1000])


def test_random_query_workloads(oracle, lattice_pair):
    """GenerateLSIQueries / GeneratePIPQueries (run_query.cu:102-167): spatially incoherent."""
    ctx, _ = lattice_pair
    qmap = synth.generate_lsi_queries(ctx.bb, ctx.scaling, 3000, 2.0, seed=5)
    c2 = maps.Context([ctx.planar_graphs[0], None])
    c2.bb, c2.scaling = ctx.bb, ctx.scaling
    c2.maps = [ctx.maps[0], qmap]
    d2 = ops.DeviceContext(c2).LoadToDevice()
    d2.BuildIndex(0)
    m0 = _omap(oracle, ctx.maps[0])
    mq = oracle.Map(qmap.pts)
    want = oracle.lsi_brute(m0, mq)
    assert len(want) > 20
    lsi = _lsi(d2, 1, 10000)
    assert np.array_equal(lsi.get_pairs(), want)
    pts = synth.generate_pip_queries(ctx.bb, ctx.scaling, 5000, seed=6)
    pip = ops.PIPLBVH(d2)
    pip.Init(len(pts))
    pip.Query(1, query_points=pts)
    assert np.array_equal(pip.get_closest_eids(), oracle.pip_brute(m0, 1, pts))
    d2.close()


@pytest.mark.parametrize("seed,span,extreme", [(1, 4, False), (2, 9, False), (3, 6, True)])
def test_adversarial_integer_lattice(oracle, seed, span, extreme):
    """Shared endpoints, T-junctions, collinear overlaps, duplicates, axis-parallel edges, and the
    +-2^46 corners: every simulation-of-simplicity branch, LSI and PIP, both directions."""
    a = synth.adversarial_segments(700, span, seed, extreme)
    b = synth.adversarial_segments(900, span, seed + 50, extreme)
    ma, mb = maps.ScaledMap.from_segments(0, a), maps.ScaledMap.from_segments(1, b)
    ctx = maps.Context([None, None])
    ctx.maps = [ma, mb]
    dctx = ops.DeviceContext(ctx).LoadToDevice()
    dctx.BuildIndex(0)
    dctx.BuildIndex(1)
    o0, o1 = oracle.Map(a), oracle.Map(b)
    want = oracle.lsi_brute(o0, o1)
    assert len(want) > 500
    for qm in (1, 0):
        lsi = _lsi(dctx, qm, 4 * len(want))
        assert np.array_equal(lsi.get_pairs(), want)
        xs = lsi.get_xsects()
        ref = oracle.lsi_points(o0, o1, want)
        for f in ("x_num", "y_num", "eid"):
            assert np.array_equal(xs[f], ref[f]), f
    rng = np.random.default_rng(seed)
    off = a.reshape(-1, 4)[:, :2].mean(axis=0).astype(np.int64) if extreme else 0
    pts = rng.integers(-span - 1, span + 2, size=(3000, 2))
    if extreme:  # sample around every corner cluster
        corners = a[rng.integers(0, len(a), 3000)]
        pts = corners + rng.integers(-3, 4, size=(3000, 2))
        pts = np.clip(pts, -(1 << 46), (1 << 46) - 1)
    for qm, ob in ((1, o0), (0, o1)):
        pip = ops.PIPLBVH(dctx)
        pip.Init(len(pts))
        pip.Query(qm, query_points=pts)
        assert np.array_equal(pip.get_closest_eids(), oracle.pip_brute(ob, qm, pts))
    dctx.close()


def test_query_ordering_modes_agree(oracle, lattice_pair):
    """Incoherent query sets are re-ordered along the Morton curve inside the query
    (rj_set_option "query_order"); never / auto / always must give identical results."""
    ctx, dctx = lattice_pair
    h = dctx.handle
    rng = np.random.default_rng(17)
    pts = ctx.maps[1].pts[rng.permutation(ctx.maps[1].n_points)]  # shuffled vertices
    want = oracle.pip_brute(_omap(oracle, ctx.maps[0]), 1, pts)
    got = {}
    for mode in (0, 1, 2):
        h.set_option("query_order", mode)
        pip = ops.PIPLBVH(dctx)
        pip.Init(len(pts))
        pip.Query(1, query_points=pts)
        got[mode] = (pip.get_closest_eids(), pip.get_face_ids())
        assert np.array_equal(got[mode][0], want), mode
        lsi = _lsi(dctx, 1, 100000)
        got[mode] += (lsi.get_pairs(),)
    h.set_option("query_order", 1)
    for mode in (1, 2):
        for a, b in zip(got[0], got[mode]):
            assert np.array_equal(a, b)
    assert h.last_ms(_capi.RJ_T_ORDER) > 0  # the always-mode ran the ordering pass


def test_pip_on_the_second_stream_beside_lsi(oracle, lattice_pair):
    """rj_set_option "pip_concurrent": the PIP kernel runs on the handle's second stream while the
    LSI kernel runs on the main one; both results are those of the serial calls."""
    ctx, dctx = lattice_pair
    h = dctx.handle
    m0 = _omap(oracle, ctx.maps[0])
    want_pairs = oracle.lsi_brute(m0, _omap(oracle, ctx.maps[1]))
    want_eids = oracle.pip_brute(m0, 1, ctx.maps[1].pts)
    cap = 4 * len(want_pairs)
    pairs = h.alloc(8 * cap)
    closest = h.alloc(4 * ctx.maps[1].n_points)
    faces = h.alloc(4 * ctx.maps[1].n_points)
    try:
        for mode in (1, 2):
            h.set_option("pip_concurrent", mode)
            for _ in range(3):
                h.lsi_query_async(0, 1, 0, ctx.maps[1].n_edges, cap, pairs)
                h.pip_query(0, 1, None, 0, ctx.maps[1].n_points, closest, faces, sync=False)
                n = h.lsi_query_finish(cap)
                h.sync()
                assert n == len(want_pairs)
                assert np.array_equal(closest.to_host(np.uint32), want_eids)
                assert np.array_equal(faces.to_host(np.int32), m0.face_ids(want_eids))
            h.sort_pairs(pairs, n)
            assert np.array_equal(pairs.to_host(np.uint32, 2 * n).reshape(-1, 2), want_pairs)
            assert h.last_ms(_capi.RJ_T_PIP_KERNEL) > 0 and h.last_ms(_capi.RJ_T_LSI_KERNEL) > 0
    finally:
        h.set_option("pip_concurrent", 0)


def test_auto_schedule_tries_all_and_settles(oracle, lattice_pair):
    """ "pip_concurrent" 2: the first LSI + PIP pairs run in three ways (one after the other / sharing
    the chip / beside each other on full grids), then the fastest schedule is kept; every step's results
    are the same; a new query size or the option itself starts the decision again."""
    ctx, dctx = lattice_pair
    h = dctx.handle
    m0 = _omap(oracle, ctx.maps[0])
    want_pairs = oracle.lsi_brute(m0, _omap(oracle, ctx.maps[1]))
    want_eids = oracle.pip_brute(m0, 1, ctx.maps[1].pts)
    cap = 4 * len(want_pairs)
    pairs = h.alloc(8 * cap)
    xs = h.alloc(48 * cap)
    closest = h.alloc(4 * ctx.maps[1].n_points)

    def pair(ne):
        h.lsi_query_async(0, 1, 0, ne, cap, pairs)
        h.lsi_points_async(pairs, cap, xs)
        h.pip_query(0, 1, None, 0, ctx.maps[1].n_points, closest, None, sync=False)
        n = h.lsi_query_finish(cap)
        h.sync()
        return n
    try:
        h.set_option("query_order", 0)  # (re-ordered queries are never paired: shared scratch)
        h.set_option("pip_concurrent", 2)
        assert h.get_option("pip_concurrent") == 2 and h.get_option("pip_schedule") == -1
        undecided = 0
        for i in range(16):
            n = pair(ctx.maps[1].n_edges)
            assert n == len(want_pairs)
            assert np.array_equal(closest.to_host(np.uint32), want_eids)
            trials, choice = h.get_option("pip_schedule_trials"), h.get_option("pip_schedule")
            assert trials <= i and (choice == -1) == (trials < 4), (i, trials, choice)  # (a pair is measured when the next one is launched)
            undecided += choice == -1
        # four measured pairs settle it: the fifth pair already runs the chosen schedule
        assert 4 <= undecided < 16 and h.get_option("pip_schedule") in (0, 1, 2)
        h.sort_pairs(pairs, n)
        assert np.array_equal(pairs.to_host(np.uint32, 2 * n).reshape(-1, 2), want_pairs)
        pair(ctx.maps[1].n_edges // 3)  # another query size: decide again
        assert h.get_option("pip_schedule") == -1
        h.set_option("pip_concurrent", 1)
        assert h.get_option("pip_schedule") == 1
    finally:
        h.set_option("pip_concurrent", 0)
        h.set_option("query_order", 1)
    assert h.get_option("pip_schedule") == 0


def test_counter_sets_alternate_without_fill_kernels(oracle, lattice_pair):
    """No fill kernel clears the result count or the scheduler counters: every query kernel clears the set
    the NEXT launch on its stream uses (two sets, alternating).  Sequences that would expose a stale
    set: an odd number of launches, back-to-back asynchronous queries of different ranges without a
    finish in between, empty queries, the grid LSI (its own count word) in between, PIP on both streams."""
    ctx, dctx = lattice_pair
    h = dctx.handle
    q = ctx.maps[1]
    m0, m1 = _omap(oracle, ctx.maps[0]), _omap(oracle, q)
    want = oracle.lsi_brute(m0, m1)
    want_eids = oracle.pip_brute(m0, 1, q.pts)
    half = q.n_edges // 2
    n_half = int((want[:, 1] < half).sum())
    cap = 4 * len(want)
    pairs, pairs2 = h.alloc(8 * cap), h.alloc(8 * cap)
    xs = h.alloc(48 * cap)
    closest = h.alloc(4 * q.n_points)
    h.build_grid(0, 64); h.build_grid(1, 64)
    try:
        for rep in range(5):  # (odd: the sets end up swapped for the next test)
            # two launches in flight, the second one's count is the one that is read
            h.lsi_query_async(0, 1, 0, q.n_edges, cap, pairs2)
            h.lsi_query_async(0, 1, 0, half, cap, pairs)
            h.lsi_points_async(pairs, cap, xs)
            assert h.lsi_query_finish(cap) == n_half
            # an empty query reports 0 and does not disturb the sets
            h.lsi_query_async(0, 1, 7, 7, cap, pairs)
            assert h.lsi_query_finish(cap) == 0
            assert h.lsi_query(0, 1, 0, q.n_edges, cap, pairs) == len(want)
            # the grid LSI has its own count word
            assert h.lsi_query_grid(cap, pairs2) == len(want)
            assert h.lsi_query(0, 1, 0, half, cap, pairs) == n_half
            # PIP: main stream, second stream, empty, in a row
            for mode in (0, 1, 1, 0):
                h.set_option("pip_concurrent", mode)
                h.pip_query(0, 1, None, 0, 0, closest, None, sync=False)
                h.pip_query(0, 1, None, 0, q.n_points, closest, None, sync=False)
                h.sync()
                assert np.array_equal(closest.to_host(np.uint32), want_eids), (rep, mode)
        h.lsi_query(0, 1, 0, q.n_edges, cap, pairs)
        h.sort_pairs(pairs, len(want))
        assert np.array_equal(pairs.to_host(np.uint32, 2 * len(want)).reshape(-1, 2), want)
    finally:
        h.set_option("pip_concurrent", 0)


def test_two_handles_from_two_threads(oracle, lattice_pair):
    """A handle is not thread-safe, but different handles may be used from different threads
    (include/rayjoin_amd.h conventions); ctypes drops the GIL during the calls."""
    import threading
    ctx, _ = lattice_pair
    want = oracle.lsi_brute(_omap(oracle, ctx.maps[0]), _omap(oracle, ctx.maps[1]))
    out = {}

    def work(tag, base):
        d = ops.DeviceContext(ctx).LoadToDevice()
        d.BuildIndex(base)
        for _ in range(5):
            lsi = _lsi(d, 1 - base, 100000)
            out[tag] = lsi.get_pairs()
        d.close()

    ts = [threading.Thread(target=work, args=(i, i % 2)) for i in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert len(out) == 4
    for v in out.values():
        assert np.array_equal(v, want)


def test_adversarial_chains_face_ids(oracle):
    pa = synth.adversarial_chains(60, 9, 12, 21)
    pb = synth.adversarial_chains(80, 5, 12, 22)
    ctx = maps.Context([None, None])
    ctx.maps = [maps.ScaledMap(0, *pa), maps.ScaledMap(1, *pb)]
    dctx = ops.DeviceContext(ctx).LoadToDevice()
    dctx.BuildIndex(0)
    o0, o1 = oracle.Map(*pa), oracle.Map(*pb)
    want = oracle.lsi_brute(o0, o1)
    assert np.array_equal(_lsi(dctx, 1, 100000).get_pairs(), want)
    pip = ops.PIPLBVH(dctx)
    pip.Init(len(pb[0]))
    pip.Query(1)
    eids = oracle.pip_brute(o0, 1, pb[0])
    assert np.array_equal(pip.get_closest_eids(), eids)
    assert np.array_equal(pip.get_face_ids(), o0.face_ids(eids))
    dctx.close()


@pytest.mark.parametrize("n0,n1", [(1, 1), (63, 65), (64, 64), (65, 4097), (0, 5), (5, 0)])
def test_ragged_and_empty_sizes(oracle, n0, n1):
    a = synth.adversarial_segments(n0, 30, 70 + n0)
    b = synth.adversarial_segments(n1, 30, 90 + n1)
    ctx = maps.Context([None, None])
    ctx.maps = [maps.ScaledMap.from_segments(0, a), maps.ScaledMap.from_segments(1, b)]
    dctx = ops.DeviceContext(ctx).LoadToDevice()
    dctx.BuildIndex(0)
    dctx.BuildIndex(1)
    o0, o1 = oracle.Map(a), oracle.Map(b)
    want = oracle.lsi_brute(o0, o1)
    for qm in (1, 0):
        assert np.array_equal(_lsi(dctx, qm, 100000).get_pairs(), want)
    pts = np.random.default_rng(n0 + n1).integers(-31, 32, size=(257, 2))
    pip = ops.PIPLBVH(dctx)
    pip.Init(len(pts))
    pip.Query(1, query_points=pts)
    assert np.array_equal(pip.get_closest_eids(), oracle.pip_brute(o0, 1, pts))
    pip.Query(1, query_points=pts[:0])
    assert pip.get_closest_eids().shape == (0,)
    dctx.close()


def test_call_order_errors_are_reported_not_fatal():
    h = _capi.Handle(0)
    with pytest.raises(_capi.RayJoinError) as ei:
        h.build_lbvh(0)
    assert ei.value.code == _capi.RJ_E_INVALID
    seg = synth.adversarial_segments(10, 5, 1)
    m = maps.ScaledMap.from_segments(0, seg)
    h.upload_map(0, m.pts, m.row_index, m.left, m.right)
    h.upload_map(1, m.pts, m.row_index, m.left, m.right)
    buf = h.alloc(800)
    with pytest.raises(_capi.RayJoinError):
        h.lsi_query(0, 1, 0, 10, 100, buf)  # index not built
    h.build_lbvh(0)
    with pytest.raises(_capi.RayJoinError):
        h.lsi_query(0, 1, 0, 11, 100, buf)  # eid range out of bounds
    with pytest.raises(_capi.RayJoinError):
        h.upload_map(0, np.array([[1 << 46, 0], [0, 0]]), [0, 2], [0], [0])  # outside scaled range
    with pytest.raises(_capi.RayJoinError):
        h.upload_map(0, np.array([[0, 0]]), [0, 1], [0], [0])  # 1-point chain (planar_graph.h:71)
    h.close()


def test_medium_maps_against_grid_oracle(oracle):
    """~0.2 M x 0.55 M edges: the CPU grid finishes in seconds; brute force no longer does."""
    ctx = maps.Context([synth.lattice_map(24, 350, 31), synth.lattice_map(130, 16, 32)]).load()
    dctx = ops.DeviceContext(ctx).LoadToDevice()
    dctx.BuildIndex(0)
    m0, m1 = _omap(oracle, ctx.maps[0]), _omap(oracle, ctx.maps[1])
    grid = oracle.lsi_grid(m0, m1, 2048)
    lsi = _lsi(dctx, 1, 4 * len(grid) + 1000)
    assert lsi.n_xsects == len(grid) > 3000
    assert np.array_equal(lsi.get_pairs(), grid["eid"])
    xs = lsi.get_xsects()
    assert np.array_equal(xs["x_num"], grid["x_num"]) and np.array_equal(xs["y_num"], grid["y_num"])
    pts = ctx.maps[1].pts
    pip = ops.PIPLBVH(dctx)
    pip.Init(len(pts))
    pip.Query(1)
    want = oracle.pip_grid(m0, 0, pts, 2048)
    assert np.array_equal(pip.get_closest_eids(), want)
    assert np.array_equal(pip.get_face_ids(), m0.face_ids(want))
    dctx.close()


def test_domain_spanning_segments(oracle):
    """A few base segments as long as the map (boxes covering millions of bitmap cells) among many
    small ones: the occupancy bitmap is flagged non-exhaustive instead of being rasterised for
    them, results stay exact, and the build stays fast."""
    import time
    rng = np.random.default_rng(8)
    small = synth.adversarial_segments(4000, 1 << 30, 81)
    big = np.array([[-(1 << 45), -(1 << 45) + 7, (1 << 45), (1 << 45) - 3],
                    [-(1 << 45), (1 << 44), (1 << 45), -(1 << 44)],
                    [5, -(1 << 45), 9, (1 << 45)]], dtype=np.int64).reshape(-1, 2)
    a = np.concatenate([small, big])
    b = synth.adversarial_segments(5000, 1 << 30, 82)
    ctx = maps.Context([None, None])
    ctx.maps = [maps.ScaledMap.from_segments(0, a), maps.ScaledMap.from_segments(1, b)]
    dctx = ops.DeviceContext(ctx).LoadToDevice()
    t0 = time.perf_counter()
    dctx.BuildIndex(0)
    dctx.BuildIndex(1)
    assert time.perf_counter() - t0 < 2.0
    o0, o1 = oracle.Map(a), oracle.Map(b)
    want = oracle.lsi_brute(o0, o1)
    assert len(want) > 1000
    for qm in (1, 0):
        assert np.array_equal(_lsi(dctx, qm, 4 * len(want)).get_pairs(), want)
    pts = rng.integers(-(1 << 30), 1 << 30, size=(4000, 2))
    pip = ops.PIPLBVH(dctx)
    pip.Init(len(pts))
    pip.Query(1, query_points=pts)
    assert np.array_equal(pip.get_closest_eids(), oracle.pip_brute(o0, 1, pts))
    dctx.close()


def test_async_query_leaves_complete_records(oracle, lattice_pair):
    """rj_lsi_query_async + rj_lsi_points_async + rj_lsi_query_finish: the 48-byte Intersection records
    of the query are produced behind it on the stream (count read on the device) -- what
    LSILBVH::Query leaves in its queue (src/app/lsi_lbvh.h:71-78) -- and equal the oracle's."""
    ctx, dctx = lattice_pair
    h = dctx.handle
    m0, m1 = _omap(oracle, ctx.maps[0]), _omap(oracle, ctx.maps[1])
    want = oracle.lsi_grid(m0, m1, 512)
    cap = 3 * len(want)
    pairs = h.alloc(8 * cap)
    recs = h.alloc(48 * cap)
    for _ in range(2):
        h.lsi_query_async(0, 1, 0, ctx.maps[1].n_edges, cap, pairs)
        h.lsi_points_async(pairs, cap, recs)
        n = h.lsi_query_finish(cap)
        assert n == len(want)
        got = recs.to_host(_capi.XSECT_DTYPE, n)
        key = (got["eid"][:, 0].astype(np.uint64) << np.uint64(32)) | got["eid"][:, 1]
        got = got[np.argsort(key, kind="stable")]
        for f in ("x_num", "x_den", "y_num", "y_den", "eid", "mid_point_polygon_id"):
            assert np.array_equal(got[f], want[f]), f
    # a capacity smaller than the result: the first `capacity` records are produced, the overflow is reported
    small = 16
    with pytest.raises(_capi.QueueOverflow) as ei:
        h.lsi_query_async(0, 1, 0, ctx.maps[1].n_edges, small, pairs)
        h.lsi_points_async(pairs, small, recs)
        h.lsi_query_finish(small)
    assert ei.value.n_found == len(want)
    got = recs.to_host(_capi.XSECT_DTYPE, small)
    ref = oracle.lsi_points(m0, m1, np.ascontiguousarray(got["eid"]))
    assert np.array_equal(got["x_num"], ref["x_num"]) and np.array_equal(got["y_num"], ref["y_num"])


def test_pipelined_steps_equal_the_oracle(oracle):
    """Step k + 1 launched before step k's count is looked at (rj_lsi_count_async / rj_lsi_count_wait, two sets of result
    buffers): what bench.py's pipelined loop and an N > 1 job run.  Every step's pairs, records, closest eids and faces
    against the oracle, for the schedule that shares the chip, for "auto" and for taking turns; then a plain
    synchronised query behind the pipelined ones, other buffers and another range."""
    ctx = maps.Context([synth.lattice_map(9, 120, 31), synth.lattice_map(21, 50, 32)]).load()
    b, q = ctx.maps
    m0, m1 = _omap(oracle, b), _omap(oracle, q)
    want_pairs = oracle.lsi_grid(m0, m1, 256)["eid"]
    want_eids = oracle.pip_grid(m0, 0, q.pts, 256)
    h = _capi.Handle(0)
    try:
        h.upload_map(0, b.pts, b.row_index, b.left, b.right)
        h.upload_map(1, q.pts, q.row_index, q.left, q.right)
        h.build_lbvh(0)
        cap = 4 * len(want_pairs) + 64
        pairs = [h.alloc(8 * cap) for _ in range(2)]
        xs = [h.alloc(48 * cap) for _ in range(2)]
        closest = [h.alloc(4 * q.n_points) for _ in range(2)]
        faces = [h.alloc(4 * q.n_points) for _ in range(2)]

        def check(k, n, what):
            assert n == len(want_pairs), what
            got = pairs[k].to_host(np.uint32, 2 * n).reshape(-1, 2)
            assert np.array_equal(oracle.sort_pairs(got.copy()), want_pairs), what
            rec = xs[k].to_host(_capi.XSECT_DTYPE, n)
            ref = oracle.lsi_points(m0, m1, np.ascontiguousarray(rec["eid"]))
            assert np.array_equal(rec["eid"], got) and np.array_equal(rec["x_num"], ref["x_num"]) and np.array_equal(rec["y_num"], ref["y_num"]), what
            assert np.array_equal(closest[k].to_host(np.uint32), want_eids), what
            assert np.array_equal(faces[k].to_host(np.int32), m0.face_ids(want_eids)), what

        for conc in (1, 2, 0):
            h.set_option("pip_concurrent", conc)
            for timers in (1, 0):
                h.set_option("timers", timers)
                for k in (0, 1):
                    closest[k].from_host(np.full(q.n_points, 0xDEADBEEF, dtype=np.uint32))
                steps = 7
                for j in range(steps):
                    k = j % 2
                    h.lsi_query_async(0, 1, 0, q.n_edges, cap, pairs[k])
                    h.pip_query(0, 1, None, 0, q.n_points, closest[k], faces[k], sync=False)
                    h.lsi_points_async(pairs[k], cap, xs[k])
                    h.lsi_count_async(k)
                    if j > 0:
                        n = h.lsi_count_wait(1 - k, cap)
                        if j in (1, steps - 1):   # (the buffers of step j - 1 are complete once step j's successor... is not yet launched: look now)
                            h.sync()
                            check(1 - k, n, (conc, timers, j - 1))
                n = h.lsi_count_wait((steps - 1) % 2, cap)
                h.sync()
                check((steps - 1) % 2, n, (conc, timers, "last"))
            h.set_option("timers", 1)
            # a plain query behind them
            half_e, half_p = q.n_edges // 2, q.n_points // 3
            n2 = h.lsi_query(0, 1, 0, half_e, cap, pairs[0])
            h.pip_query(0, 1, None, 0, half_p, closest[0], faces[0])
            want_half = want_pairs[want_pairs[:, 1] < half_e]
            assert n2 == len(want_half)
            assert np.array_equal(oracle.sort_pairs(pairs[0].to_host(np.uint32, 2 * n2).reshape(-1, 2).copy()), want_half)
            assert np.array_equal(closest[0].to_host(np.uint32)[:half_p], want_eids[:half_p])
        with pytest.raises(_capi.RayJoinError):
            h.lsi_count_wait(0, cap)  # nothing in flight
    finally:
        h.close()


@pytest.mark.parametrize("split", [0, 1])
def test_records_by_either_form_equal_the_oracle(oracle, split):
    """rj_lsi_points: k_lsi_points_gcd alone ("lsi_points_split" 0) and k_lsi_points + k_lsi_points_gcd over the pairs
    the gcd-free leg declines (1) leave the same 48-byte records as the oracle -- on a nested pair at map magnitudes
    (coordinates ~2^44: exact hits on shared vertices AND coordinates within 2^-7 of an integer, so the two-kernel form
    uses both of its legs), on the small integer lattice and in the +-2^46 corner."""
    cases = [("nested", maps.Context([synth.standin("USCounty", 0.04), synth.standin("NestedBlockGroup", 0.04)]).load())]
    for name, extreme in (("lattice", False), ("corner", True)):
        c = maps.Context([None, None])
        c.maps = [maps.ScaledMap.from_segments(0, synth.adversarial_segments(300, 7, 3, extreme)),
                  maps.ScaledMap.from_segments(1, synth.adversarial_segments(300, 7, 4, extreme))]
        cases.append((name, c))
    for name, ctx in cases:
        dctx = ops.DeviceContext(ctx).LoadToDevice()
        dctx.BuildIndex(0)
        h = dctx.handle
        h.set_option("lsi_points_split", split)
        m0, m1 = _omap(oracle, ctx.maps[0]), _omap(oracle, ctx.maps[1])
        want = oracle.lsi_brute(m0, m1)
        cap = len(want) + 64
        pairs, recs = h.alloc(8 * cap), h.alloc(48 * cap)
        n = h.lsi_query(0, 1, 0, ctx.maps[1].n_edges, cap, pairs)
        assert n == len(want) and n > 100, name
        h.lsi_points(pairs, n, recs)
        assert h.get_option("lsi_points_last_split") == split
        got = recs.to_host(_capi.XSECT_DTYPE, n)
        ref = oracle.lsi_points(m0, m1, np.ascontiguousarray(got["eid"]))
        for f in ("x_num", "x_den", "y_num", "y_den"):
            assert np.array_equal(got[f], ref[f]), (name, f)
        if split:
            left = h.get_option("lsi_points_gcd_pairs")
            assert 0 <= left < n
            if name == "nested":
                assert 0 < left < n // 8, (left, n)  # both legs worked, the gcd is the exception
            if name == "lattice":
                assert left == 0  # every product exact, every quotient far from the next integer or ON it
        # ... and behind an asynchronous query, the count read on the device
        h.lsi_query_async(0, 1, 0, ctx.maps[1].n_edges, cap, pairs)
        h.lsi_points_async(pairs, cap, recs)
        assert h.lsi_query_finish(cap) == n
        got2 = recs.to_host(_capi.XSECT_DTYPE, n)
        ref2 = oracle.lsi_points(m0, m1, np.ascontiguousarray(got2["eid"]))
        assert np.array_equal(got2["x_num"], ref2["x_num"]) and np.array_equal(got2["y_num"], ref2["y_num"]), name
        dctx.close()


def test_timers_off_changes_nothing_but_the_timers(oracle, lattice_pair):
    """"timers" 0: the stage timers are not recorded (rj_last_ms keeps the last recorded values), results are the same;
    while "pip_concurrent" 2 is still trying schedules they are recorded regardless -- it decides by them."""
    ctx, dctx = lattice_pair
    h = dctx.handle
    m0, m1 = _omap(oracle, ctx.maps[0]), _omap(oracle, ctx.maps[1])
    want = oracle.lsi_grid(m0, m1, 512)
    cap = 2 * len(want)
    pairs = h.alloc(8 * cap)
    n_pts = ctx.maps[1].n_points
    closest, faces = h.alloc(4 * n_pts), h.alloc(4 * n_pts)
    assert h.lsi_query(0, 1, 0, ctx.maps[1].n_edges, cap, pairs) == len(want)
    before = h.last_ms(_capi.RJ_T_LSI_KERNEL)
    assert before > 0
    try:
        h.set_option("timers", 0)
        assert h.get_option("timers") == 0
        assert h.lsi_query(0, 1, 0, ctx.maps[1].n_edges, cap, pairs) == len(want)
        assert h.last_ms(_capi.RJ_T_LSI_KERNEL) == before  # not re-recorded
        h.pip_query(0, 1, None, 0, n_pts, closest, faces)
        assert np.array_equal(closest.to_host(np.uint32), oracle.pip_brute(m0, 1, ctx.maps[1].pts))
        with pytest.raises(_capi.RayJoinError):
            h.set_option("timers", 2)
    finally:
        h.set_option("timers", 1)
    assert h.lsi_query(0, 1, 0, ctx.maps[1].n_edges, cap, pairs) == len(want)
    assert h.last_ms(_capi.RJ_T_LSI_KERNEL) > 0
