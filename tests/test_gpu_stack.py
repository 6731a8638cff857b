"""Traversal-stack stress: a 4-level tree (WaterBodies stand-in, 24.4 M segments,
382 k leaf blocks > 64^3) queried by groups that contain domain-spanning diagonals and long random
GenerateLSIQueries segments (run_query.cu:102-144, large -gen_t) next to ordinary short ones, so
that every child of every node overlaps the group and the LDS stacks reach their deepest.
Bit-exact against the oracle: grid for the short segments, brute force for the long ones (in the
reference's wrap regime -- edges longer than 2^39.5 units -- the grid drops true intersections,
DESIGN.md section 2, so the predicate itself is the answer there).  Plus the fault path: with the
stacks shrunk (instrumented kernels only) an overflow is reported as RJ_E_INTERNAL, not corrupted."""
import numpy as np
import pytest

from rayjoin_amd import _capi, maps, ops, synth

pytestmark = pytest.mark.gpu


def _wide_queries(ctx, n_short, n_long, seed):
    """-> (int64[n,4] segments, bool[n] is_long).  Short ones: a slice of a BlockGroup-like lattice.
    Long ones: GenerateLSIQueries with gen_t = 40 degrees, plus 8 domain-spanning diagonals; six of
    the diagonals share one 64-query group, the others sit alone among short segments."""
    sm = ctx.maps[1].segments()[:n_short]
    rnd = synth.generate_lsi_queries(ctx.bb, ctx.scaling, n_long, 40.0, seed).segments()
    # (the generator does not keep p2 inside the bounding box, run_query.cu:129-130; the C ABI only
    # accepts the scaled range, so the far ends are pulled back onto its border)
    rnd = np.clip(rnd, -(1 << 46) + 1, (1 << 46) - 1)
    big = (1 << 46) - 5
    span = np.array([[-big, -big + 11, big, big - 7], [-big, big - 3, big, -big + 2],
                     [-big, -big // 2, big, big // 3], [-big + 9, big // 5, big - 1, -big // 7],
                     [-big // 2, -big, big // 2 + 1, big], [big // 3, -big, -big // 3 - 2, big],
                     [-big, 1234567, big, -7654321], [-999, -big, 1001, big]], dtype=np.int64)
    n = n_short + n_long + len(span)
    segs = np.empty((n, 4), dtype=np.int64)
    is_long = np.zeros(n, dtype=bool)
    pos_span = np.array([640 + 10, 640 + 11, 640 + 12, 640 + 13, 640 + 14, 640 + 15, 64 * 40 + 3, 64 * 90 + 9])
    rest = np.setdiff1d(np.arange(n), pos_span)
    pos_rnd = rest[np.linspace(0, len(rest) - 1, n_long).astype(np.int64)]
    is_long[pos_span] = True
    is_long[pos_rnd] = True
    segs[pos_span] = span
    segs[pos_rnd] = rnd
    segs[~is_long] = sm[: n - n_long - len(span)]
    return segs, is_long


def test_four_level_tree_with_domain_spanning_queries(oracle):
    oracle.lib().rjo_set_num_threads(16)
    ctx = maps.Context([synth.standin("WaterBodies"), synth.lattice_map(40, 33, 9)]).load()  # (joint bbox / scaling)
    base = ctx.maps[0]
    assert base.n_edges > 64 ** 3 * 64  # 4 levels
    segs, is_long = _wide_queries(ctx, 20000, 160, seed=77)
    qmap = maps.ScaledMap.from_segments(1, segs.reshape(-1, 2))
    h = _capi.Handle(0)
    h.upload_map(0, base.pts, base.row_index, base.left, base.right)
    h.upload_map(1, qmap.pts, qmap.row_index, qmap.left, qmap.right)
    h.build_lbvh(0)
    h.build_lbvh(1)
    # the answer: grid for the short queries, brute force for the long ones
    m0 = oracle.Map(base.pts, base.row_index, base.left, base.right)
    ids_s, ids_l = np.flatnonzero(~is_long), np.flatnonzero(is_long)
    ms = oracle.Map(np.ascontiguousarray(segs[ids_s].reshape(-1, 2)))
    ml = oracle.Map(np.ascontiguousarray(segs[ids_l].reshape(-1, 2)))
    ws = oracle.lsi_grid(m0, ms, 4096)["eid"].astype(np.int64)
    wl = oracle.lsi_brute(m0, ml, cap=4_000_000).astype(np.int64)
    ws[:, 1] = ids_s[ws[:, 1]]
    wl[:, 1] = ids_l[wl[:, 1]]
    want = oracle.sort_pairs(np.concatenate([ws, wl]).astype(np.uint32))
    assert len(wl) > 20000 and len(ws) > 100  # the diagonals alone cross thousands of chains
    cap = 2 * len(want) + 1024
    pairs = h.alloc(8 * cap)
    for order in (0, 1):  # as given (wide and narrow lanes share groups) and Morton re-ordered
        h.set_option("query_order", order)
        for base_id in (0, 1):  # both roles: the wide segments as queries, then as indexed base segments
            n = h.lsi_query(base_id, 1 - base_id, 0, (qmap if base_id == 0 else base).n_edges, cap, pairs)
            h.sort_pairs(pairs, n)
            got = pairs.to_host(np.uint32, 2 * n).reshape(-1, 2)
            assert n == len(want) and np.array_equal(got, want), (order, base_id)
    # PIP: 64 unrelated columns per group (no re-ordering) -- every top-level child is wanted by some lane
    pts = synth.generate_pip_queries(ctx.bb, ctx.scaling, 30000, seed=5)
    pts[:64, 1] = -(1 << 46) + 3  # one group of rays that start below the whole map
    d = h.alloc(16 * len(pts)).from_host(pts)
    closest = h.alloc(4 * len(pts))
    faces = h.alloc(4 * len(pts))
    we = oracle.pip_grid(m0, 0, pts, 4096)
    for order in (0, 1):
        h.set_option("query_order", order)
        h.pip_query(0, 1, d, 0, len(pts), closest, faces)
        assert np.array_equal(closest.to_host(np.uint32), we), order
        assert np.array_equal(faces.to_host(np.int32), m0.face_ids(we))
    h.close()


def test_stack_overflow_is_reported_not_corrupted(oracle):
    """Shrink the stacks (honoured by the instrumented kernels only): the query fails with
    RJ_E_INTERNAL; with the real capacity the same handle answers correctly again."""
    ctx = maps.Context([synth.lattice_map(12, 100, 3), synth.lattice_map(30, 20, 4)]).load()
    dctx = ops.DeviceContext(ctx).LoadToDevice()
    dctx.BuildIndex(0)
    h = dctx.handle
    q = ctx.maps[1]
    m0 = oracle.Map(ctx.maps[0].pts, ctx.maps[0].row_index, ctx.maps[0].left, ctx.maps[0].right)
    m1 = oracle.Map(q.pts, q.row_index, q.left, q.right)
    want = oracle.lsi_brute(m0, m1)
    cap = 4 * len(want)
    pairs = h.alloc(8 * cap)
    closest = h.alloc(4 * q.n_points)
    h.set_option("stats", 1)
    h.set_debug_option("stack_cap", 1)
    with pytest.raises(_capi.RayJoinError) as ei:
        h.lsi_query(0, 1, 0, q.n_edges, cap, pairs)
    assert ei.value.code == _capi.RJ_E_INTERNAL and "k_lsi" in str(ei.value)
    with pytest.raises(_capi.RayJoinError) as ei:
        h.pip_query(0, 1, None, 0, q.n_points, closest, None)
    assert ei.value.code == _capi.RJ_E_INTERNAL and "k_pip" in str(ei.value)
    h.pip_query(0, 1, None, 0, q.n_points, closest, None, sync=False)
    with pytest.raises(_capi.RayJoinError) as ei:
        h.sync()  # the async form reports at the next sync point
    assert ei.value.code == _capi.RJ_E_INTERNAL
    for stats in (1, 0):
        h.set_option("stats", stats)
        h.set_debug_option("stack_cap", 1 << 30)
        n = h.lsi_query(0, 1, 0, q.n_edges, cap, pairs)
        h.sort_pairs(pairs, n)
        assert np.array_equal(pairs.to_host(np.uint32, 2 * n).reshape(-1, 2), want)
        h.pip_query(0, 1, None, 0, q.n_points, closest, None)
        assert np.array_equal(closest.to_host(np.uint32), oracle.pip_brute(m0, 1, q.pts))
    dctx.close()


@pytest.mark.parametrize("walk_points", [1, 2])
def test_groups_that_outgrow_the_walks_stack_leave_the_walk(oracle, walk_points):
    """The walk's stack is 124 entries whatever the tree's height (rj_kernels.hip, kWalkStack: measured depths stay
    below 94), and a group that would need more hands ALL its points to the rest list, which k_pip_exact's first
    blocks locate with k_pip's worst-case stack.  With the stack shrunk to a few entries most groups take that way
    out: closest edges and face ids (pip_lbvh.h:25-142 semantics) must not change, with one and with two points per
    lane, synchronous and asynchronous."""
    oracle.lib().rjo_set_num_threads(16)
    scale = 0.12 if walk_points == 1 else 0.4  # (two points per lane: from four 128-position groups per resident wave on)
    ctx = maps.Context([synth.standin("USCounty", scale), synth.standin("BlockGroup", scale)]).load()
    b, q = ctx.maps
    m0 = oracle.Map(b.pts, b.row_index, b.left, b.right)
    want = oracle.pip_grid(m0, 0, q.pts, 512)
    want_faces = m0.face_ids(want)
    h = _capi.Handle(0)
    try:
        h.upload_map(0, b.pts, b.row_index, b.left, b.right)
        h.upload_map(1, q.pts, q.row_index, q.left, q.right)
        h.build_lbvh(0)
        h.set_option("pip_walk", 2)
        h.set_option("pip_walk_points", walk_points)
        closest, faces = h.alloc(4 * q.n_points), h.alloc(4 * q.n_points)
        left = {}
        for cap in (0, 6, 2, 1):
            h.set_debug_option("walk_stack", cap)
            for sync in (True, False):
                closest.from_host(np.full(q.n_points, 0xDEADBEEF, dtype=np.uint32))
                faces.from_host(np.full(q.n_points, -7, dtype=np.int32))
                h.pip_query(0, 1, None, 0, q.n_points, closest, faces, sync=sync)
                h.sync()
                assert h.get_option("pip_last_walk_points") == walk_points
                assert np.array_equal(closest.to_host(np.uint32), want), (cap, sync)
                assert np.array_equal(faces.to_host(np.int32), want_faces), (cap, sync)
            left[cap] = h.get_option("pip_rest")
        # the full stack leaves (next to) nothing over, one entry nearly everything
        assert left[0] < q.n_points // 100
        assert left[1] > q.n_points // 2 and left[1] >= left[2] >= left[6] >= left[0]
    finally:
        h.close()
