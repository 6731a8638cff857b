"""PIP queries over a CALLER-OWNED point array (pts_dev != NULL -- the reference's interface, PIP::Query(Stream&, int,
ArrayView<point_t>), src/app/pip.h:23, src/run_query.cu:346,441-443): after the first query over an array the handle
enqueues without a host round trip, whatever the array holds by then.  One buffer, contents that CHANGE between queries
-- chain-order vertices, a shuffle of them, another shuffle, uniform random points, chain order again -- every result
against the oracle; the same interleaved with LSI queries in flight under "pip_concurrent" 1/2 (the caller-array query
pairs like a map-owned one); two buffers alternating; more buffers than the handle remembers; rj_invalidate."""
import numpy as np
import pytest

from rayjoin_amd import _capi, maps, synth

pytestmark = pytest.mark.gpu


def _omap(oracle, m):
    return oracle.Map(m.pts, m.row_index, m.left, m.right)


@pytest.fixture(scope="module")
def setup(oracle):
    ctx = maps.Context([synth.lattice_map(24, 90, 41), synth.lattice_map(70, 30, 42)]).load()
    b, q = ctx.maps
    m0, m1 = _omap(oracle, b), _omap(oracle, q)
    rng = np.random.default_rng(5)
    n = q.n_points
    lo, hi = q.pts.min(axis=0), q.pts.max(axis=0)
    contents = {
        "chain order": q.pts,
        "shuffled": q.pts[rng.permutation(n)],
        "shuffled again": q.pts[rng.permutation(n)],
        "uniform": np.stack([rng.integers(lo[0], hi[0], n), rng.integers(lo[1], hi[1], n)], 1).astype(np.int64),
        "on base vertices": b.pts[rng.integers(0, b.n_points, n)],
    }
    want = {k: oracle.pip_grid(m0, 0, np.ascontiguousarray(v), 256) for k, v in contents.items()}
    want_pairs = oracle.lsi_grid(m0, m1, 256)["eid"]
    return b, q, m0, contents, want, want_pairs


def _handle(b, q):
    h = _capi.Handle(0)
    h.upload_map(0, b.pts, b.row_index, b.left, b.right)
    h.upload_map(1, q.pts, q.row_index, q.left, q.right)
    h.build_lbvh(0)
    return h


SEQUENCE = ["chain order", "chain order", "shuffled", "shuffled", "shuffled", "shuffled again", "uniform", "uniform", "chain order",
            "chain order", "chain order", "on base vertices", "shuffled", "chain order"]


def test_one_buffer_changing_contents(setup):
    b, q, m0, contents, want, _ = setup
    n = q.n_points
    h = _handle(b, q)
    try:
        buf, closest, faces = h.alloc(16 * n), h.alloc(4 * n), h.alloc(4 * n)
        ordered = []
        for step, name in enumerate(SEQUENCE):
            buf.from_host(contents[name])
            closest.from_host(np.full(n, 0xDEADBEEF, dtype=np.uint32))
            h.pip_query(0, 1, buf, 0, n, closest, faces, sync=False)
            h.sync()
            ordered.append(h.get_option("query_last_ordered"))
            e = closest.to_host(np.uint32)
            assert np.array_equal(e, want[name]), (step, name)
            assert np.array_equal(faces.to_host(np.int32), m0.face_ids(want[name])), (step, name)
        # the array in chain order runs as it is; once shuffled contents have been SEEN (one query late) the array is
        # re-ordered, and stays so (a permutation of sorted contents is still a fine order for chain-order contents)
        assert ordered[0] == 0 and ordered[1] == 0
        assert 1 in ordered[3:6], ordered
        # a prefix of the buffer is another array (pointer, n): learned separately
        m = n // 3 + 17
        buf.from_host(contents["shuffled"])
        h.pip_query(0, 1, buf, 0, m, closest, None)
        assert np.array_equal(closest.to_host(np.uint32)[:m], want["shuffled"][:m])
        h.invalidate()
        buf.from_host(contents["chain order"])
        h.pip_query(0, 1, buf, 0, n, closest, faces)
        assert h.get_option("query_last_ordered") == 0
        assert np.array_equal(closest.to_host(np.uint32), want["chain order"])
    finally:
        h.close()


@pytest.mark.parametrize("conc", [1, 2])
def test_caller_array_pairs_with_an_lsi_query_in_flight(setup, oracle, conc):
    b, q, m0, contents, want, want_pairs = setup
    n = q.n_points
    h = _handle(b, q)
    try:
        h.set_option("pip_concurrent", conc)
        cap = 4 * len(want_pairs) + 64
        pairs, xs = h.alloc(8 * cap), h.alloc(48 * cap)
        buf, closest, faces = h.alloc(16 * n), h.alloc(4 * n), h.alloc(4 * n)
        for step, name in enumerate(SEQUENCE):
            buf.from_host(contents[name])
            closest.from_host(np.full(n, 0xDEADBEEF, dtype=np.uint32))
            h.lsi_query_async(0, 1, 0, q.n_edges, cap, pairs)
            h.pip_query(0, 1, buf, 0, n, closest, faces, sync=False)
            h.lsi_points_async(pairs, cap, xs)
            k = h.lsi_query_finish(cap)
            h.sync()
            assert k == len(want_pairs)
            got = pairs.to_host(np.uint32, 2 * k).reshape(-1, 2)
            assert np.array_equal(oracle.sort_pairs(got.copy()), want_pairs), (conc, step)
            assert np.array_equal(closest.to_host(np.uint32), want[name]), (conc, step, name)
            assert np.array_equal(faces.to_host(np.int32), m0.face_ids(want[name])), (conc, step, name)
    finally:
        h.close()


def test_more_buffers_than_the_handle_remembers(setup):
    b, q, m0, contents, want, _ = setup
    n = q.n_points
    h = _handle(b, q)
    try:
        names = list(contents)
        bufs = [h.alloc(16 * n).from_host(contents[k]) for k in names] + [h.alloc(16 * n).from_host(contents["shuffled"])]
        names.append("shuffled")
        closest = h.alloc(4 * n)
        for rnd in range(3):
            for k, d in zip(names, bufs):   # six arrays round robin over four remembered sets: every query a "first sight"
                h.pip_query(0, 1, d, 0, n, closest, None, sync=False)
                h.sync()
                assert np.array_equal(closest.to_host(np.uint32), want[k]), (rnd, k)
    finally:
        h.close()
