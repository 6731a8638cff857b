""""pip_exact_stream": the exact kernel of a PIP query on its own stream, beside the next query's walk.  Several steps
in flight, every step a DIFFERENT query (another range of the query map's vertices, another output array), nothing
synchronised in between -- each step's answers must equal the synchronous query's, bit for bit; a walk-less query in
between and a switch of the option mid-way must not disturb that."""
import numpy as np
import pytest

from rayjoin_amd import _capi, maps, synth

pytestmark = pytest.mark.gpu


def _setup(n0=(60, 400), n1=(140, 170)):
    ctx = maps.Context([synth.lattice_map(n0[0], n0[1], 171), synth.lattice_map(n1[0], n1[1], 172)]).load()
    b, q = ctx.maps
    h = _capi.Handle(0)
    h.upload_map(0, b.pts, b.row_index, b.left, b.right)
    h.upload_map(1, q.pts, q.row_index, q.left, q.right)
    h.build_lbvh(0)
    return h, b, q


def _ranges(q, steps, m):
    span = q.n_points - m
    return [((k * 7919) % span, m) for k in range(steps)]


@pytest.mark.parametrize("concurrent", [1, 2])
def test_steps_in_flight_equal_synchronous_queries(concurrent):
    h, b, q = _setup()
    steps, m = 14, q.n_points * 3 // 5
    cap = int(0.5 * (b.n_edges + q.n_edges))
    pairs = [h.alloc(8 * cap) for _ in range(2)]
    want = []
    c, f = h.alloc(4 * m), h.alloc(4 * m)
    h.set_option("pip_walk", 2)
    for p0, n in _ranges(q, steps, m):
        h.pip_query(0, 1, None, p0, n, c, f)
        want.append((c.to_host(np.uint32, n).copy(), f.to_host(np.int32, n).copy()))
    assert h.get_plan()["pip"]["passes"] == 3 and h.get_option("pip_rest") >= 0
    n_x = h.lsi_query(0, 1, 0, q.n_edges, cap, pairs[0])
    outs = [(h.alloc(4 * m), h.alloc(4 * m)) for _ in range(steps)]
    h.set_option("pip_concurrent", concurrent)
    h.set_option("pip_exact_stream", 1)
    assert h.get_option("pip_exact_stream") == 1
    for k, (p0, n) in enumerate(_ranges(q, steps, m)):
        h.lsi_query_async(0, 1, 0, q.n_edges, cap, pairs[k % 2])
        if k == 9:
            h.set_option("pip_walk", 0)  # one walk-less query among them: k_pip behind the exact kernels in flight
        h.pip_query(0, 1, None, p0, n, outs[k][0], outs[k][1], sync=False)
        if k == 9:
            h.set_option("pip_walk", 2)
        h.lsi_count_async(k % 2)
        if k > 0:
            assert h.lsi_count_wait(1 - k % 2, cap) == n_x
    assert h.lsi_count_wait((steps - 1) % 2, cap) == n_x
    h.sync()
    for k in range(steps):
        assert np.array_equal(outs[k][0].to_host(np.uint32, m), want[k][0]), k
        assert np.array_equal(outs[k][1].to_host(np.int32, m), want[k][1]), k
    assert h.get_plan()["pip"]["stream"].startswith("second")
    # the option off again mid-way (drains), the same steps once more on the walk's own stream
    h.set_option("pip_exact_stream", 0)
    for k, (p0, n) in enumerate(_ranges(q, 4, m)):
        h.lsi_query_async(0, 1, 0, q.n_edges, cap, pairs[k % 2])
        h.pip_query(0, 1, None, p0, n, outs[k][1], outs[k][0], sync=False)
        assert h.lsi_query_finish(cap) == n_x
    h.sync()
    for k in range(4):
        assert np.array_equal(outs[k][1].to_host(np.uint32, m), want[k][0]), k
        assert np.array_equal(outs[k][0].to_host(np.int32, m), want[k][1]), k
    h.close()


def test_queries_that_share_their_output_arrays_run_one_after_the_other():
    """Round 5's advisor: with "pip_exact_stream" 1 the walk of query k + 1 ran beside the exact kernel of query k, and both
    store into closest / face -- every host wrapper here hands every query ONE array.  The handle now remembers the last
    query's arrays and makes such a pair wait: different point ranges into one array, all in flight, then the array holds the
    LAST query's answers and nothing of the earlier ones."""
    h, b, q = _setup()
    steps, m = 8, q.n_points * 3 // 5
    cap = int(0.5 * (b.n_edges + q.n_edges))
    pairs = [h.alloc(8 * cap) for _ in range(2)]
    c, f = h.alloc(4 * m), h.alloc(4 * m)
    h.set_option("pip_walk", 2)
    rng = _ranges(q, steps, m)
    h.pip_query(0, 1, None, rng[-1][0], m, c, f)
    want = (c.to_host(np.uint32, m).copy(), f.to_host(np.int32, m).copy())
    h.set_option("pip_concurrent", 1)
    h.set_option("pip_exact_stream", 1)
    for rep in range(3):
        for k, (p0, n) in enumerate(rng):
            h.lsi_query_async(0, 1, 0, q.n_edges, cap, pairs[k % 2])
            h.pip_query(0, 1, None, p0, n, c, f, sync=False)
            h.lsi_count_async(k % 2)
            if k > 0:
                h.lsi_count_wait(1 - k % 2, cap)
        h.lsi_count_wait((steps - 1) % 2, cap)
        h.sync()
        assert np.array_equal(c.to_host(np.uint32, m), want[0]), rep
        assert np.array_equal(f.to_host(np.int32, m), want[1]), rep
    h.close()


def test_rejects_other_values():
    h, _, _ = _setup((8, 20), (9, 14))
    with pytest.raises(_capi.RayJoinError):
        h.set_option("pip_exact_stream", 2)
    h.close()
