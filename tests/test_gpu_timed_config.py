"""The configuration bench.py TIMES, under the oracle at full size (VERDICT r3 "weak" 1b).

Every other full-size oracle test uses the synchronous rj_lsi_query / rj_pip_query: whole-chip grids, one kernel at a
time, the record count read on the host.  The timed step is another regime: rj_lsi_query_async on its share of the
chip (k_lsi2, two query segments per lane) + rj_pip_query_async on the handle's second stream BESIDE it (k_pip_walk2, two
points per lane, then k_pip_exact) + rj_lsi_points_async with the count read on the device, on the schedule
"pip_concurrent" 2 settles on during the reference's five warm-up queries (run_query.cu:292-296).  Here that step runs
verbatim -- bench.py's call sequence, five warm-up pairs, then a pair on the settled schedule -- on USCounty x BlockGroup
(the headline) and USCounty x NestedBlockGroup, and pairs, 48-byte records, closest eids and face ids are compared with
oracle.lsi_grid / pip_grid (lsi_lbvh.h:27-98, pip_lbvh.h:25-142 semantics as -mode=grid computes them); then bench.py's
PIPELINED loop (step k + 1 launched before step k's count is read: rj_lsi_count_async / rj_lsi_count_wait, two sets of
result buffers), whose last two steps are compared the same way."""
import numpy as np
import pytest

from rayjoin_amd import _capi, maps, synth

pytestmark = pytest.mark.gpu


def _step(h, q, cap, pairs, xs, closest, faces):
    """bench.py::run_workload::step at N = 1, call for call"""
    early = h.get_option("pip_schedule") in (1, 2)  # (bench.py asks before the launch, and not again once settled)
    h.lsi_query_async(0, 1, 0, q.n_edges, cap, pairs)
    if early:
        h.pip_query(0, 1, None, 0, q.n_points, closest, faces, sync=False)
    h.lsi_points_async(pairs, cap, xs)
    if not early:
        h.pip_query(0, 1, None, 0, q.n_points, closest, faces, sync=False)
    n = h.lsi_query_finish(cap)
    h.sync()
    return n


def _check(oracle, m0, m1, want, want_e, n, pairs, xs, closest, faces, what):
    assert n == len(want), (what, n, len(want))
    got = pairs.to_host(np.uint32, 2 * n).reshape(-1, 2)
    order = np.lexsort((got[:, 1], got[:, 0]))
    assert np.array_equal(got[order], want["eid"]), what
    rec = xs.to_host(_capi.XSECT_DTYPE, n)
    assert np.array_equal(rec["eid"], got), what               # the records sit beside their pairs, queue order
    rec = rec[order]
    assert np.array_equal(rec["x_num"], want["x_num"]) and np.array_equal(rec["y_num"], want["y_num"]), what
    assert np.all(rec["x_den"] == 1) and np.all(rec["y_den"] == 1), what
    e = closest.to_host(np.uint32)
    assert np.array_equal(e, want_e), what
    assert np.array_equal(faces.to_host(np.int32), m0.face_ids(want_e)), what


# (CrossingZipcode: the query map whose intersection density is the published County x Zipcode pair's, 3.5 % of the query
#  segments -- synth._crossing_zipcode; 0.65 % on the headline's independent lattices, 7.5 % on the nested refinement)
@pytest.mark.parametrize("query_name,min_x", [("BlockGroup", 100_000), ("NestedBlockGroup", 1_000_000), ("CrossingZipcode", 700_000)])
def test_the_timed_step_equals_the_oracle_at_full_size(oracle, query_name, min_x):
    oracle.lib().rjo_set_num_threads(16)
    ctx = maps.Context([synth.standin("USCounty"), synth.standin(query_name)]).load()
    b, q = ctx.maps
    m0 = oracle.Map(b.pts, b.row_index, b.left, b.right)
    m1 = oracle.Map(q.pts, q.row_index, q.left, q.right)
    cap = int(0.1 * (b.n_edges + q.n_edges))  # run_query.cu:226-228, -xsect_factor 0.1
    want = oracle.lsi_grid(m0, m1, 2048, cap=cap)
    want_e = oracle.pip_grid(m0, 0, q.pts, 2048)
    assert len(want) > min_x
    del m1
    h = _capi.Handle(0)
    try:
        h.upload_map(0, b.pts, b.row_index, b.left, b.right)
        h.upload_map(1, q.pts, q.row_index, q.left, q.right)
        h.build_lbvh(0)
        h.build_lbvh(0)
        h.set_option("pip_concurrent", 2)
        pairs, xs = h.alloc(8 * cap), h.alloc(48 * cap)
        closest, faces = h.alloc(4 * q.n_points), h.alloc(4 * q.n_points)
        # bench.py's setup: one synchronous query of each kind, then the warm-up steps
        h.lsi_query(0, 1, 0, q.n_edges, cap, pairs)
        h.pip_query(0, 1, None, 0, q.n_points, closest, faces)
        for _ in range(5):
            _step(h, q, cap, pairs, xs, closest, faces)
        auto_choice = h.get_option("pip_schedule")
        assert auto_choice >= 0, "five warm-up pairs settle the schedule"
        if auto_choice != 1:
            # this box measured another schedule faster: the shared one is the regime to put under the oracle all the same
            h.set_option("pip_concurrent", 1)
            _step(h, q, cap, pairs, xs, closest, faces)

        def wipe():
            closest.from_host(np.full(q.n_points, 0xDEADBEEF, dtype=np.uint32))
            faces.from_host(np.full(q.n_points, -7, dtype=np.int32))
            pairs.from_host(np.zeros(1 << 20, dtype=np.uint32))

        # --- the timed regime: stage timers off (as in three of four timed steps), settled shared schedule
        for timers in (0, 1):
            wipe()
            h.set_option("timers", timers)
            n = _step(h, q, cap, pairs, xs, closest, faces)
            assert h.get_option("pip_schedule") == 1
            assert h.get_option("lsi_last_segments") == 2 and h.get_option("pip_last_walk_points") == 2
            assert h.get_option("pip_last_passes") == 3
            assert h.get_option("lsi_share_blocks") < 4 * 256 and h.get_option("pip_share_blocks") < 8 * 256  # (both on a share of the chip)
            _check(oracle, m0, None, want, want_e, n, pairs, xs, closest, faces, (query_name, "timers", timers))
        h.set_option("timers", 1)

        # --- bench.py::pipelined_steps: no host sync at the end of a step, two buffer sets, timers off
        pairs2, xs2 = h.alloc(8 * cap), h.alloc(48 * cap)
        closest2, faces2 = h.alloc(4 * q.n_points), h.alloc(4 * q.n_points)
        bufs = [(pairs, xs, closest, faces), (pairs2, xs2, closest2, faces2)]
        wipe()
        closest2.from_host(np.full(q.n_points, 0xDEADBEEF, dtype=np.uint32))
        h.set_option("timers", 0)
        steps, counts = 6, {}
        for j in range(steps):
            k = j % 2
            P, X, C_, F = bufs[k]
            h.lsi_query_async(0, 1, 0, q.n_edges, cap, P)
            early = h.get_option("pip_schedule") in (1, 2)
            if early:
                h.pip_query(0, 1, None, 0, q.n_points, C_, F, sync=False)
            h.lsi_points_async(P, cap, X)
            h.lsi_count_async(k)
            if not early:
                h.pip_query(0, 1, None, 0, q.n_points, C_, F, sync=False)
            if j > 0:
                counts[j - 1] = h.lsi_count_wait(1 - k, cap)
        counts[steps - 1] = h.lsi_count_wait((steps - 1) % 2, cap)
        h.sync()
        h.set_option("timers", 1)
        assert h.get_option("lsi_last_segments") == 2 and h.get_option("pip_last_walk_points") == 2
        for j in (steps - 2, steps - 1):   # the last step into either buffer set
            P, X, C_, F = bufs[j % 2]
            _check(oracle, m0, None, want, want_e, counts[j], P, X, C_, F, (query_name, "pipelined", j))
        assert all(n == len(want) for n in counts.values())
        # ... and a plain step behind the pipelined ones
        wipe()
        n = _step(h, q, cap, pairs, xs, closest, faces)
        _check(oracle, m0, None, want, want_e, n, pairs, xs, closest, faces, (query_name, "after the pipelined steps"))
    finally:
        h.close()
