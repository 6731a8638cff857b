"""The polyline runs rj_build_lbvh cuts ON THE DEVICE (rj_stitch.hip, "leaf_order" 1) against the host pass of rounds 1-3
(tests/hosttwin/stitch_twin.cc::stitch_runs, the reference): the same pieces and runs, array for array -- lattices,
chains cut short, isolated rings, closed loops of paired chains, hubs, adversarial integer chains -- and no read-back of the
map on the way (the first build of a 0.8 M-chain map stays within milliseconds)."""
import ctypes as C
import time

import numpy as np
import pytest

from rayjoin_amd import _capi, maps, synth
from test_stitch_twin import cut_chains, edge_begin, scaled, split_rings, stitch_lib

pytestmark = pytest.mark.gpu


def reference_runs(L, pts, row_index, cap):
    pts = np.ascontiguousarray(pts, dtype=np.int64)
    eb = np.ascontiguousarray(edge_begin(row_index))
    nc, ne = len(eb) - 1, int(eb[-1])
    pb = np.zeros(2 * nc + ne // cap + 2, dtype=np.uint32)
    pl = np.zeros_like(pb)
    rf = np.zeros(nc + ne // cap + 3, dtype=np.uint32)
    nr, npc = C.c_uint64(0), C.c_uint64(0)
    u32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint32))
    assert L.stitch_ref(pts.ctypes.data_as(C.POINTER(C.c_int64)), u32(eb), nc, cap, u32(pb), u32(pl), u32(rf), C.byref(nr), C.byref(npc)) == 0
    return pb[:npc.value], pl[:npc.value], rf[:nr.value + 1]


def device_runs(pts, row_index, cap=0, columns=-1):
    pts = np.ascontiguousarray(pts, dtype=np.int64)
    row_index = np.ascontiguousarray(row_index, dtype=np.uint32)
    nc = len(row_index) - 1
    h = _capi.Handle(0)
    try:
        h.upload_map(0, pts, row_index, np.zeros(nc, dtype=np.int64), np.ones(nc, dtype=np.int64))
        h.set_option("leaf_order", 1)
        h.set_option("pip_columns", columns)
        if cap:
            h.set_debug_option("run_cap", cap)
        h.build_lbvh(0)
        return h.map_runs(0), (h.get_option("stitch_rounds"), h.get_option("stitch_loop_ends"))
    finally:
        h.close()


def default_cap(row_index, columns=-1):
    """rj_build_lbvh: full runs of 64 edges -- since round 6 also where the chains average fewer than 16 edges, because such a
    map gets the column index for its upward rays and the leaves serve LSI alone; runs of 32 there only when the caller
    forbids that index ("pip_columns" 0) and the rays walk the tree"""
    eb = edge_begin(row_index)
    nc = len(eb) - 1
    return 32 if columns == 0 and nc and int(eb[-1]) // nc < 16 else 64


def same(ref, got, what):
    for a, b, name in zip(ref, got, ("piece_begin", "piece_len", "run_first")):
        assert len(a) == len(b), (what, name, len(a), len(b))
        assert np.array_equal(a, b), (what, name, int(np.argmax(a != b)))


@pytest.fixture(scope="module")
def lib():
    return stitch_lib()


@pytest.mark.parametrize("G,k,cap", [(7, 150, 0), (30, 7, 0), (30, 7, 64), (40, 1, 0), (3, 1, 8), (12, 33, 17), (200, 5, 0)])
def test_lattices(lib, G, k, cap):
    pts, rows = scaled(synth.lattice_map(G, k, 100 + G))
    got, (rounds, loops) = device_runs(pts, rows, cap)
    same(reference_runs(lib, pts, rows, cap or default_cap(rows)), got, ("lattice", G, k, cap))
    if not cap and k < 16:   # (the short-chain lattice with the column index forbidden: the round-4 cap)
        got0, _ = device_runs(pts, rows, 0, columns=0)
        same(reference_runs(lib, pts, rows, default_cap(rows, 0)), got0, ("lattice, pip_columns 0", G, k))
        assert default_cap(rows, 0) == 32
    assert loops == 0 and rounds <= 12


@pytest.mark.parametrize("piece", [15, 3, 1])
def test_cut_chains_are_stitched_back(lib, piece):
    pts, rows = cut_chains(*scaled(synth.lattice_map(9, 60, 5)), piece)
    got, _ = device_runs(pts, rows)
    same(reference_runs(lib, pts, rows, default_cap(rows)), got, ("cut", piece))


def test_isolated_rings(lib):
    pts, rows = scaled(synth.gaussian_polygons(20000, 5))
    got, (rounds, loops) = device_runs(pts, rows)
    same(reference_runs(lib, pts, rows, default_cap(rows)), got, "rings")
    assert loops == 0 and len(got[2]) - 1 == len(rows) - 1


@pytest.mark.parametrize("parts,seed", [(2, 1), (5, 3), (9, 4)])
def test_closed_loops_of_chains(lib, parts, seed):
    rng = np.random.default_rng(seed)
    pts, rows = split_rings(*scaled(synth.gaussian_polygons(3000, 10 + seed, maxseg=24, polysize=0.01)), parts, rng)
    got, (rounds, loops) = device_runs(pts, rows, 8)
    assert loops > 0
    same(reference_runs(lib, pts, rows, 8), got, ("loops", parts))


def test_loops_open_paths_and_rings_together(lib):
    rng = np.random.default_rng(7)
    a_pts, a_rows = split_rings(*scaled(synth.gaussian_polygons(500, 3, maxseg=30, polysize=0.02)), 4, rng)
    b_pts, b_rows = cut_chains(*scaled(synth.lattice_map(11, 40, 8)), 9)
    c_pts, c_rows = scaled(synth.gaussian_polygons(300, 4))
    pts = np.concatenate([b_pts, a_pts, c_pts])
    rows = np.concatenate([b_rows, b_rows[-1] + a_rows[1:], b_rows[-1] + a_rows[-1] + c_rows[1:]]).astype(np.uint32)
    for cap in (64, 5):
        got, (rounds, loops) = device_runs(pts, rows, cap)
        assert loops > 0
        same(reference_runs(lib, pts, rows, cap), got, ("mixed", cap))


@pytest.mark.parametrize("seed", range(4))
def test_adversarial_integer_chains(lib, seed):
    pts, rows, _, _ = synth.adversarial_chains(3000, 2 + seed % 4, 6 + seed, seed)
    got, _ = device_runs(pts, rows, 4 + seed)
    same(reference_runs(lib, pts, rows, 4 + seed), got, ("adversarial", seed))


def test_blockgroup_first_build_is_a_device_build(lib):
    """0.87 M chains, 28.8 M segments: the runs of the reference pass, and a first build (runs + allocations + index)
    that is milliseconds, not the 170 ms of the host pass"""
    g = synth.standin("BlockGroup", 1.0)
    ctx = maps.Context([g, None]).load()
    m = ctx.maps[0]
    want = reference_runs(lib, m.pts, m.row_index, 64)
    h = _capi.Handle(0)
    try:
        h.upload_map(0, m.pts, m.row_index, m.left, m.right)
        t0 = time.perf_counter()
        h.build_lbvh(0)
        wall = (time.perf_counter() - t0) * 1e3
        first = h.last_ms(_capi.RJ_T_BUILD)
        same(want, h.map_runs(0), "BlockGroup")
        assert h.get_option("leaf_order_used0") == 1
        h.build_lbvh(0)
        again = h.last_ms(_capi.RJ_T_BUILD)
        print("BlockGroup first build %.2f ms (wall %.2f), rebuild %.2f ms" % (first, wall, again))
        assert first < 40.0 and wall < 60.0, (first, wall)
    finally:
        h.close()
