"""Polyline runs, the data-parallel form (rayjoin_amd/csrc/rj_stitch.h -- the source the HIP kernels of rj_stitch.hip run)
against the host pass of rounds 1-3 (tests/hosttwin/stitch_twin.cc::stitch_runs, kept as the reference): the SAME
pieces and runs, array for array -- lattices (rows and columns stitched through the junctions), chains cut into short
pieces, isolated rings, rings split into several chains (closed loops of paired chains), hubs, adversarial integer
chains dense in shared end points, and the full-size stand-ins' chain counts at reduced edge counts.
The GPU side of the same comparison is tests/test_gpu_stitch.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from rayjoin_amd import maps, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "hosttwin", "stitch_twin.cc")
HDR = os.path.join(ROOT, "rayjoin_amd", "csrc", "rj_stitch.h")
OUT = os.path.join(ROOT, "tests", "hosttwin", "_build", "libstitch_twin.so")


def stitch_lib():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if not os.path.exists(OUT) or os.path.getmtime(OUT) < max(os.path.getmtime(SRC), os.path.getmtime(HDR)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-pthread",
                               "-I", os.path.dirname(HDR), "-o", OUT, SRC])
    L = C.CDLL(OUT)
    u32p, u64p, i64p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_int64)
    L.stitch_ref.argtypes = [i64p, u32p, C.c_uint64, C.c_uint64, u32p, u32p, u32p, u64p, u64p]
    L.stitch_twin.argtypes = [i64p, u32p, C.c_uint64, C.c_uint64, u32p, u32p, u32p, u64p, u64p, u64p]
    return L


@pytest.fixture(scope="module")
def lib():
    return stitch_lib()


def edge_begin(row_index):
    return (np.asarray(row_index, dtype=np.int64) - np.arange(len(row_index), dtype=np.int64)).astype(np.uint32)


def run_both(L, pts, row_index, cap):
    """-> (reference, twin), each (piece_begin, piece_len, run_first), + the twin's stats"""
    pts = np.ascontiguousarray(pts, dtype=np.int64)
    eb = np.ascontiguousarray(edge_begin(row_index))
    nc = len(eb) - 1
    ne = int(eb[-1])
    out = []
    stats = np.zeros(4, dtype=np.uint64)
    for which in ("ref", "twin"):
        pb = np.zeros(2 * nc + ne // cap + 2, dtype=np.uint32)
        pl = np.zeros_like(pb)
        rf = np.zeros(nc + ne // cap + 3, dtype=np.uint32)
        nr, npc = C.c_uint64(0), C.c_uint64(0)
        u32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint32))
        args = [pts.ctypes.data_as(C.POINTER(C.c_int64)), u32(eb), nc, cap, u32(pb), u32(pl), u32(rf), C.byref(nr), C.byref(npc)]
        if which == "ref":
            assert L.stitch_ref(*args) == 0
        else:
            assert L.stitch_twin(*args, stats.ctypes.data_as(C.POINTER(C.c_uint64))) == 0
        out.append((pb[:npc.value].copy(), pl[:npc.value].copy(), rf[:nr.value + 1].copy()))
    return out[0], out[1], stats


def assert_same(ref, twin, what):
    for a, b, name in zip(ref, twin, ("piece_begin", "piece_len", "run_first")):
        assert len(a) == len(b), (what, name, len(a), len(b))
        assert np.array_equal(a, b), (what, name, int(np.argmax(a != b)))


def check_cover(pb, pl, rf, ne, cap):
    """every edge in exactly one piece, every run non-empty and at most `cap` edges"""
    seen = np.zeros(ne + 1, dtype=np.int64)
    np.add.at(seen, pb, 1)
    np.add.at(seen, pb + pl, -1)
    assert np.all(np.cumsum(seen)[:ne] == 1)
    csum = np.concatenate([[0], np.cumsum(pl, dtype=np.int64)])
    per_run = csum[rf[1:]] - csum[rf[:-1]]
    assert per_run.min() >= 1 and per_run.max() <= cap


def scaled(g, other=None):
    ctx = maps.Context([g, other if other is not None else synth.lattice_map(2, 2, 99)]).load()
    m = ctx.maps[0]
    return m.pts, m.row_index


def cut_chains(pts, row_index, piece):
    """every chain cut into chains of <= `piece` edges that continue each other (digitised junction to junction)"""
    P, rows = [], [0]
    for c in range(len(row_index) - 1):
        a, b = int(row_index[c]), int(row_index[c + 1])
        Q = pts[a:b]
        for s0 in range(0, len(Q) - 1, piece):
            part = Q[s0:s0 + piece + 1]
            P.append(part)
            rows.append(rows[-1] + len(part))
    return np.concatenate(P), np.array(rows, dtype=np.uint32)


def split_rings(pts, row_index, parts, rng, reverse_some=True, shuffle=True):
    """closed chains cut into `parts` chains each, some of them reversed, in shuffled file order: the pieces pair up
    into closed loops of chains that no free end leads into"""
    P = []
    for c in range(len(row_index) - 1):
        Q = pts[int(row_index[c]):int(row_index[c + 1])]
        n = len(Q) - 1
        cuts = sorted(set([0, n] + list(rng.choice(np.arange(1, n), size=min(parts - 1, n - 1), replace=False))))
        for a, b in zip(cuts[:-1], cuts[1:]):
            part = Q[a:b + 1]
            P.append(part[::-1] if reverse_some and rng.random() < 0.5 else part)
    if shuffle:
        P = [P[i] for i in rng.permutation(len(P))]
    rows = np.concatenate([[0], np.cumsum([len(p) for p in P])]).astype(np.uint32)
    return np.concatenate(P), rows


@pytest.mark.parametrize("G,k,cap", [(7, 150, 64), (16, 70, 64), (30, 7, 32), (30, 7, 64), (5, 200, 64), (40, 1, 64), (3, 1, 8), (12, 33, 17)])
def test_lattices(lib, G, k, cap):
    pts, rows = scaled(synth.lattice_map(G, k, 100 + G))
    ref, twin, stats = run_both(lib, pts, rows, cap)
    assert_same(ref, twin, ("lattice", G, k, cap))
    check_cover(*twin, int(edge_begin(rows)[-1]), cap)
    assert stats[1] == 0  # rows and columns end at the map's border: no closed loops


@pytest.mark.parametrize("piece,cap", [(15, 64), (7, 32), (3, 64), (1, 16)])
def test_cut_chains_are_stitched_back(lib, piece, cap):
    pts, rows = scaled(synth.lattice_map(9, 60, 5))
    pts, rows = cut_chains(pts, rows, piece)
    ref, twin, _ = run_both(lib, pts, rows, cap)
    assert_same(ref, twin, ("cut", piece, cap))
    ne = int(edge_begin(rows)[-1])
    check_cover(*twin, ne, cap)
    assert len(twin[2]) - 1 <= ne / cap * 1.25 + 40  # (whatever the chain length, the runs come out nearly full)


def test_isolated_rings(lib):
    pts, rows = scaled(synth.gaussian_polygons(3000, 5))
    ref, twin, stats = run_both(lib, pts, rows, 64)
    assert_same(ref, twin, "rings")
    assert len(twin[2]) - 1 == len(rows) - 1 and stats[1] == 0  # one run per ring, nothing to rank


@pytest.mark.parametrize("parts,seed", [(2, 1), (3, 2), (5, 3), (9, 4)])
def test_closed_loops_of_chains(lib, parts, seed):
    rng = np.random.default_rng(seed)
    g = synth.gaussian_polygons(800, 10 + seed, maxseg=24, polysize=0.01)
    pts, rows = scaled(g)
    pts, rows = split_rings(pts, rows, parts, rng)
    ref, twin, stats = run_both(lib, pts, rows, 8)
    assert stats[1] > 0, "this case is meant to hold closed loops"
    assert_same(ref, twin, ("loops", parts))
    check_cover(*twin, int(edge_begin(rows)[-1]), 8)


def test_loops_and_open_paths_together(lib):
    rng = np.random.default_rng(7)
    a_pts, a_rows = scaled(synth.gaussian_polygons(500, 3, maxseg=30, polysize=0.02))
    a_pts, a_rows = split_rings(a_pts, a_rows, 4, rng)
    b_pts, b_rows = cut_chains(*scaled(synth.lattice_map(11, 40, 8)), 9)
    c_pts, c_rows = scaled(synth.gaussian_polygons(300, 4))
    pts = np.concatenate([b_pts, a_pts, c_pts])
    rows = np.concatenate([b_rows, b_rows[-1] + a_rows[1:], b_rows[-1] + a_rows[-1] + c_rows[1:]]).astype(np.uint32)
    for cap in (64, 32, 5):
        ref, twin, stats = run_both(lib, pts, rows, cap)
        assert stats[1] > 0
        assert_same(ref, twin, ("mixed", cap))


def test_hubs_and_duplicates(lib):
    """spokes of one point: 12 (paired by straightness), 16 (still paired) and 17, 40 (hubs: left alone); duplicate
    chains; chains of one edge"""
    rng = np.random.default_rng(11)
    P = []
    for n_spokes, centre in ((12, (0, 0)), (16, (10 ** 6, 0)), (17, (0, 10 ** 6)), (40, (10 ** 6, 10 ** 6)), (2, (-10 ** 6, 0)), (3, (0, -10 ** 6))):
        ang = np.sort(rng.uniform(0, 2 * np.pi, n_spokes))
        for a in ang:
            npt = int(rng.integers(2, 6))
            r = np.arange(npt)[:, None] * 1000.0
            line = np.round(np.array(centre)[None, :] + r * np.array([np.cos(a), np.sin(a)])[None, :]).astype(np.int64)
            P.append(line if rng.random() < 0.5 else line[::-1])
    P.append(P[0].copy())        # the same chain twice
    P.append(P[1][::-1].copy())  # ... and one reversed
    rows = np.concatenate([[0], np.cumsum([len(p) for p in P])]).astype(np.uint32)
    pts = np.concatenate(P)
    for cap in (64, 3):
        ref, twin, _ = run_both(lib, pts, rows, cap)
        assert_same(ref, twin, ("hubs", cap))
        check_cover(*twin, int(edge_begin(rows)[-1]), cap)


@pytest.mark.parametrize("seed", range(6))
def test_adversarial_integer_chains(lib, seed):
    """random polylines on a tiny lattice: most end points are shared, many directions tie exactly"""
    pts, rows, _, _ = synth.adversarial_chains(1500, 2 + seed % 4, 6 + seed, seed)
    ref, twin, _ = run_both(lib, pts, rows, 4 + seed)
    assert_same(ref, twin, ("adversarial", seed))


@pytest.mark.parametrize("name,scale", [("USCounty", 1.0), ("WaterBodies", 0.3), ("BlockGroup", 0.4)])
def test_standin_maps(lib, name, scale):
    G, k, seed, bbox = synth.STANDINS[name]
    g = synth.lattice_map(max(2, int(round(G * scale))), k, seed, bbox)
    pts, rows = scaled(g)
    cap = 32 if k < 16 else 64
    ref, twin, stats = run_both(lib, pts, rows, cap)
    assert_same(ref, twin, (name, scale))
    assert stats[0] <= 14  # ranking rounds: log2 of the longest path, not of the chain count
