"""rj_get_plan: the handle's decisions are inspectable, carry the epoch they were taken in, stay once settled, and are
taken again after rj_upload_map / rj_build_lbvh / another query size."""
import numpy as np
import pytest

from rayjoin_amd import _capi, maps, synth

pytestmark = pytest.mark.gpu


def _pairs(h, q, cap, pairs, closest, faces, n, npts=None):
    npts = q.n_points if npts is None else npts
    for _ in range(n):
        h.lsi_query_async(0, 1, 0, q.n_edges, cap, pairs)
        h.pip_query(0, 1, None, 0, npts, closest, faces, sync=False)
        h.lsi_query_finish(cap)
        h.sync()


def test_plan_reports_and_rederives():
    ctx = maps.Context([synth.lattice_map(40, 300, 71), synth.lattice_map(90, 130, 72)]).load()
    b, q = ctx.maps
    h = _capi.Handle(0)
    h.upload_map(0, b.pts, b.row_index, b.left, b.right)
    h.upload_map(1, q.pts, q.row_index, q.left, q.right)
    h.build_lbvh(0)
    cap = int(0.5 * (b.n_edges + q.n_edges))
    pairs, closest, faces = h.alloc(8 * cap), h.alloc(4 * q.n_points), h.alloc(4 * q.n_points)
    p0 = h.get_plan()
    assert p0["lsi"] is None and p0["pip"] is None and p0["index"][0]["built"] and not p0["index"][1]["built"]
    assert p0["schedule"]["choice"] == "turns"  # "pip_concurrent" 0: nothing to decide
    # a synchronous query of each kind: full grids, main stream
    n = h.lsi_query(0, 1, 0, q.n_edges, cap, pairs)
    h.pip_query(0, 1, None, 0, q.n_points, closest, faces)
    p = h.get_plan()
    assert p["lsi"]["segments"] == q.n_edges and p["lsi"]["kernel"] in ("k_lsi", "k_lsi2", "k_lsix", "k_lsi2x") and p["lsi"]["blocks"] > 0
    assert not p["lsi"]["paired_with_pip"] and p["lsi"]["current"]
    assert p["pip"]["points"] == q.n_points and p["pip"]["stream"] == "main" and p["pip"]["passes"] == 3
    assert p["pip"]["first_pass"]["kernel"].startswith("k_pip_walk") and p["pip"]["second_pass"]["kernel"] == "k_pip_exact"
    assert p["records"] is None
    # auto schedule: undecided for four pairs, settled from then on, and it STAYS (same text twenty pairs later)
    h.set_option("pip_concurrent", 2)
    e0 = h.get_plan()["epoch"]
    assert e0 > p["epoch"] and h.get_plan()["schedule"]["choice"] == "undecided"
    _pairs(h, q, cap, pairs, closest, faces, 6)
    s1 = h.get_plan()
    assert s1["schedule"]["settled"] and s1["schedule"]["in_force"] and s1["schedule"]["settled_in_epoch"] == s1["epoch"] == e0
    assert s1["schedule"]["trials"] == 4 and s1["lsi"]["paired_with_pip"] and s1["lsi"]["schedule"] == s1["schedule"]["choice"]
    assert all(v > 0 for v in s1["schedule"]["best_span_us"].values())
    _pairs(h, q, cap, pairs, closest, faces, 20)
    s2 = h.get_plan()
    assert s2["schedule"] == s1["schedule"] and s2["lsi"] == s1["lsi"] and s2["pip"]["first_pass"] == s1["pip"]["first_pass"]
    if s1["schedule"]["choice"] == "shared":
        assert s1["pip"]["on_shared_grid"] and s1["lsi"]["on_shared_grid"] and s1["lsi"]["blocks"] <= s1["schedule"]["shared_grids"]["lsi_blocks"]
    # a new index: every decision is taken again
    h.build_lbvh(0)
    r = h.get_plan()
    assert r["epoch"] > s2["epoch"] and r["schedule"]["choice"] == "undecided" and not r["lsi"]["current"] and not r["pip"]["current"]
    _pairs(h, q, cap, pairs, closest, faces, 6)
    r2 = h.get_plan()
    assert r2["schedule"]["settled"] and r2["schedule"]["settled_in_epoch"] == r2["epoch"] and r2["lsi"]["current"]
    # another query size (a shard): decided again
    _pairs(h, q, cap, pairs, closest, faces, 2, npts=q.n_points // 3)
    r3 = h.get_plan()
    assert r3["epoch"] > r2["epoch"] and r3["pip"]["points"] == q.n_points // 3
    # a new map: again; and the results never depended on any of it
    h.upload_map(1, q.pts, q.row_index, q.left, q.right)
    r4 = h.get_plan()
    assert r4["epoch"] > r3["epoch"] and r4["schedule"]["choice"] == "undecided"
    n2 = h.lsi_query(0, 1, 0, q.n_edges, cap, pairs)
    assert n2 == n
    h.close()


def test_plan_says_why_the_first_pass_differs():
    g0 = synth.ring_map(3000, 40000, seed=81)
    g1 = synth.lattice_map(60, 100, 82)
    ctx = maps.Context([g0, g1]).load()
    b, q = ctx.maps
    h = _capi.Handle(0)
    h.upload_map(0, b.pts, b.row_index, b.left, b.right)
    h.upload_map(1, q.pts, q.row_index, q.left, q.right)
    h.build_lbvh(0)
    closest = h.alloc(4 * q.n_points)
    h.pip_query(0, 1, None, 0, q.n_points, closest, None)
    p = h.get_plan()
    assert p["index"][0]["columns"] and p["pip"]["first_pass"]["kernel"] == "k_pip_strip" and "column index" in p["pip"]["why"]
    assert "closed rings" in p["index"][0]["columns_why"]
    # round 6, the measured rule: a lattice of SHORT chains (mean below 16 edges) gets the column index too, one of long chains
    # does not, and the plan says on what grounds either way
    for k, want in ((9, True), (40, False)):
        s = maps.Context([synth.lattice_map(50, k, 83), g1]).load().maps[0]
        h.upload_map(0, s.pts, s.row_index, s.left, s.right)
        h.build_lbvh(0)
        h.pip_query(0, 1, None, 0, q.n_points, closest, None)
        p = h.get_plan()
        ix = p["index"][0]
        assert ix["columns"] is want and abs(ix["mean_chain_edges"] - k) < 0.01 and ("short chains" in ix["columns_why"]) is want, ix
        assert p["pip"]["first_pass"]["kernel"].startswith("k_pip_strip" if want else "k_pip_walk"), p["pip"]   # (a small query set walks one point per lane)
    h.upload_map(0, b.pts, b.row_index, b.left, b.right)
    h.build_lbvh(0)
    h.set_option("pip_walk", 0)
    h.pip_query(0, 1, None, 0, q.n_points, closest, None)
    p = h.get_plan()
    assert p["pip"]["passes"] == 1 and p["pip"]["first_pass"]["kernel"] == "k_pip" and "pip_walk" in p["pip"]["why"]
    h.close()
