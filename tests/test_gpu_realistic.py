"""Full-size parity on geometry that is not a plain lattice:

* NESTED: the query map refines the base map the way census block groups nest in counties -- every
  base boundary is also a query boundary, 80 % of those sub-chains vertex for vertex (7.1 M base
  edges meet an identical query edge and two end-sharing neighbours: millions of candidate pairs go
  through the simulation-of-simplicity and identical-edge branches, lsi.h:42-100), the rest as a
  differently generalised copy that crosses the base chain every few segments; a quarter of the
  query vertices lie exactly ON base vertices (pip.h:44-93 tie rules).
* CLUSTERED: the reference's own synthetic scalability workload (misc/gen_polys.sh, gaussian,
  5 M x 1 M polygons; expr/draw/scal_lsi_synthetic/gaussian_1000000.log), chains in generation
  order, i.e. spatially shuffled.

* LAKE-SHAPED (round 4): the topology of the reference's water-body / lake / park inputs -- millions of ISOLATED,
  disjoint closed rings with log-normal edge counts (10.5 per chain for the WaterBodies statistics, 35 and 44.5 for
  lakes and parks, single rings of thousands of edges), clustered, with fractal-looking shores (synth.ring_map).
  Nothing to stitch, most chains far shorter than a leaf (several neighbouring rings share one), a third of a
  lattice's vertices with NOTHING above them (the skyline answers those), ring maps as base and as query.

All bit-exact against the oracle's -mode=grid (it finishes in seconds on the box's cores)."""
import numpy as np
import pytest

from rayjoin_amd import _capi, maps, synth

pytestmark = pytest.mark.gpu


def _run_pair(oracle, base_name, query_name, gsize, min_xsects):
    oracle.lib().rjo_set_num_threads(16)
    ctx = maps.Context([synth.standin(base_name), synth.standin(query_name)]).load()
    base, query = ctx.maps
    h = _capi.Handle(0)
    h.upload_map(0, base.pts, base.row_index, base.left, base.right)
    h.upload_map(1, query.pts, query.row_index, query.left, query.right)
    h.build_lbvh(0)
    cap = int(0.2 * (base.n_edges + query.n_edges))
    pairs = h.alloc(8 * cap)
    n = h.lsi_query(0, 1, 0, query.n_edges, cap, pairs)
    h.sort_pairs(pairs, n)
    got = pairs.to_host(np.uint32, 2 * n).reshape(-1, 2)
    xs_dev = h.alloc(48 * n)
    h.lsi_points(pairs, n, xs_dev)
    xs = xs_dev.to_host(_capi.XSECT_DTYPE, n)
    closest = h.alloc(4 * query.n_points)
    faces = h.alloc(4 * query.n_points)
    h.pip_query(0, 1, None, 0, query.n_points, closest, faces)
    eids, fids = closest.to_host(np.uint32), faces.to_host(np.int32)
    m0 = oracle.Map(base.pts, base.row_index, base.left, base.right)
    m1 = oracle.Map(query.pts, query.row_index, query.left, query.right)
    want = oracle.lsi_grid(m0, m1, gsize, cap=cap)
    assert len(want) == n >= min_xsects
    assert np.array_equal(want["eid"], got)
    assert np.array_equal(want["x_num"], xs["x_num"]) and np.array_equal(want["y_num"], xs["y_num"])
    we = oracle.pip_grid(m0, 0, query.pts, gsize)
    assert np.array_equal(we, eids)
    assert np.array_equal(m0.face_ids(we), fids)
    del m1, want
    # role symmetry (the predicate's operand order is fixed, so the pair set does not depend on the indexed side)
    h.build_lbvh(1)
    n2 = h.lsi_query(1, 0, 0, base.n_edges, cap, pairs)
    h.sort_pairs(pairs, n2)
    assert n2 == n and np.array_equal(pairs.to_host(np.uint32, 2 * n2).reshape(-1, 2), got)
    h.close()
    return ctx, n, eids


def test_nested_county_blockgroup_full_size(oracle):
    ctx, n, eids = _run_pair(oracle, "USCounty", "NestedBlockGroup", 2048, 300000)
    # the nesting is real: a large share of the query vertices ARE base vertices
    base_pts = np.unique(ctx.maps[0].pts.view([("x", "<i8"), ("y", "<i8")]))
    q = ctx.maps[1].pts.view([("x", "<i8"), ("y", "<i8")]).ravel()
    assert np.isin(q, base_pts).mean() > 0.15


def test_gaussian_polygons_full_size(oracle):
    ctx, n, eids = _run_pair(oracle, "Gaussian5M", "Gaussian1M", 4096, 100000)
    assert ctx.maps[0].n_edges > 30_000_000 and ctx.maps[1].n_edges > 6_000_000
    assert (eids != _capi.MISS_EID).mean() > 0.05


def test_lake_shaped_base_full_size(oracle):
    """WaterBodiesLike (2.44 M rings, 25.7 M edges) as the BASE map of a join with the BlockGroup lattice: LSI pairs and
    records, the closest edge / face of all 29.7 M lattice vertices (a third of them misses), role symmetry."""
    ctx, n, eids = _run_pair(oracle, "WaterBodiesLike", "BlockGroup", 4096, 100000)
    assert ctx.maps[0].n_chains > 2_400_000 and ctx.maps[0].n_edges > 25_000_000
    miss = (eids == _capi.MISS_EID).mean()
    assert 0.2 < miss < 0.6, miss


def test_lakes_parks_like_pip_full_size(oracle):
    """LakesLike (1.91 M rings, 67.4 M edges) x ParksLike (0.6 M rings, 26.9 M edges): both maps ring-shaped, the query
    vertices in ring order (spatially shuffled: the handle re-orders them) -- every vertex's closest edge and face."""
    import psutil
    avail = psutil.virtual_memory().available
    if avail < 40 << 30:
        pytest.skip("needs ~30 GB of host memory for the 67 M-segment map and the oracle's grid")
    oracle.lib().rjo_set_num_threads(16)
    ctx = maps.Context([synth.standin("LakesLike"), synth.standin("ParksLike")]).load()
    base, query = ctx.maps
    pts = np.ascontiguousarray(query.pts if avail > (60 << 30) else query.pts[::13])
    h = _capi.Handle(0)
    h.upload_map(0, base.pts, base.row_index, base.left, base.right)
    h.build_lbvh(0)
    assert h.get_option("leaf_order_used0") == 1 and h.get_option("skyline_used0") == 1
    assert h.get_option("leaf_slots0") <= 1.5 * base.n_edges  # (short rings share leaves)
    d = h.alloc(16 * len(pts)).from_host(pts)
    closest, faces = h.alloc(4 * len(pts)), h.alloc(4 * len(pts))
    h.pip_query(0, 1, d, 0, len(pts), closest, faces)
    m0 = oracle.Map(base.pts, base.row_index, base.left, base.right)
    we = oracle.pip_grid(m0, 0, pts, 4096)
    eids = closest.to_host(np.uint32)
    assert np.array_equal(we, eids)
    assert np.array_equal(m0.face_ids(we), faces.to_host(np.int32))
    assert 0.05 < (eids != _capi.MISS_EID).mean() < 0.95
    h.close()
