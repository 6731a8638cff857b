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

Both bit-exact against the oracle's -mode=grid (it finishes in seconds on the box's cores)."""
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
