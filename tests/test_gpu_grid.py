"""-mode=grid on the device against the CPU oracle's restatement of the same grid
algorithm -- and against the LBVH path: the three must agree bit for bit."""
import numpy as np
import pytest

from rayjoin_amd import _capi, maps, ops, synth

pytestmark = pytest.mark.gpu


def _omap(oracle, m):
    return oracle.Map(m.pts, m.row_index, m.left, m.right)


@pytest.fixture(scope="module")
def pair():
    ctx = maps.Context([synth.lattice_map(9, 120, 11), synth.lattice_map(21, 50, 12)]).load()
    dctx = ops.DeviceContext(ctx).LoadToDevice()
    dctx.BuildIndex(0)
    dctx.BuildIndex(1)
    yield ctx, dctx
    dctx.close()


@pytest.mark.parametrize("g", [7, 64, 512, 2048])
def test_grid_lsi_equals_oracle_grid_and_lbvh(oracle, pair, g):
    ctx, dctx = pair
    m0, m1 = _omap(oracle, ctx.maps[0]), _omap(oracle, ctx.maps[1])
    want = oracle.lsi_grid(m0, m1, g)
    assert dctx.BuildGrid(g) > 0
    lsi = ops.LSIGrid(dctx)
    lsi.Init(4 * len(want) + 16)
    assert lsi.Query() == len(want)
    assert np.array_equal(lsi.get_pairs(), want["eid"])
    x = lsi.get_xsects()
    assert np.array_equal(x["x_num"], want["x_num"]) and np.array_equal(x["y_num"], want["y_num"])
    lb = ops.LSILBVH(dctx)
    lb.Init(4 * len(want) + 16)
    lb.Query(1)
    assert np.array_equal(lb.get_pairs(), want["eid"])
    small = ops.LSIGrid(dctx)
    small.Init(5)
    with pytest.raises(_capi.QueueOverflow) as ei:
        small.Query()
    assert ei.value.n_found == len(want)


@pytest.mark.parametrize("g", [1, 97, 2048])
@pytest.mark.parametrize("qm", [1, 0])
def test_grid_pip_equals_oracle_grid_and_lbvh(oracle, pair, g, qm):
    ctx, dctx = pair
    base = 1 - qm
    mb = _omap(oracle, ctx.maps[base])
    pts = ctx.maps[qm].pts
    want = oracle.pip_grid(mb, base, pts, g)
    dctx.BuildGrid(g, (base,))
    pip = ops.PIPGrid(dctx)
    pip.Init(len(pts))
    pip.Query(qm)
    assert np.array_equal(pip.get_closest_eids(), want)
    assert np.array_equal(pip.get_face_ids(), mb.face_ids(want))
    lb = ops.PIPLBVH(dctx)
    lb.Init(len(pts))
    lb.Query(qm)
    assert np.array_equal(lb.get_closest_eids(), want)
    # free-standing points and a point sub-range
    rng = np.random.default_rng(5)
    free = rng.integers(-(1 << 45), 1 << 45, size=(3000, 2))
    pip.Query(qm, query_points=free)
    assert np.array_equal(pip.get_closest_eids(), oracle.pip_grid(mb, base, free, g))
    pip.Query(qm, point_range=(100, 1100))
    assert np.array_equal(pip.get_closest_eids(), want[100:1100])


def test_grid_adversarial_lattice_ties(oracle):
    """integer lattice with shared vertices, collinear overlaps, points on vertices and edges: the
    SoS substitutions and the visit-order tie rule, grid vs oracle grid"""
    a = synth.adversarial_segments(3000, 6, 91)
    b = synth.adversarial_segments(3500, 6, 92)
    ctx = maps.Context([None, None])
    ctx.maps = [maps.ScaledMap.from_segments(0, a), maps.ScaledMap.from_segments(1, b)]
    dctx = ops.DeviceContext(ctx).LoadToDevice()
    o0, o1 = oracle.Map(a), oracle.Map(b)
    for g in (3, 256):
        dctx.BuildGrid(g)
        want = oracle.lsi_grid(o0, o1, g)
        lsi = ops.LSIGrid(dctx)
        lsi.Init(4 * len(want) + 16)
        lsi.Query()
        assert np.array_equal(lsi.get_pairs(), want["eid"])
        pts = np.concatenate([a.reshape(-1, 2)[:2000], b.reshape(-1, 2)[:2000]])
        for qm in (1, 0):
            base = 1 - qm
            pip = ops.PIPGrid(dctx)
            pip.Init(len(pts))
            pip.Query(qm, query_points=pts)
            assert np.array_equal(pip.get_closest_eids(), oracle.pip_grid((o0, o1)[base], base, pts, g)), (g, qm)
    dctx.close()


def test_grid_call_order_errors():
    ctx = maps.Context([synth.lattice_map(3, 5, 1), synth.lattice_map(4, 4, 2)]).load()
    dctx = ops.DeviceContext(ctx).LoadToDevice()
    h = dctx.handle
    buf = h.alloc(64)
    with pytest.raises(_capi.RayJoinError):
        h.lsi_query_grid(4, buf)  # no grids yet
    h.build_grid(0, 16)
    h.build_grid(1, 32)
    with pytest.raises(_capi.RayJoinError):
        h.lsi_query_grid(4, buf)  # different grid sizes
    with pytest.raises(_capi.RayJoinError):
        h.build_grid(0, 0)
    with pytest.raises(_capi.RayJoinError):
        h.pip_query_grid(0, 0, None, 0, 1, buf)
    dctx.close()
