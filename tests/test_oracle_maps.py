"""Oracle self-consistency on maps: host scaling == oracle scaling, grid == brute force
(LSI and PIP), and the documented wrap regime of the reference's intersection point."""
import numpy as np

from rayjoin_amd import maps, synth


def _omap(oracle, m):
    return oracle.Map(m.pts, m.row_index, m.left, m.right)


def _pair(G0=6, k0=80, G1=13, k1=40):
    ctx = maps.Context([synth.lattice_map(G0, k0, 11), synth.lattice_map(G1, k1, 12)]).load()
    return ctx


def test_scaling_matches_oracle(oracle):
    ctx = _pair()
    s = oracle.make_scaling(*ctx.bb)
    for im in range(2):
        assert (oracle.scale_points(s, ctx.planar_graphs[im].points) == ctx.maps[im].pts).all()
        assert np.array_equal(oracle.unscale_points(s, ctx.maps[im].pts), ctx.scaling.unscale(ctx.maps[im].pts))
    assert ctx.maps[0].pts.min() >= maps.INTERNAL_MIN and ctx.maps[0].pts.max() <= maps.INTERNAL_MAX
    assert s.imin == maps.INTERNAL_MIN and s.irange == maps.INTERNAL_RANGE == 140737488355327


def test_edge_layout(oracle):
    ctx = _pair(3, 4, 2, 5)
    m = ctx.maps[0]
    om = _omap(oracle, m)
    p1 = m.edge_p1()
    segs = m.segments()
    assert om.ne == m.n_edges == m.n_points - m.n_chains
    for eid in range(m.n_edges):
        e = om.edge(eid)
        assert e["eid"] == eid and e["p1"] == p1[eid] and e["p2"] == p1[eid] + 1
        x1, y1, x2, y2 = (int(v) for v in segs[eid])
        a, b = y1 - y2, x2 - x1
        c = -x1 * a - y1 * b
        if b < 0:
            a, b, c = -a, -b, -c
        assert (e["a"], e["b"], e["c"]) == (a, b, c) and e["b"] >= 0


def test_grid_equals_brute_lsi_pip(oracle):
    ctx = _pair()
    m0, m1 = _omap(oracle, ctx.maps[0]), _omap(oracle, ctx.maps[1])
    brute = oracle.lsi_brute(m0, m1)
    assert len(brute) > 50
    for g in (1, 64, 2048):
        gr = oracle.lsi_grid(m0, m1, g)
        assert np.array_equal(gr["eid"], brute)
    pts = ctx.maps[1].pts
    pb = oracle.pip_brute(m0, 1, pts)
    assert (pb != oracle.MISS).sum() > 1000 and (pb == oracle.MISS).sum() > 0
    for g in (1, 64, 2048):
        assert np.array_equal(oracle.pip_grid(m0, 0, pts, g), pb)
    # the other direction (query map 0 against base map 1) flips the SoS signs
    pb0 = oracle.pip_brute(m1, 0, ctx.maps[0].pts)
    assert np.array_equal(oracle.pip_grid(m1, 1, ctx.maps[0].pts, 64), pb0)


def test_adversarial_grid_equals_brute(oracle):
    for seed in (1, 2, 3):
        a = synth.adversarial_segments(300, 6, seed)
        b = synth.adversarial_segments(300, 6, seed + 100)
        m0, m1 = oracle.Map(a), oracle.Map(b)
        brute = oracle.lsi_brute(m0, m1)
        assert len(brute) > 100
        gr = oracle.lsi_grid(m0, m1, 2048)
        assert np.array_equal(gr["eid"], brute)
        pts = np.random.default_rng(seed).integers(-7, 8, size=(500, 2))
        for base, bid in ((m0, 0), (m1, 1)):
            assert np.array_equal(oracle.pip_grid(base, bid, pts, 2048), oracle.pip_brute(base, 1 - bid, pts))


def test_reference_wrap_regime_documented(oracle):
    """Edges longer than 2^39 units: the reference's numx/numy wrap in int128, the clamped point
    lands in a wrong cell and -mode=grid drops hits (DESIGN.md 'validity domain').  The predicate
    itself never wraps, so brute force (and the HIP path) still reports every true pair."""
    ctx = maps.Context([synth.lattice_map(8, 6, 11), synth.lattice_map(13, 3, 12)]).load()
    seg = ctx.maps[0].segments()
    assert np.abs(seg[:, 2:] - seg[:, :2]).max() > (1 << 40)
    m0, m1 = _omap(oracle, ctx.maps[0]), _omap(oracle, ctx.maps[1])
    brute = oracle.lsi_brute(m0, m1)
    gr = oracle.lsi_grid(m0, m1, 64)
    assert len(gr) < len(brute)
    assert set(map(tuple, gr["eid"].tolist())) <= set(map(tuple, brute.tolist()))
