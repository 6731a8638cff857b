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


def test_ring_map_is_isolated_disjoint_rings():
    """synth.ring_map (the lake-shaped stand-ins): every chain is a closed ring, rings are simple (star-shaped) and
    pairwise disjoint (bounding boxes do not even touch), edge counts are heavy-tailed with the requested mean, the
    big rings are big -- and two such maps with different seeds do cross each other."""
    from rayjoin_amd import synth
    g = synth.ring_map(1500, 1500 * 12, 3, big_share=0.01)
    ri = g.row_index.astype(np.int64)
    P = g.points
    assert g.n_chains == 1500 and abs(g.n_edges / g.n_chains - 12) < 1.0
    assert np.array_equal(P[ri[:-1]], P[ri[1:] - 1])                      # closed
    nv = np.diff(ri) - 1
    assert nv.min() >= 3 and nv.max() > 8 * np.median(nv)                 # heavy tail
    lo = np.array([P[a:b].min(axis=0) for a, b in zip(ri[:-1], ri[1:])])
    hi = np.array([P[a:b].max(axis=0) for a, b in zip(ri[:-1], ri[1:])])
    apart = (lo[:, None, 0] > hi[None, :, 0]) | (hi[:, None, 0] < lo[None, :, 0]) | (lo[:, None, 1] > hi[None, :, 1]) | (hi[:, None, 1] < lo[None, :, 1])
    np.fill_diagonal(apart, True)
    assert apart.all()                                                    # disjoint boxes: disjoint rings
    size = (hi - lo).max(axis=1)
    assert np.corrcoef(np.log(nv), np.log(size))[0, 1] > 0.3              # more edges, larger ring
    # star-shaped around the box centre: the vertices' angles around it are monotone (one wrap)
    for c in (0, int(np.argmax(nv)), 7):
        Q = P[ri[c]:ri[c + 1] - 1]
        ctr = (lo[c] + hi[c]) / 2
        ang = np.unwrap(np.arctan2(Q[:, 1] - ctr[1], Q[:, 0] - ctr[0]))
        assert np.all(np.diff(ang) > 0) or np.all(np.diff(ang) < 0)


def test_two_ring_maps_intersect(oracle):
    from rayjoin_amd import maps, synth
    ctx = maps.Context([synth.ring_map(800, 9000, 1), synth.ring_map(700, 9000, 2)]).load()
    m0 = oracle.Map(ctx.maps[0].pts, ctx.maps[0].row_index, ctx.maps[0].left, ctx.maps[0].right)
    m1 = oracle.Map(ctx.maps[1].pts, ctx.maps[1].row_index, ctx.maps[1].left, ctx.maps[1].right)
    a, b = oracle.lsi_brute(m0, m1), oracle.lsi_grid(m0, m1, 64)["eid"]
    assert len(a) > 20 and np.array_equal(a, b)
    inside = oracle.pip_grid(m0, 0, ctx.maps[1].pts, 64)
    assert 0 < (inside != 0xFFFFFFFF).sum() < len(inside)
