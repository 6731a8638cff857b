"""Randomised pairs against the oracle (the reference's own tests are the sample datasets; this is the net
under everything the index builds differently by map shape): ring maps of random density and ring size (shared leaves,
skyline, column index with the strip width the map asks for), gaussian polygons, jittered lattices and nested
refinements, in both roles and with both query-map ids -- LSI pairs, closest edges and face ids (lsi_lbvh.h:27-98,
pip_lbvh.h:25-142 semantics) bit for bit, through the default options and through the walk forced onto the ring maps."""
import numpy as np
import pytest

from rayjoin_amd import _capi, maps, synth

pytestmark = pytest.mark.gpu


def _maps(rng):
    """a generator's output that is a planar graph the loader accepts (chains of >= 2 points: planar_graph.h:71)"""
    while True:
        g = _draw(rng)
        probe = maps.Context([g]).load().maps[0]
        if probe.n_edges > 0 and np.all(np.diff(probe.row_index) >= 2) and np.all(np.isfinite(probe.pts.astype(np.float64))):
            return g


def _draw(rng):
    kind = rng.integers(0, 5)
    seed = int(rng.integers(1, 1 << 30))
    if kind == 0:
        n = int(rng.integers(50, 4000))
        return synth.ring_map(n, n * int(rng.integers(4, 40)), seed, fill=float(rng.uniform(0.05, 0.6)), sigma=float(rng.uniform(0.3, 1.3)))
    if kind == 1:
        return synth.gaussian_polygons(int(rng.integers(200, 8000)), seed, polysize=float(rng.uniform(0.002, 0.05)))
    if kind == 2:
        return synth.lattice_map(int(rng.integers(3, 40)), int(rng.integers(2, 60)), seed)
    if kind == 3:
        G, k = int(rng.integers(3, 12)), int(rng.integers(3, 20))
        return synth.nested_refinement(synth.lattice_map(G, k, seed), G, k, int(rng.integers(2, 5)), int(rng.integers(3, 12)), seed=seed + 1)
    n = int(rng.integers(2000, 30000))   # many tiny rings: most leaves shared, narrow strips
    return synth.ring_map(n, n * 4, seed, fill=0.5, sigma=0.4)


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15, 16])
def test_random_pairs_equal_the_oracle(oracle, seed):
    rng = np.random.default_rng(seed)
    for _ in range(4):
        ctx = maps.Context([_maps(rng), _maps(rng)]).load()
        m = ctx.maps
        om = [oracle.Map(x.pts, x.row_index, x.left, x.right) for x in m]
        want_pairs = oracle.lsi_brute(om[0], om[1])
        h = _capi.Handle(0)
        try:
            for i in (0, 1):
                h.upload_map(i, m[i].pts, m[i].row_index, m[i].left, m[i].right)
            cap = max(1024, 2 * len(want_pairs))
            pairs = h.alloc(8 * cap)
            for base in (0, 1):
                q = m[1 - base]
                want = oracle.pip_brute(om[base], 1 - base, q.pts)
                closest, faces = h.alloc(4 * max(1, q.n_points)), h.alloc(4 * max(1, q.n_points))
                for columns in (-1, 0):
                    h.set_option("pip_columns", columns)
                    h.build_lbvh(base)
                    what = (seed, base, columns, m[0].n_edges, m[1].n_edges)
                    n = h.lsi_query(base, 1 - base, 0, q.n_edges, cap, pairs)
                    h.sort_pairs(pairs, n)
                    got = pairs.to_host(np.uint32, 2 * n).reshape(-1, 2)
                    assert np.array_equal(got, want_pairs), what   # (the pair is (map-0 edge, map-1 edge) whichever map is indexed)
                    h.pip_query(base, 1 - base, None, 0, q.n_points, closest, faces)
                    assert np.array_equal(closest.to_host(np.uint32)[:q.n_points], want), what
                    assert np.array_equal(faces.to_host(np.int32)[:q.n_points], om[base].face_ids(want)), what
        finally:
            h.close()
