"""Seeded synthetic CDB maps: jittered-lattice planar subdivisions sized like the
paper's datasets (the real files and the reference's own sample pair are unobtainable here), the
reference's random query workloads (run_query.cu:102-167) and adversarial integer fixtures."""
import numpy as np

from .maps import PlanarGraph, ScaledMap

US_BBOX = (-179.149, -14.5487, 179.778, 71.3905)       # expr/draw/query_lsi/*_lbvh.log:4
CONUS_BBOX = (-179.149, 18.9107, -66.9496, 71.3652)

# name -> (G, k, seed, bbox): SURVEY 8d table
STANDINS = {
    "USCounty": (103, 332, 1, US_BBOX),
    "BlockGroup": (660, 33, 2, US_BBOX),
    "Zipcode": (259, 176, 3, US_BBOX),
    "WaterBodies": (1105, 10, 4, US_BBOX),
    "LakesNA": (977, 35, 5, CONUS_BBOX),
    "ParksNA": (549, 45, 6, CONUS_BBOX),
}


def lattice_map(G, k, seed, bbox=US_BBOX, vertex_jitter=0.3, seg_jitter=0.1):
    """G x G cells over bbox; every lattice edge is one chain of k segments.
    chains = 2*G*(G+1), edges = chains*k.  left/right = adjacent cell ids (1-based, 0 = exterior)."""
    rng = np.random.default_rng(seed)
    x0, y0, x1, y1 = bbox
    cw, ch = (x1 - x0) / G, (y1 - y0) / G
    gx, gy = np.meshgrid(np.arange(G + 1, dtype=np.float64), np.arange(G + 1, dtype=np.float64))
    vx = x0 + (gx + rng.uniform(-vertex_jitter, vertex_jitter, gx.shape)) * cw   # [j, i]
    vy = y0 + (gy + rng.uniform(-vertex_jitter, vertex_jitter, gy.shape)) * ch
    jj, ii = np.meshgrid(np.arange(G + 1), np.arange(G + 1), indexing="ij")
    # horizontal edges (i,j)->(i+1,j): i<G ; vertical edges (i,j)->(i,j+1): j<G
    hm = ii < G
    vm = jj < G
    hi, hj = ii[hm], jj[hm]
    vi, vj = ii[vm], jj[vm]
    # chain order: row-major over lattice nodes, horizontal then vertical edge of each node
    key_h = (hj * (G + 1) + hi) * 2
    key_v = (vj * (G + 1) + vi) * 2 + 1
    P0 = np.concatenate([np.stack([vx[hj, hi], vy[hj, hi]], 1), np.stack([vx[vj, vi], vy[vj, vi]], 1)])
    P1 = np.concatenate([np.stack([vx[hj, hi + 1], vy[hj, hi + 1]], 1),
                         np.stack([vx[vj + 1, vi], vy[vj + 1, vi]], 1)])
    cell = lambda ci, cj: np.where((ci >= 0) & (ci < G) & (cj >= 0) & (cj < G), cj * G + ci + 1, 0)
    left = np.concatenate([cell(hi, hj), cell(vi - 1, vj)])
    right = np.concatenate([cell(hi, hj - 1), cell(vi, vj)])
    order = np.argsort(np.concatenate([key_h, key_v]), kind="stable")
    P0, P1, left, right = P0[order], P1[order], left[order], right[order]
    nch = P0.shape[0]
    t = np.linspace(0.0, 1.0, k + 1)
    d = P1 - P0
    seglen = np.hypot(d[:, 0], d[:, 1]) / k
    perp = np.stack([-d[:, 1], d[:, 0]], 1) / (np.hypot(d[:, 0], d[:, 1])[:, None])
    jit = rng.uniform(-seg_jitter, seg_jitter, (nch, k + 1)) * seglen[:, None]
    jit[:, 0] = 0.0
    jit[:, -1] = 0.0
    pts = P0[:, None, :] + t[None, :, None] * d[:, None, :] + jit[:, :, None] * perp[:, None, :]
    pts = pts.reshape(-1, 2)
    row_index = (np.arange(nch + 1, dtype=np.int64) * (k + 1)).astype(np.uint32)
    first = row_index[:-1].astype(np.int64)
    chains = np.stack([np.arange(nch, dtype=np.int64), first, first + k, left.astype(np.int64),
                       right.astype(np.int64)], 1)
    return PlanarGraph(chains, row_index, pts)


def _ragged(counts):
    """-> (owner of every element, index of every element inside its owner) for ragged rows"""
    counts = np.asarray(counts, dtype=np.int64)
    starts = np.cumsum(counts) - counts
    owner = np.repeat(np.arange(len(counts), dtype=np.int64), counts)
    return owner, np.arange(int(counts.sum()), dtype=np.int64) - np.repeat(starts, counts)


def nested_refinement(base, G, k, s, k2, seed, share=0.8, node_jitter=0.25, seg_jitter=0.1, detach=0.0):
    """A query map NESTED in the lattice map `base` = lattice_map(G, k, ...): every base cell is cut
    into s x s sub-cells and every base boundary is also a boundary of the refinement, the way census
    block groups nest in counties.  Sub-lattice nodes on a base line are vertices OF the base chain;
    a fraction `share` of the sub-chains along base lines reuse the base chain's vertices one for one
    (identical and end-sharing edge pairs: the simulation-of-simplicity and identical-edge branches of
    lsi.h:42-100, and PIP points that lie exactly on base vertices, pip.h:44-93), the others run
    between the same two shared vertices with their own jitter (a differently generalised copy of the
    boundary: it crosses the base chain every few segments).  Interior sub-chains have k2 segments and
    end ON base vertices where they meet a base line (T-junctions at shared vertices).
    `detach`: of the sub-chains along base lines that do not reuse the base vertices, this fraction runs BESIDE the base
    chain (its interior pushed to one side by 0.6-0.9 of a segment's length) and meets it only at its two end vertices --
    a boundary drawn from another source; the rest weave across it.  With `share` it sets the pair's intersection
    density (CrossingZipcode: the published County x Zipcode density)."""
    rng = np.random.default_rng(seed)
    G2 = s * G
    bp = base.points.reshape(-1, k + 1, 2)                      # base chain c = bp[c]
    # base chain index of the horizontal / vertical chain that starts at lattice node (i, j)
    jj, ii = np.meshgrid(np.arange(G + 1), np.arange(G + 1), indexing="ij")
    keys = np.sort(np.concatenate([((jj * (G + 1) + ii) * 2)[ii < G], ((jj * (G + 1) + ii) * 2 + 1)[jj < G]]))
    hchain = lambda i, j: np.searchsorted(keys, (j * (G + 1) + i) * 2)
    vchain = lambda i, j: np.searchsorted(keys, (j * (G + 1) + i) * 2 + 1)
    m = np.rint(np.arange(s + 1) * k / s).astype(np.int64)       # base-chain vertex of sub-node a: m[0] = 0, m[s] = k
    # ---- sub-lattice nodes
    j2, i2 = np.meshgrid(np.arange(G2 + 1), np.arange(G2 + 1), indexing="ij")
    i, a, j, b = i2 // s, i2 % s, j2 // s, j2 % s
    Q = np.empty((G2 + 1, G2 + 1, 2))
    corner = lambda di, dj: bp[np.where(i + di < G, hchain(np.minimum(i + di, G - 1), np.minimum(j + dj, G)), hchain(G - 1, np.minimum(j + dj, G))),
                               np.where(i + di < G, 0, k)]     # lattice node (i+di, j+dj) as a base vertex
    u, v = (a / s)[..., None], (b / s)[..., None]
    c00, c10, c01, c11 = corner(0, 0), corner(1, 0), corner(0, 1), corner(1, 1)
    cell = np.stack([np.hypot(*(c10 - c00).transpose(2, 0, 1)), np.hypot(*(c01 - c00).transpose(2, 0, 1))], -1) / s
    Q[:] = (1 - u) * (1 - v) * c00 + u * (1 - v) * c10 + (1 - u) * v * c01 + u * v * c11
    Q += rng.uniform(-node_jitter, node_jitter, Q.shape) * cell
    on_h = (b == 0) & (i < G)                                     # on a horizontal base line (between nodes or at one)
    Q[on_h] = bp[hchain(i[on_h], j[on_h]), m[a[on_h]]]
    on_v = (a == 0) & (j < G)
    Q[on_v] = bp[vchain(i[on_v], j[on_v]), m[b[on_v]]]
    last = (i == G) & (b == 0)                                    # right border nodes: end of the last horizontal chain
    Q[last] = bp[hchain(G - 1, j[last]), k]
    last = (j == G) & (a == 0)
    Q[last] = bp[vchain(i[last], G - 1), k]
    # ---- sub-chains in the order of lattice_map: row-major over nodes, horizontal then vertical
    hm, vm = i2 < G2, j2 < G2
    key = np.concatenate([((j2 * (G2 + 1) + i2) * 2)[hm], ((j2 * (G2 + 1) + i2) * 2 + 1)[vm]])
    order = np.argsort(key, kind="stable")
    cat = lambda fh, fv: np.concatenate([fh, fv])[order]
    P0 = cat(Q[hm], Q[vm])
    P1 = cat(Q[j2[hm], i2[hm] + 1], Q[j2[vm] + 1, i2[vm]])
    on_base = cat((b == 0)[hm], (a == 0)[vm])
    bchain = cat(hchain(np.minimum(i, G - 1), j)[hm], vchain(i, np.minimum(j, G - 1))[vm])
    bfirst = cat(m[a][hm], m[b][vm])                              # first base vertex of an on-base sub-chain
    blen = cat((m[np.minimum(a + 1, s)] - m[a])[hm], (m[np.minimum(b + 1, s)] - m[b])[vm])
    shared = on_base & (rng.random(on_base.shape) < share)
    nseg = np.where(on_base, blen, k2)
    c2 = lambda ci, cj: np.where((ci >= 0) & (ci < G2) & (cj >= 0) & (cj < G2), cj * G2 + ci + 1, 0)
    left = cat(c2(i2, j2)[hm], c2(i2 - 1, j2)[vm])
    right = cat(c2(i2, j2 - 1)[hm], c2(i2, j2)[vm])
    owner, t = _ragged(nseg + 1)
    f = (t / nseg[owner])[:, None]
    d = P1 - P0
    ln = np.hypot(d[:, 0], d[:, 1])
    perp = np.stack([-d[:, 1], d[:, 0]], 1) / ln[:, None]
    jit = rng.uniform(-seg_jitter, seg_jitter, len(owner)) * (ln / nseg)[owner]
    jit[(t == 0) | (t == nseg[owner])] = 0.0
    if detach > 0:
        away = on_base & ~shared & (rng.random(on_base.shape) < detach)
        side = np.where(rng.random(on_base.shape) < 0.5, -1.0, 1.0)
        aw = away[owner] & (t > 0) & (t < nseg[owner])
        jit[aw] = (side[owner] * (0.6 + 0.3 * rng.random(len(owner))) * (ln / nseg)[owner])[aw]
    pts = (1 - f) * P0[owner] + f * P1[owner] + jit[:, None] * perp[owner]
    ends0, ends1 = t == 0, t == nseg[owner]
    pts[ends0] = P0[owner[ends0]]                                 # shared vertices bit for bit
    pts[ends1] = P1[owner[ends1]]
    sh = shared[owner]
    pts[sh] = bp[bchain[owner[sh]], bfirst[owner[sh]] + t[sh]]
    row_index = np.concatenate([[0], np.cumsum(nseg + 1)]).astype(np.uint32)
    first = row_index[:-1].astype(np.int64)
    chains = np.stack([np.arange(len(nseg), dtype=np.int64), first, first + nseg, left.astype(np.int64),
                       right.astype(np.int64)], 1)
    return PlanarGraph(chains, row_index, pts)


def gaussian_polygons(n, seed, maxseg=10, polysize=0.001, affine=(50.0, 0.0, -119.0, 0.0, 30.0, 35.0)):
    """The reference's synthetic scalability workload (misc/gen_polys.sh: `generator.py
    distribution=gaussian geometry=polygon polysize=0.001 maxseg=10 affinematrix=50,0,-119,0,30,35`,
    expr/draw/scal_lsi_synthetic/gaussian_*.log): n polygons whose centres are N(0.5, 0.1)^2 points of
    the unit square (redrawn until inside it, generator.py:276-300) mapped by the affine matrix; a
    polygon has dice(maxseg - 3) + 3 vertices at sorted uniform angles on a circle of radius polysize
    around its centre (generator.py:185-199).  Every polygon is one closed chain (left = its 1-based
    id, right = 0), in generation order -- consecutive chains are spatially unrelated.  The
    reference planarises the overlapping polygons on the way to CDB (misc/ scripts, geopandas), which
    is not reproduced: polygons of one map may overlap here, and edge counts are ~15 % lower."""
    rng = np.random.default_rng(seed)
    c = np.empty((0, 2))
    while len(c) < n:
        g = rng.normal(0.5, 0.1, (int((n - len(c)) * 1.01) + 16, 2))
        c = np.concatenate([c, g[((g >= 0) & (g <= 1)).all(axis=1)]])
    c = c[:n]
    cx = affine[0] * c[:, 0] + affine[1] * c[:, 1] + affine[2]
    cy = affine[3] * c[:, 0] + affine[4] * c[:, 1] + affine[5]
    nv = rng.integers(1, maxseg - 3 + 1, n) + 3 if maxseg > 3 else np.full(n, 3)
    ang = rng.uniform(0, 2 * np.pi, (n, maxseg))
    ang[np.arange(maxseg)[None, :] >= nv[:, None]] = np.inf
    ang.sort(axis=1)
    owner, t = _ragged(nv + 1)                                    # closed: the first vertex again at the end
    a = ang[owner, np.where(t == nv[owner], 0, t)]
    pts = np.stack([cx[owner] + polysize * np.cos(a), cy[owner] + polysize * np.sin(a)], 1)
    row_index = np.concatenate([[0], np.cumsum(nv + 1)]).astype(np.uint32)
    first = row_index[:-1].astype(np.int64)
    ids = np.arange(n, dtype=np.int64)
    chains = np.stack([ids, first, first + nv, ids + 1, np.zeros(n, dtype=np.int64)], 1)
    return PlanarGraph(chains, row_index, pts)


def ring_map(n_rings, total_edges, seed, bbox=US_BBOX, sigma=1.0, max_edges=10_000, clusters=400, fill=0.25,
             big_share=0.004, order="random"):
    """Lake-shaped maps: `n_rings` ISOLATED, mutually NON-OVERLAPPING closed rings, one closed chain each (left = the
    ring's 1-based id, right = 0) -- the topology of the reference's USADetailedWaterBodies / lakes / parks inputs
    (README.md:66-69: "2,442,900 chains"; expr/draw/query_lsi/*_lbvh.log:2-3), which the jittered lattices do not have:
    no shared end points, nothing to stitch, most chains far shorter than a 64-edge leaf.

    * edge counts: log-normal (sigma 1.0: a heavy tail), clipped to [3, max_edges], scaled so that they sum to about
      `total_edges` (WaterBodies: 10.5 per chain, Lakes 35, Parks 44.5, single rings up to 10^4);
    * places: a fine grid of cells (`fill` of them occupied), each ring confined to its own cell -- that is what makes
      the rings disjoint -- the occupied cells drawn from a clustered density (a few hundred gaussian bumps of mixed
      widths over a thin uniform floor); the `big_share` largest rings live in cells of an 8 x coarser grid whose fine
      cells stay empty, so a ring with thousands of edges is also a LARGE ring;
    * shapes: star-shaped around the cell centre, radius = a power-law cosine series in the angle (beta 1.3: rough,
      fractal-looking shores), vertices at jittered uniform angles -- single-valued in the angle, hence simple;
    * file order: "random" (the order the cells were drawn in: spatially shuffled, like the reference's synthetic
      workload) or "hilbert" (along a space-filling curve)."""
    rng = np.random.default_rng(seed)
    x0, y0, x1, y1 = bbox
    nv = np.exp(rng.normal(0.0, sigma, n_rings))
    nv = np.clip(np.rint(nv * (total_edges / nv.sum())), 3, max_edges).astype(np.int64)
    for _ in range(3):  # (the clip moves the sum: pull the unclipped ones back towards it)
        free = (nv > 3) & (nv < max_edges)
        nv[free] = np.clip(np.rint(nv[free] * (total_edges - nv[~free].sum()) / max(1, nv[free].sum())), 3, max_edges)
    G = int(np.ceil(np.sqrt(n_rings / fill) / 8.0)) * 8          # fine cells per side (a multiple of the coarse factor)
    Gc = G // 8
    n_big = min(int(n_rings * big_share), Gc * Gc // 8)
    big = np.zeros(n_rings, dtype=bool)
    if n_big:
        big[np.argpartition(nv, -n_big)[-n_big:]] = True
    # clustered density over the unit square
    cx, cy, cw = rng.random(clusters), rng.random(clusters), 10 ** rng.uniform(-2.2, -0.8, clusters)
    amp = rng.random(clusters) ** 2

    D = 512                                                       # the density, tabulated once on a D x D raster
    dv, du = np.meshgrid((np.arange(D) + 0.5) / D, (np.arange(D) + 0.5) / D, indexing="ij")
    dens = np.full((D, D), 0.02)
    for k in range(clusters):
        dens += amp[k] * np.exp(-((du - cx[k]) ** 2 + (dv - cy[k]) ** 2) / (2 * cw[k] ** 2))
    logd = np.log(dens)

    def draw(g, n, banned=None):
        """n distinct cells of a g x g grid, probability ~ density (Gumbel top-k)"""
        jj, ii = np.divmod(np.arange(g * g), g)
        w = logd[np.minimum((jj * D) // g, D - 1), np.minimum((ii * D) // g, D - 1)] + rng.gumbel(size=g * g)
        if banned is not None:
            w[banned] = -np.inf
        sel = np.argpartition(w, -n)[-n:]
        return sel[rng.permutation(n)]

    coarse = draw(Gc, n_big) if n_big else np.zeros(0, dtype=np.int64)
    banned = np.zeros((G, G), dtype=bool)
    cj, ci = np.divmod(coarse, Gc)
    for a in range(8):
        for b in range(8):
            banned[cj * 8 + a, ci * 8 + b] = True
    fine = draw(G, n_rings - n_big, banned.ravel())
    # centre and largest radius of every ring, in units of the bbox
    cxr, cyr, rad = np.empty(n_rings), np.empty(n_rings), np.empty(n_rings)
    fj, fi = np.divmod(fine, G)
    cxr[~big], cyr[~big], rad[~big] = (fi + 0.5) / G, (fj + 0.5) / G, 0.46 / G
    cxr[big], cyr[big], rad[big] = (ci + 0.5) / Gc, (cj + 0.5) / Gc, 0.46 / Gc
    # small rings are small: the radius follows the edge count up to the cell's limit
    rad *= np.clip(np.sqrt(nv / np.where(big, max_edges, 64.0)), 0.25, 1.0)
    if order == "hilbert":
        gx, gy = (cxr * 65535).astype(np.uint32), (cyr * 65535).astype(np.uint32)
        d = np.zeros(n_rings, dtype=np.uint64)
        sx, sy = gx.copy(), gy.copy()
        for k in range(15, -1, -1):
            rx, ry = (sx >> k) & 1, (sy >> k) & 1
            d += (np.uint64(1) << np.uint64(2 * k)) * ((3 * rx) ^ ry).astype(np.uint64)
            flip = (ry == 0) & (rx == 1)
            sx, sy = np.where(flip, 65535 - sx, sx), np.where(flip, 65535 - sy, sy)
            swap = ry == 0
            sx, sy = np.where(swap, sy, sx), np.where(swap, sx, sy)
        perm = np.argsort(d, kind="stable")
        nv, cxr, cyr, rad = nv[perm], cxr[perm], cyr[perm], rad[perm]
    cnt = nv + 1                                                  # closed: the first vertex again at the end
    starts_all = np.cumsum(cnt) - cnt
    # r(angle) = 1 + sum_k a_k cos(k angle + phase_k), a_k ~ k^-1.3, per ring, kept inside (0.35, 1]
    K = 6
    a = rng.normal(0, 1, (n_rings, K)) * (np.arange(1, K + 1)[None, :] ** -1.3)
    a *= 0.45 / np.maximum(np.abs(a).sum(axis=1, keepdims=True), 1e-9)
    ph = rng.uniform(0, 2 * np.pi, (n_rings, K))
    pts = np.empty((int(cnt.sum()), 2))
    # (ring blocks of ~2 M vertices: the temporaries stay cache- and page-friendly; 67 M vertices at once took 8 x longer per vertex)
    cuts = np.searchsorted(starts_all, np.arange(0, len(pts), 1 << 21))
    for lo, hi in zip(cuts, list(cuts[1:]) + [n_rings]):
        if hi <= lo:
            continue
        c = cnt[lo:hi]
        rep = lambda v: np.repeat(v[lo:hi], c)                    # (per-ring value -> per-vertex: sequential, unlike v[owner])
        v0 = int(starts_all[lo])
        starts = starts_all[lo:hi] - v0
        t = np.arange(int(c.sum()), dtype=np.int64) - np.repeat(starts, c)
        nvr = rep(nv)
        last = t == nvr
        ang = 2 * np.pi * (np.where(last, 0, t) + rng.uniform(-0.35, 0.35, len(t))) / nvr
        ang[last] = ang[starts]                                   # ... bit for bit
        r = np.ones(len(ang))
        for k in range(K):
            r += rep(a[:, k]) * np.cos((k + 1) * ang + rep(ph[:, k]))
        r *= rep(rad) / 1.45
        pts[v0:v0 + len(t), 0] = x0 + (rep(cxr) + r * np.cos(ang)) * (x1 - x0)
        pts[v0:v0 + len(t), 1] = y0 + (rep(cyr) + r * np.sin(ang)) * (y1 - y0)
    row_index = np.concatenate([[0], np.cumsum(nv + 1)]).astype(np.uint32)
    first = row_index[:-1].astype(np.int64)
    ids = np.arange(n_rings, dtype=np.int64)
    chains = np.stack([ids, first, first + nv, ids + 1, np.zeros(n_rings, dtype=np.int64)], 1)
    return PlanarGraph(chains, row_index, pts)


# the chain statistics the reference's logs print for its ring-shaped inputs (expr/draw/query_lsi/*_lbvh.log:2-3)
RING_STANDINS = {
    "WaterBodiesLike": (2_442_900, 25_564_755, 14, US_BBOX),
    "LakesLike": (1_910_184, 67_363_477, 15, CONUS_BBOX),
    "ParksLike": (603_325, 26_861_683, 16, CONUS_BBOX),
}


# realistic-geometry stand-ins: name -> builder(scale)
def _nested_blockgroup(scale):
    G, k, seed, bbox = STANDINS["USCounty"]
    G = max(2, int(round(G * scale)))
    return nested_refinement(lattice_map(G, k, seed, bbox), G, k, 6, 33, seed=7)


def _crossing_zipcode(scale):
    """A Zipcode-sized query map (23.7 M segments) whose boundaries CROSS the USCounty stand-in's at the density the
    reference's logs show for the real County x Zipcode pair (833 470 intersections / 23.76 M query segments = 3.5 %,
    BASELINE.md; the independent lattices give 0.3 %, the nested refinement 7.5 %): a 3 x 3 refinement of the county cells
    whose boundaries along county lines are mostly drawn beside the county's own (another source's generalisation), one in
    five weaves across it, one in twenty reuses its vertices."""
    G, k, seed, bbox = STANDINS["USCounty"]
    G = max(2, int(round(G * scale)))
    return nested_refinement(lattice_map(G, k, seed, bbox), G, k, 3, 130, seed=9, share=0.05, detach=0.8)


EXTRA = {
    "CrossingZipcode": _crossing_zipcode,
    "NestedBlockGroup": _nested_blockgroup,                       # refines the USCounty stand-in (same seed): 28.1 M segments
    "Gaussian5M": lambda scale: gaussian_polygons(max(16, int(5_000_000 * scale * scale)), 1),
    "Gaussian1M": lambda scale: gaussian_polygons(max(16, int(1_000_000 * scale * scale)), 2),
}


def standin(name, scale=1.0):
    """Synthetic stand-in for a paper dataset; scale<1 shrinks G (fewer cells, same k)."""
    if name in EXTRA:
        return EXTRA[name](scale)
    if name in RING_STANDINS:
        n, e, seed, bbox = RING_STANDINS[name]
        return ring_map(max(16, int(n * scale * scale)), max(64, int(e * scale * scale)), seed, bbox)
    G, k, seed, bbox = STANDINS[name]
    G = max(2, int(round(G * scale)))
    return lattice_map(G, k, seed, bbox)


def generate_lsi_queries(bb, scaling, gen_n, gen_t, seed):
    """GenerateLSIQueries (run_query.cu:102-144): uniform start, uniform direction,
    length U(0, gen_t); returns a ScaledMap of free-standing segments (map id 1)."""
    rng = np.random.default_rng(seed)
    x1 = rng.uniform(bb[0], bb[2], gen_n)
    y1 = rng.uniform(bb[1], bb[3], gen_n)
    x2 = rng.uniform(bb[0], bb[2], gen_n)
    y2 = rng.uniform(bb[1], bb[3], gen_n)
    ln = np.hypot(x2 - x1, y2 - y1)
    t = rng.uniform(0, gen_t, gen_n)
    pts = np.empty((2 * gen_n, 2))
    pts[0::2, 0], pts[0::2, 1] = x1, y1
    pts[1::2, 0], pts[1::2, 1] = x1 + t * (x2 - x1) / ln, y1 + t * (y2 - y1) / ln
    return ScaledMap.from_segments(1, scaling.scale(pts))


def generate_pip_queries(bb, scaling, gen_n, seed):
    """GeneratePIPQueries (run_query.cu:147-167): uniform points, scaled."""
    rng = np.random.default_rng(seed)
    pts = np.stack([rng.uniform(bb[0], bb[2], gen_n), rng.uniform(bb[1], bb[3], gen_n)], 1)
    return scaling.scale(pts)


def adversarial_segments(n, span, seed, extreme=False):
    """Random integer segments on a tiny lattice (|coord| <= span): dense in shared endpoints,
    T-junctions, collinear overlaps, vertical/horizontal edges and duplicates -- the cases the
    simulation-of-simplicity branches of lsi.h:42-100 and pip.h:44-93 exist for.
    extreme=True moves the lattice to the +-2^46 corners of the internal range."""
    rng = np.random.default_rng(seed)
    p = rng.integers(-span, span + 1, size=(n, 4), dtype=np.int64)
    same = (p[:, 0] == p[:, 2]) & (p[:, 1] == p[:, 3])
    p[same, 2] += 1  # no zero-length edges (the loader rejects them, planar_graph.h:85)
    if extreme:
        big = (1 << 46) - 1 - span - 1
        sx = rng.choice(np.array([-big, big], dtype=np.int64), size=n)
        sy = rng.choice(np.array([-big, big], dtype=np.int64), size=n)
        p[:, 0] += sx
        p[:, 2] += sx
        p[:, 1] += sy
        p[:, 3] += sy
    return p.reshape(-1, 2)  # points (2i, 2i+1)


def adversarial_chains(n_chains, pts_per_chain, span, seed):
    """Random integer polyline chains on a tiny lattice (shared vertices galore)."""
    rng = np.random.default_rng(seed)
    pts = np.empty((n_chains, pts_per_chain, 2), dtype=np.int64)
    pts[:, 0, :] = rng.integers(-span, span + 1, size=(n_chains, 2))
    for k in range(1, pts_per_chain):
        step = rng.integers(-3, 4, size=(n_chains, 2))
        zero = (step[:, 0] == 0) & (step[:, 1] == 0)
        step[zero, 0] = 1
        pts[:, k, :] = pts[:, k - 1, :] + step
    row_index = (np.arange(n_chains + 1) * pts_per_chain).astype(np.uint32)
    left = rng.integers(0, 50, n_chains).astype(np.int64)
    right = rng.integers(0, 50, n_chains).astype(np.int64)
    return pts.reshape(-1, 2), row_index, left, right
