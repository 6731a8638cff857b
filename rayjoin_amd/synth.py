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


def standin(name, scale=1.0):
    """Synthetic stand-in for a paper dataset; scale<1 shrinks G (fewer cells, same k)."""
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
