"""Host-side data model: CDB planar graphs, scaling to exact integers, chain -> edge layout.

Mirrors the reference loader surface (file:line relative to /root/reference):
  * PlanarGraph / read_cdb / write_cdb     src/map/planar_graph.h:24-126, README.md:78-90
  * serialize_bin / deserialize_bin        src/map/planar_graph.h:129-220 (byte-compatible cache)
  * load_from                              src/map/planar_graph.h:223-252
  * Scaling                                src/map/scaling.h:32-136
  * Context                                src/context.h:31-88 (joint bbox -> one Scaling)
  * ScaledMap                              src/map/map.h:162-233 (eid = p_idx - ichain)

Scaled int64 coordinates are computed exactly once, on the host, as a separate IEEE multiply
then add then truncation (no FMA), and shipped to every GPU as integers (SURVEY 7 hard part 7).
"""
import os
import struct

import numpy as np

INTERNAL_MAX = (2 ** 63 - 1) >> 17   # scaling.h:44  ->  2^46 - 1
INTERNAL_MIN = -(2 ** 63) >> 17      # scaling.h:45  -> -2^46
INTERNAL_RANGE = INTERNAL_MAX - INTERNAL_MIN
SCALING_BOUNDING_BOX_MARGIN = 1      # config.h:4
EXTERIOR_FACE_ID = 0                 # config.h:8
MISS = 0xFFFFFFFF                    # static_cast<index_t>(DONTKNOW)
_BIN_MAGIC = 0xABCDABCD              # planar_graph.h:139


class CDBFormatError(ValueError):
    """Raised where the reference CHECK-fails on malformed input (planar_graph.h:71,85,100,105)."""


class PlanarGraph:
    """chains: int64[nc,5] = (id, first_point_idx, last_point_idx, left, right);
    row_index: uint32[nc+1]; points: float64[np,2]; bb = (min_x, min_y, max_x, max_y)."""

    def __init__(self, chains, row_index, points, bb=None):
        self.chains = np.ascontiguousarray(chains, dtype=np.int64).reshape(-1, 5)
        self.row_index = np.ascontiguousarray(row_index, dtype=np.uint32)
        self.points = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 2)
        if bb is None:
            if self.points.shape[0]:
                bb = (self.points[:, 0].min(), self.points[:, 1].min(),
                      self.points[:, 0].max(), self.points[:, 1].max())
            else:  # BoundingBox() default, bounding_box.h:8-12
                m = np.finfo(np.float64).max
                bb = (m, m, -m, -m)
        self.bb = tuple(float(v) for v in bb)

    @property
    def n_chains(self):
        return self.chains.shape[0]

    @property
    def n_points(self):
        return self.points.shape[0]

    @property
    def n_edges(self):
        return self.n_points - self.n_chains  # map.h:165


def read_cdb(path):
    """Text CDB parser with the reference's rules (planar_graph.h:42-126)."""
    chains, row_index, pts = [], [], []
    np_left = 0
    last = None
    with open(path, "r", newline="\n") as f:  # (std::getline splits at '\n' only: a '\r' stays in the line)
        for lno, line in enumerate(f, 1):
            line = line.rstrip("\n")
            if not line or line[0] in "#%":
                continue
            tok = line.split()
            try:
                if np_left == 0:
                    if len(tok) < 6:
                        raise ValueError
                    cid, n, first, lastp, left, right = (int(t) for t in tok[:6])
                    if n < 2 or any(not -(1 << 63) <= v < (1 << 63) for v in (cid, n, first, lastp, left, right)):
                        raise ValueError  # (an istream fails on an integer that does not fit int64_t)
                    chains.append((cid, first, lastp, left, right))
                    row_index.append(len(pts))
                    np_left = n
                    last = None
                else:
                    if len(tok) < 2:
                        raise ValueError
                    p = (float(tok[0]), float(tok[1]))
                    if not (np.isfinite(p[0]) and np.isfinite(p[1])):
                        raise ValueError  # (an istream reads neither "inf"/"nan" nor an out-of-range number)
                    if last is not None and p == last:
                        raise ValueError
                    pts.append(p)
                    last = p
                    np_left -= 1
            except ValueError:
                raise CDBFormatError("Bad line. Check your dataset! %s[%d]: %s" % (path, lno, line))
    if np_left != 0:
        raise CDBFormatError("%s: trailing incomplete chain" % path)
    if pts:
        row_index.append(len(pts))
    return PlanarGraph(np.array(chains, dtype=np.int64).reshape(-1, 5),
                       np.array(row_index, dtype=np.uint32), np.array(pts, dtype=np.float64))


def write_cdb(path, g, fmt="%.9f"):
    with open(path, "w") as f:
        for ic in range(g.n_chains):
            b, e = int(g.row_index[ic]), int(g.row_index[ic + 1])
            c = g.chains[ic]
            f.write("%d %d %d %d %d %d\n" % (c[0], e - b, c[1], c[2], c[3], c[4]))
            for k in range(b, e):
                f.write((fmt + " " + fmt + "\n") % (g.points[k, 0], g.points[k, 1]))


def serialize_bin(g, path):
    """planar_graph.h:129-167 layout: u64 magic, n_chains, n_row_index, n_points, chains 5xi64,
    row_index u32[], points 2xf64[], bbox (min_x, min_y, max_x, max_y) f64, u64 magic."""
    with open(path, "wb") as f:
        f.write(struct.pack("<QQQQ", _BIN_MAGIC, g.n_chains, g.row_index.shape[0], g.n_points))
        f.write(g.chains.astype("<i8").tobytes())
        f.write(g.row_index.astype("<u4").tobytes())
        f.write(g.points.astype("<f8").tobytes())
        f.write(struct.pack("<dddd", *g.bb))
        f.write(struct.pack("<Q", _BIN_MAGIC))


def deserialize_bin(path):
    with open(path, "rb") as f:
        buf = f.read()
    magic, nc, nri, npts = struct.unpack_from("<QQQQ", buf, 0)
    if magic != _BIN_MAGIC:
        raise CDBFormatError("%s: bad checksum" % path)
    off = 32
    chains = np.frombuffer(buf, "<i8", nc * 5, off).reshape(-1, 5)
    off += nc * 40
    row_index = np.frombuffer(buf, "<u4", nri, off)
    off += nri * 4
    points = np.frombuffer(buf, "<f8", npts * 2, off).reshape(-1, 2)
    off += npts * 16
    bb = struct.unpack_from("<dddd", buf, off)
    off += 32
    if struct.unpack_from("<Q", buf, off)[0] != _BIN_MAGIC:
        raise CDBFormatError("%s: bad trailing checksum" % path)
    return PlanarGraph(chains, row_index, points, bb)


def load_from(path, serialize_prefix=""):
    """planar_graph.h:223-252: cache file = <prefix>/<path with '/' -> '-'>.bin"""
    if serialize_prefix:
        os.makedirs(serialize_prefix, exist_ok=True)
    ser = serialize_prefix + "/" + path.replace("/", "-") + ".bin"
    if os.access(ser, os.R_OK):
        return deserialize_bin(ser)
    g = read_cdb(path)
    if serialize_prefix and os.access(serialize_prefix, os.W_OK):
        serialize_bin(g, ser)
    return g


def sample_map_from(g, sample_rate, seed=0):
    """planar_graph.h:255-317 ("-sample map"): keep every chain and its two end points; of the
    interior points keep a random fraction (at least one), in order.  Same semantics as the C++
    host's sampler, numpy's generator instead of std::mt19937 (a different but equally valid draw)."""
    rng = np.random.default_rng(None if seed == 0 else seed)
    keep = []
    row = [0]
    for ic in range(g.n_chains):
        b, e = int(g.row_index[ic]), int(g.row_index[ic + 1])
        pids = [b]
        if e - b > 2:
            mid = rng.permutation(np.arange(b + 1, e - 1))
            n_keep = max(2, int((1 + len(mid)) * np.float32(sample_rate)))  # counts the first point too
            pids += sorted(int(v) for v in mid[:n_keep - 1])
        pids.append(e - 1)
        keep += pids
        row.append(len(keep))
    return PlanarGraph(g.chains.copy(), np.array(row, dtype=np.uint32), g.points[keep])


def sample_edges_from(g, sample_rate, seed=0):
    """planar_graph.h:319-399 ("-sample edges"): a random fraction of all edges; survivors of one
    chain are re-packed into one chain, empty chains vanish, chain ids are renumbered from 0."""
    rng = np.random.default_rng(None if seed == 0 else seed)
    chain_of = np.repeat(np.arange(g.n_chains), np.diff(g.row_index.astype(np.int64)) - 1)
    first_pid = np.concatenate([np.arange(int(g.row_index[ic]), int(g.row_index[ic + 1]) - 1)
                                for ic in range(g.n_chains)]) if g.n_chains else np.zeros(0, np.int64)
    pick = rng.permutation(len(first_pid))[:int(len(first_pid) * np.float32(sample_rate))]
    chains, keep, row = [], [], [0]
    for new_id, ic in enumerate(np.unique(chain_of[pick])):
        p1 = first_pid[pick[chain_of[pick] == ic]]
        pids = np.unique(np.concatenate([p1, p1 + 1]))
        c = g.chains[ic].copy()
        c[0] = new_id
        chains.append(c)
        keep += [int(v) for v in pids]
        row.append(len(keep))
    if not chains:
        return PlanarGraph(np.zeros((0, 5), np.int64), np.zeros(0, np.uint32), np.zeros((0, 2)))
    return PlanarGraph(np.array(chains), np.array(row, dtype=np.uint32), g.points[keep])


class Scaling:
    """Scaling<double, int64_t, 17> (scaling.h:32-136)."""

    def __init__(self, bb, fused=False):
        """fused=True: scale() computes x*rx + dx as ONE fma, what nvcc makes of the reference's
        device lambda (src/map/map.h:171-180); the default is the separate multiply and add that
        scaling.h spells out and its host build computes."""
        self.bb = tuple(float(v) for v in bb)
        self.fused = bool(fused)
        min_x, min_y, max_x, max_y = (np.float64(v) for v in bb)
        m = np.float64(SCALING_BOUNDING_BOX_MARGIN)
        mxx, mnx, mxy, mny = max_x + m, min_x - m, max_y + m, min_y - m
        rng = np.float64(INTERNAL_RANGE)
        s = np.float64(INTERNAL_MAX + INTERNAL_MIN)
        self.rx = rng / (mxx - mnx)
        self.ry = rng / (mxy - mny)
        self.rrx = np.float64(1) / self.rx
        self.rry = np.float64(1) / self.ry
        self.dx = np.float64(0.5) * (s - (mxx + mnx) * self.rx)
        self.dy = np.float64(0.5) * (s - (mxy + mny) * self.ry)
        self.ddx = np.float64(0.5) * ((mxx + mnx) - s * self.rrx)
        self.ddy = np.float64(0.5) * ((mxy + mny) - s * self.rry)

    def scale(self, xy):
        xy = np.asarray(xy, dtype=np.float64).reshape(-1, 2)
        if self.fused:  # numpy has no fma: the library's host helper does it (std::fma)
            from . import _capi
            return _capi.scale_points(self.bb, xy, fused=True)
        out = np.empty(xy.shape, dtype=np.int64)
        # separate multiply and add (numpy never fuses), C-style truncation toward zero
        out[:, 0] = (xy[:, 0] * self.rx + self.dx).astype(np.int64)
        out[:, 1] = (xy[:, 1] * self.ry + self.dy).astype(np.int64)
        return out

    def unscale(self, xy):
        xy = np.asarray(xy, dtype=np.int64).reshape(-1, 2)
        out = np.empty(xy.shape, dtype=np.float64)
        out[:, 0] = xy[:, 0].astype(np.float64) * self.rrx + self.ddx
        out[:, 1] = xy[:, 1].astype(np.float64) * self.rry + self.ddy
        return out


class ScaledMap:
    """What is uploaded to a GPU: int64 points + chain layout (the 80-byte dev::Edge of
    map.h:42-46 is never materialised; a,b,c are recomputed from the endpoints)."""

    def __init__(self, map_id, pts, row_index, left, right):
        self.map_id = int(map_id)
        self.pts = np.ascontiguousarray(pts, dtype=np.int64).reshape(-1, 2)
        self.row_index = np.ascontiguousarray(row_index, dtype=np.uint32)
        self.left = np.ascontiguousarray(left, dtype=np.int64)
        self.right = np.ascontiguousarray(right, dtype=np.int64)
        nc = self.left.shape[0]
        if self.row_index.shape[0] != (nc + 1 if nc else self.row_index.shape[0]):
            raise ValueError("row_index must have n_chains + 1 entries")
        if nc and int(self.row_index[-1]) != self.pts.shape[0]:
            raise ValueError("row_index sentinel must equal the number of points")

    @classmethod
    def from_segments(cls, map_id, seg_pts):
        """Free-standing segments (GenerateLSIQueries, run_query.cu:102-144): edge i = points
        (2i, 2i+1), every edge its own 2-point chain, faces 0."""
        seg_pts = np.ascontiguousarray(seg_pts, dtype=np.int64).reshape(-1, 2)
        ne = seg_pts.shape[0] // 2
        ri = (np.arange(ne + 1, dtype=np.uint32) * 2).astype(np.uint32)
        z = np.zeros(ne, dtype=np.int64)
        return cls(map_id, seg_pts, ri, z, z)

    @property
    def n_points(self):
        return self.pts.shape[0]

    @property
    def n_chains(self):
        return self.left.shape[0]

    @property
    def n_edges(self):
        return self.n_points - self.n_chains

    def edge_p1(self):
        """p1 point index of every edge (eid = p_idx - ichain, map.h:198-207)."""
        counts = np.diff(self.row_index.astype(np.int64)) - 1
        chain_of_edge = np.repeat(np.arange(self.n_chains, dtype=np.int64), counts)
        return (np.arange(self.n_edges, dtype=np.int64) + chain_of_edge).astype(np.uint32)

    def segments(self):
        """int64[ne,4] = (x1,y1,x2,y2) in eid order."""
        p1 = self.edge_p1().astype(np.int64)
        return np.concatenate([self.pts[p1], self.pts[p1 + 1]], axis=1)

    def chain_range_to_eids(self, c0, c1):
        """A contiguous chain range is a contiguous eid range."""
        return int(self.row_index[c0]) - c0, int(self.row_index[c1]) - c1

    def shard_chain_ranges(self, n):
        """Split into n contiguous chain ranges balanced by edge count."""
        nc = self.n_chains
        cum = self.row_index.astype(np.int64) - np.arange(nc + 1, dtype=np.int64)  # edges before chain i
        cuts = [0]
        for r in range(1, n):
            cuts.append(int(np.searchsorted(cum, r * self.n_edges / n, side="left")))
        cuts.append(nc)
        cuts = np.maximum.accumulate(np.clip(cuts, 0, nc))
        return [(int(cuts[i]), int(cuts[i + 1])) for i in range(n)]


class Context:
    """src/context.h:31-88: owns the (up to) two planar graphs, the joint bbox and the scaling."""

    def __init__(self, pgraphs, fused_scaling=False):
        if isinstance(pgraphs, PlanarGraph):
            pgraphs = [pgraphs, None]
        self.planar_graphs = list(pgraphs) + [None] * (2 - len(pgraphs))
        m = np.finfo(np.float64).max
        bb = [m, m, -m, -m]
        for g in self.planar_graphs:
            if g is not None:
                bb = [min(bb[0], g.bb[0]), min(bb[1], g.bb[1]), max(bb[2], g.bb[2]), max(bb[3], g.bb[3])]
        self.bb = tuple(bb)
        # no planar graph (maps injected later through set_map / .maps): leave the scaling unset
        self.scaling = Scaling(self.bb, fused_scaling) if any(g is not None for g in self.planar_graphs) else None
        self.maps = [None, None]

    def load(self):
        """Host half of Context::LoadToDevice (context.h:76-88): scale each graph."""
        for im, g in enumerate(self.planar_graphs):
            if g is not None:
                self.maps[im] = ScaledMap(im, self.scaling.scale(g.points), g.row_index,
                                          g.chains[:, 3], g.chains[:, 4])
        return self

    def set_map(self, im, m):
        self.maps[im] = m

    def get_map(self, im):
        return self.maps[im]
