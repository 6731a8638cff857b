"""Host-side mirror of the reference's query operators for -mode=lbvh (and -mode=grid), on top of the C ABI.

  LSILBVH(ctx).Init(max_n_xsects); .Query(query_map_id); .get_xsects(); .CopyTo()
      -- src/app/lsi.h:8-43, src/app/lsi_lbvh.h:17-98
  PIPLBVH(ctx).Init(n_points); .Query(query_map_id, query_points); .get_closest_eids()
      -- src/app/pip.h:9-38, src/app/pip_lbvh.h:14-142
  DeviceContext.LoadToDevice()/BuildIndex()
      -- src/context.h:76-88, src/run_query.cu:273-290 ("Build Index")

  LSIGrid / PIPGrid + DeviceContext.BuildGrid(grid_size): the uniform-grid operators
      -- src/app/lsi_grid.h:80-131, src/app/pip_grid.h:14-70, src/grid/uniform_grid.h:132-349

Same names, argument meaning and error behaviour, with two deliberate differences recorded in
DESIGN.md: LSI pairs are always evaluated as (e1 = map-0 edge, e2 = map-1 edge) so results
equal -mode=grid bit for bit, and a full queue raises QueueOverflow instead of being undefined
behaviour (src/util/queue.h:37 only asserts)."""
import numpy as np

from . import _capi
from .maps import Context


class DeviceContext:
    """A maps.Context plus its GPU residency: uploads both maps, owns the rj_handle."""

    def __init__(self, ctx, device_id=0):
        assert isinstance(ctx, Context)
        self.ctx = ctx
        self.handle = _capi.Handle(device_id)
        self.loaded = [False, False]
        self.indexed = [False, False]

    def LoadToDevice(self):
        for im in range(2):
            m = self.ctx.get_map(im)
            if m is not None:
                self.handle.upload_map(im, m.pts, m.row_index, m.left, m.right)
                self.loaded[im] = True
                self.indexed[im] = False
        return self

    def BuildIndex(self, base_map_id):
        self.handle.build_lbvh(base_map_id)
        self.indexed[base_map_id] = True
        return self.handle.last_ms(_capi.RJ_T_BUILD)

    def BuildGrid(self, grid_size, map_ids=(0, 1)):
        """UniformGrid::AddMapsToGrid (both maps, LSI) / AddMapToGrid (base map only, PIP)."""
        ms = 0.0
        for im in map_ids:
            self.handle.build_grid(im, grid_size)
            ms += self.handle.last_ms(_capi.RJ_T_BUILD)
        return ms

    def get_map(self, im):
        return self.ctx.get_map(im)

    def close(self):
        self.handle.close()


class LSILBVH:
    def __init__(self, dctx):
        self.ctx_ = dctx
        self.h = dctx.handle
        self.capacity = 0
        self.queue = None
        self.n_xsects = 0

    def Init(self, max_n_xsects):
        self.capacity = int(max_n_xsects)
        self.queue = self.h.alloc(8 * max(1, self.capacity))

    def Query(self, query_map_id, eid_range=None):
        """Intersect every edge (or the eid sub-range: a shard) of map `query_map_id` with the
        indexed other map.  Synchronous.  Returns the number of intersections."""
        base = 1 - query_map_id
        qb, qe = eid_range if eid_range is not None else (0, self.ctx_.get_map(query_map_id).n_edges)
        try:
            self.n_xsects = self.h.lsi_query(base, query_map_id, qb, qe, self.capacity, self.queue)
        except _capi.QueueOverflow as e:
            self.n_xsects = min(e.n_found, self.capacity)
            raise
        return self.n_xsects

    def get_pairs(self, sort=True):
        """(eid map 0, eid map 1) pairs on the host; sorted = the checker's canonical order."""
        if sort:
            self.h.sort_pairs(self.queue, self.n_xsects)
        return self.queue.to_host(np.uint32, 2 * self.n_xsects).reshape(-1, 2)

    def get_xsects(self, sort=True):
        """The reference's 48-byte Intersection records for the current result."""
        if sort:
            self.h.sort_pairs(self.queue, self.n_xsects)
        out = self.h.alloc(48 * max(1, self.n_xsects))
        self.h.lsi_points(self.queue, self.n_xsects, out)
        rec = out.to_host(_capi.XSECT_DTYPE, self.n_xsects)
        out.free()
        return rec

    CopyTo = get_xsects


class PIPLBVH:
    def __init__(self, dctx):
        self.ctx_ = dctx
        self.h = dctx.handle
        self.closest = None
        self.faces = None
        self.n = 0

    def Init(self, n_points):
        self.n_alloc = int(n_points)
        self.closest = self.h.alloc(4 * max(1, self.n_alloc))
        self.faces = self.h.alloc(4 * max(1, self.n_alloc))

    def Query(self, query_map_id, query_points=None, point_range=None):
        """query_points: int64[n,2] scaled host points, or None for the query map's own vertices
        (RunPIPQuery, src/run_query.cu:346), optionally a [begin, end) sub-range (a shard)."""
        base = 1 - query_map_id
        if query_points is not None:
            pts = np.ascontiguousarray(query_points, dtype=np.int64).reshape(-1, 2)
            n = pts.shape[0]
            buf = self.h.alloc(16 * max(1, n)).from_host(pts)
            assert n <= self.n_alloc
            self.h.pip_query(base, query_map_id, buf, 0, n, self.closest, self.faces)
            buf.free()
        else:
            b, e = point_range if point_range is not None else (0, self.ctx_.get_map(query_map_id).n_points)
            n = e - b
            assert n <= self.n_alloc
            self.h.pip_query(base, query_map_id, None, b, n, self.closest, self.faces)
        self.n = n
        return n

    def get_closest_eids(self):
        return self.closest.to_host(np.uint32, self.n)

    def get_face_ids(self):
        return self.faces.to_host(np.int32, self.n)


class LSIGrid(LSILBVH):
    """-mode=grid LSI (src/app/lsi_grid.h): needs DeviceContext.BuildGrid(g) for both maps.
    The grid holds both maps, so the result does not depend on query_map_id and there is no
    eid sub-range (the reference's LSIGrid::Query ignores its query_map_id too, lsi_grid.h:103-104)."""

    def Query(self, query_map_id=1, eid_range=None):
        if eid_range is not None:
            raise ValueError("LSIGrid has no eid sub-ranges: the grid joins the two whole maps")
        try:
            self.n_xsects = self.h.lsi_query_grid(self.capacity, self.queue)
        except _capi.QueueOverflow as e:
            self.n_xsects = min(e.n_found, self.capacity)
            raise
        return self.n_xsects


class PIPGrid(PIPLBVH):
    """-mode=grid PIP (src/app/pip_grid.h): needs DeviceContext.BuildGrid(g, (base_map_id,))."""

    def Query(self, query_map_id, query_points=None, point_range=None):
        base = 1 - query_map_id
        if query_points is not None:
            pts = np.ascontiguousarray(query_points, dtype=np.int64).reshape(-1, 2)
            n = pts.shape[0]
            buf = self.h.alloc(16 * max(1, n)).from_host(pts)
            assert n <= self.n_alloc
            self.h.pip_query_grid(base, query_map_id, buf, 0, n, self.closest, self.faces)
            buf.free()
        else:
            b, e = point_range if point_range is not None else (0, self.ctx_.get_map(query_map_id).n_points)
            n = e - b
            assert n <= self.n_alloc
            self.h.pip_query_grid(base, query_map_id, None, b, n, self.closest, self.faces)
        self.n = n
        return n
