"""ctypes loader for oracle/librj_oracle.so (our C restatement) and, when present,
oracle/_ref/liblsi_ref.so (the reference's own predicate headers compiled in place).

TEST INFRASTRUCTURE: importable only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under rayjoin_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "librj_oracle.so")
REF_PATH = os.path.join(HERE, "_ref", "liblsi_ref.so")
MISS = 0xFFFFFFFF

_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")

XSECT_DTYPE = np.dtype(
    [("x_num", "<i8"), ("x_den", "<i8"), ("y_num", "<i8"), ("y_den", "<i8"),
     ("eid", "<u4", (2,)), ("mid_point_polygon_id", "<i4"), ("_pad", "<i4")])
assert XSECT_DTYPE.itemsize == 48


def build(force=False):
    """Compile the oracle (and oracle/_ref when /root/reference exists)."""
    src = os.path.join(HERE, "rj_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-s", os.path.join(HERE, "librj_oracle.so")])
    if os.path.isdir("/root/reference/src") and (force or not os.path.exists(REF_PATH)):
        subprocess.check_call(["make", "-C", HERE, "-s", "ref"])


class Scaling(C.Structure):
    _fields_ = [("imax", C.c_int64), ("imin", C.c_int64), ("irange", C.c_int64),
                ("rx", C.c_double), ("ry", C.c_double), ("rrx", C.c_double), ("rry", C.c_double),
                ("dx", C.c_double), ("dy", C.c_double), ("ddx", C.c_double), ("ddy", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(LIB_PATH)
    L.rjo_scaling_init.argtypes = [C.POINTER(Scaling)] + [C.c_double] * 4
    L.rjo_scale_points.argtypes = [C.POINTER(Scaling), _f64p, C.c_size_t, _i64p]
    L.rjo_unscale_points.argtypes = [C.POINTER(Scaling), _i64p, C.c_size_t, _f64p]
    L.rjo_map_create.restype = C.c_void_p
    L.rjo_map_create.argtypes = [_i64p, C.c_size_t, _u32p, _i64p, _i64p, C.c_size_t]
    L.rjo_map_create_segments.restype = C.c_void_p
    L.rjo_map_create_segments.argtypes = [_i64p, C.c_size_t]
    L.rjo_map_free.argtypes = [C.c_void_p]
    L.rjo_map_num_edges.restype = C.c_size_t
    L.rjo_map_num_edges.argtypes = [C.c_void_p]
    L.rjo_map_get_edge.argtypes = [C.c_void_p, C.c_size_t, _i64p]
    L.rjo_face_ids.argtypes = [C.c_void_p, _u32p, C.c_size_t, _i32p]
    L.rjo_intersect_test_segs.argtypes = [_i64p, _i64p]
    L.rjo_intersect_point_segs.argtypes = [_i64p, _i64p, C.c_int, _i64p]
    L.rjo_cell_of_int.argtypes = [C.c_int, C.c_int64]
    L.rjo_cell_of_double.argtypes = [C.c_int, C.c_double]
    L.rjo_pip_single.argtypes = [_i64p, _i64p, C.c_int, C.POINTER(C.c_double)]
    L.rjo_lsi_brute.restype = C.c_uint64
    L.rjo_lsi_brute.argtypes = [C.c_void_p, C.c_void_p, _u32p, C.c_uint64]
    L.rjo_pip_brute.argtypes = [C.c_void_p, C.c_int, _i64p, C.c_size_t, _u32p]
    L.rjo_lsi_points.argtypes = [C.c_void_p, C.c_void_p, _u32p, C.c_uint64, C.c_void_p]
    L.rjo_grid_build.restype = C.c_void_p
    L.rjo_grid_build.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.rjo_grid_free.argtypes = [C.c_void_p]
    L.rjo_grid_total.restype = C.c_uint64
    L.rjo_grid_total.argtypes = [C.c_void_p]
    L.rjo_lsi_grid.restype = C.c_uint64
    L.rjo_lsi_grid.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    L.rjo_pip_grid.argtypes = [C.c_void_p, C.c_int, C.c_void_p, _i64p, C.c_size_t, _u32p]
    L.rjo_overlay_edge_xsects.argtypes = [C.c_void_p, C.c_void_p, C.c_int, _u32p, C.c_uint64, C.c_int, C.c_void_p]
    L.rjo_num_threads.restype = C.c_int
    L.rjo_set_num_threads.argtypes = [C.c_int]
    _lib = L
    return L


_ref = None


def ref_lib():
    """The reference's own predicate headers (oracle/_ref); None when not built/available."""
    global _ref
    if _ref is not None:
        return _ref
    if not os.path.exists(REF_PATH):
        if os.path.isdir("/root/reference/src"):
            build()
        if not os.path.exists(REF_PATH):
            return None
    R = C.CDLL(REF_PATH)
    R.ref_intersect_test_segs.argtypes = [_i64p, _i64p]
    R.ref_intersect_point_segs.argtypes = [_i64p, _i64p, C.c_int, _i64p]
    R.ref_cell_of_int.argtypes = [C.c_int, C.c_int64]
    R.ref_cell_of_double.argtypes = [C.c_int, C.c_double]
    R.ref_scale_points.argtypes = [_f64p, _f64p, C.c_uint64, _i64p]
    R.ref_unscale_points.argtypes = [_f64p, _i64p, C.c_uint64, _f64p]
    R.ref_scaling_consts.argtypes = [_i64p]
    _ref = R
    return R


def make_scaling(min_x, min_y, max_x, max_y):
    s = Scaling()
    lib().rjo_scaling_init(C.byref(s), min_x, min_y, max_x, max_y)
    return s


def scale_points(s, xy):
    xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1, 2)
    out = np.empty(xy.shape, dtype=np.int64)
    lib().rjo_scale_points(C.byref(s), xy, xy.shape[0], out)
    return out


def unscale_points(s, xy):
    xy = np.ascontiguousarray(xy, dtype=np.int64).reshape(-1, 2)
    out = np.empty(xy.shape, dtype=np.float64)
    lib().rjo_unscale_points(C.byref(s), xy, xy.shape[0], out)
    return out


class Map:
    """Scaled map: pts int64[np,2], row_index uint32[nc+1], left/right int64[nc]."""

    def __init__(self, pts, row_index=None, left=None, right=None):
        L = lib()
        pts = np.ascontiguousarray(pts, dtype=np.int64).reshape(-1, 2)
        self.pts = pts
        if row_index is None:  # free-standing segments: edge i = points (2i, 2i+1)
            self.h = L.rjo_map_create_segments(pts, pts.shape[0] // 2)
        else:
            row_index = np.ascontiguousarray(row_index, dtype=np.uint32)
            nc = row_index.shape[0] - 1 if row_index.shape[0] else 0
            left = np.ascontiguousarray(left, dtype=np.int64)
            right = np.ascontiguousarray(right, dtype=np.int64)
            self.h = L.rjo_map_create(pts, pts.shape[0], row_index, left, right, nc)
        self.ne = L.rjo_map_num_edges(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().rjo_map_free(self.h)
            self.h = None

    def edge(self, eid):
        o = np.zeros(11, dtype=np.int64)
        lib().rjo_map_get_edge(self.h, eid, o)

        def i128(lo, hi):
            return (int(hi) << 64) | (int(lo) & 0xFFFFFFFFFFFFFFFF)
        return dict(a=i128(o[0], o[1]), b=i128(o[2], o[3]), c=i128(o[4], o[5]), eid=int(o[6]),
                    p1=int(o[7]), p2=int(o[8]), left=int(o[9]), right=int(o[10]))

    def face_ids(self, eids):
        eids = np.ascontiguousarray(eids, dtype=np.uint32)
        out = np.empty(eids.shape[0], dtype=np.int32)
        lib().rjo_face_ids(self.h, eids, eids.shape[0], out)
        return out


def sort_pairs(pairs):
    """Canonical order used by the reference's own checker: by (eid[0], eid[1])
    (src/run_overlay.cu:38-52)."""
    pairs = np.asarray(pairs, dtype=np.uint32).reshape(-1, 2)
    if pairs.shape[0] == 0:
        return pairs
    key = (pairs[:, 0].astype(np.uint64) << np.uint64(32)) | pairs[:, 1].astype(np.uint64)
    return pairs[np.argsort(key, kind="stable")]


def lsi_brute(m0, m1, cap=None):
    cap = cap or max(1024, 4 * (m0.ne + m1.ne))
    pairs = np.empty((cap, 2), dtype=np.uint32)
    n = lib().rjo_lsi_brute(m0.h, m1.h, pairs.reshape(-1), cap)
    if n > cap:
        return lsi_brute(m0, m1, cap=int(n))
    return sort_pairs(pairs[:n])


def lsi_grid(m0, m1, gsize=2048, cap=None, grid=None):
    """-mode=grid LSI.  Returns the 48-byte Intersection records sorted by (eid0, eid1)."""
    L = lib()
    g = grid or L.rjo_grid_build(m0.h, m1.h, gsize)
    try:
        cap = cap or max(1024, 2 * (m0.ne + m1.ne))
        out = np.zeros(cap, dtype=XSECT_DTYPE)
        n = L.rjo_lsi_grid(m0.h, m1.h, g, out.ctypes.data, cap)
        if n > cap:
            out = np.zeros(int(n), dtype=XSECT_DTYPE)
            n = L.rjo_lsi_grid(m0.h, m1.h, g, out.ctypes.data, int(n))
    finally:
        if grid is None:
            L.rjo_grid_free(g)
    out = out[:n]
    key = (out["eid"][:, 0].astype(np.uint64) << np.uint64(32)) | out["eid"][:, 1].astype(np.uint64)
    return out[np.argsort(key, kind="stable")]


def lsi_points(m0, m1, pairs):
    pairs = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
    out = np.zeros(pairs.shape[0], dtype=XSECT_DTYPE)
    lib().rjo_lsi_points(m0.h, m1.h, pairs.reshape(-1), pairs.shape[0], out.ctypes.data)
    return out


def overlay_edge_xsects(m0, m1, im, pairs, gsize=2048):
    """Intersection records ordered for map `im` with mid-point faces (ComputeOutputPolygons)."""
    pairs = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
    out = np.zeros(pairs.shape[0], dtype=XSECT_DTYPE)
    lib().rjo_overlay_edge_xsects(m0.h, m1.h, im, pairs.reshape(-1), pairs.shape[0], gsize, out.ctypes.data)
    return out


def pip_brute(base, query_map_id, pts):
    pts = np.ascontiguousarray(pts, dtype=np.int64).reshape(-1, 2)
    out = np.empty(pts.shape[0], dtype=np.uint32)
    lib().rjo_pip_brute(base.h, query_map_id, pts, pts.shape[0], out)
    return out


def pip_grid(base, base_map_id, pts, gsize=2048):
    L = lib()
    pts = np.ascontiguousarray(pts, dtype=np.int64).reshape(-1, 2)
    g = L.rjo_grid_build(base.h if base_map_id == 0 else None,
                         base.h if base_map_id == 1 else None, gsize)
    try:
        out = np.empty(pts.shape[0], dtype=np.uint32)
        L.rjo_pip_grid(base.h, base_map_id, g, pts, pts.shape[0], out)
    finally:
        L.rjo_grid_free(g)
    return out


def intersect_test(s1, s2, which="oracle"):
    s1 = np.ascontiguousarray(s1, dtype=np.int64)
    s2 = np.ascontiguousarray(s2, dtype=np.int64)
    if which == "ref":
        return int(ref_lib().ref_intersect_test_segs(s1, s2))
    return int(lib().rjo_intersect_test_segs(s1, s2))


def intersect_point(s1, s2, gsize=2048, which="oracle"):
    """None on miss; else dict(x=(num,den), y=(num,den), stored=(x,y), cell=(cx,cy))."""
    s1 = np.ascontiguousarray(s1, dtype=np.int64)
    s2 = np.ascontiguousarray(s2, dtype=np.int64)
    o = np.zeros(16, dtype=np.int64)
    if which == "ref":
        hit = ref_lib().ref_intersect_point_segs(s1, s2, gsize, o)
    else:
        hit = lib().rjo_intersect_point_segs(s1, s2, gsize, o)
    if not hit:
        return None

    def i128(lo, hi):
        return (int(hi) << 64) | (int(lo) & 0xFFFFFFFFFFFFFFFF)
    return dict(x=(i128(o[0], o[1]), i128(o[2], o[3])), y=(i128(o[4], o[5]), i128(o[6], o[7])),
                stored=(int(o[8]), int(o[9])), cell=(int(o[10]), int(o[11])))


def pip_single(seg, pt, query_map_id):
    seg = np.ascontiguousarray(seg, dtype=np.int64)
    pt = np.ascontiguousarray(pt, dtype=np.int64)
    yy = C.c_double()
    r = lib().rjo_pip_single(seg, pt, query_map_id, C.byref(yy))
    return int(r), yy.value


def num_threads():
    return int(lib().rjo_num_threads())
