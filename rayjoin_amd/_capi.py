"""ctypes binding of librayjoin_amd.so (include/rayjoin_amd.h).  No torch types cross this
boundary: device buffers are raw pointers (a torch tensor's data_ptr() or rj_dev_alloc memory).

The library is the only compute path: if it is missing or fails to load, importing the query
operators raises -- there is no CPU fallback."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RAYJOIN_AMD_LIB") or os.path.join(HERE, "librayjoin_amd.so")  # override: A/B builds

RJ_OK, RJ_E_INVALID, RJ_E_HIP, RJ_E_OVERFLOW, RJ_E_NOMEM, RJ_E_INTERNAL = 0, 1, 2, 3, 4, 5
RJ_EXCHANGE_HEAD_WORDS = 4
RJ_T_BUILD, RJ_T_LSI_KERNEL, RJ_T_PIP_KERNEL, RJ_T_LSI_POINTS, RJ_T_SORT, RJ_T_ORDER = 0, 1, 2, 3, 4, 5
RJ_T_BUILD_KEYS, RJ_T_BUILD_SORT, RJ_T_BUILD_LEAVES, RJ_T_BUILD_LEVELS, RJ_T_PIP_WALK, RJ_T_BUILD_RUNS = 6, 7, 8, 9, 10, 11
MISS_EID = 0xFFFFFFFF

XSECT_DTYPE = np.dtype(
    [("x_num", "<i8"), ("x_den", "<i8"), ("y_num", "<i8"), ("y_den", "<i8"),
     ("eid", "<u4", (2,)), ("mid_point_polygon_id", "<i4"), ("_pad", "<i4")])

# every symbol include/rayjoin_amd.h declares: name -> (restype, argtypes)
_vp, _u64, _i64, _int = C.c_void_p, C.c_uint64, C.c_int64, C.c_int
SYMBOLS = {
    "rj_create": (_int, [_int, C.POINTER(_vp)]),
    "rj_destroy": (_int, [_vp]),
    "rj_set_stream": (_int, [_vp, _vp]),
    "rj_sync": (_int, [_vp]),
    "rj_last_error_string": (C.c_char_p, [_vp]),
    "rj_version": (C.c_char_p, []),
    "rj_upload_map": (_int, [_vp, _int, _vp, _u64, _vp, _vp, _vp, _u64]),
    "rj_scale_points": (_int, [_vp, _vp, _u64, _vp, _int]),
    "rj_map_num_edges": (_int, [_vp, _int, C.POINTER(_u64)]),
    "rj_map_num_points": (_int, [_vp, _int, C.POINTER(_u64)]),
    "rj_map_points_dev": (_int, [_vp, _int, C.POINTER(_vp)]),
    "rj_invalidate": (_int, [_vp]),
    "rj_map_runs": (_int, [_vp, _int, _vp, _vp, _vp, C.POINTER(_u64), C.POINTER(_u64)]),
    "rj_build_lbvh": (_int, [_vp, _int]),
    "rj_lsi_query": (_int, [_vp, _int, _int, _u64, _u64, _u64, _vp, C.POINTER(_u64)]),
    "rj_lsi_query_async": (_int, [_vp, _int, _int, _u64, _u64, _u64, _vp]),
    "rj_lsi_query_finish": (_int, [_vp, _u64, C.POINTER(_u64)]),
    "rj_lsi_count_to": (_int, [_vp, _vp]),
    "rj_lsi_count_async": (_int, [_vp, _int]),
    "rj_lsi_count_wait": (_int, [_vp, _int, _u64, C.POINTER(_u64)]),
    "rj_lsi_points": (_int, [_vp, _vp, _u64, _vp]),
    "rj_lsi_points_async": (_int, [_vp, _vp, _u64, _vp]),
    "rj_sort_pairs": (_int, [_vp, _vp, _u64]),
    "rj_comm_unique_id": (_int, [_vp]),
    "rj_comm_init": (_int, [_vp, _int, _int, _vp]),
    "rj_comm_destroy": (_int, [_vp]),
    "rj_allgather_pairs": (_int, [_vp, _vp, _u64, _vp, _u64, _vp, C.POINTER(_u64)]),
    "rj_allgather_u32": (_int, [_vp, _vp, _u64, _vp, _u64, _vp, C.POINTER(_u64)]),
    "rj_allgatherv_plan": (_int, [_vp, _int, _u64, _vp, C.POINTER(_u64)]),
    "rj_exchange_init": (_int, [_vp, _u64, _u64, _vp, _vp]),
    "rj_exchange_pairs_begin": (_int, [_vp, _int]),
    "rj_exchange_pairs_finish": (_int, [_vp, _int, _vp, _vp, C.POINTER(_u64)]),
    "rj_exchange_u32_begin": (_int, [_vp, _vp, _u64, _vp]),
    "rj_exchange_u32_finish": (_int, [_vp]),
    "rj_exchange_verdict": (_int, [_vp, _vp, _int, C.POINTER(_u64), C.POINTER(_int)]),
    "rj_last_ms_all": (_int, [_vp, _vp, _int]),
    "rj_overlay_edge_xsects": (_int, [_vp, _int, _vp, _u64, _vp]),
    "rj_pip_query": (_int, [_vp, _int, _int, _vp, _u64, _u64, _vp, _vp]),
    "rj_pip_query_async": (_int, [_vp, _int, _int, _vp, _u64, _u64, _vp, _vp]),
    "rj_build_grid": (_int, [_vp, _int, _int]),
    "rj_lsi_query_grid": (_int, [_vp, _u64, _vp, C.POINTER(_u64)]),
    "rj_pip_query_grid": (_int, [_vp, _int, _int, _vp, _u64, _u64, _vp, _vp]),
    "rj_last_ms": (_int, [_vp, _int, C.POINTER(C.c_float)]),
    "rj_last_stats": (_int, [_vp, C.POINTER(_u64)]),
    "rj_set_option": (_int, [_vp, C.c_char_p, _i64]),
    "rj_set_debug_option": (_int, [_vp, C.c_char_p, _i64]),
    "rj_get_debug_option": (_int, [_vp, C.c_char_p, C.POINTER(_i64)]),
    "rj_get_option": (_int, [_vp, C.c_char_p, C.POINTER(_i64)]),
    "rj_get_plan": (_int, [_vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "rj_dev_alloc": (_int, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "rj_dev_free": (_int, [_vp, _vp]),
    "rj_memcpy_h2d": (_int, [_vp, _vp, _vp, C.c_size_t]),
    "rj_memcpy_d2h": (_int, [_vp, _vp, _vp, C.c_size_t]),
}


def kernel_source_hash():
    """sha256 (16 hex digits) of the HIP sources the query kernels are built from: profile artefacts
    (profiles/traffic.json) carry it, and bench.py only quotes them for the sources they measured."""
    import hashlib
    hsh = hashlib.sha256()
    for name in ("rj_kernels.hip", "rj_kernels.h", "rj_device.h", "rj_predicates.h", "rj_strip.hip"):
        with open(os.path.join(HERE, "csrc", name), "rb") as f:
            hsh.update(f.read())
    return hsh.hexdigest()[:16]


def exchange_verdict(counts, capacities):
    """rj_exchange_verdict: (status, max_count, first_bad) every rank concludes from the gathered (count, capacity) words"""
    counts = np.ascontiguousarray(counts, dtype=np.uint64)
    caps = np.ascontiguousarray(capacities, dtype=np.uint64)
    mx, bad = _u64(0), _int(-1)
    rc = load().rj_exchange_verdict(counts.ctypes.data, caps.ctypes.data, int(counts.shape[0]), C.byref(mx), C.byref(bad))
    return rc, mx.value, bad.value


def allgatherv_plan(counts, capacity):
    """rj_allgatherv_plan: (offsets, total, status) of an all-gather-v of `counts` elements per rank (host only)."""
    counts = np.ascontiguousarray(counts, dtype=np.uint64)
    off = np.zeros(max(1, counts.shape[0]), dtype=np.uint64)
    total = _u64()
    rc = load().rj_allgatherv_plan(counts.ctypes.data, int(counts.shape[0]), int(capacity), off.ctypes.data, C.byref(total))
    return off[:counts.shape[0]], int(total.value), rc


def scale_points(bb, xy, fused=False):
    """rj_scale_points: Scaling(bb).ScaleX/ScaleY over xy (host code of the library, no GPU).
    fused=True reproduces the FMA that nvcc contracts the reference's device lambda into."""
    xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1, 2)
    b = np.ascontiguousarray(bb, dtype=np.float64)
    out = np.empty(xy.shape, dtype=np.int64)
    rc = load().rj_scale_points(b.ctypes.data, xy.ctypes.data, xy.shape[0], out.ctypes.data, int(bool(fused)))
    if rc != RJ_OK:
        raise RayJoinError(rc, "rj_scale_points failed")
    return out


class RayJoinError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("rayjoin_amd error %d: %s" % (code, msg))
        self.code = code


class QueueOverflow(RayJoinError):
    """RJ_E_OVERFLOW: n_found holds the true count (the reference only asserts, queue.h:37)."""

    def __init__(self, msg, n_found):
        super().__init__(RJ_E_OVERFLOW, msg)
        self.n_found = n_found


_lib = None


def load():
    """dlopen the HIP library.  Fails loudly when it has not been built (python -c 'import
    __graft_entry__ as g; g.build()' or make -C rayjoin_amd/csrc)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("rayjoin_amd: %s is missing -- build the HIP extension first "
                              "(make -C rayjoin_amd/csrc); there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        variant = bool(os.environ.get("RAYJOIN_AMD_LIB"))  # an A/B build of another revision may lack the newer entry points
        for name, (res, args) in SYMBOLS.items():
            if variant and not hasattr(L, name):
                continue
            fn = getattr(L, name)  # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _ptr(a):
    """void* of a numpy array (host), an int, None, or anything with data_ptr() (torch tensor)."""
    if a is None:
        return None
    if isinstance(a, int):
        return a
    if hasattr(a, "data_ptr"):
        return a.data_ptr()
    return a.ctypes.data


class DeviceBuffer:
    """Device memory owned through the C ABI (rj_dev_alloc) for hosts without torch."""

    def __init__(self, handle, nbytes):
        self.handle = handle
        self.nbytes = int(nbytes)
        p = _vp()
        handle._check(load().rj_dev_alloc(handle.h, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def data_ptr(self):
        return self.ptr

    def to_host(self, dtype, count=None):
        dtype = np.dtype(dtype)
        count = self.nbytes // dtype.itemsize if count is None else int(count)
        out = np.empty(count, dtype=dtype)
        if count:
            self.handle._check(load().rj_memcpy_d2h(self.handle.h, out.ctypes.data, self.ptr,
                                                    count * dtype.itemsize))
        return out

    def from_host(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        if arr.nbytes:
            self.handle._check(load().rj_memcpy_h2d(self.handle.h, self.ptr, arr.ctypes.data, arr.nbytes))
        return self

    def free(self):
        if self.ptr:
            load().rj_dev_free(self.handle.h, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            if self.handle.h:
                self.free()
        except Exception:
            pass


class Handle:
    """One per GPU (rj_create).  Thin, 1:1 with the C ABI."""

    def __init__(self, device_id=0):
        self.L = load()
        h = _vp()
        rc = self.L.rj_create(int(device_id), C.byref(h))
        if rc != RJ_OK:
            raise RayJoinError(rc, "rj_create(device %d) failed -- no usable HIP device?" % device_id)
        self.h = h.value
        self.device_id = device_id

    def close(self):
        if getattr(self, "h", None):
            self.L.rj_destroy(self.h)
            self.h = None

    __del__ = close

    def _check(self, rc):
        if rc != RJ_OK:
            raise RayJoinError(rc, self.L.rj_last_error_string(self.h).decode())

    def set_stream(self, stream_ptr):
        self._check(self.L.rj_set_stream(self.h, stream_ptr))
        self._stream_ptr = stream_ptr  # (dist.PairExchange checks that its event and the kernels share a stream)

    def sync(self):
        self._check(self.L.rj_sync(self.h))

    def set_option(self, name, value):
        self._check(self.L.rj_set_option(self.h, name.encode(), int(value)))

    def get_option(self, name):
        v = _i64()
        self._check(self.L.rj_get_option(self.h, name.encode(), C.byref(v)))
        return int(v.value)

    def set_debug_option(self, name, value):
        """experiment knobs (grids, chunk sizes, run lengths: rj_set_debug_option)"""
        self._check(self.L.rj_set_debug_option(self.h, name.encode(), int(value)))

    def get_debug_option(self, name):
        v = _i64()
        self._check(self.L.rj_get_debug_option(self.h, name.encode(), C.byref(v)))
        return int(v.value)

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def upload_map(self, map_id, pts, row_index, left, right):
        pts = np.ascontiguousarray(pts, dtype=np.int64).reshape(-1, 2)
        row_index = np.ascontiguousarray(row_index, dtype=np.uint32)
        left = np.ascontiguousarray(left, dtype=np.int64)
        right = np.ascontiguousarray(right, dtype=np.int64)
        nc = left.shape[0]
        self._check(self.L.rj_upload_map(self.h, map_id, pts.ctypes.data, pts.shape[0],
                                         row_index.ctypes.data, left.ctypes.data, right.ctypes.data, nc))

    def map_num_edges(self, map_id):
        n = _u64()
        self._check(self.L.rj_map_num_edges(self.h, map_id, C.byref(n)))
        return n.value

    def map_num_points(self, map_id):
        n = _u64()
        self._check(self.L.rj_map_num_points(self.h, map_id, C.byref(n)))
        return n.value

    def map_points_dev(self, map_id):
        p = _vp()
        self._check(self.L.rj_map_points_dev(self.h, map_id, C.byref(p)))
        return p.value

    def invalidate(self):
        self._check(self.L.rj_invalidate(self.h))

    def map_runs(self, map_id):
        """-> (piece_begin, piece_len, run_first) of the polyline runs cut for this map"""
        nr, npc = _u64(), _u64()
        self._check(self.L.rj_map_runs(self.h, map_id, None, None, None, C.byref(nr), C.byref(npc)))
        pb = np.zeros(npc.value, dtype=np.uint32)
        pl = np.zeros(npc.value, dtype=np.uint32)
        rf = np.zeros(nr.value + 1, dtype=np.uint32)
        self._check(self.L.rj_map_runs(self.h, map_id, pb.ctypes.data, pl.ctypes.data, rf.ctypes.data, None, None))
        return pb, pl, rf

    def build_lbvh(self, base_map_id):
        self._check(self.L.rj_build_lbvh(self.h, base_map_id))

    def lsi_query(self, base_map_id, query_map_id, qb, qe, capacity, pairs_dev):
        n = _u64()
        rc = self.L.rj_lsi_query(self.h, base_map_id, query_map_id, qb, qe, capacity, _ptr(pairs_dev), C.byref(n))
        if rc == RJ_E_OVERFLOW:
            raise QueueOverflow(self.L.rj_last_error_string(self.h).decode(), n.value)
        self._check(rc)
        return n.value

    def lsi_query_async(self, base_map_id, query_map_id, qb, qe, capacity, pairs_dev):
        self._check(self.L.rj_lsi_query_async(self.h, base_map_id, query_map_id, qb, qe, capacity, _ptr(pairs_dev)))

    def lsi_query_finish(self, capacity):
        n = _u64()
        rc = self.L.rj_lsi_query_finish(self.h, capacity, C.byref(n))
        if rc == RJ_E_OVERFLOW:
            raise QueueOverflow(self.L.rj_last_error_string(self.h).decode(), n.value)
        self._check(rc)
        return n.value

    def build_grid(self, map_id, grid_size):
        self._check(self.L.rj_build_grid(self.h, map_id, grid_size))

    def lsi_query_grid(self, capacity, pairs_dev):
        n = _u64()
        rc = self.L.rj_lsi_query_grid(self.h, capacity, _ptr(pairs_dev), C.byref(n))
        if rc == RJ_E_OVERFLOW:
            raise QueueOverflow(self.L.rj_last_error_string(self.h).decode(), n.value)
        self._check(rc)
        return n.value

    def pip_query_grid(self, base_map_id, query_map_id, pts_dev, pt_begin, n, closest_dev, face_dev=None):
        self._check(self.L.rj_pip_query_grid(self.h, base_map_id, query_map_id, _ptr(pts_dev), pt_begin, n,
                                             _ptr(closest_dev), _ptr(face_dev)))

    def lsi_count_async(self, slot):
        self._check(self.L.rj_lsi_count_async(self.h, slot))

    def lsi_count_wait(self, slot, capacity):
        n = _u64()
        rc = self.L.rj_lsi_count_wait(self.h, slot, capacity, C.byref(n))
        if rc == RJ_E_OVERFLOW:
            raise QueueOverflow(self.L.rj_last_error_string(self.h).decode(), n.value)
        self._check(rc)
        return n.value

    def lsi_count_to(self, n_found_dev):
        """device-side Queue::size: copy the last async LSI's count (u64) to device memory, on the stream"""
        self._check(self.L.rj_lsi_count_to(self.h, _ptr(n_found_dev)))

    def lsi_points(self, pairs_dev, n, out_dev):
        self._check(self.L.rj_lsi_points(self.h, _ptr(pairs_dev), n, _ptr(out_dev)))

    def lsi_points_async(self, pairs_dev, capacity, out_dev):
        """records of the last lsi_query_async, behind it on the stream (count read on the device)"""
        self._check(self.L.rj_lsi_points_async(self.h, _ptr(pairs_dev), capacity, _ptr(out_dev)))

    @staticmethod
    def comm_unique_id():
        buf = (C.c_uint8 * 128)()
        rc = load().rj_comm_unique_id(buf)
        if rc != RJ_OK:
            raise RayJoinError(rc, "rj_comm_unique_id failed")
        return bytes(buf)

    def comm_init(self, nranks, rank, uid):
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        self._check(self.L.rj_comm_init(self.h, nranks, rank, buf))
        self.nranks = nranks

    def comm_destroy(self):
        self._check(self.L.rj_comm_destroy(self.h))

    def exchange_init(self, capacity, slot, buf0_dev, buf1_dev=None):
        self._check(self.L.rj_exchange_init(self.h, capacity, slot, _ptr(buf0_dev), _ptr(buf1_dev)))

    def exchange_pairs_begin(self, buf):
        self._check(self.L.rj_exchange_pairs_begin(self.h, buf))

    def exchange_pairs_finish(self, buf, nranks):
        """-> (counts list, device pointers of every rank's pairs, total); raises QueueOverflow on every rank alike"""
        counts = (_u64 * nranks)()
        slices = (_vp * nranks)()
        total = _u64()
        rc = self.L.rj_exchange_pairs_finish(self.h, buf, counts, slices, C.byref(total))
        if rc == RJ_E_OVERFLOW:
            raise QueueOverflow(self.L.rj_last_error_string(self.h).decode(), total.value)
        self._check(rc)
        return list(counts), [s or 0 for s in slices], total.value

    def exchange_u32_begin(self, src_dev, n_per_rank, recv_dev):
        self._check(self.L.rj_exchange_u32_begin(self.h, _ptr(src_dev), n_per_rank, _ptr(recv_dev)))

    def exchange_u32_finish(self):
        self._check(self.L.rj_exchange_u32_finish(self.h))

    def _allgather(self, fn, src_dev, n_local, out_dev, out_capacity):
        counts = (_u64 * getattr(self, "nranks", 1))()
        tot = _u64()
        rc = fn(self.h, _ptr(src_dev), n_local, _ptr(out_dev), out_capacity, counts, C.byref(tot))
        if rc == RJ_E_OVERFLOW:
            raise QueueOverflow(self.L.rj_last_error_string(self.h).decode(), tot.value)
        self._check(rc)
        return tot.value, list(counts)

    def allgather_pairs(self, pairs_dev, n_local, out_dev, out_capacity):
        return self._allgather(self.L.rj_allgather_pairs, pairs_dev, n_local, out_dev, out_capacity)

    def allgather_u32(self, src_dev, n_local, out_dev, out_capacity):
        return self._allgather(self.L.rj_allgather_u32, src_dev, n_local, out_dev, out_capacity)

    def overlay_edge_xsects(self, im, pairs_dev, n, xsects_dev):
        self._check(self.L.rj_overlay_edge_xsects(self.h, im, _ptr(pairs_dev), n, _ptr(xsects_dev)))

    def sort_pairs(self, pairs_dev, n):
        self._check(self.L.rj_sort_pairs(self.h, _ptr(pairs_dev), n))

    def pip_query(self, base_map_id, query_map_id, pts_dev, pt_begin, n, closest_dev, face_dev=None, sync=True):
        fn = self.L.rj_pip_query if sync else self.L.rj_pip_query_async
        self._check(fn(self.h, base_map_id, query_map_id, _ptr(pts_dev), pt_begin, n, _ptr(closest_dev), _ptr(face_dev)))

    def last_ms(self, which):
        ms = C.c_float()
        self._check(self.L.rj_last_ms(self.h, which, C.byref(ms)))
        return ms.value

    def last_ms_all(self):
        """-> list of the last duration of every stage (RJ_T_* order), -1 where a stage has not run"""
        buf = (C.c_float * 12)()
        self._check(self.L.rj_last_ms_all(self.h, buf, 12))
        return list(buf)

    def get_plan(self):
        """rj_get_plan: what the last query of each kind ran and why (a dict)"""
        import json
        need = C.c_size_t(0)
        self._check(self.L.rj_get_plan(self.h, None, 0, C.byref(need)))
        buf = C.create_string_buffer(need.value + 1)
        self._check(self.L.rj_get_plan(self.h, buf, need.value + 1, None))
        return json.loads(buf.value.decode())

    def last_stats_raw(self):
        """the 16 counters as a list (what each slot means depends on the instrumented kernel that ran: rj_kernels.hip)"""
        s = (_u64 * 16)()
        self._check(self.L.rj_last_stats(self.h, s))
        return list(s)

    def last_stats(self):
        s = (_u64 * 16)()
        self._check(self.L.rj_last_stats(self.h, s))
        return dict(leaf_blocks=s[0], exact_tests=s[1], nodes_expanded=s[2], leaf_box_tests=s[3],
                    cyc_total=s[4], cyc_node=s[5], cyc_leaf=s[6], cyc_drain=s[7], merge_rounds=s[8],
                    cyc_max_wave=s[9], cyc_sched=s[10], cyc_head=s[11], cyc_tail=s[12], stale_pops=s[13], leaf_visits_without_candidates=s[14], leaf_interested_lanes=s[15],
                    walk_stack_max=s[8])  # (k_pip_walk<STATS>: the deepest stack any group reached; k_lsi: merge_rounds)
