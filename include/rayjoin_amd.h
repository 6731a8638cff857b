/*
 * rayjoin_amd.h -- C ABI of the MI355X-native LSI / PIP query path (librayjoin_amd.so).
 *
 * The reference (pwrliang/RayJoin @ /root/reference) has no FFI layer: its seam is C++ virtual
 * dispatch chosen by the -mode string -- LSI<CONTEXT_T>::{Init,Query,get_xsects,CopyTo}
 * (src/app/lsi.h:8-43), PIP<CONTEXT_T>::{Init,Query,get_closest_eids} (src/app/pip.h:9-38),
 * with the per-mode index handed over through QueryConfigLBVH (src/app/query_config.h:24-28) and
 * built in RunLSIQuery/RunPIPQuery (src/run_query.cu:273-290,422-438).  A "-mode=lbvh" drop-in
 * therefore needs exactly the entry points below; INTEGRATION.md shows the LSILBVH/PIPLBVH
 * subclasses a maintainer would write on top of them.
 *
 * Conventions
 *   - every function returns an rj_status (0 = ok) and never throws or aborts;
 *     rj_last_error_string() describes the last failure on that handle;
 *   - one handle per device; a handle is not thread-safe, different handles may be used from
 *     different threads (the reference is single-threaded with one stream, src/context.h:119);
 *   - plain pointers and sizes only; "_dev" pointers are device memory owned by the CALLER
 *     (hipMalloc / rj_dev_alloc / a torch tensor's data_ptr), all others are host memory;
 *   - coordinates are the reference's scaled integers: int64 in [-2^46, 2^46) produced on the
 *     host by Scaling (src/map/scaling.h:79-93); the library never sees floating-point input;
 *   - queries are synchronous like the reference's (Query ends with stream.Sync(),
 *     src/app/lsi_lbvh.h:89, src/app/pip_lbvh.h:136) unless the _async form is used.
 */
#ifndef RAYJOIN_AMD_H
#define RAYJOIN_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rj_handle_s* rj_handle;

typedef enum {
  RJ_OK = 0,
  RJ_E_INVALID = 1,   /* bad argument / call order (e.g. query before rj_build_lbvh) */
  RJ_E_HIP = 2,       /* a HIP runtime call failed; see rj_last_error_string */
  RJ_E_OVERFLOW = 3,  /* result queue capacity exceeded; *n_found holds the true count.
                         (The reference only asserts here: src/util/queue.h:37.) */
  RJ_E_NOMEM = 4,
  RJ_E_INTERNAL = 5   /* a traversal stack of the query kernels overflowed: results incomplete.  The
                         stacks cover the worst case of every index rj_build_lbvh accepts, so this is
                         a defect report, not an input error.  (The reference: a fixed 64-entry stack
                         per thread, unchecked -- deps/lbvh/lbvh/query.cuh:16.)  Reported by the call
                         that synchronises: rj_lsi_query(_finish), rj_pip_query, rj_sync. */
} rj_status;

#define RJ_MISS_EID 0xFFFFFFFFu /* static_cast<index_t>(DONTKNOW), src/app/pip_lbvh.h:44 */
#define RJ_EXTERIOR_FACE_ID 0   /* src/config.h:8 */

/* dev::Intersection<int64_t> as the reference's queue stores it (src/algo/lsi.h:10-26):
 * two rational<int64_t> (denominators are always 1 after the narrowing store,
 * src/util/rational.h:84-85,190-192), eid[2] = (map-0 edge, map-1 edge), 48 bytes. */
typedef struct {
  int64_t x_num, x_den;
  int64_t y_num, y_den;
  uint32_t eid[2];
  int32_t mid_point_polygon_id; /* DONTKNOW (-1) */
  int32_t _pad;
} rj_xsect;

/* ---- lifetime ------------------------------------------------------------------------- */
/* replaces: Context() + Stream (src/context.h:31-74,119; src/util/stream.h:13-27) */
int rj_create(int device_id, rj_handle* out);
int rj_destroy(rj_handle h);
/* run all work of this handle on a caller-owned hipStream_t; NULL is HIP's null (legacy default)
 * stream.  A new handle uses a private non-blocking stream (rj_set_option(h,"own_stream",1)
 * returns to it).  Queries on one stream are ordered by the stream, and each query kernel prepares
 * the counters of the next, so a CHANGE of stream first drains the old one (and the handle's second
 * stream); setting the stream the handle already uses costs nothing. */
int rj_set_stream(rj_handle h, void* hip_stream);
int rj_sync(rj_handle h);
const char* rj_last_error_string(rj_handle h);
const char* rj_version(void);

/* ---- maps ----------------------------------------------------------------------------- */
/* replaces: Context::LoadToDevice -> Map::LoadFrom (src/context.h:76-88, src/map/map.h:162-233).
 * xy: np scaled points (x,y interleaved); row_index[nc+1]: first point of each chain + sentinel;
 * left/right[nc]: face ids of each chain (truncated to 32 bits like dev::Edge, map.h:45).
 * Edge eid of chain c, point p is p - c (map.h:200-203).  Host arrays are copied. */
int rj_upload_map(rj_handle h, int map_id, const int64_t* xy, uint64_t np,
                  const uint32_t* row_index, const int64_t* left, const int64_t* right,
                  uint64_t nc);
int rj_map_num_edges(rj_handle h, int map_id, uint64_t* ne);
int rj_map_num_points(rj_handle h, int map_id, uint64_t* np);
/* device pointer to the uploaded scaled points (int64 x,y pairs), owned by the handle */
int rj_map_points_dev(rj_handle h, int map_id, const int64_t** pts_dev);
/* inspection: the polyline runs rj_build_lbvh cut for this map on the device ("leaf_order" 1; the analogue of the
 * reference's RT grouping, src/rt/primitive.h:120-260) -- piece p = eids [piece_begin[p], + piece_len[p]), run r =
 * pieces [run_first[r], run_first[r + 1]).  Host arrays, any of them NULL to fetch only the counts. */
int rj_map_runs(rj_handle h, int map_id, uint32_t* piece_begin, uint32_t* piece_len, uint32_t* run_first,
                uint64_t* nruns, uint64_t* npieces);

/* replaces: Scaling(bb) + the scale pass of Map::LoadFrom (src/map/scaling.h:56-93, src/map/map.h:171-180)
 * for hosts that do not scale themselves: xy[2n] doubles -> out_xy[2n] scaled int64, bb = {min_x, min_y,
 * max_x, max_y} of BOTH maps (src/context.h:37-47).  Host code, no GPU involved.
 * fused = 0: a separate multiply and add, what scaling.h spells out and what its host build computes
 * (the arithmetic every parity test of this repository is pinned to); fused = 1: one std::fma, what
 * nvcc's default -fmad=true makes of the same expression inside the reference's device lambda -- the
 * two differ by one unit for about 0.14 % of points.  Use the same setting for both maps. */
int rj_scale_points(const double bb[4], const double* xy, uint64_t n, int64_t* out_xy, int fused);

/* ---- index ---------------------------------------------------------------------------- */
/* replaces: FillPrimitivesLBVH + lbvh::bvh::assign/construct (src/tree/primtive.h:34-57,
 * deps/lbvh/lbvh/bvh.cuh:277-481) as called at src/run_query.cu:273-290,422-438. */
int rj_build_lbvh(rj_handle h, int base_map_id);

/* ---- LSI ------------------------------------------------------------------------------ */
/* replaces: LSILBVH::Query (src/app/lsi_lbvh.h:27-98) + Queue::Clear/size (src/util/queue.h).
 * Intersects query-map edges [query_eid_begin, query_eid_end) with all edges of the base map.
 * Writes (eid of map 0, eid of map 1) pairs, unordered, into pairs_dev[2*capacity].
 * The predicate is always evaluated as intersect_test(e1 = map-0 edge, e2 = map-1 edge)
 * whichever side is indexed, which is -mode=grid's operand order (src/app/lsi_grid.h:103-104). */
int rj_lsi_query(rj_handle h, int base_map_id, int query_map_id, uint64_t query_eid_begin,
                 uint64_t query_eid_end, uint64_t capacity, uint32_t* pairs_dev,
                 uint64_t* n_found);
/* same without the final count read-back/sync; pair with rj_lsi_query_finish */
int rj_lsi_query_async(rj_handle h, int base_map_id, int query_map_id, uint64_t query_eid_begin,
                       uint64_t query_eid_end, uint64_t capacity, uint32_t* pairs_dev);
int rj_lsi_query_finish(rj_handle h, uint64_t capacity, uint64_t* n_found);
/* Device-side Queue::size (src/util/queue.h:125-129 reads the tail counter back to the host):
 * enqueue, on the handle's stream, a copy of the last rj_lsi_query_async's result count (the TRUE
 * count, which exceeds the capacity after an overflow) into n_found_dev[0] (device memory, 8 bytes).
 * A multi-GPU caller puts it at the head of its exchange buffer and ships count + pairs in one
 * collective with no host round trip between the LSI kernel and the exchange. */
int rj_lsi_count_to(rj_handle h, uint64_t* n_found_dev);
/* The count of the last rj_lsi_query_async read back behind an event instead of a stream sync (two slots): a caller
 * that pipelines steps launches step k + 1 -- into other buffers -- before it looks at step k's count, so the GPU
 * never idles while the host wakes up.  rj_lsi_count_wait: *n_found, RJ_E_OVERFLOW / RJ_E_INTERNAL as
 * rj_lsi_query_finish; it waits for that query (and whatever was enqueued before rj_lsi_count_async) only. */
int rj_lsi_count_async(rj_handle h, int slot);
int rj_lsi_count_wait(rj_handle h, int slot, uint64_t capacity, uint64_t* n_found);

/* replaces: the intersection-point half of dev::intersect_test + the narrowing store into
 * Intersection<int64_t> (src/algo/lsi.h:107-143, src/app/lsi_lbvh.h:71-78).
 * pairs_dev: n (eid map 0, eid map 1) pairs; out_dev: n rj_xsect records. */
int rj_lsi_points(rj_handle h, const uint32_t* pairs_dev, uint64_t n, rj_xsect* out_dev);
/* The same for the result of the last rj_lsi_query_async, enqueued behind it on the handle's stream:
 * the number of records is read on the device from the queue's counter (min(count, capacity)), so
 * that "Query" leaves complete 48-byte records like the reference's (src/app/lsi_lbvh.h:71-78)
 * without a host round trip between the two kernels.  pairs_dev / capacity: what the query was given;
 * out_dev[capacity].  Complete after rj_lsi_query_finish / rj_sync. */
int rj_lsi_points_async(rj_handle h, const uint32_t* pairs_dev, uint64_t capacity, rj_xsect* out_dev);

/* sort n pairs in place by (eid0, eid1) -- the canonical order of the reference's checker
 * (src/run_overlay.cu:38-52) */
int rj_sort_pairs(rj_handle h, uint32_t* pairs_dev, uint64_t n);

/* ---- PIP ------------------------------------------------------------------------------ */
/* replaces: PIPLBVH::Query (src/app/pip_lbvh.h:25-142) and the get_face_id transform
 * (src/map/map.h:79-87, src/app/map_overlay_lbvh.h:96-104).
 * pts_dev: n scaled query points, or NULL to use points [pt_begin, pt_begin+n) of the query map
 * (RunPIPQuery queries every vertex of map 1, src/run_query.cu:346).
 * closest_eid_dev[n]: eid of the lowest base-map edge above each point, RJ_MISS_EID when none.
 * face_id_dev[n] (nullable): face below that edge, RJ_EXTERIOR_FACE_ID on a miss. */
int rj_pip_query(rj_handle h, int base_map_id, int query_map_id, const int64_t* pts_dev,
                 uint64_t pt_begin, uint64_t n, uint32_t* closest_eid_dev, int32_t* face_id_dev);
int rj_pip_query_async(rj_handle h, int base_map_id, int query_map_id, const int64_t* pts_dev,
                       uint64_t pt_begin, uint64_t n, uint32_t* closest_eid_dev,
                       int32_t* face_id_dev);
/* Caller-owned point arrays (pts_dev != NULL -- the reference's PIP::Query(Stream&, int, ArrayView<point_t>),
 * src/app/pip.h:23, handed a separate device vector by src/run_query.cu:346,441-443): the handle remembers, per
 * (pointer, n), whether the array needs re-ordering along the Morton curve and the permutation if so.  The first
 * query over an array pays one host round trip for the estimate; every later one is enqueued without any
 * synchronisation, pairs with an LSI query in flight exactly like a map-owned range, and refreshes the estimate
 * with a one-block kernel behind its own kernels, so contents that change are followed one query late.  None of
 * it can affect results (any permutation of [0, n) is a valid processing order).  rj_invalidate forgets what was
 * learned, e.g. before a buffer is reused for unrelated points. */
int rj_invalidate(rj_handle h);

/* ---- -mode=grid on the device (the reference's third index, for the grid / lbvh / rt comparison)
 * replaces: UniformGrid::AddMapsToGrid / AddMapToGrid (src/grid/uniform_grid.h:132-349): one CSR per
 * map over grid_size x grid_size cells of the scaled domain (calculate_cell, src/grid/cell.h:16-22);
 * inside a cell the eids are ascending.  Costs one entry per (cell, edge) incidence; fails with
 * RJ_E_INVALID when those exceed 2^32 (grid too fine for the map's longest edges). */
int rj_build_grid(rj_handle h, int map_id, int grid_size);
/* replaces: LSIGrid::Query + intersect_one_cell (src/app/lsi_grid.h:19-131): needs the grids of BOTH
 * maps at the same grid_size; every cell tests its (map-0 edge, map-1 edge) pairs and reports a hit
 * only from the cell that contains the computed intersection point.  Output and overflow behaviour
 * as rj_lsi_query. */
int rj_lsi_query_grid(rj_handle h, uint64_t capacity, uint32_t* pairs_dev, uint64_t* n_found);
/* replaces: PIPGrid::Query (src/app/pip_grid.h:37-70, cell acceptance src/algo/pip.h:98-114): needs
 * the grid of the base map.  Arguments and outputs as rj_pip_query. */
int rj_pip_query_grid(rj_handle h, int base_map_id, int query_map_id, const int64_t* pts_dev,
                      uint64_t pt_begin, uint64_t n, uint32_t* closest_eid_dev, int32_t* face_id_dev);

/* ---- multi-GPU: RCCL over xGMI ---------------------------------------------------------- */
/* New design (the reference is single-GPU: no NCCL/MPI anywhere).  One process and
 * one handle per GPU; the query map is sharded by chain range (rj_lsi_query's eid range / the point
 * range of rj_pip_query), the base map + LBVH are replicated, and the result queues are exchanged
 * with an all-gather-v.  Every exchange below is made of ncclAllGather alone and takes no
 * rank-dependent branch between two collectives: what a rank does next is a function of the
 * gathered words, which are the same everywhere.
 * rj_comm_unique_id: call on ONE rank, hand the 128 bytes to the others out of band (file, env,
 * MPI, torch.distributed store ...).  rj_comm_init is collective (it makes two communicators: pair
 * queues and point queues travel on streams of their own and never wait for each other). */
#define RJ_COMM_ID_BYTES 128
int rj_comm_unique_id(uint8_t id[RJ_COMM_ID_BYTES]);
int rj_comm_init(rj_handle h, int nranks, int rank, const uint8_t id[RJ_COMM_ID_BYTES]);
int rj_comm_destroy(rj_handle h);

/* The exchange of a STEP, off the critical path (bench.py, query_exec -nranks).  rj_exchange_init registers one or two
 * CALLER-OWNED exchange buffers of RJ_EXCHANGE_HEAD_WORDS + 2 * capacity 32-bit words each (buf1_dev nullable; every rank
 * must pass the same capacity and slot): the handle keeps a head in front -- count (u64), capacity (u64) -- and the
 * caller hands bufK_dev + RJ_EXCHANGE_HEAD_WORDS to rj_lsi_query_async as pairs_dev.  Per step:
 *   rj_lsi_query_async(h, ..., capacity, bufK_dev + RJ_EXCHANGE_HEAD_WORDS);
 *   rj_exchange_pairs_begin(h, K);        the device-side count goes into the buffer's head and ONE ncclAllGather of
 *                                         head + `slot` pairs per rank starts on the communication stream behind an
 *                                         event: no host round trip between the LSI kernel and the collective
 *   ... rj_lsi_points_async, rj_pip_query_async, the next step into the other buffer ...
 *   rj_exchange_pairs_finish(h, K, counts, slices, &total);   the step's one host sync on the LSI side: counts_out
 *                                         [nranks] (host), slices_dev[nranks] = device pointers to every rank's pairs
 *                                         (handle-owned, valid until the buffer's next begin), *n_total = sum.
 * `slot` (pairs shipped per rank) follows twice the largest count seen; a step in which some rank found more is gathered
 * again with a larger slot -- by every rank, which all see the same counts.  RJ_E_OVERFLOW, on EVERY rank, when some
 * rank's queue overflowed its capacity. */
#define RJ_EXCHANGE_HEAD_WORDS 4
int rj_exchange_init(rj_handle h, uint64_t capacity, uint64_t slot, uint32_t* buf0_dev, uint32_t* buf1_dev);
int rj_exchange_pairs_begin(rj_handle h, int buf);
int rj_exchange_pairs_finish(rj_handle h, int buf, uint64_t* counts_out, const uint32_t** slices_dev, uint64_t* n_total);
/* The PIP result queues of contiguous point shards: every rank ships n_per_rank words (its shard, padded: the same
 * n on every rank), recv_dev[nranks * n_per_rank].  Starts behind everything enqueued so far on the handle's streams, on
 * the second communicator's stream; rj_exchange_u32_finish waits for it. */
int rj_exchange_u32_begin(rj_handle h, const uint32_t* src_dev, uint64_t n_per_rank, uint32_t* recv_dev);
int rj_exchange_u32_finish(rj_handle h);
/* what every rank concludes from the gathered (count, capacity) words -- a pure host function (no GPU, no
 * communicator), so that the branch after a collective can be tested without one: RJ_E_OVERFLOW when some rank's count
 * exceeds its capacity (*first_bad = the lowest such rank, else -1), *max_count = the largest count. */
int rj_exchange_verdict(const uint64_t* counts, const uint64_t* capacities, int nranks, uint64_t* max_count, int* first_bad);

/* Synchronous all-gather-v with exact, contiguous output (hosts that want one flat queue).
 * pairs_dev: this rank's n_local (eid0, eid1) pairs; out_dev[2 * out_capacity]: all ranks' pairs in
 * rank order; counts_out[nranks] (host, nullable); *n_total = sum.  Two collectives: (count, capacity) of every rank,
 * then every slice padded to the largest.  RJ_E_OVERFLOW (with *n_total set) on EVERY rank when the total exceeds the
 * SMALLEST out_capacity any rank passed -- no rank enters the second collective alone. */
int rj_allgather_pairs(rj_handle h, const uint32_t* pairs_dev, uint64_t n_local, uint32_t* out_dev,
                       uint64_t out_capacity, uint64_t* counts_out, uint64_t* n_total);
/* The layout of an all-gather-v, as a pure host function (no GPU, no communicator; what rj_allgather_* compute
 * between their two collectives): offsets[r] = counts[0] + ... + counts[r-1] (rank r's slice starts there, zero
 * counts take no room), *total = the sum.  RJ_E_OVERFLOW (offsets and *total still set) when total > capacity,
 * RJ_E_INVALID for nranks < 1, null arrays or a sum that does not fit 64 bits. */
int rj_allgatherv_plan(const uint64_t* counts, int nranks, uint64_t capacity, uint64_t* offsets, uint64_t* total);
/* same for a queue of 32-bit values (closest eids / face ids of a point shard) */
int rj_allgather_u32(rj_handle h, const uint32_t* src_dev, uint64_t n_local, uint32_t* out_dev,
                     uint64_t out_capacity, uint64_t* counts_out, uint64_t* n_total);

/* ---- overlay support ------------------------------------------------------------------- */
/* replaces: the per-map body of MapOverlayLBVH::ComputeOutputPolygons
 * (src/app/map_overlay_lbvh.h:109-265).  For map `im`: the n intersections (eid map 0, eid map 1)
 * become 48-byte records ordered by eid[im] and, on one edge, by squared distance of the stored
 * point from that edge's first endpoint (:204-213; the reference's unstable ties are resolved by
 * the other map's eid); the mid-point of each consecutive pair on an edge (:215-227) is located
 * in the other map (query map id = im, :232-236) and the face found is stored in the FIRST record
 * of the pair (mid_point_polygon_id, :238-262); the last record of an edge keeps DONTKNOW (-1).
 * Needs rj_build_lbvh(1 - im).  xsects_dev[n] is caller-owned device memory. */
int rj_overlay_edge_xsects(rj_handle h, int im, const uint32_t* pairs_dev, uint64_t n,
                           rj_xsect* xsects_dev);

/* ---- measurement ---------------------------------------------------------------------- */
typedef enum {
  RJ_T_BUILD = 0,     /* whole rj_build_lbvh */
  RJ_T_LSI_KERNEL = 1,/* the LSI traversal+predicate kernel of the last rj_lsi_query* */
  RJ_T_PIP_KERNEL = 2,/* the PIP kernel of the last rj_pip_query* */
  RJ_T_LSI_POINTS = 3,
  RJ_T_SORT = 4,
  RJ_T_ORDER = 5,     /* Morton re-ordering of an incoherent query set inside the last query, if any */
  /* stages of the last rj_build_lbvh (the reference prints its own under -profile,
   * deps/lbvh/lbvh/bvh.cuh:464-474): sort keys, radix sort, leaf pass, upper levels + sibling order */
  RJ_T_BUILD_KEYS = 6,
  RJ_T_BUILD_SORT = 7,
  RJ_T_BUILD_LEAVES = 8,
  RJ_T_BUILD_LEVELS = 9,
  RJ_T_BUILD_RUNS = 11, /* cutting the polyline runs on the device (the first rj_build_lbvh of a map with "leaf_order" 1; inside RJ_T_BUILD) */
  RJ_T_PIP_WALK = 10  /* k_pip_walk, the integer-only first pass of the last rj_pip_query* (RJ_T_PIP_KERNEL spans both passes) */
} rj_timer;
/* HIP-event time (ms) of the last launch of that stage on the handle's stream; syncs. */
int rj_last_ms(rj_handle h, int which, float* ms);
/* every stage at once: ms[i] = rj_last_ms(h, i) for i < n (at most the number of stages), -1 for a stage that has
 * not run -- one call instead of one per stage between two steps of a timed loop */
int rj_last_ms_all(rj_handle h, float* ms, int n);
/* traversal statistics of the last LSI/PIP query (diagnostic; mirrors the reference's
 * "Total tests"/"Visited nodes" debug counters, src/app/lsi_lbvh.h:37-42,93-94):
 * stats[0] = leaf blocks visited, [1] = candidate pairs tested exactly, [2] = nodes expanded,
 * [3] = box tests in the leaf loop; [4..9] = summed per-wave cycle stamps of the instrumented
 * build (total, node expansion, leaf loop, dense predicate phase, merge rounds, max wave total).
 * Collected only after rj_set_option(h,"stats",1), which selects a separate, slower kernel. */
int rj_last_stats(rj_handle h, uint64_t stats[16]);

/* ---- options ---------------------------------------------------------------------------
 * rj_set_option(h, name, value).  NO option changes a result: they choose kernels, leaves and schedules.
 *
 * name               values (default first)   what it does
 * ------------------ ------------------------ ------------------------------------------------------------------------
 * "leaf_order"       1 / 0                    what the NEXT rj_build_lbvh makes a leaf of.  1: polyline runs -- chains
 *                                             stitched through their shared end points by straightest continuation and
 *                                             cut into near-equal runs of <= 64 edges ON THE DEVICE by the first build of
 *                                             an uploaded map (the analogue of the reference's RT grouping,
 *                                             src/rt/primitive.h:120-260); consecutive short runs of the sorted order
 *                                             share a leaf (maps of isolated rings).  0: 64 neighbours along the Hilbert
 *                                             curve.  (Environment RJ_LEAF_ORDER=0 changes the default.)
 * "skyline"          -1 / 0 / 1               the per-x-bucket top of the map that proves a PIP miss without a traversal:
 *                                             -1 built where most chains are closed rings, 0 never, 1 always.
 * "pip_columns"      -1 / 0 / 1               a second index for PIP on the NEXT rj_build_lbvh: the map's runs listed per
 *                                             vertical strip (2^15..2^17 quanta, by the mean width of a segment), sorted by height, with 1024 height buckets
 *                                             per strip -- a query point scans the entries above it in its own strip
 *                                             (k_pip_strip) instead of walking the tree.  -1 built where most chains are
 *                                             closed rings (lakes, parks: many small isolated faces) or where the map's
 *                                             chains are SHORT (mean below 16 edges: fat leaves; measured rule, round 6),
 *                                             0 never, 1 always.  -1 also builds it LATER, at the first PIP query whose point
 *                                             set (>= 2^22 points) turns out spatially incoherent -- the reference's
 *                                             GeneratePIPQueries, uniform random points: the index answers every point on its
 *                                             own where the tree walk first sorts the points and still shares little (8.4 M
 *                                             random points: 1.07 -> 0.70 ms USCounty, 1.9 -> 0.43 LakesNA; that query pays the
 *                                             build).  rj_get_plan's index[].columns_why says which rule applied.
 *                                             (Environment RJ_PIP_COLUMNS=0/1 changes the default: A/B runs.)
 * "leaf_ysort"       1 / 0                    the order inside a leaf block on the NEXT rj_build_lbvh.  Blocks lie sorted by x0 with
 *                                             a bucket table on x (what an upward ray needs).  1: a block TALLER than wide also gets
 *                                             a SECOND order, by y0, with its own table -- a query segment then scans the slots over
 *                                             its y-range (a steep run of a polyline folds back and forth in x: most of its edges lie
 *                                             over every query's x-range); the PIP traversals keep the x order.  8 bytes per slot.
 *                                             0: x order only.  (Environment RJ_LEAF_YSORT=0 changes the default: A/B runs.)
 * "pip_walk"         1 / 0 / 2                a PIP query = k_pip_walk* (integer-only traversal) + k_pip_exact (exact
 *                                             predicate over the candidate lists; its first blocks locate the points
 *                                             whose list overflowed).  1: unless the last query of this size left > 30 %
 *                                             of its points over; 0: k_pip alone; 2: always the passes.
 * "pip_walk_points"  2 / 1                    query points per lane of the walk (k_pip_walk2 / k_pip_walk); 2 applies to
 *                                             query sets that fill 64-position groups.
 * "lsi_segments"     2 / 1                    query segments per lane of the LSI kernel (k_lsi2 / k_lsi); 2 applies to
 *                                             query sets of at least two full groups per resident wave.
 * "lsi_points_split" -1 / 0 / 1               how rj_lsi_points* make the 48-byte records: 1 a gcd-free kernel + the gcd
 *                                             kernel over the pairs it declines, 0 the gcd kernel for every pair, -1 two
 *                                             kernels from 384 Ki pairs on.
 * "query_order"      1 / 0 / 2                re-order a query set along the Morton curve: 1 when consecutive queries are
 *                                             spatially scattered (e.g. the generated workloads of
 *                                             src/run_query.cu:102-167), 0 never, 2 always.
 * "pip_concurrent"   0 / 1 / 2                the caller issues rj_lsi_query_async and rj_pip_query_async in PAIRS (the
 *                                             step of a join: both only read the maps and the index).  1: the two sides
 *                                             share the chip (the LSI kernel on a reduced grid, the PIP kernels on the
 *                                             handle's second stream fill the rest); 2: the handle measures the first
 *                                             four pairs -- taking turns / sharing / sharing with the neighbouring split
 *                                             / beside each other on full grids -- and keeps the fastest from the fifth
 *                                             on (the reference's five warm-up queries settle it), deciding again when
 *                                             the index, a map or the query size changes; 0: every kernel on the whole
 *                                             chip (a caller that issues one kind of query).  The PIP query's inputs must
 *                                             be complete when the call is made; its outputs are complete after rj_sync,
 *                                             rj_pip_query or rj_last_ms(RJ_T_PIP_KERNEL).
 * "pip_exact_stream" 0 / 1                    for a caller that keeps several steps in flight (step k + 1 issued before
 *                                             step k's results are read): the exact kernel of a PIP query that runs on
 *                                             the handle's second stream (see "pip_concurrent") goes to a third stream
 *                                             behind its walk, so the NEXT query's walk starts beside it instead of after
 *                                             it.  Consecutive asynchronous PIP queries must then write to DIFFERENT
 *                                             output arrays (a pipelined caller double-buffers them anyway: step k's
 *                                             answers are still being read); which kernels ran is unchanged.
 * "timers"           1 / 0                    record the stage timers behind rj_last_ms (two event records per stage,
 *                                             ~1 % of a 0.9 ms step); while "pip_concurrent" 2 is still trying schedules
 *                                             they are recorded regardless.
 * "stats"            0 / 1                    the instrumented kernels (visit counters for rj_last_stats; slower).
 * "own_stream"       1                        back to the handle's private stream (see rj_set_stream).
 *
 * rj_get_option reads any of these, and what the handle did or decided:
 *   "leaf_order_used0/1", "leaf_slots0/1", "leaf_runs0/1", "skyline_used0/1", "closed_chains0/1",
 *   "pip_columns_used0/1", "pip_column_entries0/1", "pip_column_shift0/1"   the index of map 0 / 1
 *   "stitch_rounds", "stitch_loop_ends"                         the last run cutting (pointer-jumping rounds; closed loops)
 *   "pip_schedule" (0 turns, 1 shared, 2 full grids, -1 still trying), "pip_schedule_trials", "pip_schedule_us0/1/2",
 *   "lsi_share_blocks", "pip_share_blocks"                      what "pip_concurrent" 2 measured and settled on
 *   "lsi_last_segments", "pip_last_walk_points", "pip_last_passes", "lsi_points_last_split", "lsi_points_gcd_pairs",
 *   "pip_rest", "pip_rest_aux", "query_last_ordered", "pip_last_columns"            what the last query ran
 *   "comm_ranks"                                                the ranks RCCL counts in the handle's communicator (0: none) */
int rj_set_option(rj_handle h, const char* name, int64_t value);
int rj_get_option(rj_handle h, const char* name, int64_t* value);
/* rj_get_plan: what the handle ran for the last query of each kind and on what grounds, as ONE JSON object (text):
 *   "epoch"     counts the events after which the handle decides again -- rj_upload_map, rj_build_lbvh, another query size,
 *               "pip_concurrent"; every record below carries the epoch it was made in ("current": still that epoch)
 *   "index"     per map: levels, slots, what a leaf is, skyline, column index
 *   "schedule"  of an LSI + PIP pair: the choice ("turns" / "shared" / "full grids" / "undecided"), the trials run, the best
 *               span per schedule, the epoch it was settled in and whether it is in force, the shared grids
 *   "lsi", "records", "pip"   kernel, grid, queries per lane, stream, processing order, passes, and -- where the
 *               handle took another path than the default -- why
 * Once settled a decision stays until the epoch moves.  Host-side state only: no synchronisation, no launch.  The text
 * is written to buf (NUL-terminated, truncated to cap) and its full length to *need; either may be NULL / 0. */
int rj_get_plan(rj_handle h, char* buf, size_t cap, size_t* need);
/* Experiment knobs for tools/ and the fault-path tests -- grids, chunk sizes, run lengths ("chunk_groups",
 * "group_lanes", "max_blocks", "lsi_share_blocks", "pip_share_blocks", "stack_cap", "walk_stack", "strip_shift", "run_cap", "pack_solo",
 * "pack_spread"; rj_api.hip lists their ranges).  Not needed by a host of the library, never a correctness input,
 * no promise that a name survives a round. */
int rj_set_debug_option(rj_handle h, const char* name, int64_t value);
int rj_get_debug_option(rj_handle h, const char* name, int64_t* value);

/* ---- device memory helpers (for hosts without their own allocator) -------------------- */
int rj_dev_alloc(rj_handle h, size_t bytes, void** out_dev);
int rj_dev_free(rj_handle h, void* dev);
int rj_memcpy_h2d(rj_handle h, void* dst_dev, const void* src, size_t bytes);
int rj_memcpy_d2h(rj_handle h, void* dst, const void* src_dev, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* RAYJOIN_AMD_H */
