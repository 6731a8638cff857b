// rj_kernels.h -- kernel argument blocks and launch wrappers (implemented in rj_kernels.hip)
#pragma once
#include "rj_device.h"

namespace rj {

struct XsectRec {  // == rj_xsect (include/rayjoin_amd.h) == dev::Intersection<int64_t>, 48 bytes
  int64_t x_num, x_den, y_num, y_den;
  uint32_t eid0, eid1;
  int32_t mid, pad;
};
static_assert(sizeof(XsectRec) == 48, "Intersection record must be 48 bytes");
static_assert(sizeof(Seg) == 32 && sizeof(QBox) == 16, "layout");

// what the first `blocks` blocks of k_pip_exact need to run k_pip's traversal over the walk's overflowed lists
// (everything else is the PipArgs of the launch)
struct PipRestArgs {
  const uint32_t* order;                 // the walk's rest list
  const unsigned long long* n_dev;       // ... and its count, on the device
  unsigned long long* rest_count;        // mapped host word: the count, for the next launch's grid
  unsigned int* work_counter;            // a scheduler block of its own (the exact blocks use none)
  unsigned int* next_work_counter;
  uint32_t group_lanes, chunk_groups;
  uint32_t blocks;                       // 0: none (the kernel is k_pip_exact proper)
};

struct LsiArgs {
  DeviceBvh bvh;
  const Seg* qseg;     // query map segments in eid order
  const uint32_t* qcode; // their 4-byte cell codes (the pre-filter's stream)
  const uint32_t* order; // nullable: Morton-sorted positions (relative to qbeg) for incoherent query sets
  uint64_t qbeg, qend; // query eid range
  int base_is_map0;
  uint32_t* out;       // pairs out [2*cap]
  uint64_t cap;
  unsigned long long* counter;  // result count
  unsigned int* work_counter;   // dynamic chunk scheduler (zero when the kernel starts)
  // the counters of the NEXT launch on this stream: this kernel clears them (no fill kernels between launches)
  unsigned long long* next_counter;
  unsigned int* next_work_counter;
  uint32_t chunk_groups;        // consecutive groups per chunk
  uint32_t group_lanes;         // queries per wave (0 = choose from the query count)
  int stack_cap;                // instrumented kernel only: use fewer stack entries (tests of the fault path)
  unsigned long long* stats;    // [4] or nullptr
};

struct PipArgs {
  DeviceBvh bvh;
  DeviceMap base;
  const int64_t* pts;  // query points (x,y interleaved), already offset to the first point
  const uint32_t* order; // nullable: Morton-sorted point indices for incoherent query sets
  uint64_t n;
  int query_map_id;
  uint32_t* closest;   // [n]
  int32_t* face;       // [n] or nullptr
  unsigned int* work_counter;
  unsigned int* next_work_counter;  // the next launch's scheduler counters on this stream: cleared by this kernel
  uint32_t chunk_groups;
  uint32_t group_lanes;  // points per wave (0 = choose from the point count)
  int stack_cap;         // instrumented kernel only: use fewer stack entries (tests of the fault path)
  int walk_stack;        // k_pip_walk*: use fewer than kWalkStack entries (0: all; tests of the groups that leave the walk)
  unsigned long long* stats;
  // k_pip_walk hand-over.  A point the walk could not settle with integer tests alone leaves its complete candidate
  // list in `todo` (room for one 24-byte record per query position; a group's records lie side by side at the head of
  // the group's region, in the order of the bits of its 64-bit mask: no atomics, no capacity to run out of, and a
  // group with a dozen listed points is three lines) and k_pip_exact evaluates it, no traversal; a point whose list overflowed goes to
  // `rest` and k_pip locates it from scratch (order = rest, n_dev = the count as the walk left it on the device).
  uint32_t* todo;                        // [n][pip_walk_list_slots()] sorted slots of the index, 0xFFFFFFFF = unused (walk, k_pip_exact)
  unsigned long long* todo_mask;         // [groups] (walk, k_pip_exact)
  uint32_t* rest;                        // [n] (walk only)
  unsigned long long* rest_count;        // (walk) zero when the kernel starts; (k_pip) nullable: where it reports the count it found
  unsigned long long* next_rest_count;   // (walk) the next walk's counter on this stream: cleared by this walk
  const unsigned long long* n_dev;       // (k_pip only) nullable: process min(n, *n_dev) queries
};

// ---- -mode=grid on the device (rj_grid.hip) ------------------------------------------------
struct GridLsiArgs {
  int g;
  double scale;  // grid_size / INTERNAL_RANGE * 0.999 (cell.h:16-22), computed once on the host
  const uint32_t *begin0, *eids0, *begin1, *eids1;  // per-map CSR over g*g cells, eids ascending inside a cell
  const Seg *seg0, *seg1;
  uint32_t* out;  // (eid map 0, eid map 1) pairs
  uint64_t cap;
  unsigned long long* counter;
};
struct GridPipArgs {
  int g;
  double scale;
  const uint32_t *begin, *eids;  // the base map's CSR
  DeviceMap base;
  const int64_t* pts;
  uint64_t n;
  int query_map_id;
  uint32_t* closest;
  int32_t* face;  // nullable
};
hipError_t launch_grid_count(hipStream_t st, const Seg* seg, uint64_t ne, int g, double scale, uint32_t* counts,
                             unsigned long long* total);
hipError_t launch_grid_emit(hipStream_t st, const Seg* seg, uint64_t ne, int g, double scale, uint64_t* keys,
                            unsigned long long* cursor);
hipError_t launch_grid_unpack(hipStream_t st, const uint64_t* keys, uint64_t n, uint32_t* eids);
hipError_t scan_cell_counts(hipStream_t st, void* temp, size_t& temp_bytes, const uint32_t* counts, uint32_t* begin, uint64_t n);
hipError_t launch_lsi_grid(hipStream_t st, const GridLsiArgs& a);
hipError_t launch_pip_grid(hipStream_t st, const GridPipArgs& a);

hipError_t launch_build_segs(hipStream_t st, const int64_t* pts, const uint32_t* edge_begin,
                             uint32_t nc, uint64_t ne, Seg* seg, uint32_t* edge_chain, uint32_t* ccode);
hipError_t launch_morton(hipStream_t st, const Seg* seg, uint64_t ne, MortonKey* keys, uint32_t* vals);
hipError_t sort_morton_pairs(hipStream_t st, void* temp, size_t& temp_bytes, const MortonKey* kin, MortonKey* kout,
                             const uint32_t* vin, uint32_t* vout, uint64_t n);
hipError_t sort_pairs_u64_u32(hipStream_t st, void* temp, size_t& temp_bytes, const uint64_t* kin,
                              uint64_t* kout, const uint32_t* vin, uint32_t* vout, uint64_t n, unsigned begin_bit = 0,
                              unsigned end_bit = 64);
hipError_t sort_keys_u64(hipStream_t st, void* temp, size_t& temp_bytes, const uint64_t* kin,
                         uint64_t* kout, uint64_t n);
hipError_t launch_xsect_keys(hipStream_t st, const XsectRec* rec, uint64_t n, int im, uint64_t* keys, uint32_t* vals);
hipError_t launch_xsect_gather(hipStream_t st, const XsectRec* in, const uint32_t* order, uint64_t n, XsectRec* out);
hipError_t launch_xsect_order_runs(hipStream_t st, XsectRec* rec, uint64_t n, int im, const Seg* seg_im, int64_t* midpts);
hipError_t launch_xsect_set_mid(hipStream_t st, XsectRec* rec, uint64_t n, int im, const int32_t* face);
hipError_t launch_swap_halves(hipStream_t st, uint64_t* v, uint64_t n);
// One empty kernel per translation unit: the first launch from a file loads that file's code object (10-60 ms under a
// profiler, ~15 ms without) -- rj_create launches these so that no upload, build or query pays it.
hipError_t warm_query_kernels(hipStream_t st);
hipError_t warm_grid_kernels(hipStream_t st);
hipError_t warm_stitch_kernels(hipStream_t st);
hipError_t warm_strip_kernels(hipStream_t st);
// The column index of an indexed map (rj_strip.hip, rj_device.h DeviceStrips) and the PIP pass on it.  Both build passes
// answer a call with temp == nullptr with the temporary bytes they need.
hipError_t launch_strip_width(hipStream_t st, const QBox* box0, const uint32_t* seid, uint64_t n0p, unsigned long long* out2);
hipError_t launch_strip_count(hipStream_t st, const QBox* box0, const uint32_t* seid, uint64_t n0p, int shift, uint32_t* cnt, uint32_t* offs,
                              void* temp, size_t& temp_bytes, uint32_t* flag);
hipError_t launch_strip_fill(hipStream_t st, const QBox* box0, const uint32_t* seid, const int32_t* sface, const uint32_t* cnt,
                             const uint32_t* offs, uint64_t n0p, int shift, uint64_t entries, uint32_t* key, uint32_t* eslot, uint32_t* key_tmp,
                             uint32_t* slot_tmp, uint32_t* tall, uint32_t* ytab, QBox* ebox, uint4* einfo, uint2* tall_end, uint32_t* sky, void* temp,
                             size_t& temp_bytes);
hipError_t launch_pip_strip(hipStream_t st, const PipArgs& a, int max_blocks, int cus);
// Polyline runs of a map, cut on the device (rj_stitch.hip): pieces and runs into caller-owned arrays sized by
// stitch_output_bounds; two host syncs (closed loops left? -- the totals), no read-back of the map.
void stitch_output_bounds(uint64_t nc, uint64_t ne, uint32_t cap, uint64_t* max_pieces, uint64_t* max_runs);
hipError_t stitch_runs_device(hipStream_t st, const int64_t* pts, const uint32_t* edge_begin, uint64_t nc, uint64_t ne, uint32_t cap,
                              uint32_t* piece_begin, uint32_t* piece_len, uint32_t* run_first, uint64_t* nruns, uint64_t* npieces,
                              uint32_t* stats /* [4] nullable: ranking rounds, incidences on closed loops, rounds of the second ranking, closed chains */,
                              char** scratch, size_t* scratch_bytes /* the caller's grow-only device block: grown here when too small, never freed here */);
hipError_t launch_run_keys(hipStream_t st, const Seg* seg, const uint32_t* piece_begin, const uint32_t* piece_len, const uint32_t* run_first,
                           uint64_t nruns, MortonKey* keys, uint32_t* vals, uint32_t* run_len, QBox* run_box, uint32_t box_upto);
uint64_t pack_runs_chunks(uint64_t nruns);
hipError_t launch_pack_runs(hipStream_t st, const uint32_t* order, const uint32_t* run_len, const QBox* run_box, uint64_t nruns,
                            uint32_t solo_above, uint32_t spread, uint32_t* chunk_leaves, uint32_t* chunk_base, uint32_t* leaf_first,
                            unsigned long long* total_out);
hipError_t launch_build_leaves(hipStream_t st, const Seg* seg, const uint32_t* order, const uint32_t* edge_chain,
                               const uint32_t* left, const uint32_t* right, uint64_t ne, const uint32_t* piece_begin,
                               const uint32_t* piece_len, const uint32_t* run_first, const uint32_t* run_len, const uint32_t* leaf_first, uint64_t nblocks,
                               uint64_t n_parent_alloc, Seg* sseg, uint32_t* seid, int32_t* sface, QBox* box0,
                               int32_t* pmx1, uint2* xtab, QBox* lvl1, uint32_t* occ, uint2* ytab2 = nullptr);
hipError_t launch_occ_count(hipStream_t st, const uint32_t* occ, unsigned long long* part16, unsigned long long* out_mapped);
hipError_t launch_build_sky(hipStream_t st, const QBox* box0, const uint32_t* seid, uint64_t n0p, uint32_t* sky);
hipError_t launch_sibling_order(hipStream_t st, const QBox* box, uint64_t n_alloc, uint64_t* higher);
hipError_t launch_reduce_level(hipStream_t st, const QBox* child, uint64_t n_child_alloc, QBox* parent,
                               uint64_t n_parent_alloc);
// what the last launch_* wrapper of this thread launched (the kernel's name and its grid): rj_get_plan reports it
struct LaunchNote { const char* kernel; int grid; int per_lane; };
LaunchNote last_launch();
LaunchNote last_strip_launch();  // (k_pip_strip lives in its own translation unit)
hipError_t launch_lsi(hipStream_t st, const LsiArgs& a, bool stats, int max_blocks, int segs_per_lane = 1, int* segs_used = nullptr);
hipError_t launch_group_extent(hipStream_t st, bool points, const int64_t* pts, const Seg* segs, uint64_t begin,
                               uint64_t n, unsigned long long* out2);
hipError_t launch_group_extent_tail(hipStream_t st, const int64_t* pts, const uint32_t* order, uint64_t n, unsigned long long* out_mapped);
hipError_t launch_query_keys(hipStream_t st, bool points, const int64_t* pts, const Seg* segs, uint64_t begin,
                             uint64_t n, MortonKey* keys, uint32_t* vals, int strip_shift = 0);
hipError_t launch_lsi_points(hipStream_t st, const Seg* seg0, const Seg* seg1, const uint32_t* pairs,
                             uint64_t n, const unsigned long long* n_dev, XsectRec* out, uint32_t* slow_list,
                             unsigned long long* slow_count, unsigned long long* next_slow_count,
                             unsigned long long* count_hint);
hipError_t launch_pip(hipStream_t st, const PipArgs& a, bool stats, int max_blocks);
hipError_t launch_pip_walk(hipStream_t st, const PipArgs& a, bool stats, int max_blocks);
hipError_t launch_pip_walk2(hipStream_t st, const PipArgs& a, int max_blocks, int cus, bool stats = false, int points = 2);
int pip_walk2_blocks_per_cu(int top, int points = 2);
int pip_walk2_blocks_beside(int top, int lsi_blocks_per_cu, int points = 2);
hipError_t launch_pip_exact(hipStream_t st, const PipArgs& a, int blocks, const PipRestArgs& r);
int pip_walk_list_slots();  // candidates a todo record holds
uint32_t pip_walk_group_lanes(uint64_t n, int top, int cus);  // points per wave k_pip_walk uses when the caller leaves it open
int pip_walk_blocks_per_cu(int top);  // resident 256-thread blocks of k_pip_walk per compute unit for a tree of this height

}  // namespace rj
