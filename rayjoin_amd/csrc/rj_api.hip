// rj_api.hip -- the C ABI of include/rayjoin_amd.h on top of the kernels in rj_kernels.hip.
// Host-side orchestration only (allocation, upload, launch order, timing events).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rayjoin_amd.h"
#include "rj_kernels.h"

using namespace rj;

namespace {

struct MapState {
  bool present = false;
  uint64_t np = 0, ne = 0, nc = 0;
  int64_t* pts = nullptr;
  Seg* seg = nullptr;
  uint32_t* edge_chain = nullptr;
  uint32_t* ccode = nullptr;  // 4-byte cell code per edge (k_lsi's pre-filter stream)
  uint32_t* left = nullptr;
  uint32_t* right = nullptr;
  uint32_t* edge_begin = nullptr;  // [nc + 1] first eid of every chain (+ sentinel): row_index[c] - c
  // "leaf_order" 1: the polyline runs of this map, cut ON THE DEVICE (rj_stitch.hip) the first time an index of it is
  // built and kept: piece p = eids [piece_begin[p], + piece_len[p]), run r = pieces [run_first[r], run_first[r + 1])
  bool runs_cut = false;
  uint32_t run_cap = 0;  // edges a run may hold (what the runs were cut with)
  uint64_t closed_chains = 0;  // rings among the chains (counted while the runs were cut)
  // ... and how the sorted runs share leaves (k_pack_runs): run_len[nruns], leaf_first[leaves + 1] (positions in the sorted
  // order), scratch of the packing passes; the leaf count, known after the first build
  uint32_t *run_len = nullptr, *leaf_first = nullptr, *pack_tmp = nullptr;
  QBox* run_box = nullptr;
  uint64_t packed_leaves = 0;
  uint32_t packed_solo = 0;
  uint64_t nruns = 0, npieces = 0;
  uint32_t *piece_begin = nullptr, *piece_len = nullptr, *run_first = nullptr;
};

struct BvhState {
  bool built = false;
  int leaf_order = 0;  // how this index's leaves were formed (0 Hilbert neighbours, 1 chain runs)
  uint64_t n0 = 0, n0p = 0;
  char* pool = nullptr;  // ONE device block for the arrays below and the levels where they are small (eleven hipMallocs were 0.2 ms of a 1.5 ms
                         // first build); nullptr: an allocation each (large maps: ONE hipMalloc of 4.6 GB took 120 ms where the eleven take 1)
  Seg* sseg = nullptr;
  uint32_t* seid = nullptr;
  int32_t* sface = nullptr;
  QBox* box0 = nullptr;
  int32_t* pmx1 = nullptr;
  uint2* xtab = nullptr;
  uint2* ytab2 = nullptr;   // the second order of blocks taller than wide (k_build_leaves; "leaf_ysort")
  uint32_t* occ = nullptr;
  uint32_t* sky = nullptr;  // skyline of the map (rj_device.h kSkyShift): what the PIP kernels prove a miss with
  bool use_sky = false;     // ... filled and used (maps of isolated rings)
  // the column index (rj_device.h DeviceStrips; maps of isolated rings): what the PIP query's first pass runs on instead
  // of the tree
  uint32_t* strip_ytab = nullptr;
  uint4* strip_info = nullptr;   // [strip_cap] {slot, edge id, face id, 0} per entry
  uint2* strip_tall = nullptr;   // [strips] {tallest box, end of the strip's entries}
  int strip_shift = 0;           // a strip is 2^strip_shift quanta wide (chosen by the map's segments, build_strips)
  uint32_t occ_permille = 0;     // set bits of the occupancy bitmap per thousand cells (the build counts them)
  bool ysort = false;            // the leaves' blocks may be ordered by y (k_build_leaves: blocks taller than wide)
  const char* columns_why = "";  // rj_get_plan: on what grounds the last build made (or did not make) the column index
  int strip_tab_shift = 0;       // ... and the width strip_tall / strip_ytab are allocated for (0: not yet)
  QBox* strip_box = nullptr;
  uint64_t strip_entries = 0, strip_cap = 0;
  bool strips_built = false;
  QBox* lvl[kMaxLevels] = {nullptr};  // boxes of level l, then one sibling-order word per node (rj_device.h)
  uint64_t nlvl[kMaxLevels] = {0};
  uint64_t alloc[kMaxLevels] = {0};
  int top = 0;
};

struct GridState {  // -mode=grid: one CSR per map (rj_grid.hip)
  bool built = false;
  int g = 0;
  double scale = 0;
  uint32_t* begin = nullptr;  // [g*g + 1]
  uint32_t* eids = nullptr;   // [total], ascending inside a cell
  uint64_t total = 0;
};

constexpr int kNumTimers = 12;
// average (w + h) of a 64-query group's quantised box above which the query set is re-ordered
// along the Morton curve before the kernels run (the domain is 2^31 wide per axis)
constexpr unsigned long long kIncoherentExtent = 1ull << 28;
constexpr uint64_t kExchHead = RJ_EXCHANGE_HEAD_WORDS;  // 32-bit words in front of the pairs of an exchange buffer: count (u64), capacity (u64)
constexpr int kCoTrials = 4;  // measured LSI + PIP pairs before "pip_concurrent" 2 settles on a schedule

}  // namespace

struct rj_handle_s {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t aux_stream = nullptr;  // "pip_concurrent": PIP launches go here so that they overlap the LSI kernel
  int pip_concurrent = 0;  // 0 never; 1: the caller issues LSI and PIP queries in pairs and they share the chip; 2: auto (see co_* below)
  int cus = 256;           // compute units of the device
  bool lsi_inflight = false;  // an asynchronous LSI query was launched and not yet finished / synced
  bool lsi_shared = false;    // ... on a reduced grid, waiting for its PIP partner
  // "pip_concurrent" 2: three schedules for an LSI + PIP pair -- 0: one after the other on the main
  // stream; 1: beside each other on shared grids (k_lsi 1.25-2.25 blocks per CU, k_pip 5-4, see co_ratio); 2: beside each
  // other, each on its own full grid (whichever starts first fills the chip, the other fills its
  // ramp and tail).  Which one wins depends on the workload (headline pair: 1 by 9 %; 24-67 M-segment
  // trees: 0 by 35 % over 1; a 1/8 shard against the 24 M-segment tree: 2), so the first pairs run
  // 0, 1, 1 (split corrected), 2 and the fastest is kept until the index, the maps or the query size change:
  // settled from the fifth pair on.
  // The span of a pair = start of its LSI kernel .. end of the last of k_lsi, k_lsi_points, k_pip.
  int co_trials = 0;          // pairs measured so far
  float co_best[3] = {1e30f, 1e30f, 1e30f};  // best span [ms] per schedule
  int co_choice = -1;         // decided schedule, -1 while trying
  int co_mode = 0;            // schedule of the pair in flight
  bool co_measure = false;    // the pair in flight is complete (LSI + PIP): its span can be read
  bool co_points = false;     // ... and k_lsi_points ran between them
  uint64_t co_n = 0;          // query size the decision was made for (LSI segments)
  uint64_t co_np = 0;         // ... and the PIP side's (points)
  uint64_t co_np_other = 0;   // a point count that differed from co_np once: the trials start again when it is seen a second time
  // how the chip is split when shared.  Both kernels are persistent, so what counts is what fits a CU together: the
  // register file takes 6 walk blocks + 2 k_lsi blocks, and that is where the step is shortest wherever the walk may keep
  // 8 blocks per CU (tree of <= 4 levels); where LDS limits it to 7 (5 levels) the best k_lsi grid is half a block per CU
  // larger.  Around that, a mild dependence on the ratio of the two sides' solo times seen in the "taking turns" trial,
  // r = (k_lsi + its records) / (the PIP kernels).  Swept per pair with tools/share_probe.py on the round-3 kernels
  // (profiles/r03_d_share_sweep.txt; optimum k_lsi grid at the warm r): Zipcode 0.52 -> 448-512, headline 0.53 -> 512,
  // nested 0.72 -> 512, WaterBodies 0.77 -> 640-704, LakesNA 0.87 -> 640-704 blocks of 256 CUs' worth:
  // 512 + 128 (8 - walk blocks per CU) + 200 (r - 0.6), within 1 % of the best step on all five; the second shared
  // trial then tries the neighbouring grid on the side the first one's imbalance points to, and the better one stays.
  // The PIP side (launched second) fills what is left.
  float co_ratio = 0.46f;
  int co_wpc = 8;  // walk blocks a CU can hold for the base tree of the pair being scheduled
  int lsi_share_set = 0, pip_share_set = 0;  // "lsi_share_set" / "pip_share_set": fixed grids for schedule 1 (0 = derive them, the default)
  int co_L = 0, co_best_L = 0;  // the k_lsi grid of the next shared pair / of the best one measured (0: the formula below)
  float co_best_imb = 0;        // ... and how far apart its two sides ended, as a fraction of its span
  int lsi_share_blocks() const {
    if (lsi_share_set) return lsi_share_set;
    if (co_L) return co_L;
    // (in steps of half a block per CU since round 5: at 2.25 blocks per CU a quarter of the CUs hold a third k_lsi2 block
    //  and one walk block less than fits -- the nested pair at 448 / 512 / 576 / 640 blocks: 1.34 / 1.29 / 1.44 / 1.30 ms --
    //  and a fit that lands on either side of such a step from run to run is not a schedule)
    const int half = cus / 2 > 0 ? cus / 2 : 1;
    const int b = ((int) ((512.0f + 128.0f * (float) (8 - co_wpc) + 200.0f * (co_ratio - 0.6f)) * (float) cus / 256.0f) + half / 2) / half * half;
    return b < cus ? cus : (b > cus * 4 ? cus * 4 : b);
  }
  int last_pip_share = 0;  // the PIP side's grid in the last shared pair (what "pip_share_blocks" reports)
  int pip_share_blocks() const { return pip_share_set ? pip_share_set : cus * 5; }  // (k_pip without the walk; the walk gets its full grid less one block per CU)
  bool aux_pending = false;          // something was enqueued on aux_stream since it was last joined
  hipEvent_t ev_order = nullptr;     // "taking turns" under "pip_concurrent" 2: the PIP kernels still use aux_stream, behind this event
  hipStream_t stream = nullptr;
  MapState map[2];
  BvhState bvh[2];
  // d_counter (u64 words): [0],[1] LSI result count (alternating: flip_lsi), [6] the grid LSI's; [2],[3] grid build; [4],[5] group-extent estimate;
  // then three scheduler blocks of 8 counters 128 B apart: LSI, PIP on the main stream, PIP on aux
  unsigned long long* d_counter = nullptr;
  int flip_lsi = 0, flip_pip[2] = {0, 0};  // which of the two counter sets the next launch uses (LSI; PIP on main / aux stream)
  // PIP in two passes (rj_kernels.hip, k_pip_walk): the integer-only walk settles what it can, k_pip takes the rest
  int pip_walk = 1;                        // "pip_walk": 1 auto (default), 0 k_pip alone, 2 always both passes
  int last_walk_points = 1;                // ... and how many points a lane of its walk took
  int last_passes = 0;                     // kernels of the last PIP query (3 or 1)
  int flip_walk[2] = {0, 0};
  // list sets: [0] main stream, [1] aux, [2] aux again -- "pip_exact_stream" alternates [1] and [2], so that the walk of
  // query k + 1 fills one set while the exact kernel of query k, on its own stream, still reads the other
  uint32_t* rest[3] = {nullptr, nullptr, nullptr};  // points the walk left to k_pip (grow-only)
  uint64_t rest_cap[3] = {0, 0, 0};
  uint32_t* todo[3] = {nullptr, nullptr, nullptr};  // candidate lists the walk left to k_pip_exact, one slot per query position (grow-only)
  unsigned long long* todo_mask[3] = {nullptr, nullptr, nullptr};  // ... and which slots of a group are filled
  // "pip_exact_stream" 1: the exact kernel of a PIP query on the aux stream runs on a third stream, behind its walk by an
  // event -- the walk of the NEXT query (aux stream) starts while it runs instead of after it.  Three rest-count words in
  // rotation (a walk clears the next one's: the one after that may still be read by the exact kernel two queries back,
  // which the walk waits for -- it is also the last reader of the list set the walk is about to fill).
  int exact_own_stream = 0;
  hipStream_t exact_stream = nullptr;
  hipEvent_t ev_walk_done = nullptr, ev_exact_done[2] = {nullptr, nullptr};
  bool exact_recorded[2] = {false, false};  // ev_exact_done[b] has been recorded at least once
  int exact_buf = 0, exact_last = -1;       // the list set (1 + exact_buf) the next such query uses; the one the last used
  int exact_rot = 0;                        // ... and its rest-count word (kExactRestWord)
  bool exact_pending = false;               // something was enqueued on exact_stream since it was last joined
  const void *exact_out_closest = nullptr, *exact_out_face = nullptr;  // the output arrays of the last query whose exact kernel went to exact_stream
  unsigned long long* h_rest = nullptr;    // mapped host words [0],[1]: the rest count of the last finished query per stream (a hint; [2]: see lsi_points_on_stream
  unsigned long long* d_rest = nullptr;    // for the next launch's grid and for "pip_rest"; the same memory as the device sees it)
  uint64_t walk_n[2] = {0, 0};             // size of the query the hint belongs to
  size_t count_word = 0;                   // where the latest LSI query's result count lives
  // k_lsi_points: the pairs its gcd-free leg declines, for k_lsi_points_gcd (grow-only; counts at d_counter[12],[13], alternating)
  uint32_t* slow_list = nullptr;
  uint64_t slow_cap = 0;
  int flip_slow = 0;
  int last_lsi_segments = 1;               // what the last LSI query ran
  int lsi_segments = 2;                    // "lsi_segments": 2 (default): k_lsi2, two query segments per lane, where it applies; 1: k_lsi always
  int walk_points = 2;                     // "pip_walk_points": 2 (default): k_pip_walk2, two query points per lane, where it applies; 1: k_pip_walk always
  int timers = 1;                          // "timers": 1 (default) the stage timers behind rj_last_ms are recorded, 0 they are not
  int points_split = -1;                   // "lsi_points_split": -1 by the last count (default), 0 never, 1 always
  int last_points_split = 0;               // what the last records launch did
  unsigned long long* d_stats = nullptr;    // [16]
  unsigned long long* d_occ_part = nullptr; // [16] partial sums of the occupancy bitmap's count (k_occ_count)
  unsigned long long* h_pinned = nullptr;   // [32] pinned read-back area
  // Traversal-stack fault words, [0] LSI [1] PIP: pinned host memory the kernels write directly
  // (never in practice: rj_device.h), so every sync point can check them without a copy
  uint32_t* h_fault = nullptr;
  uint32_t* d_fault = nullptr;              // the same memory as the device sees it
  int debug_stack_cap = 1 << 30;            // tests of the fault path only
  int debug_walk_stack = 0;                 // tests of the groups that leave the walk (0: kWalkStack entries)
  int debug_strip_shift = 0;                // the column index of the next build on strips of 2^this quanta (0: by the map)
  int debug_query_key_strips = 0;           // experiment: a re-ordered PIP query set over a column index is sorted strip-major (strip, then y), not by Morton key
  int order_strip_shift = 0;                // ... the strip width the next query-key pass sorts by (0: Morton keys)
  // Round 6: a spatially INCOHERENT point set (the reference's GeneratePIPQueries: uniform random points) is answered by the
  // column index -- every point on its own, nothing to share, nothing to sort -- built at the first such query of at least
  // `lazy_columns_min` points when "pip_columns" is auto and the base map has none (8.4 M uniform random points, PIP query:
  // USCounty 1.07 -> 0.70 ms, BlockGroup 1.63 -> 0.75, LakesNA 1.91 -> 0.43 -- tools/incoherent_columns_probe.py).
  int debug_lazy_columns_min = 0;           // (0: 2^22 points)
  bool lazy_columns_ok = false;             // set around the coherence estimate of a PIP query: "say so instead of sorting"
  bool lazy_columns_want = false;           // ... the estimate's answer
  hipEvent_t ev[kNumTimers][2];
  bool ev_valid[kNumTimers] = {false};
  bool stats_on = false;
  GridState grid[2];
  int query_order = 1;  // 0 never, 1 auto (estimate coherence), 2 always
  bool last_ordered = false;
  // the permutation of the query being issued was sorted IN THIS CALL (main stream, shared sort scratch): such a query
  // runs alone; a permutation that comes from a cache (a private buffer, complete long ago) pairs like no permutation
  bool order_fresh = false;
  // coherence decisions for map-owned query sets (immutable after upload): [kind 0=segs,1=points][map]
  struct CohCache { bool valid = false; uint64_t begin = 0, n = 0; bool incoherent = false; } coh[2][2];
  // ... and the Morton permutation of such a set, kept until the map is uploaded again: repeated
  // queries over the same immutable edges / vertices sort once (an index on the query side)
  struct OrdCache { uint32_t* perm = nullptr; uint64_t cap = 0, begin = 0, n = 0; bool valid = false; } ordc[2][2];
  // Caller-owned point arrays (rj_pip_query* with pts_dev, the reference's PIP::Query(Stream&, int, ArrayView<point_t>),
  // src/app/pip.h:23): what the handle has learned about the array at (pointer, size).  The contents may change between
  // queries, so nothing here is ever a correctness input -- a permutation of [0, n) stays one whatever the points are --
  // only a choice of processing order: the first query over an array estimates its coherence with one host round trip
  // (like the first-use allocations), every later one reads the estimate a one-block kernel behind an earlier query's
  // kernels left in mapped host memory (k_group_extent_tail, measured through the permutation in use): no
  // synchronisation between rj_pip_query_async and its kernels, the query pairs with an LSI query in flight like a
  // map-owned one, and an array whose contents turned incoherent is re-sorted by the query after the one that saw it.
  struct CallerSet {
    const void* p = nullptr;
    uint64_t n = 0;
    bool valid = false, has_perm = false, fresh_perm = false;
    uint32_t* perm = nullptr;
    uint64_t perm_cap = 0;
    unsigned long long sorted_extent = 0;  // mean group extent right after the last sort (what "still sorted" looks like)
    uint64_t queries = 0, stamp = 0;
  };
  static constexpr int kCallerSets = 4;
  CallerSet caller[kCallerSets];
  unsigned long long* h_est = nullptr;  // mapped host words, one per set: mean group extent + 1 of the last estimate (0: none yet)
  unsigned long long* d_est = nullptr;
  uint64_t caller_clock = 0;
  int cur_caller = -1;                  // the set of the query being issued (-1: map-owned or too small to matter)
  const uint32_t* cur_order = nullptr;
  // grow-only scratch of the query-ordering pass
  uint64_t ord_cap = 0;
  MortonKey *ord_kin = nullptr, *ord_kout = nullptr;
  uint32_t *ord_vin = nullptr, *ord_vout = nullptr;
  void* ord_temp = nullptr;
  size_t ord_temp_bytes = 0;
  int leaf_order = 1;        // "leaf_order" (default 1, or RJ_LEAF_ORDER): what the NEXT rj_build_lbvh makes a leaf of
  int debug_run_cap = 0;     // experiments: edges per polyline run (0: by the mean chain length)
  int debug_pack_solo = 0;   // experiments: a run longer than this never shares its leaf (0: the default, 48)
  int debug_pack_spread = 0; // experiments: how many times larger than its runs a shared leaf may be (0: the default, 8)
  uint32_t stitch_stats[4] = {0, 0, 0, 0};  // the last run cutting: ranking rounds, incidences on closed loops, rounds of the second ranking, closed chains
  char* strip_scratch = nullptr;  // grow-only temporaries of the column index's build
  size_t strip_scratch_bytes = 0;
  int pip_columns = -1;      // "pip_columns": -1 auto (maps of closed rings or of short chains), 0 never, 1 always -- the column index of the NEXT rj_build_lbvh
  int last_columns = 0;      // the last PIP query's first pass ran on the column index
  int leaf_ysort = 1;        // "leaf_ysort": 1 (default) the NEXT rj_build_lbvh gives blocks taller than wide a second order, by y, for the LSI kernels; 0: x order only
  int skyline = -1;          // "skyline": -1 auto (maps of isolated rings), 0 never, 1 always -- what the NEXT rj_build_lbvh does
  int max_blocks = 1 << 20;  // cap on the persistent grid (default: whatever is resident)
  int chunk_groups = 0;      // consecutive groups handed to a wave at a time; 0 = per kernel (k_lsi 8, k_pip 6: measured optima; k_pip's waves share a chunk's rest inside the block)
  int group_lanes = 0;       // queries per wave: 0 = automatic (64 unless the query set is small)
  uint64_t last_stats[16] = {0};
  // grow-only arena for the overlay pass and for the run cutting of a first index build (carved per call, no per-call hipMalloc/hipFree)
  char* arena = nullptr;
  size_t arena_bytes = 0;
  ncclComm_t comm = nullptr;
  // The exchange of a step (rj_exchange_*): ONE collective per queue, symmetric on every rank by construction (nothing
  // but ncclAllGather, no rank-dependent branch between collectives).  The pair queues travel as [count u64 | capacity
  // u64 | pairs ...] -- the count is stamped on the device, so no host round trip sits between the LSI kernel and the
  // collective -- on a communication stream of their own behind an event; the PIP result queues on a second
  // communicator and stream (two collectives of one communicator must not be in flight together), so that a step's
  // point gather never queues in front of the next step's pair exchange.  Two buffers each: step k + 1 can be launched
  // before step k's heads have been read.
  ncclComm_t comm2 = nullptr;
  hipStream_t comm_stream = nullptr, comm_stream2 = nullptr;
  hipEvent_t ev_comm = nullptr, ev_comm2 = nullptr, ev_comm2_aux = nullptr;
  struct Exch {
    uint32_t* send = nullptr;   // caller-owned, [RJ_EXCHANGE_HEAD_WORDS + 2 capacity] words: the LSI query writes its pairs behind the head
    uint32_t* recv = nullptr;   // [nranks][kExchHead + 2 slot]
    uint64_t recv_words = 0, slot = 0;
    bool pending = false;
  } ex[2];
  uint64_t ex_capacity = 0, ex_slot = 0;
  int ex_last_begin = -1;   // the buffer of the latest LSI query handed to the exchange
  // rj_lsi_count_async / rj_lsi_count_wait: the count of an asynchronous LSI query read back behind an event, so that the
  // host can launch the next step before it looks at this one's count (two slots)
  hipEvent_t ev_count[2] = {nullptr, nullptr};
  bool count_pending[2] = {false, false};
  unsigned long long* h_heads = nullptr;  // pinned [3][nranks][2]: the gathered (count, capacity) words -- of exchange buffer 0 / 1, of rj_allgather_*
  uint32_t* ag_send = nullptr;  // grow-only staging of the synchronous rj_allgather_* forms (padded slices)
  uint32_t* ag_recv = nullptr;
  uint64_t ag_send_words = 0, ag_recv_words = 0;
  int nranks = 1, rank = 0;
  unsigned long long* d_counts = nullptr;  // [nranks] gathered counts
  // rj_get_plan: what the last query of each kind ran, on what grid, and on what grounds.  `epoch` counts the events that
  // make the handle decide again -- a map upload, an index build, another query size, "pip_concurrent" -- and every record
  // carries the epoch it was made in: a decision older than the handle's epoch is not in force any more.
  struct PlanRec {
    uint64_t epoch = 1;
    uint64_t sched_epoch = 0;  // the epoch in which the schedule in force (co_choice) was settled
    struct Lsi { bool ran = false; uint64_t epoch = 0, n = 0; LaunchNote k = {"", 0, 0}; int order = 0, co_mode = 0; bool paired = false, shared = false; } lsi;
    struct Rec { bool ran = false; uint64_t epoch = 0; bool two_kernels = false, count_on_device = false; uint64_t seen = ~0ull; } rec;
    struct Pip {
      bool ran = false, aux = false, caller = false, shared = false;
      uint64_t epoch = 0, n = 0, rest_hint = ~0ull;
      LaunchNote first = {"", 0, 0};
      int passes = 0, order = 0, exact_blocks = 0, locate_blocks = 0;
      const char* why = "";
    } pip;
  } plan;
  std::string err;
};

namespace {

int fail(rj_handle h, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (h) h->err = buf;
  return code;
}

#define RJ_HIP(h, expr)                                                                        \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess)                                                                      \
      return fail(h, _e == hipErrorOutOfMemory ? RJ_E_NOMEM : RJ_E_HIP, "%s failed: %s (%s:%d)", \
                  #expr, hipGetErrorString(_e), __FILE__, __LINE__);                           \
  } while (0)

#define RJ_CHECK_H(h) \
  if (!(h)) return RJ_E_INVALID

template <typename T>
int dev_alloc(rj_handle h, T** p, uint64_t count) {
  *p = nullptr;
  RJ_HIP(h, hipMalloc((void**) p, (count ? count : 1) * sizeof(T)));
  return RJ_OK;
}

void free_map(MapState& m) {
  (void) hipFree(m.pts); (void) hipFree(m.seg); (void) hipFree(m.edge_chain); (void) hipFree(m.ccode); (void) hipFree(m.left); (void) hipFree(m.right); (void) hipFree(m.edge_begin);
  (void) hipFree(m.piece_begin); (void) hipFree(m.piece_len); (void) hipFree(m.run_first);
  (void) hipFree(m.run_len); (void) hipFree(m.leaf_first); (void) hipFree(m.pack_tmp); (void) hipFree(m.run_box);
  m = MapState();
}

void free_grid(GridState& g) {
  (void) hipFree(g.begin); (void) hipFree(g.eids);
  g = GridState();
}

void free_bvh(BvhState& b) {
  if (b.pool) {
    (void) hipFree(b.pool);  // (sseg ... sky and the levels are carved out of it)
  } else {
    (void) hipFree(b.sseg); (void) hipFree(b.seid); (void) hipFree(b.sface); (void) hipFree(b.box0); (void) hipFree(b.pmx1); (void) hipFree(b.xtab); (void) hipFree(b.ytab2); (void) hipFree(b.occ); (void) hipFree(b.sky);
    for (int l = 0; l < kMaxLevels; l++) (void) hipFree(b.lvl[l]);
  }
  (void) hipFree(b.strip_ytab); (void) hipFree(b.strip_info); (void) hipFree(b.strip_tall); (void) hipFree(b.strip_box);
  b = BvhState();
}

static bool g_lsi_root_skip = true;  // (environment RJ_LSI_ROOT_SKIP=0: A/B runs)

DeviceBvh bvh_view(const BvhState& b) {
  DeviceBvh d;
  d.sseg = b.sseg; d.seid = b.seid; d.sface = b.sface; d.box0 = b.box0; d.pmx1 = b.pmx1; d.xtab = b.xtab; d.ytab2 = b.ysort ? b.ytab2 : nullptr; d.occ = b.occ; d.sky = b.use_sky ? b.sky : nullptr;
  for (int l = 0; l < kMaxLevels; l++) {
    d.lvl[l] = b.lvl[l]; d.nlvl[l] = (uint32_t) b.nlvl[l];
    d.ord[l] = b.lvl[l] ? reinterpret_cast<const uint64_t*>(b.lvl[l] + b.alloc[l]) : nullptr;
  }
  d.top = b.top; d.n0 = b.n0; d.occ_permille = b.occ_permille;
  // (the LSI traversals start one level below a top level that holds a handful of nodes: rj_device.h DeviceBvh::lsi_root)
  d.lsi_root = g_lsi_root_skip && b.top >= 2 && b.nlvl[b.top] <= 4 && b.nlvl[b.top - 1] <= 128 ? b.top - 1 : b.top;
  d.strips.ytab = b.strips_built ? b.strip_ytab : nullptr;
  d.strips.ebox = b.strip_box; d.strips.einfo = b.strip_info; d.strips.tall = b.strip_tall; d.strips.shift = b.strip_shift;
  return d;
}

DeviceMap map_view(const MapState& m) {
  DeviceMap d;
  d.pts = m.pts; d.seg = m.seg; d.edge_chain = m.edge_chain; d.left = m.left; d.right = m.right;
  d.np = m.np; d.ne = m.ne; d.nc = m.nc;
  return d;
}

int set_device(rj_handle h) {
  RJ_HIP(h, hipSetDevice(h->device));
  return RJ_OK;
}

// ("timers" 0: no event records -- eight per step of a join, ~1.2 % of the headline step; while "pip_concurrent" 2 is
//  still trying schedules the records stay, it decides by them)
static inline bool timers_off(rj_handle h) { return !h->timers && !(h->pip_concurrent == 2 && h->co_choice < 0); }
void tic(rj_handle h, int t, hipStream_t st = nullptr) { if (timers_off(h)) return; (void) hipEventRecord(h->ev[t][0], st ? st : h->stream); }
void toc(rj_handle h, int t, hipStream_t st = nullptr) { if (timers_off(h)) return; (void) hipEventRecord(h->ev[t][1], st ? st : h->stream); h->ev_valid[t] = true; }
// wait for a concurrent PIP before anything that frees, rebuilds or consumes what it touches
hipError_t join_aux(rj_handle h) {
  if (!h->aux_pending) return hipSuccess;
  h->aux_pending = false;
  hipError_t e = hipStreamSynchronize(h->aux_stream);
  if (h->exact_pending) {
    h->exact_pending = false;
    const hipError_t e2 = hipStreamSynchronize(h->exact_stream);
    if (e == hipSuccess) e = e2;
  }
  return e;
}

// ---- "pip_concurrent" 2: which schedule for this pair? ------------------------------------------
static void co_reset(rj_handle h) {
  h->plan.epoch++;  // (every decision taken so far is void: rj_get_plan)
  h->co_trials = 0; h->co_best[0] = h->co_best[1] = h->co_best[2] = 1e30f; h->co_choice = -1; h->co_measure = false; h->co_n = 0; h->co_np = 0; h->co_np_other = 0;
  h->co_ratio = 0.46f;
  h->co_L = h->co_best_L = 0;
  h->co_best_imb = 0;
}
static void co_collect(rj_handle h) {  // read the span of the previous pair, if it has completed
  if (!h->co_measure) return;
  h->co_measure = false;
  if (h->co_choice >= 0) return;  // settled: nothing left to learn, and this sits in front of every step's first launch
  if (hipEventQuery(h->ev[RJ_T_LSI_KERNEL][1]) != hipSuccess || hipEventQuery(h->ev[RJ_T_PIP_KERNEL][1]) != hipSuccess) return;
  float a = 0, b = 0;
  if (hipEventElapsedTime(&a, h->ev[RJ_T_LSI_KERNEL][0], h->ev[RJ_T_LSI_KERNEL][1]) != hipSuccess) return;
  if (hipEventElapsedTime(&b, h->ev[RJ_T_LSI_KERNEL][0], h->ev[RJ_T_PIP_KERNEL][1]) != hipSuccess) return;
  float span = a > b ? a : b, c = 0;
  if (h->co_points && hipEventQuery(h->ev[RJ_T_LSI_POINTS][1]) == hipSuccess &&
      hipEventElapsedTime(&c, h->ev[RJ_T_LSI_KERNEL][0], h->ev[RJ_T_LSI_POINTS][1]) == hipSuccess && c > span) span = c;
  if (h->co_mode == 0) {  // taking turns: the solo times of the two sides set the split of the shared schedule
    float pip = 0, pts = 0;
    if (hipEventElapsedTime(&pip, h->ev[RJ_T_PIP_KERNEL][0], h->ev[RJ_T_PIP_KERNEL][1]) == hipSuccess && pip > 0) {
      if (h->co_points && c > 0) (void) hipEventElapsedTime(&pts, h->ev[RJ_T_LSI_POINTS][0], h->ev[RJ_T_LSI_POINTS][1]);
      h->co_ratio = (a + pts) / pip;
    }
  }
  if (h->co_mode == 1 && h->co_choice < 0 && !h->lsi_share_set) {
    // sharing: both sides should end together.  The first split comes from the turns pair's ratio -- the first, cold
    // pair of a workload -- so the second shared trial corrects it by what the first one showed: the side that ended
    // later gets more of the chip.
    const int used = h->lsi_share_blocks();
    const float lsi_side = a > c ? a : c;
    const float imb = (lsi_side - b) / (span > 0 ? span : 1.0f);
    // Which of the two shared grids stays: the neighbour has to win by 3 % -- one sample's noise plus what the first shared
    // pair of a workload pays for being the first: near a tie the grid the sweep's fit gives stays -- or by 1 % if its two
    // sides also end closer together (round 5: the span alone left the nested pair on 576 or 512 blocks from run to run,
    // 1.45 or 1.31 ms per step; balance alone took the headline to 448 + 1 792, whose pipelined step is 4 % slower).
    const float aimb = imb < 0 ? -imb : imb;
    if (h->co_best_L == 0 || span < h->co_best[1] * 0.97f || (span < h->co_best[1] * 0.99f && aimb < h->co_best_imb)) {
      h->co_best_L = used;
      h->co_best_imb = aimb;
    }
    // (one step of half a block per CU: a measured neighbour is worth more than an extrapolated one)
    int next = used + (imb > 0 ? 1 : -1) * (h->cus / 2 > 0 ? h->cus / 2 : 1);
    next = next < h->cus ? h->cus : (next > h->cus * 4 ? h->cus * 4 : next);
    h->co_L = next;
  }
  if (span < h->co_best[h->co_mode]) h->co_best[h->co_mode] = span;
  if (++h->co_trials >= kCoTrials && h->co_choice < 0) {
    h->co_choice = 0;
    for (int m = 1; m < 3; m++) if (h->co_best[m] < h->co_best[h->co_choice]) h->co_choice = m;
    // (one sample per schedule: within 3 % of the best, sharing stays -- it is the schedule both kernels were shaped for,
    //  and the one whose pairs were measured twice)
    if (h->co_choice != 1 && h->co_best[1] <= h->co_best[h->co_choice] * 1.03f) h->co_choice = 1;
    if (h->co_best_L) h->co_L = h->co_best_L;  // (the split of the best shared pair stays)
    h->plan.sched_epoch = h->plan.epoch;
  }
}
static int co_pick(rj_handle h, uint64_t n) {
  co_collect(h);
  if (h->co_n && (n > h->co_n + h->co_n / 4 || n + n / 4 < h->co_n)) co_reset(h);  // another query size: decide again
  h->co_n = n;
  if (h->co_choice >= 0) return h->co_choice;
  // Four measured pairs -- turns (it also measures the split for "shared" and takes the cold start), shared, shared
  // again with the split corrected, full grids -- so the fifth pair already runs the settled schedule: the
  // reference's five warm-up queries (run_query.cu:292-296) are enough.
  static const int kTrial[kCoTrials] = {0, 1, 1, 2};
  return kTrial[h->co_trials < kCoTrials ? h->co_trials : kCoTrials - 1];
}

uint64_t pad64(uint64_t n) { return (n + 63) / 64 * 64; }

constexpr size_t kSchedBlockWords = 8 * 128 / 8 + 16;            // one scheduler block, in u64 words (+ the fault-pointer line)
// Every kernel kind has TWO scheduler blocks per stream (and k_lsi two result counts, words [0] and [1]): a
// launch uses one and clears the other for the next launch on that stream, so no fill kernel sits between
// the host's call and the kernel (each cost the step 6-8 us of launch gap).
constexpr size_t kSchedLsi = 16, kSchedPipMain = kSchedLsi + 2 * kSchedBlockWords, kSchedPipAux = kSchedPipMain + 2 * kSchedBlockWords;
constexpr size_t kSchedWalkMain = kSchedPipAux + 2 * kSchedBlockWords, kSchedWalkAux = kSchedWalkMain + 2 * kSchedBlockWords;
constexpr size_t kCounterBytes = (kSchedWalkAux + 2 * kSchedBlockWords) * 8;
constexpr size_t kSlowCountWord = 12;    // [12],[13] (alternating): how many pairs k_lsi_points left to k_lsi_points_gcd
constexpr size_t kRestCountWord = 8;     // [8],[9] main stream (alternating), [10],[11] aux: how many points k_pip_walk left to k_pip
constexpr size_t kExactRestWord[3] = {10, 11, 14};  // ... the aux stream's under "pip_exact_stream": three in rotation
constexpr size_t kGridLsiCountWord = 6;  // rj_lsi_query_grid's result count (cleared by a fill: not on the hot path)

// after a stream sync: did a traversal stack overflow?  (cannot for an index rj_build_lbvh accepted)
int check_fault(rj_handle h) {
  const uint32_t l = h->h_fault[0], p = h->h_fault[1];
  if (!l && !p) return RJ_OK;
  h->h_fault[0] = h->h_fault[1] = 0;
  return fail(h, RJ_E_INTERNAL, "traversal stack overflow in %s%s%s: results are incomplete", l ? "k_lsi" : "",
              l && p ? " and " : "", p ? "k_pip" : "");
}

}  // namespace

extern "C" {

const char* rj_version(void) { return "rayjoin_amd 0.1.0 (gfx950)"; }

int rj_create(int device_id, rj_handle* out) {
  if (!out) return RJ_E_INVALID;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) return RJ_E_HIP;
  rj_handle h = new (std::nothrow) rj_handle_s();
  if (!h) return RJ_E_NOMEM;
  h->device = device_id;
  if (const char* e = getenv("RJ_LEAF_ORDER")) h->leaf_order = atoi(e) == 0 ? 0 : 1;
  if (const char* e = getenv("RJ_LSI_SEGMENTS")) h->lsi_segments = atoi(e) == 1 ? 1 : 2;  // (A/B runs)
  if (const char* e = getenv("RJ_WALK_POINTS")) h->walk_points = atoi(e) == 1 ? 1 : (atoi(e) == 4 ? 4 : 2);  // (A/B runs)
  if (const char* e = getenv("RJ_POINTS_SPLIT")) { const int v = atoi(e); h->points_split = v < -1 || v > 1 ? -1 : v; }  // (A/B runs, like the above)
  if (const char* e = getenv("RJ_LSI_ROOT_SKIP")) g_lsi_root_skip = atoi(e) != 0;                                          // (A/B runs)
  if (const char* e = getenv("RJ_LEAF_YSORT")) h->leaf_ysort = atoi(e) == 0 ? 0 : 1;                                    // (A/B runs)
  if (const char* e = getenv("RJ_PIP_COLUMNS")) { const int v = atoi(e); h->pip_columns = v < -1 || v > 1 ? -1 : v; }    // (A/B runs: the column index on / off whatever the map)
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0) h->cus = prop.multiProcessorCount;
  }
  if (hipSetDevice(device_id) != hipSuccess) { delete h; return RJ_E_HIP; }
  if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess) { delete h; return RJ_E_HIP; }
  h->stream = h->own_stream;
  if (hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking) != hipSuccess) { (void) hipStreamDestroy(h->own_stream); delete h; return RJ_E_HIP; }
  // (highest priority: its blocks are short and few -- they take the slots a draining walk frees before the next walk's do,
  //  instead of waiting behind a chip full of the next walk's resident blocks)
  int prio_least = 0, prio_greatest = 0;
  (void) hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
  if (hipStreamCreateWithPriority(&h->exact_stream, hipStreamNonBlocking, prio_greatest) != hipSuccess) {
    (void) hipStreamDestroy(h->own_stream); (void) hipStreamDestroy(h->aux_stream); delete h; return RJ_E_HIP;
  }
  bool ok = hipMalloc((void**) &h->d_counter, kCounterBytes) == hipSuccess &&
            hipMalloc((void**) &h->d_stats, 128) == hipSuccess &&
            hipMalloc((void**) &h->d_occ_part, 128) == hipSuccess &&
            hipHostMalloc((void**) &h->h_pinned, 256) == hipSuccess &&
            hipHostMalloc((void**) &h->h_fault, 64, hipHostMallocMapped) == hipSuccess &&
            hipHostGetDevicePointer((void**) &h->d_fault, h->h_fault, 0) == hipSuccess &&
            hipHostMalloc((void**) &h->h_rest, 64, hipHostMallocMapped) == hipSuccess &&
            hipHostGetDevicePointer((void**) &h->d_rest, h->h_rest, 0) == hipSuccess &&
            hipHostMalloc((void**) &h->h_est, 64, hipHostMallocMapped) == hipSuccess &&
            hipHostGetDevicePointer((void**) &h->d_est, h->h_est, 0) == hipSuccess;
  if (ok) {
    h->h_fault[0] = h->h_fault[1] = 0;
    h->h_rest[0] = h->h_rest[1] = ~0ull;  // (no finished two-pass query yet)
    h->h_rest[2] = ~0ull;                 // (... and no records produced yet: k_lsi_points' pair count)
    for (int k = 0; k < rj_handle_s::kCallerSets; k++) h->h_est[k] = 0;
    ok = hipMemset(h->d_counter, 0, kCounterBytes) == hipSuccess;
    for (size_t blk : {kSchedLsi, kSchedLsi + kSchedBlockWords, kSchedPipMain, kSchedPipMain + kSchedBlockWords, kSchedPipAux,
                       kSchedPipAux + kSchedBlockWords, kSchedWalkMain, kSchedWalkMain + kSchedBlockWords, kSchedWalkAux,
                       kSchedWalkAux + kSchedBlockWords})  // behind each block's counters: where its kernel reports a fault
      ok = ok && hipMemcpy((char*) (h->d_counter + blk) + kSchedFaultPtrWord * 4, &h->d_fault, sizeof(h->d_fault), hipMemcpyHostToDevice) == hipSuccess;
  }
  for (int t = 0; ok && t < kNumTimers; t++)
    ok = hipEventCreate(&h->ev[t][0]) == hipSuccess && hipEventCreate(&h->ev[t][1]) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&h->ev_count[0], hipEventDisableTiming) == hipSuccess &&
       hipEventCreateWithFlags(&h->ev_count[1], hipEventDisableTiming) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&h->ev_order, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&h->ev_walk_done, hipEventDisableTiming) == hipSuccess &&
       hipEventCreateWithFlags(&h->ev_exact_done[0], hipEventDisableTiming) == hipSuccess &&
       hipEventCreateWithFlags(&h->ev_exact_done[1], hipEventDisableTiming) == hipSuccess;
  // load every code object of the library now (a kernel's first launch loads its file's code object: milliseconds
  // that would otherwise land in the first upload, the first index build and the first query)
  ok = ok && warm_query_kernels(h->stream) == hipSuccess && warm_grid_kernels(h->stream) == hipSuccess &&
       warm_stitch_kernels(h->stream) == hipSuccess && warm_strip_kernels(h->stream) == hipSuccess &&
       hipStreamSynchronize(h->stream) == hipSuccess;
  if (!ok) { delete h; return RJ_E_HIP; }
  *out = h;
  return RJ_OK;
}

int rj_destroy(rj_handle h) {
  RJ_CHECK_H(h);
  (void) hipSetDevice(h->device);
  (void) hipStreamSynchronize(h->stream);
  (void) hipStreamSynchronize(h->aux_stream);
  (void) hipStreamSynchronize(h->exact_stream);
  for (int i = 0; i < 2; i++) { free_map(h->map[i]); free_bvh(h->bvh[i]); free_grid(h->grid[i]); }
  for (int k = 0; k < 2; k++) for (int i = 0; i < 2; i++) (void) hipFree(h->ordc[k][i].perm);
  (void) hipFree(h->d_counter); (void) hipFree(h->d_stats); (void) hipFree(h->d_occ_part); (void) hipHostFree(h->h_pinned); (void) hipHostFree(h->h_fault);
  (void) hipFree(h->slow_list);
  (void) hipHostFree(h->h_est);
  for (int k = 0; k < rj_handle_s::kCallerSets; k++) (void) hipFree(h->caller[k].perm);
  (void) hipHostFree(h->h_rest); for (int k = 0; k < 3; k++) { (void) hipFree(h->rest[k]); (void) hipFree(h->todo[k]); (void) hipFree(h->todo_mask[k]); }
  (void) hipFree(h->ord_kin); (void) hipFree(h->ord_kout); (void) hipFree(h->ord_vin); (void) hipFree(h->ord_vout); (void) hipFree(h->ord_temp);
  for (int t = 0; t < kNumTimers; t++) { (void) hipEventDestroy(h->ev[t][0]); (void) hipEventDestroy(h->ev[t][1]); }
  for (int k = 0; k < 2; k++) if (h->ev_count[k]) (void) hipEventDestroy(h->ev_count[k]);
  if (h->ev_order) (void) hipEventDestroy(h->ev_order);
  if (h->ev_walk_done) (void) hipEventDestroy(h->ev_walk_done);
  for (int k = 0; k < 2; k++) if (h->ev_exact_done[k]) (void) hipEventDestroy(h->ev_exact_done[k]);
  (void) hipFree(h->arena);
  (void) hipFree(h->strip_scratch);
  if (h->comm) (void) rj_comm_destroy(h);
  (void) hipStreamDestroy(h->own_stream);
  (void) hipStreamDestroy(h->aux_stream);
  (void) hipStreamDestroy(h->exact_stream);
  delete h;
  return RJ_OK;
}

// Every query kernel clears the scheduler counters of the NEXT launch of its kind, so launches of one
// kind must stay ordered on one stream: a switch drains the old stream (and the aux stream) first.
static int switch_stream(rj_handle h, hipStream_t s) {
  if (s == h->stream) return RJ_OK;
  if (int r = set_device(h)) return r;
  RJ_HIP(h, hipStreamSynchronize(h->stream));
  RJ_HIP(h, join_aux(h));
  h->lsi_shared = h->lsi_inflight = false;
  h->co_measure = false;
  h->stream = s;
  return RJ_OK;
}

int rj_set_stream(rj_handle h, void* s) {
  RJ_CHECK_H(h);
  return switch_stream(h, (hipStream_t) s);  // NULL is HIP's null (legacy default) stream, e.g. torch's default
}

int rj_sync(rj_handle h) {
  RJ_CHECK_H(h);
  if (int r = set_device(h)) return r;
  RJ_HIP(h, hipStreamSynchronize(h->stream));
  h->lsi_shared = h->lsi_inflight = false;
  RJ_HIP(h, join_aux(h));
  return check_fault(h);
}

const char* rj_last_error_string(rj_handle h) { return h ? h->err.c_str() : "null handle"; }

int rj_invalidate(rj_handle h) {
  RJ_CHECK_H(h);
  if (int r = set_device(h)) return r;
  RJ_HIP(h, hipStreamSynchronize(h->stream));  // (estimates in flight write the sets' words)
  RJ_HIP(h, join_aux(h));
  for (int k = 0; k < rj_handle_s::kCallerSets; k++) {
    h->caller[k].valid = h->caller[k].has_perm = false;
    h->h_est[k] = 0;
  }
  return RJ_OK;
}

int rj_get_option(rj_handle h, const char* name, int64_t* value) {
  RJ_CHECK_H(h);
  if (!name || !value) return fail(h, RJ_E_INVALID, "rj_get_option: null argument");
  if (!strcmp(name, "stats")) *value = h->stats_on;
  else if (!strcmp(name, "query_order")) *value = h->query_order;
  else if (!strcmp(name, "query_last_ordered")) *value = h->last_ordered ? 1 : 0;  // the last query ran through a Morton permutation of its queries
  else if (!strcmp(name, "pip_concurrent")) *value = h->pip_concurrent;
  else if (!strcmp(name, "pip_schedule")) *value = h->pip_concurrent == 2 ? h->co_choice : (h->pip_concurrent == 1 ? 1 : 0);
  else if (!strcmp(name, "pip_schedule_trials")) *value = h->co_trials;
  else if (!strncmp(name, "pip_schedule_us", 15) && name[15] >= '0' && name[15] <= '2' && !name[16])  // best span seen per schedule, microseconds (-1: not measured)
    *value = h->co_best[name[15] - '0'] < 1e29f ? (int64_t) (h->co_best[name[15] - '0'] * 1000.0f) : -1;
  else if (!strcmp(name, "pip_walk")) *value = h->pip_walk;
  else if (!strcmp(name, "timers")) *value = h->timers;
  else if (!strcmp(name, "pip_walk_points")) *value = h->walk_points;
  else if (!strcmp(name, "pip_exact_stream")) *value = h->exact_own_stream;
  else if (!strcmp(name, "lsi_segments")) *value = h->lsi_segments;
  else if (!strcmp(name, "lsi_last_segments")) *value = h->last_lsi_segments;
  else if (!strcmp(name, "pip_last_walk_points")) *value = h->last_walk_points;
  else if (!strcmp(name, "lsi_points_split")) *value = h->points_split;
  else if (!strcmp(name, "lsi_points_last_split")) *value = h->last_points_split;
  else if (!strcmp(name, "lsi_points_gcd_pairs")) {  // pairs the last two-kernel records launch left to the gcd leg (synchronises the main stream)
    if (int r = set_device(h)) return r;
    unsigned long long c = 0;
    RJ_HIP(h, hipStreamSynchronize(h->stream));
    RJ_HIP(h, hipMemcpy(&c, h->d_counter + kSlowCountWord + (1 - h->flip_slow), 8, hipMemcpyDeviceToHost));
    *value = h->last_points_split ? (int64_t) c : -1;
  }
  else if (!strcmp(name, "pip_last_passes")) *value = h->last_passes;  // how the last PIP query ran: 3 = walk + exact + k_pip, 1 = k_pip alone
  else if (!strcmp(name, "leaf_order")) *value = h->leaf_order;
  else if (!strcmp(name, "leaf_order_used0") || !strcmp(name, "leaf_order_used1")) *value = h->bvh[name[15] - '0'].leaf_order;  // what the index of map 0 / 1 was built with
  else if (!strcmp(name, "leaf_slots0") || !strcmp(name, "leaf_slots1")) *value = (int64_t) h->bvh[name[10] - '0'].n0p;  // slots of the index of map 0 / 1 (64 per leaf, padding included)
  else if (!strcmp(name, "leaf_runs0") || !strcmp(name, "leaf_runs1")) *value = h->map[name[9] - '0'].runs_cut ? (int64_t) h->map[name[9] - '0'].nruns : -1;  // polyline runs cut for map 0 / 1 (-1: none cut)
  else if (!strcmp(name, "stitch_rounds")) *value = h->stitch_stats[0];      // the last run cutting: pointer-jumping rounds that had work
  else if (!strcmp(name, "stitch_loop_ends")) *value = h->stitch_stats[1];   // ... chain ends on closed loops of paired chains
  else if (!strcmp(name, "skyline")) *value = h->skyline;
  else if (!strcmp(name, "pip_columns")) *value = h->pip_columns;
  else if (!strcmp(name, "leaf_ysort")) *value = h->leaf_ysort;
  else if (!strcmp(name, "occ_permille0") || !strcmp(name, "occ_permille1")) *value = h->bvh[name[12] - '0'].occ_permille;
  else if (!strcmp(name, "leaf_ysort_used0") || !strcmp(name, "leaf_ysort_used1")) *value = h->bvh[name[15] - '0'].ysort ? 1 : 0;
  else if (!strcmp(name, "pip_columns_used0") || !strcmp(name, "pip_columns_used1")) *value = h->bvh[name[16] - '0'].strips_built ? 1 : 0;
  else if (!strcmp(name, "pip_column_entries0") || !strcmp(name, "pip_column_entries1")) *value = (int64_t) h->bvh[name[18] - '0'].strip_entries;
  else if (!strcmp(name, "pip_column_shift0") || !strcmp(name, "pip_column_shift1")) *value = h->bvh[name[16] - '0'].strips_built ? h->bvh[name[16] - '0'].strip_shift : 0;
  else if (!strcmp(name, "pip_last_columns")) *value = h->last_columns;
  else if (!strcmp(name, "skyline_used0") || !strcmp(name, "skyline_used1")) *value = h->bvh[name[12] - '0'].use_sky ? 1 : 0;
  else if (!strcmp(name, "closed_chains0") || !strcmp(name, "closed_chains1")) *value = (int64_t) h->map[name[13] - '0'].closed_chains;
  else if (!strcmp(name, "pip_rest")) *value = (int64_t) h->h_rest[0];  // points the last finished two-pass query on the main stream left to k_pip (-1: none yet)
  else if (!strcmp(name, "pip_rest_aux")) *value = (int64_t) h->h_rest[1];
  else if (!strcmp(name, "comm_ranks")) {  // the ranks RCCL itself counts in the handle's communicator (0: rj_comm_init was not called)
    int c = 0;
    if (h->comm && ncclCommCount(h->comm, &c) != ncclSuccess) return fail(h, RJ_E_HIP, "ncclCommCount failed");
    *value = c;
  }
  else if (!strcmp(name, "lsi_share_blocks")) *value = h->lsi_share_blocks();
  else if (!strcmp(name, "pip_share_blocks")) *value = h->last_pip_share ? h->last_pip_share : h->pip_share_blocks();
  else return fail(h, RJ_E_INVALID, "unknown option '%s'", name);
  return RJ_OK;
}

// rj_get_plan: one JSON object (text) -- what the last query of each kind ran and why.  Host-side state only: nothing is
// synchronised or launched.
int rj_get_plan(rj_handle h, char* buf, size_t cap, size_t* need) {
  RJ_CHECK_H(h);
  const rj_handle_s::PlanRec& P = h->plan;
  char t[1024];
  std::string o = "{";
  auto add = [&](const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t, sizeof(t), fmt, ap);
    va_end(ap);
    o += t;
  };
  static const char* const kOrder[3] = {"as given", "through a cached Morton permutation", "through a Morton permutation sorted in this call (runs alone)"};
  static const char* const kSched[3] = {"turns", "shared", "full grids"};
  add("\"epoch\": %llu", (unsigned long long) P.epoch);
  add(", \"index\": [");
  for (int m = 0; m < 2; m++) {
    const BvhState& b = h->bvh[m];
    add("%s{\"map\": %d, \"built\": %s, \"levels\": %d, \"slots\": %llu, \"leaves\": \"%s\", \"skyline\": %s, \"columns\": %s, \"column_shift\": %d, "
        "\"steep_blocks_sorted_by_y\": %s, \"occupancy_permille\": %u, \"mean_chain_edges\": %.1f, \"slots_per_segment\": %.3f, \"columns_why\": \"",
        m ? ", " : "", m, b.built ? "true" : "false", b.top, (unsigned long long) b.n0p, b.leaf_order == 1 ? "polyline runs" : "Hilbert neighbours",
        b.use_sky ? "true" : "false", b.strips_built ? "true" : "false", b.strips_built ? b.strip_shift : 0, b.ysort ? "true" : "false", b.occ_permille,
        h->map[m].nc ? (double) h->map[m].ne / (double) h->map[m].nc : 0.0, b.n0 ? (double) b.n0p / (double) b.n0 : 0.0);
    for (const char* c = b.built ? b.columns_why : ""; *c; c++) { if (*c == '"') o += '\\'; o += *c; }  // (JSON string: quotes escaped)
    o += "\"}";
  }
  add("]");
  // the schedule of an LSI + PIP pair
  add(", \"schedule\": {\"pip_concurrent\": %d, \"trials\": %d, \"of\": %d", h->pip_concurrent, h->co_trials, kCoTrials);
  if (h->pip_concurrent == 2) {
    add(", \"choice\": \"%s\", \"settled\": %s, \"settled_in_epoch\": %llu, \"in_force\": %s, \"for_query_size\": %llu", h->co_choice >= 0 ? kSched[h->co_choice] : "undecided",
        h->co_choice >= 0 ? "true" : "false", (unsigned long long) P.sched_epoch, h->co_choice >= 0 && P.sched_epoch == P.epoch ? "true" : "false",
        (unsigned long long) h->co_n);
    add(", \"best_span_us\": {");
    for (int m = 0; m < 3; m++) add("%s\"%s\": %d", m ? ", " : "", kSched[m], h->co_best[m] < 1e29f ? (int) (h->co_best[m] * 1000.0f) : -1);
    add("}, \"turns_ratio_lsi_over_pip\": %.3f", (double) h->co_ratio);
  } else {
    add(", \"choice\": \"%s\", \"settled\": true", h->pip_concurrent == 1 ? "shared" : "turns");
  }
  add(", \"shared_grids\": {\"lsi_blocks\": %d, \"pip_blocks\": %d, \"fixed_by_debug_option\": %s}}", h->lsi_share_blocks(),
      h->last_pip_share ? h->last_pip_share : h->pip_share_blocks(), h->lsi_share_set || h->pip_share_set ? "true" : "false");
  // the last LSI query
  if (P.lsi.ran) {
    add(", \"lsi\": {\"epoch\": %llu, \"current\": %s, \"segments\": %llu, \"kernel\": \"%s\", \"blocks\": %d, \"segments_per_lane\": %d, \"order\": \"%s\", "
        "\"paired_with_pip\": %s, \"schedule\": \"%s\", \"on_shared_grid\": %s}",
        (unsigned long long) P.lsi.epoch, P.lsi.epoch == P.epoch ? "true" : "false", (unsigned long long) P.lsi.n, P.lsi.k.kernel, P.lsi.k.grid, P.lsi.k.per_lane,
        kOrder[P.lsi.order], P.lsi.paired ? "true" : "false", P.lsi.paired ? kSched[P.lsi.co_mode] : "alone (a synchronous or unpaired query: full grid)",
        P.lsi.shared ? "true" : "false");
  } else {
    add(", \"lsi\": null");
  }
  if (P.rec.ran) {
    add(", \"records\": {\"epoch\": %llu, \"kernels\": \"%s\", \"count\": \"%s\", \"pairs_expected\": %lld, \"rule\": \"%s\"}", (unsigned long long) P.rec.epoch,
        P.rec.two_kernels ? "k_lsi_points + k_lsi_points_gcd over what it declines" : "k_lsi_points_gcd alone",
        P.rec.count_on_device ? "read on the device" : "given by the caller", P.rec.seen == ~0ull ? -1ll : (long long) P.rec.seen,
        h->points_split >= 0 ? "fixed by the lsi_points_split option" : "two kernels from 393216 pairs on (the count of the last query; unknown yet: two)");
  } else {
    add(", \"records\": null");
  }
  if (P.pip.ran) {
    add(", \"pip\": {\"epoch\": %llu, \"current\": %s, \"points\": %llu, \"points_from\": \"%s\", \"stream\": \"%s\", \"on_shared_grid\": %s, \"order\": \"%s\", "
        "\"passes\": %d, \"first_pass\": {\"kernel\": \"%s\", \"blocks\": %d, \"points_per_lane\": %d}",
        (unsigned long long) P.pip.epoch, P.pip.epoch == P.epoch ? "true" : "false", (unsigned long long) P.pip.n, P.pip.caller ? "a caller-owned array" : "the query map's vertices",
        P.pip.aux ? "second (beside the LSI query)" : "main", P.pip.shared ? "true" : "false", kOrder[P.pip.order], P.pip.passes, P.pip.first.kernel, P.pip.first.grid,
        P.pip.first.per_lane);
    if (P.pip.passes == 3)
      add(", \"second_pass\": {\"kernel\": \"k_pip_exact\", \"blocks\": %d, \"locate_blocks\": %d}, \"left_over_by_last_query_of_this_size\": %lld", P.pip.exact_blocks,
          P.pip.locate_blocks, P.pip.rest_hint == ~0ull ? -1ll : (long long) P.pip.rest_hint);
    add(", \"why\": \"");
    for (const char* c = P.pip.why; *c; c++) { if (*c == '"' || *c == '\\') o += '\\'; o += *c; }
    add("\"}");
  } else {
    add(", \"pip\": null");
  }
  o += "}";
  if (need) *need = o.size();
  if (buf && cap) {
    const size_t k = o.size() < cap - 1 ? o.size() : cap - 1;
    memcpy(buf, o.data(), k);
    buf[k] = 0;
  }
  return RJ_OK;
}

int rj_set_option(rj_handle h, const char* name, int64_t value) {
  RJ_CHECK_H(h);
  if (!name) return fail(h, RJ_E_INVALID, "null option name");
  if (!strcmp(name, "stats")) { h->stats_on = value != 0; return RJ_OK; }
  if (!strcmp(name, "own_stream")) return switch_stream(h, h->own_stream);
  if (!strcmp(name, "pip_concurrent")) {
    if (join_aux(h) != hipSuccess) return fail(h, RJ_E_HIP, "pip_concurrent: stream sync failed");
    if (value < 0 || value > 2) return fail(h, RJ_E_INVALID, "pip_concurrent: 0 never, 1 LSI and PIP queries come in pairs and share the chip, 2 the same if it measures faster");
    h->pip_concurrent = (int) value;
    co_reset(h);
    return RJ_OK;
  }
  if (!strcmp(name, "leaf_order")) {
    if (value < 0 || value > 1) return fail(h, RJ_E_INVALID, "leaf_order: 0 Hilbert neighbours, 1 chain runs");
    h->leaf_order = (int) value;
    return RJ_OK;
  }
  if (!strcmp(name, "skyline")) {
    if (value < -1 || value > 1) return fail(h, RJ_E_INVALID, "skyline: -1 auto (maps of isolated rings), 0 never, 1 always");
    h->skyline = (int) value;
    return RJ_OK;
  }
  if (!strcmp(name, "leaf_ysort")) {
    if (value < 0 || value > 1) return fail(h, RJ_E_INVALID, "leaf_ysort: 1 blocks taller than wide get a second order by y (LSI), 0 x order only");
    h->leaf_ysort = (int) value;
    return RJ_OK;
  }
  if (!strcmp(name, "pip_columns")) {
    if (value < -1 || value > 1) return fail(h, RJ_E_INVALID, "pip_columns: -1 auto (maps of closed rings or of short chains), 0 never, 1 always");
    h->pip_columns = (int) value;
    return RJ_OK;
  }
  if (!strcmp(name, "lsi_segments")) {
    if (value != 1 && value != 2) return fail(h, RJ_E_INVALID, "lsi_segments: 1 or 2");
    h->lsi_segments = (int) value;
    return RJ_OK;
  }
  if (!strcmp(name, "pip_walk_points")) {
    if (value != 1 && value != 2 && value != 4) return fail(h, RJ_E_INVALID, "pip_walk_points: 1, 2 or 4");
    h->walk_points = (int) value;
    return RJ_OK;
  }
  if (!strcmp(name, "pip_exact_stream")) {
    if (value < 0 || value > 1) return fail(h, RJ_E_INVALID, "pip_exact_stream: 0 or 1");
    if ((int) value == h->exact_own_stream) return RJ_OK;
    // the two modes keep the aux stream's rest counts in different words: drain, then start both from zeroed words
    if (int r = set_device(h)) return r;
    RJ_HIP(h, hipStreamSynchronize(h->stream));
    h->aux_pending = true;
    RJ_HIP(h, join_aux(h));
    RJ_HIP(h, hipStreamSynchronize(h->exact_stream));
    for (size_t wd : kExactRestWord) RJ_HIP(h, hipMemsetAsync(h->d_counter + wd, 0, 8, h->stream));
    RJ_HIP(h, hipStreamSynchronize(h->stream));
    h->exact_rot = 0; h->exact_buf = 0; h->exact_last = -1; h->exact_out_closest = h->exact_out_face = nullptr;
    h->exact_recorded[0] = h->exact_recorded[1] = false;
    h->exact_own_stream = (int) value;
    return RJ_OK;
  }
  if (!strcmp(name, "timers")) {
    if (value < 0 || value > 1) return fail(h, RJ_E_INVALID, "timers: 0 or 1");
    h->timers = (int) value;
    return RJ_OK;
  }
  if (!strcmp(name, "lsi_points_split")) {
    if (value < -1 || value > 1) return fail(h, RJ_E_INVALID, "lsi_points_split: -1 by the last count, 0 never, 1 always");
    h->points_split = (int) value;
    return RJ_OK;
  }
  if (!strcmp(name, "pip_walk")) {
    if (value < 0 || value > 2) return fail(h, RJ_E_INVALID, "pip_walk: 0 k_pip alone, 1 auto, 2 always two passes");
    h->pip_walk = (int) value;
    return RJ_OK;
  }
  if (!strcmp(name, "query_order")) {
    if (value < 0 || value > 2) return fail(h, RJ_E_INVALID, "query_order: 0 never, 1 auto, 2 always");
    h->query_order = (int) value;
    return RJ_OK;
  }
  return fail(h, RJ_E_INVALID, "unknown option '%s'", name);
}

// Experiment knobs (tools/, tests of the fault path): not part of what a host of the library needs, never a
// correctness input, no promise that they survive a round.
int rj_set_debug_option(rj_handle h, const char* name, int64_t value) {
  RJ_CHECK_H(h);
  if (!name) return fail(h, RJ_E_INVALID, "null option name");
  struct { const char* name; int* var; int64_t lo, hi; } knobs[] = {
      {"chunk_groups", &h->chunk_groups, 0, 4096},        // consecutive groups handed to a wave at a time (0: per kernel, 8 / 6)
      {"max_blocks", &h->max_blocks, 1, 1 << 20},         // cap on the persistent grids
      {"lsi_share_blocks", &h->lsi_share_set, 0, 1 << 20},  // fixed grids of the shared schedule (0: derived)
      {"pip_share_blocks", &h->pip_share_set, 0, 1 << 20},
      {"stack_cap", &h->debug_stack_cap, 1, 1 << 30},     // instrumented kernels: fewer traversal-stack entries (fault path)
      {"walk_stack", &h->debug_walk_stack, 0, 1 << 30},   // k_pip_walk*: fewer stack entries (groups that need more leave the walk)
      {"strip_shift", &h->debug_strip_shift, 0, 20},      // the column index on strips of 2^this quanta (0: chosen by the map; 15..20)
      {"lazy_columns_min", &h->debug_lazy_columns_min, 0, 1 << 30},  // points from which an incoherent PIP query set makes the base map's column index (0: 2^22)
      {"query_key_strips", &h->debug_query_key_strips, 0, 1},  // "query_order" 2 over a column index: sort the points strip-major instead of by Morton key
      {"run_cap", &h->debug_run_cap, 0, 64},              // edges per polyline run of the next first build of a map (0: 64)
      {"pack_solo", &h->debug_pack_solo, 0, 64},          // a run longer than this never shares its leaf (0: 48)
      {"pack_spread", &h->debug_pack_spread, 0, 1000000}, // a shared leaf may be this many times as large as its runs (0: 8)
  };
  for (auto& k : knobs)
    if (!strcmp(name, k.name)) {
      if (value < k.lo || value > k.hi) return fail(h, RJ_E_INVALID, "%s: %lld..%lld", name, (long long) k.lo, (long long) k.hi);
      if (!strcmp(name, "run_cap") && value == 1) return fail(h, RJ_E_INVALID, "run_cap: 0 or 2..64");
      if (!strcmp(name, "strip_shift") && value != 0 && value < 15) return fail(h, RJ_E_INVALID, "strip_shift: 0 or 15..20 (the sort key holds 16 bits of strip)");
      *k.var = (int) value;
      return RJ_OK;
    }
  if (!strcmp(name, "group_lanes")) {  // queries per wave (0: 64 unless the query set is small)
    if (value != 0 && value != 4 && value != 8 && value != 16 && value != 32 && value != 64)
      return fail(h, RJ_E_INVALID, "group_lanes: 0 (auto), 4, 8, 16, 32 or 64");
    h->group_lanes = (int) value;
    return RJ_OK;
  }
  return fail(h, RJ_E_INVALID, "unknown debug option '%s'", name);
}

int rj_get_debug_option(rj_handle h, const char* name, int64_t* value) {
  RJ_CHECK_H(h);
  if (!name || !value) return fail(h, RJ_E_INVALID, "rj_get_debug_option: null argument");
  if (!strcmp(name, "chunk_groups")) *value = h->chunk_groups;
  else if (!strcmp(name, "group_lanes")) *value = h->group_lanes;
  else if (!strcmp(name, "max_blocks")) *value = h->max_blocks;
  else if (!strcmp(name, "lsi_share_blocks")) *value = h->lsi_share_set;
  else if (!strcmp(name, "pip_share_blocks")) *value = h->pip_share_set;
  else if (!strcmp(name, "stack_cap")) *value = h->debug_stack_cap;
  else if (!strcmp(name, "walk_stack")) *value = h->debug_walk_stack;
  else if (!strcmp(name, "strip_shift")) *value = h->debug_strip_shift;
  else if (!strcmp(name, "run_cap")) *value = h->debug_run_cap;
  else if (!strcmp(name, "pack_solo")) *value = h->debug_pack_solo;
  else if (!strcmp(name, "pack_spread")) *value = h->debug_pack_spread;
  else return fail(h, RJ_E_INVALID, "unknown debug option '%s'", name);
  return RJ_OK;
}

int rj_upload_map(rj_handle h, int map_id, const int64_t* xy, uint64_t np, const uint32_t* row_index,
                  const int64_t* left, const int64_t* right, uint64_t nc) {
  RJ_CHECK_H(h);
  if (map_id < 0 || map_id > 1) return fail(h, RJ_E_INVALID, "map_id must be 0 or 1");
  RJ_HIP(h, join_aux(h));
  co_reset(h);
  h->h_rest[0] = h->h_rest[1] = ~0ull;  // (what the last walk over another map left to k_pip says nothing about this one)
  h->walk_n[0] = h->walk_n[1] = 0;
  if ((np && !xy) || (nc && (!row_index || !left || !right))) return fail(h, RJ_E_INVALID, "null input array");
  if (np >= (1ull << 32) || nc > np) return fail(h, RJ_E_INVALID, "index_t is 32-bit: np < 2^32, nc <= np");
  if (nc && (row_index[0] != 0 || row_index[nc] != np)) return fail(h, RJ_E_INVALID, "row_index must start at 0 and end at np");
  for (uint64_t c = 0; c < nc; c++)
    if (row_index[c + 1] < row_index[c] + 2)
      return fail(h, RJ_E_INVALID, "chain %llu has fewer than 2 points (planar_graph.h:71)", (unsigned long long) c);
  if (nc == 0 && np != 0) return fail(h, RJ_E_INVALID, "points without chains");
  for (uint64_t i = 0; i < 2 * np; i++)
    if (xy[i] < -((int64_t) 1 << 46) || xy[i] >= ((int64_t) 1 << 46))
      return fail(h, RJ_E_INVALID, "coordinate %llu outside the scaled range [-2^46, 2^46)", (unsigned long long) i);
  if (int r = set_device(h)) return r;
  MapState& m = h->map[map_id];
  free_map(m);
  free_bvh(h->bvh[map_id]);
  free_grid(h->grid[map_id]);
  h->coh[0][map_id].valid = h->coh[1][map_id].valid = false;
  h->ordc[0][map_id].valid = h->ordc[1][map_id].valid = false;
  m.np = np; m.nc = nc; m.ne = np - nc;
  std::vector<uint32_t> eb(nc + 1), l32(nc), r32(nc);
  for (uint64_t c = 0; c <= nc; c++) eb[c] = nc ? (uint32_t) (row_index[c] - c) : 0;
  for (uint64_t c = 0; c < nc; c++) { l32[c] = (uint32_t) left[c]; r32[c] = (uint32_t) right[c]; }  // map.h:45
  int rc = RJ_OK;
  if (!rc) rc = dev_alloc(h, &m.pts, 2 * np + 2);
  if (!rc) rc = dev_alloc(h, &m.seg, m.ne);
  if (!rc) rc = dev_alloc(h, &m.edge_chain, m.ne);
  if (!rc) rc = dev_alloc(h, &m.ccode, m.ne);
  if (!rc) rc = dev_alloc(h, &m.left, nc);
  if (!rc) rc = dev_alloc(h, &m.right, nc);
  if (!rc) rc = dev_alloc(h, &m.edge_begin, nc + 1);
  hipError_t e = hipSuccess;
  if (!rc) {
    if (np) e = hipMemcpyAsync(m.pts, xy, 16 * np, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess && nc) e = hipMemcpyAsync(m.edge_begin, eb.data(), 4 * (nc + 1), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess && nc) e = hipMemcpyAsync(m.left, l32.data(), 4 * nc, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess && nc) e = hipMemcpyAsync(m.right, r32.data(), 4 * nc, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = launch_build_segs(h->stream, m.pts, m.edge_begin, (uint32_t) nc, m.ne, m.seg, m.edge_chain, m.ccode);
    // whatever was enqueued reads the host vectors: drain the stream before they go away
    const hipError_t es = hipStreamSynchronize(h->stream);
    if (e == hipSuccess) e = es;
  }
  if (rc || e != hipSuccess) free_map(m);  // no half-uploaded map
  if (rc) return rc;
  RJ_HIP(h, e);
  m.present = true;
  return RJ_OK;
}

int rj_scale_points(const double bb[4], const double* xy, uint64_t n, int64_t* out_xy, int fused) {
  if (!bb || (n && (!xy || !out_xy))) return RJ_E_INVALID;
  // Scaling<double, int64_t, 17>(bb), src/map/scaling.h:43-72 (this file is built with -ffp-contract=off)
  const int64_t imax = INT64_MAX >> 17, imin = INT64_MIN >> 17;
  const double range = (double) (imax - imin);
  const double max_x = bb[2] + 1, min_x = bb[0] - 1, max_y = bb[3] + 1, min_y = bb[1] - 1;  // SCALING_BOUNDING_BOX_MARGIN
  const double rx = range / (max_x - min_x), ry = range / (max_y - min_y);
  const double dx = 0.5 * ((imax + imin) - (max_x + min_x) * rx), dy = 0.5 * ((imax + imin) - (max_y + min_y) * ry);
  for (uint64_t i = 0; i < n; i++) {
    const double x = xy[2 * i], y = xy[2 * i + 1];
    double sx, sy;
    if (fused) {
      sx = std::fma(x, rx, dx);
      sy = std::fma(y, ry, dy);
    } else {
      const double tx = x * rx, ty = y * ry;
      sx = tx + dx;
      sy = ty + dy;
    }
    out_xy[2 * i] = (int64_t) sx;
    out_xy[2 * i + 1] = (int64_t) sy;
  }
  return RJ_OK;
}

int rj_map_num_edges(rj_handle h, int map_id, uint64_t* ne) {
  RJ_CHECK_H(h);
  if (map_id < 0 || map_id > 1 || !ne || !h->map[map_id].present) return fail(h, RJ_E_INVALID, "map not uploaded");
  *ne = h->map[map_id].ne;
  return RJ_OK;
}

int rj_map_num_points(rj_handle h, int map_id, uint64_t* np) {
  RJ_CHECK_H(h);
  if (map_id < 0 || map_id > 1 || !np || !h->map[map_id].present) return fail(h, RJ_E_INVALID, "map not uploaded");
  *np = h->map[map_id].np;
  return RJ_OK;
}

int rj_map_points_dev(rj_handle h, int map_id, const int64_t** pts_dev) {
  RJ_CHECK_H(h);
  if (map_id < 0 || map_id > 1 || !pts_dev || !h->map[map_id].present) return fail(h, RJ_E_INVALID, "map not uploaded");
  *pts_dev = h->map[map_id].pts;
  return RJ_OK;
}

int rj_map_runs(rj_handle h, int map_id, uint32_t* piece_begin, uint32_t* piece_len, uint32_t* run_first, uint64_t* nruns, uint64_t* npieces) {
  RJ_CHECK_H(h);
  if (map_id < 0 || map_id > 1 || !h->map[map_id].present) return fail(h, RJ_E_INVALID, "rj_map_runs: map not uploaded");
  const MapState& m = h->map[map_id];
  if (!m.runs_cut) return fail(h, RJ_E_INVALID, "rj_map_runs: no runs cut for map %d (rj_build_lbvh with \"leaf_order\" 1 first)", map_id);
  if (nruns) *nruns = m.nruns;
  if (npieces) *npieces = m.npieces;
  if (int r = set_device(h)) return r;
  if (piece_begin) RJ_HIP(h, hipMemcpy(piece_begin, m.piece_begin, 4 * m.npieces, hipMemcpyDeviceToHost));
  if (piece_len) RJ_HIP(h, hipMemcpy(piece_len, m.piece_len, 4 * m.npieces, hipMemcpyDeviceToHost));
  if (run_first) RJ_HIP(h, hipMemcpy(run_first, m.run_first, 4 * (m.nruns + 1), hipMemcpyDeviceToHost));
  return RJ_OK;
}

// Sort scratch ((u64 key, u32 value) in/out + rocPRIM temp) shared by the index build and the
// query re-ordering; it only grows and stays with the handle, so a rebuild or a repeated query
// does not pay hipMalloc again (24 B per item + temp; 2.4 GB for the 67 M-segment map).
static int ensure_sort_scratch(rj_handle h, uint64_t n) {
  if (n > h->ord_cap) {
    (void) hipFree(h->ord_kin); (void) hipFree(h->ord_kout); (void) hipFree(h->ord_vin); (void) hipFree(h->ord_vout);
    h->ord_kin = h->ord_kout = nullptr; h->ord_vin = h->ord_vout = nullptr; h->ord_cap = 0;
    if (int r = dev_alloc(h, &h->ord_kin, n)) return r;
    if (int r = dev_alloc(h, &h->ord_kout, n)) return r;
    if (int r = dev_alloc(h, &h->ord_vin, n)) return r;
    if (int r = dev_alloc(h, &h->ord_vout, n)) return r;
    h->ord_cap = n;
  }
  size_t need = 0;
  RJ_HIP(h, sort_morton_pairs(h->stream, nullptr, need, h->ord_kin, h->ord_kout, h->ord_vin, h->ord_vout, n));
  if (need > h->ord_temp_bytes) {
    (void) hipFree(h->ord_temp);
    h->ord_temp = nullptr; h->ord_temp_bytes = 0;
    RJ_HIP(h, hipMalloc(&h->ord_temp, need));
    h->ord_temp_bytes = need;
  }
  return RJ_OK;
}

// the column index of an index whose leaves are built (b.box0, b.seid): rj_strip.hip.  Temporaries come out of one
// grow-only scratch block of the handle and the index's own arrays are kept while they are large enough: a rebuild
// allocates nothing (the first version paid six hipMalloc / hipFree of hundreds of MB per build: 50-120 ms).
static int build_strips(rj_handle h, BvhState& b, bool with_sky) {
  b.strips_built = false;
  auto up = [](size_t v) { return (v + 255) & ~(size_t) 255; };
  // (an allocation that fails is not the build's failure unless the caller FORCED the column index: the tree is built
  //  and serves the map alone -- RJ_E_NOMEM only under "pip_columns" 1)
  const bool forced = h->pip_columns == 1;
  auto scratch = [&](size_t bytes) -> int {
    if (bytes <= h->strip_scratch_bytes) return RJ_OK;
    (void) hipFree(h->strip_scratch);
    h->strip_scratch = nullptr; h->strip_scratch_bytes = 0;
    const hipError_t e = hipMalloc((void**) &h->strip_scratch, bytes);
    if (e != hipSuccess) {
      (void) hipGetLastError();
      return forced ? fail(h, RJ_E_NOMEM, "rj_build_lbvh: %zu bytes of scratch for the column index: %s", bytes, hipGetErrorString(e)) : -1;
    }
    h->strip_scratch_bytes = bytes;
    return RJ_OK;
  };
  // the strip width: widest power of two below 2.3 x the mean x-extent of a segment (rj_device.h, DeviceStrips)
  size_t scan_bytes = 0;
  RJ_HIP(h, launch_strip_count(h->stream, nullptr, nullptr, b.n0p, 0, nullptr, nullptr, nullptr, scan_bytes, nullptr));
  const size_t cnt_bytes = up(4 * (b.n0p + 1));
  // pass 1 needs: cnt, offs, flag, scan temp; pass 2 adds key_tmp, slot_tmp, sort temp (sized once the total is known:
  // an upper bound first -- twice the slots covers every map whose segments are not wider than a strip or two)
  // (the width's two sums land in d_counter[2],[3] -- the grid build's words, same stream -- so that the scratch block
  //  can be sized once the strip count is known)
  RJ_HIP(h, launch_strip_width(h->stream, b.box0, b.seid, b.n0p, h->d_counter + 2));
  RJ_HIP(h, hipMemcpyAsync(h->h_pinned + 30, h->d_counter + 2, 16, hipMemcpyDeviceToHost, h->stream));
  RJ_HIP(h, hipStreamSynchronize(h->stream));
  if (!h->h_pinned[31]) return RJ_OK;  // (no real segment)
  const double mean_dx = (double) h->h_pinned[30] / (double) h->h_pinned[31];
  int shift = kStripShiftMin;
  while (shift < kStripShiftMax && (double) (2u << shift) <= 2.3 * mean_dx) shift++;
  if (h->debug_strip_shift) shift = h->debug_strip_shift;
  const uint32_t strips = strip_count(shift);
  {
    // everything both passes need, for an estimate of 2 entries per slot (the lake-shaped maps have 1.5): the counts then
    // stay where they are when the total arrives (a first build used to count and scan twice: 0.27 ms)
    // (... per SLOT, or 2.5 per real segment where the leaves are half empty -- the short-chain lattices' 2.12 slots per
    //  segment: 2 x 51.8 M slots for 24.4 M segments that make 44 M entries)
    const size_t est = 2 * (size_t) b.n0p < 5 * (size_t) b.n0 / 2 + 64 ? 2 * (size_t) b.n0p : 5 * (size_t) b.n0 / 2 + 64;
    size_t est_sort = 0;
    RJ_HIP(h, launch_strip_fill(h->stream, nullptr, nullptr, nullptr, nullptr, nullptr, b.n0p, shift, est, nullptr, nullptr, nullptr, nullptr, nullptr,
                                nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, est_sort));
    const size_t both = 2 * cnt_bytes + 256 + 4 * up(4 * est) + up(4 * (size_t) strips) + up(est_sort);  // (keys are 32-bit since round 5)
    const size_t first = 2 * cnt_bytes + 256 + up(scan_bytes);
    if (int r = scratch(both > first ? both : first)) {
      if (r > 0 || (r = scratch(first))) return r < 0 ? RJ_OK : r;  // (the estimate did not fit: count first, size then -- or leave it to the tree)
    }
  }
  uint32_t* cnt = (uint32_t*) h->strip_scratch;
  uint32_t* offs = (uint32_t*) (h->strip_scratch + cnt_bytes);
  uint32_t* flag = (uint32_t*) (h->strip_scratch + 2 * cnt_bytes);
  void* temp = h->strip_scratch + 2 * cnt_bytes + 256;
  RJ_HIP(h, hipMemsetAsync(flag, 0, 4, h->stream));
  size_t tb = scan_bytes;
  RJ_HIP(h, launch_strip_count(h->stream, b.box0, b.seid, b.n0p, shift, cnt, offs, temp, tb, flag));
  RJ_HIP(h, hipMemcpyAsync(h->h_pinned + 30, offs + b.n0p, 4, hipMemcpyDeviceToHost, h->stream));
  RJ_HIP(h, hipMemcpyAsync((uint32_t*) (h->h_pinned + 30) + 1, flag, 4, hipMemcpyDeviceToHost, h->stream));
  RJ_HIP(h, hipStreamSynchronize(h->stream));
  const uint32_t total = (uint32_t) h->h_pinned[30], bad = (uint32_t) (h->h_pinned[30] >> 32);
  if (bad || total == 0) return RJ_OK;  // (a segment spanning more than kStripMaxSpan strips: the tree alone serves this map)
  size_t sort_bytes = 0;
  RJ_HIP(h, launch_strip_fill(h->stream, nullptr, nullptr, nullptr, nullptr, nullptr, b.n0p, shift, total, nullptr, nullptr, nullptr, nullptr, nullptr,
                              nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, sort_bytes));
  const size_t need = 2 * cnt_bytes + 256 + 4 * up(4 * (size_t) total) + up(4 * (size_t) strips) + up(sort_bytes);
  if (need > h->strip_scratch_bytes) {
    // (the counts live in the scratch block that is about to move: count again into the new one -- first build of a
    //  larger map only)
    if (int r = scratch(need + need / 8)) {
      if (r > 0 || (r = scratch(need))) return r < 0 ? RJ_OK : r;
    }
    cnt = (uint32_t*) h->strip_scratch; offs = (uint32_t*) (h->strip_scratch + cnt_bytes); flag = (uint32_t*) (h->strip_scratch + 2 * cnt_bytes);
    temp = h->strip_scratch + 2 * cnt_bytes + 256;
    tb = scan_bytes;
    RJ_HIP(h, launch_strip_count(h->stream, b.box0, b.seid, b.n0p, shift, cnt, offs, temp, tb, flag));
  }
  // (32-bit keys since round 5 -- strip above the band of y0, rj_strip.hip -- in the room the 64-bit ones had)
  uint32_t* key_tmp = (uint32_t*) (h->strip_scratch + 2 * cnt_bytes + 256);
  uint32_t* key = (uint32_t*) ((char*) key_tmp + up(4 * (size_t) total));
  uint32_t* slot_tmp = (uint32_t*) ((char*) key + up(4 * (size_t) total));
  uint32_t* slot_sorted = (uint32_t*) ((char*) slot_tmp + up(4 * (size_t) total));
  uint32_t* tall_tmp = (uint32_t*) ((char*) slot_sorted + up(4 * (size_t) total));
  temp = (char*) tall_tmp + up(4 * (size_t) strips);
  if (b.strip_cap < total) {
    (void) hipFree(b.strip_info); (void) hipFree(b.strip_box);
    b.strip_info = nullptr; b.strip_box = nullptr; b.strip_cap = 0;
    const uint64_t cap = (uint64_t) total + total / 16;
    int r = dev_alloc(h, &b.strip_box, cap);
    if (!r) r = dev_alloc(h, &b.strip_info, cap);
    if (r) {  // (no room for the lists: the caller's error only if the caller forced the index -- else the tree serves the map)
      (void) hipFree(b.strip_box); (void) hipFree(b.strip_info);
      b.strip_box = nullptr; b.strip_info = nullptr;
      return forced ? r : RJ_OK;
    }
    b.strip_cap = cap;
  }
  if (!b.strip_tall || b.strip_tab_shift > shift) {  // (narrower strips than the tables were made for)
    (void) hipFree(b.strip_tall); (void) hipFree(b.strip_ytab);
    b.strip_tall = nullptr; b.strip_ytab = nullptr; b.strip_tab_shift = 0;
    int r = dev_alloc(h, &b.strip_tall, (uint64_t) strips);
    if (!r) r = dev_alloc(h, &b.strip_ytab, ((uint64_t) strips << kStripYBits) + 1);
    if (r) {
      (void) hipFree(b.strip_tall); (void) hipFree(b.strip_ytab);
      b.strip_tall = nullptr; b.strip_ytab = nullptr;
      return forced ? r : RJ_OK;
    }
    b.strip_tab_shift = shift;
  }
  tb = sort_bytes;
  // (the skyline, where wanted, comes out of the same pass: every entry's box is in hand there)
  if (with_sky) RJ_HIP(h, hipMemsetAsync(b.sky, 0, ((size_t) kSkyBuckets + 1) * 4, h->stream));
  RJ_HIP(h, launch_strip_fill(h->stream, b.box0, b.seid, b.sface, cnt, offs, b.n0p, shift, total, key, slot_sorted, key_tmp, slot_tmp, tall_tmp,
                              b.strip_ytab, b.strip_box, b.strip_info, b.strip_tall, with_sky ? b.sky : nullptr, temp, tb));
  b.use_sky = with_sky;
  b.strip_shift = shift;
  b.strip_entries = total;
  b.strips_built = true;
  return RJ_OK;
}

int rj_build_lbvh(rj_handle h, int base_map_id) {
  RJ_CHECK_H(h);
  if (base_map_id < 0 || base_map_id > 1 || !h->map[base_map_id].present)
    return fail(h, RJ_E_INVALID, "rj_build_lbvh: map %d not uploaded", base_map_id);
  if (int r = set_device(h)) return r;
  RJ_HIP(h, join_aux(h));
  const MapState& m = h->map[base_map_id];
  BvhState& b = h->bvh[base_map_id];
  // RJ_T_BUILD spans everything this call does, the first time too: cutting the runs, allocating, sorting, the leaves
  // (the reference's "Build Index", run_query.cu, runs once per map: that is the build to quote)
  h->ev_valid[RJ_T_BUILD_RUNS] = false;
  tic(h, RJ_T_BUILD);
  // "leaf_order" 1: the leaves are polyline runs -- chains stitched through the junctions and cut into runs of <= 64
  // consecutive edges (rj_stitch.h; the reference's RT grouping, src/rt/primitive.h:120-260) -- cut ON THE DEVICE the
  // first time an index of this map wants them and kept (a later rebuild only sorts the runs).
  if (h->leaf_order == 1 && m.ne && m.nc < (1ull << 30) && !m.runs_cut) {
    MapState& mm = h->map[base_map_id];
    // How long a run may be.  A full leaf (64 edges) is right where chains are long: the strip is a piece of one smooth
    // line.  A polyline stitched from many SHORT chains wiggles through a junction every few edges, its strip is fat,
    // and where it runs steeply all of its edges overlap in x: the in-leaf scans test every slot and the PIP walk's
    // candidate lists overflow.  Measured again in round 4 on the WaterBodies lattice (10-edge chains), query alone:
    // runs of 32, one per leaf (2.12 slots per segment) k_lsi 1.11 ms, PIP 1.94; runs of 64 (1.12 slots) 1.27 / 2.39 and
    // the walk dropped for its overflowed lists in the step (3.64 ms against 2.7); runs of 32 sharing leaves in pairs
    // (1.46 slots) 1.31 / 2.37.  So: the cap follows the mean chain length, and a run that fills three quarters of its
    // cap keeps its leaf (k_pack_runs) -- twice the index there, the faster one.
    // Round 6: that holds where upward rays walk the tree.  A map of short chains now gets the column index for them
    // ("pip_columns"), its leaves serve LSI alone, and with the steep blocks in their second order ("leaf_ysort") the full
    // run wins: WaterBodies x BlockGroup, runs of 32 / 48 / 64 -- step 2.11 / 2.05 / 2.01 ms, k_lsi2 alone 1.03 / 1.02 / 1.01,
    // 2.12 / 1.45 / 1.12 slots per segment, first build 9.2 / 8.9 / 8.1 ms (profiles/r06_runcap_with_second_order.txt).
    uint32_t cap_edges = mm.nc && mm.ne / mm.nc < 16 && h->pip_columns == 0 ? 32 : 64;
    if (h->debug_run_cap) cap_edges = (uint32_t) h->debug_run_cap;
    uint64_t max_pieces = 0, max_runs = 0;
    stitch_output_bounds(mm.nc, mm.ne, cap_edges, &max_pieces, &max_runs);
    tic(h, RJ_T_BUILD_RUNS);
    int rc = dev_alloc(h, &mm.piece_begin, max_pieces);
    if (!rc) rc = dev_alloc(h, &mm.piece_len, max_pieces);
    if (!rc) rc = dev_alloc(h, &mm.run_first, max_runs + 1);
    hipError_t e = hipSuccess;
    uint64_t nruns_cut = 0, npieces_cut = 0;
    if (!rc) e = stitch_runs_device(h->stream, mm.pts, mm.edge_begin, mm.nc, mm.ne, cap_edges, mm.piece_begin, mm.piece_len, mm.run_first,
                                    &nruns_cut, &npieces_cut, h->stitch_stats, &h->arena, &h->arena_bytes);
    if (rc || e != hipSuccess) {  // no half-cut map: a retried build starts over
      (void) hipFree(mm.piece_begin); (void) hipFree(mm.piece_len); (void) hipFree(mm.run_first);
      mm.piece_begin = mm.piece_len = mm.run_first = nullptr;
      mm.nruns = mm.npieces = 0;
      if (rc) return rc;
      RJ_HIP(h, e);
    }
    toc(h, RJ_T_BUILD_RUNS);
    mm.nruns = nruns_cut;
    mm.npieces = npieces_cut;
    mm.run_cap = cap_edges;
    mm.closed_chains = h->stitch_stats[3];
    mm.runs_cut = true;
  }
  // Polyline-run leaves: the runs get a Hilbert key and are sorted FIRST -- consecutive short runs of that order then share
  // a leaf (k_pack_runs: a map of isolated rings, ten edges per chain, fills its leaves with several neighbouring rings
  // instead of one ring and 54 empty slots), and the number of leaves sizes everything allocated below.
  MortonKey *k_in = nullptr, *k_out = nullptr;
  uint32_t *v_in = nullptr, *v_out = nullptr;
  uint64_t nruns = (h->leaf_order == 1 && m.ne && m.runs_cut) ? m.nruns : 0;
  uint64_t nleaves_runs = 0;
  if (nruns) {
    MapState& mm = h->map[base_map_id];
    if (int r = ensure_sort_scratch(h, nruns)) return r;
    k_in = h->ord_kin; k_out = h->ord_kout; v_in = h->ord_vin; v_out = h->ord_vout;
    const uint64_t nchunks = pack_runs_chunks(nruns);
    if (!mm.run_len) {
      int rc = dev_alloc(h, &mm.run_len, nruns);
      if (!rc) rc = dev_alloc(h, &mm.leaf_first, nruns + 1);
      if (!rc) rc = dev_alloc(h, &mm.run_box, nruns);
      if (!rc) rc = dev_alloc(h, &mm.pack_tmp, 2 * (nchunks + 1));
      if (rc) return rc;
    }
    // a run longer than three quarters of the cap keeps its leaf to itself (a nearly full strip of one polyline gains
    // little from a tenant and its box would have to be computed); a shared leaf may be `spread` times as large as what it holds
    // (a map of short closed RINGS keeps round 4's rule -- rings of more than 24 edges alone -- whatever the run cap: with the
    //  cap's 48, the lake-shaped map packs to 1.23 slots per segment instead of 1.54 and k_lsi2 over it is 8 % slower, the
    //  gaussian polygons 12 %)
    const bool short_rings = mm.nc && mm.ne / mm.nc < 16 && 2 * mm.closed_chains >= mm.nc;
    const uint32_t solo_above = h->debug_pack_solo ? (uint32_t) h->debug_pack_solo : (short_rings ? 24u : (mm.run_cap ? mm.run_cap : 64u) * 3 / 4);
    // (measured on the ring-shaped pairs, PIP query alone: 4 / 8 / 16 -> lake-shaped base 40.2 / 41.6 / 41.8 ms, lakes x
    //  parks 18.7 / 19.7 / 24.2, gaussian polygons 2.46 / 2.25 / 2.18: 8)
    const uint32_t spread = h->debug_pack_spread ? (uint32_t) h->debug_pack_spread : 8;
    hipError_t e = hipSuccess;
    tic(h, RJ_T_BUILD_KEYS);
    e = launch_run_keys(h->stream, m.seg, m.piece_begin, m.piece_len, m.run_first, nruns, k_in, v_in, mm.run_len, mm.run_box, solo_above);
    toc(h, RJ_T_BUILD_KEYS);
    tic(h, RJ_T_BUILD_SORT);
    size_t tb = h->ord_temp_bytes;
    if (e == hipSuccess) e = sort_morton_pairs(h->stream, h->ord_temp, tb, k_in, k_out, v_in, v_out, nruns);
    if (e == hipSuccess) e = launch_pack_runs(h->stream, v_out, mm.run_len, mm.run_box, nruns, solo_above, spread, mm.pack_tmp, mm.pack_tmp + nchunks + 1, mm.leaf_first, h->d_rest + 3);
    toc(h, RJ_T_BUILD_SORT);
    RJ_HIP(h, e);
    if (!mm.packed_leaves || mm.packed_solo != solo_above + 1000 * spread) {
      // the first time (the packing is a function of the runs and their order: a rebuild finds the same leaves)
      RJ_HIP(h, hipStreamSynchronize(h->stream));
      mm.packed_leaves = h->h_rest[3];
      mm.packed_solo = solo_above + 1000 * spread;
    }
    nleaves_runs = mm.packed_leaves;
    // (leaves still mostly empty -- cannot happen with runs sharing leaves unless "debug_pack_solo" forbids it: Hilbert leaves)
    if (nleaves_runs * 64 > m.ne * 5 / 2) nruns = 0;
  }
  const uint64_t n0p_new = nruns ? nleaves_runs * 64 : pad64(m.ne ? m.ne : 1);
  if (n0p_new >= (1ull << 32)) return fail(h, RJ_E_INVALID, "rj_build_lbvh: %llu leaf slots do not fit 32-bit slot ids", (unsigned long long) n0p_new);
  const bool reuse = b.sseg && b.n0p == n0p_new;  // rebuild of a same-sized map: keep the buffers
  if (!reuse) free_bvh(b);
  b.built = false;
  b.leaf_order = nruns ? 1 : 0;
  b.n0 = m.ne;
  b.n0p = n0p_new;
  // level sizes: level l has ceil(n_{l-1}/64) nodes; top = first level with <= 64 nodes
  b.nlvl[0] = b.n0;
  b.alloc[0] = b.n0p;
  int top = 0;
  uint64_t n = b.n0p / 64;  // number of leaf blocks (>= 1)
  for (int l = 1; l < kMaxLevels; l++) {
    b.nlvl[l] = n;
    b.alloc[l] = pad64(n);
    top = l;
    if (n <= 64) break;
    n = (n + 63) / 64;
  }
  if (b.nlvl[top] > 64) return fail(h, RJ_E_INVALID, "too many segments for %d levels", kMaxLevels);
  b.top = top;
  if (!reuse) {
    // one block, carved (sizes first, then the pointers); every array starts on a 256-byte boundary
    size_t used = 0;
    auto take = [&](size_t bytes) { const size_t at = used; used = (used + bytes + 255) & ~(size_t) 255; return at; };
    const size_t o_sseg = take(sizeof(Seg) * b.n0p), o_seid = take(4 * b.n0p), o_sface = take(4 * b.n0p), o_box0 = take(sizeof(QBox) * b.n0p),
                 o_pmx1 = take(4 * b.n0p), o_xtab = take(sizeof(uint2) * b.n0p), o_ytab2 = take(sizeof(uint2) * b.n0p), o_occ = take(4 * ((size_t) kOccDim * kOccRowWords + 1)),
                 o_sky = take(4 * ((size_t) kSkyBuckets + 1));  // (the skyline's 1 MiB: allocated with the index, filled when wanted)
    size_t o_lvl[kMaxLevels] = {0};
    for (int l = 1; l <= top; l++) o_lvl[l] = take(sizeof(QBox) * (b.alloc[l] + b.alloc[l] / 2));  // 16 B box + 8 B order word per node
    if (used <= ((size_t) 5 << 29)) {  // (2.5 GiB)
      const hipError_t pe = hipMalloc((void**) &b.pool, used ? used : 1);
      if (pe != hipSuccess) {
        free_bvh(b);
        return fail(h, pe == hipErrorOutOfMemory ? RJ_E_NOMEM : RJ_E_HIP, "rj_build_lbvh: hipMalloc of %zu bytes for the index failed: %s", used, hipGetErrorString(pe));
      }
      b.sseg = (Seg*) (b.pool + o_sseg); b.seid = (uint32_t*) (b.pool + o_seid); b.sface = (int32_t*) (b.pool + o_sface);
      b.box0 = (QBox*) (b.pool + o_box0); b.pmx1 = (int32_t*) (b.pool + o_pmx1); b.xtab = (uint2*) (b.pool + o_xtab); b.ytab2 = (uint2*) (b.pool + o_ytab2);
      b.occ = (uint32_t*) (b.pool + o_occ); b.sky = (uint32_t*) (b.pool + o_sky);
      for (int l = 1; l <= top; l++) b.lvl[l] = (QBox*) (b.pool + o_lvl[l]);
    } else {
      int r = 0;
      if (!r) r = dev_alloc(h, &b.sseg, b.n0p);
      if (!r) r = dev_alloc(h, &b.seid, b.n0p);
      if (!r) r = dev_alloc(h, &b.sface, b.n0p);
      if (!r) r = dev_alloc(h, &b.box0, b.n0p);
      if (!r) r = dev_alloc(h, &b.pmx1, b.n0p);
      if (!r) r = dev_alloc(h, &b.xtab, b.n0p);
      if (!r) r = dev_alloc(h, &b.ytab2, b.n0p);
      if (!r) r = dev_alloc(h, &b.occ, (uint64_t) kOccDim * kOccRowWords + 1);
      if (!r) r = dev_alloc(h, &b.sky, (uint64_t) kSkyBuckets + 1);
      for (int l = 1; l <= top && !r; l++) r = dev_alloc(h, &b.lvl[l], b.alloc[l] + b.alloc[l] / 2);
      if (r) { free_bvh(b); return r; }
    }
  }
  // 1. Morton keys  2. radix sort (key, eid)  (Hilbert leaves; the runs were keyed and sorted above)
  // 3. leaves + occupancy + level 1 in one pass  4. upper levels
  if (!nruns && m.ne) {
    if (int r = ensure_sort_scratch(h, m.ne)) return r;
    k_in = h->ord_kin; k_out = h->ord_kout; v_in = h->ord_vin; v_out = h->ord_vout;
  }
  hipError_t e = hipSuccess;
  do {
    if (!nruns) {
      tic(h, RJ_T_BUILD_KEYS);
      if ((e = launch_morton(h->stream, m.seg, m.ne, k_in, v_in)) != hipSuccess) break;
      toc(h, RJ_T_BUILD_KEYS);
      tic(h, RJ_T_BUILD_SORT);
      if (m.ne) {
        size_t tb = h->ord_temp_bytes;
        if ((e = sort_morton_pairs(h->stream, h->ord_temp, tb, k_in, k_out, v_in, v_out, m.ne)) != hipSuccess) break;
      }
      toc(h, RJ_T_BUILD_SORT);
    }
    tic(h, RJ_T_BUILD_LEAVES);
    if ((e = hipMemsetAsync(b.occ, 0, ((size_t) kOccDim * kOccRowWords + 1) * 4, h->stream)) != hipSuccess) break;
    // The skyline (rj_device.h kSkyShift) is what proves a MISS of the upward ray without a traversal.  A planar subdivision
    // has an outer boundary over every x it covers -- a ray from inside never misses, the table would cost the build up
    // to a third of its time (one atomic per segment and bucket) and every query point a load for nothing; a map of
    // isolated rings leaves a third of a lattice's vertices with nothing above them (35 % measured on the lake-shaped
    // stand-in, 240 leaf blocks opened for each).  So: built where most chains are closed rings.
    // (filled below: by the column index's own pass where the map gets one, else by a pass over the leaves' boxes)
    b.use_sky = false;
    // "leaf_ysort": blocks taller than wide get a second order, by y, for the LSI kernels -- where the leaves are runs of
    // polylines.  Not on maps of closed rings: a block of packed rings is about as tall as wide, most of its edges lie over a
    // query's range on either axis (8 scan steps per opened block whichever the order, tools/lsi_stats_probe.py), and a step of
    // the y order costs one cross-lane read more: k_lsi2 + 3 % on the lake-shaped base map.
    b.ysort = h->leaf_ysort != 0 && !(m.runs_cut && m.nc && 2 * m.closed_chains >= m.nc);
    if ((e = launch_build_leaves(h->stream, m.seg, v_out, m.edge_chain, m.left, m.right, m.ne, nruns ? m.piece_begin : nullptr,
                                 m.piece_len, m.run_first, m.run_len, m.leaf_first, b.n0p / 64, b.alloc[1],
                                 b.sseg, b.seid, b.sface, b.box0, b.pmx1, b.xtab, b.lvl[1], b.occ, b.ysort ? b.ytab2 : nullptr)) != hipSuccess) break;
    toc(h, RJ_T_BUILD_LEAVES);
    tic(h, RJ_T_BUILD_LEVELS);
    const QBox* child = b.lvl[1];
    uint64_t child_alloc = b.alloc[1];
    for (int l = 2; l <= top; l++) {
      if ((e = launch_reduce_level(h->stream, child, child_alloc, b.lvl[l], b.alloc[l])) != hipSuccess) break;
      child = b.lvl[l];
      child_alloc = b.alloc[l];
    }
    if (e != hipSuccess) break;
    for (int l = 1; l <= top; l++)  // front-to-back sibling order of every level (k_pip)
      if ((e = launch_sibling_order(h->stream, b.lvl[l], b.alloc[l], (uint64_t*) (b.lvl[l] + b.alloc[l]))) != hipSuccess) break;
    if (e != hipSuccess) break;
    toc(h, RJ_T_BUILD_LEVELS);
  } while (0);
  RJ_HIP(h, e);
  // The column index (rj_device.h DeviceStrips), where the tree is at its worst for upward rays: maps of isolated rings
  // (the criterion of the skyline).  Two passes over the sorted slots around a radix sort of the (strip, y0) entries.
  // Round 6, a second class, from the build's own statistics: maps of SHORT CHAINS (mean chain length below 16 edges -- the
  // test that already halves the run cap above; the WaterBodies lattice: 10-edge chains, 2.12 slots per segment, fat leaves).
  // Measured on one box, WaterBodies x BlockGroup: first pass alone 1.28 (walk) -> 0.97 ms (columns), step 2.45 -> 2.17 ms;
  // the build pays 3.4 ms for it (first 5.9 -> 9.3, rebuild 2.3 -> 5.6), i.e. from the twelfth step of a map on.  On lattices
  // of LONG chains the walk stays: forced there the columns lose (USCounty 0.54 -> 1.21 ms alone, step 0.77 -> 1.35; LakesNA
  // step 2.32 -> 2.35 and 100 ms of first build) -- profiles/r06_columns_ab.txt.
  b.strips_built = false;
  const bool rings = m.runs_cut && m.nc && 2 * m.closed_chains >= m.nc;
  const bool short_chains = m.runs_cut && m.nc && m.ne / m.nc < 16;
  const bool want_sky = h->skyline == 1 || (h->skyline < 0 && rings);
  b.columns_why = h->pip_columns == 0 ? "off (\"pip_columns\" 0)" : "the map's chains are long open polylines: the tree walk is the faster first pass";
  if (h->pip_columns == 1 || (h->pip_columns < 0 && (rings || short_chains))) {
    if (int r = build_strips(h, b, want_sky)) return r;
    b.columns_why = !b.strips_built ? "wanted, not built (a segment spans too many strips, or no memory for it): the tree serves the map"
                    : h->pip_columns == 1 ? "forced (\"pip_columns\" 1)"
                    : rings ? "most chains are closed rings (the tree is at its worst for upward rays there)"
                            : "short chains (mean chain length below 16 edges: fat leaves, measured rule of round 6)";
  }
  if (want_sky && !b.use_sky) {
    RJ_HIP(h, hipMemsetAsync(b.sky, 0, ((size_t) kSkyBuckets + 1) * 4, h->stream));
    RJ_HIP(h, launch_build_sky(h->stream, b.box0, b.seid, b.n0p, b.sky));
    b.use_sky = true;
  }
  RJ_HIP(h, launch_occ_count(h->stream, b.occ, h->d_occ_part, h->d_rest + 4));  // (how dense the pre-filter's bitmap is: DeviceBvh::occ_permille)
  toc(h, RJ_T_BUILD);
  RJ_HIP(h, hipStreamSynchronize(h->stream));
  b.occ_permille = (uint32_t) (h->h_rest[4] * 1000ull / ((unsigned long long) kOccDim * kOccDim));
  b.built = true;
  co_reset(h);
  h->h_rest[0] = h->h_rest[1] = ~0ull;  // (a new index: the "auto" decision to drop the walk is taken again)
  h->walk_n[0] = h->walk_n[1] = 0;
  return RJ_OK;
}

// A caller-owned point array under "query_order" 1 (see rj_handle_s::CallerSet): which order to process it in, without a
// host round trip except the first time the array (pointer, size) is seen.
static int sort_query_points(rj_handle h, const int64_t* pts, uint64_t n) {  // -> h->ord_vout
  if (int r = ensure_sort_scratch(h, n)) return r;
  tic(h, RJ_T_ORDER);
  RJ_HIP(h, launch_query_keys(h->stream, true, pts, nullptr, 0, n, h->ord_kin, h->ord_vin, h->order_strip_shift));
  size_t tb = h->ord_temp_bytes;
  RJ_HIP(h, sort_morton_pairs(h->stream, h->ord_temp, tb, h->ord_kin, h->ord_kout, h->ord_vin, h->ord_vout, n));
  toc(h, RJ_T_ORDER);
  return RJ_OK;
}
static int order_caller_points(rj_handle h, const int64_t* pts, uint64_t n, const uint32_t** order_out) {
  rj_handle_s::CallerSet* e = nullptr;
  int slot = -1;
  for (int k = 0; k < rj_handle_s::kCallerSets; k++)
    if (h->caller[k].valid && h->caller[k].p == pts && h->caller[k].n == n) { e = &h->caller[k]; slot = k; }
  bool incoherent = false;
  if (!e) {
    // first sight of this array: one estimate with a host round trip (a captured step cannot hold one)
    slot = 0;  // a free set, else the one used longest ago
    for (int k = 0; k < rj_handle_s::kCallerSets; k++) {
      if (!h->caller[k].valid) { slot = k; break; }
      if (h->caller[k].stamp < h->caller[slot].stamp) slot = k;
    }
    e = &h->caller[slot];
    // (an estimate still in flight for the set that goes away would land in the new set's word: drain first)
    RJ_HIP(h, hipStreamSynchronize(h->stream));
    RJ_HIP(h, join_aux(h));
    uint32_t* keep = e->perm;
    const uint64_t keep_cap = e->perm_cap;
    *e = rj_handle_s::CallerSet();
    e->perm = keep; e->perm_cap = keep_cap;
    e->p = pts; e->n = n; e->valid = true;
    h->h_est[slot] = 0;
    RJ_HIP(h, launch_group_extent_tail(h->stream, pts, nullptr, n, h->d_est + slot));
    RJ_HIP(h, hipStreamSynchronize(h->stream));
    incoherent = h->h_est[slot] > kIncoherentExtent + 1;
  } else {
    const unsigned long long v = h->h_est[slot];  // whatever the last finished estimate said (through the permutation then in use)
    if (v) {
      if (e->has_perm) {
        if (e->fresh_perm) { e->sorted_extent = v - 1; e->fresh_perm = false; }
        // the order no longer fits the contents: much wider groups than right after the sort
        incoherent = v - 1 > kIncoherentExtent && v - 1 > 2 * e->sorted_extent;
      } else {
        incoherent = v - 1 > kIncoherentExtent;
      }
    }
  }
  e->stamp = ++h->caller_clock;
  e->queries++;
  if (incoherent && h->lazy_columns_ok) {  // (the caller builds the column index instead: no permutation, this query or later)
    h->lazy_columns_want = true;
    h->cur_caller = slot;
    return RJ_OK;
  }
  if (incoherent) {
    if (e->perm_cap < n) {
      RJ_HIP(h, hipStreamSynchronize(h->stream));  // (the old permutation may be in use)
      RJ_HIP(h, join_aux(h));
      (void) hipFree(e->perm);
      e->perm = nullptr; e->perm_cap = 0; e->has_perm = false;
      RJ_HIP(h, hipMalloc((void**) &e->perm, n * sizeof(uint32_t)));
      e->perm_cap = n;
    }
    if (int r = sort_query_points(h, pts, n)) return r;
    h->order_fresh = true;
    RJ_HIP(h, hipMemcpyAsync(e->perm, h->ord_vout, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, h->stream));
    e->has_perm = true;
    e->fresh_perm = true;   // the next estimate says what "sorted" looks like for these contents
    h->h_est[slot] = 0;     // (estimates of the old order say nothing about this one; one still in flight is overwritten behind this query)
    e->queries = 1;
  }
  if (e->has_perm) {
    *order_out = e->perm;
    h->last_ordered = true;
  }
  h->cur_caller = slot;
  h->cur_order = *order_out;
  return RJ_OK;
}

// Decide whether the query set [begin, begin+n) needs re-ordering and, if so, produce the
// Morton-sorted permutation (indices relative to `begin`) in h->ord_vout.
static int maybe_order_queries(rj_handle h, bool points, const int64_t* pts, const Seg* segs, uint64_t begin,
                               uint64_t n, const uint32_t** order_out, int owner_map /* -1: caller's array */,
                               uint64_t key_begin) {
  *order_out = nullptr;
  h->last_ordered = false;
  h->order_fresh = false;
  h->cur_caller = -1;
  h->cur_order = nullptr;
  if (h->query_order == 0 || n <= 64) return RJ_OK;
  if (points && owner_map < 0 && h->query_order == 1) return order_caller_points(h, pts, n, order_out);
  if (h->query_order == 1) {
    rj_handle_s::CohCache* cc = owner_map >= 0 ? &h->coh[points ? 1 : 0][owner_map] : nullptr;
    bool incoherent;
    if (cc && cc->valid && cc->begin == key_begin && cc->n == n) {
      incoherent = cc->incoherent;  // same immutable range as last time: no estimate, no sync
    } else {
      RJ_HIP(h, hipMemsetAsync(h->d_counter + 4, 0, 16, h->stream));
      RJ_HIP(h, launch_group_extent(h->stream, points, pts, segs, begin, n, h->d_counter + 4));
      RJ_HIP(h, hipMemcpyAsync(h->h_pinned + 20, h->d_counter + 4, 16, hipMemcpyDeviceToHost, h->stream));
      RJ_HIP(h, hipStreamSynchronize(h->stream));
      const unsigned long long sum = h->h_pinned[20], groups = h->h_pinned[21];
      incoherent = groups != 0 && sum / groups > kIncoherentExtent;
      if (cc) { cc->valid = true; cc->begin = key_begin; cc->n = n; cc->incoherent = incoherent; }
    }
    if (!incoherent) return RJ_OK;
    if (h->lazy_columns_ok) { h->lazy_columns_want = true; return RJ_OK; }
  }
  rj_handle_s::OrdCache* oc = owner_map >= 0 ? &h->ordc[points ? 1 : 0][owner_map] : nullptr;
  if (oc && oc->valid && oc->begin == key_begin && oc->n == n) {  // sorted before, the map has not changed
    *order_out = oc->perm;
    h->last_ordered = true;
    return RJ_OK;
  }
  if (int r = ensure_sort_scratch(h, n)) return r;
  h->order_fresh = true;
  tic(h, RJ_T_ORDER);
  RJ_HIP(h, launch_query_keys(h->stream, points, pts, segs, begin, n, h->ord_kin, h->ord_vin, points ? h->order_strip_shift : 0));
  size_t tb = h->ord_temp_bytes;
  RJ_HIP(h, sort_morton_pairs(h->stream, h->ord_temp, tb, h->ord_kin, h->ord_kout, h->ord_vin, h->ord_vout, n));
  toc(h, RJ_T_ORDER);
  *order_out = h->ord_vout;
  h->last_ordered = true;
  if (oc) {  // keep it (the sort scratch is shared with the index build and other queries)
    oc->valid = false;
    if (oc->cap < n) {
      (void) hipFree(oc->perm);
      oc->perm = nullptr; oc->cap = 0;
      if (hipMalloc((void**) &oc->perm, n * sizeof(uint32_t)) != hipSuccess) { oc->perm = nullptr; return RJ_OK; }  // no cache, still correct
      oc->cap = n;
    }
    RJ_HIP(h, hipMemcpyAsync(oc->perm, h->ord_vout, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, h->stream));
    oc->begin = key_begin; oc->n = n; oc->valid = true;
    *order_out = oc->perm;
  }
  return RJ_OK;
}

static int lsi_launch(rj_handle h, int base_map_id, int query_map_id, uint64_t qb, uint64_t qe,
                      uint64_t capacity, uint32_t* pairs_dev, bool async_call = false) {
  if (base_map_id < 0 || base_map_id > 1 || query_map_id != 1 - base_map_id)
    return fail(h, RJ_E_INVALID, "rj_lsi_query: base/query map ids must be {0,1} and differ");
  if (!h->map[query_map_id].present) return fail(h, RJ_E_INVALID, "rj_lsi_query: query map not uploaded");
  if (!h->bvh[base_map_id].built) return fail(h, RJ_E_INVALID, "rj_lsi_query: call rj_build_lbvh(base map) first");
  if (qb > qe || qe > h->map[query_map_id].ne) return fail(h, RJ_E_INVALID, "rj_lsi_query: bad query eid range");
  if (capacity && !pairs_dev) return fail(h, RJ_E_INVALID, "rj_lsi_query: null output");
  if (int r = set_device(h)) return r;
  // Queue::Clear (queue.h:125-129) and the scheduler counters: cleared by the previous launch (see kSchedLsi)
  if (h->stats_on) RJ_HIP(h, hipMemsetAsync(h->d_stats, 0, 128, h->stream));
  const uint32_t* order = nullptr;
  if (int r = maybe_order_queries(h, false, nullptr, h->map[query_map_id].seg, qb, qe - qb, &order, query_map_id, qb)) return r;
  LsiArgs a;
  a.bvh = bvh_view(h->bvh[base_map_id]);
  a.qseg = h->map[query_map_id].seg;
  a.qcode = h->map[query_map_id].ccode;
  a.order = order;
  a.qbeg = qb; a.qend = qe;
  a.base_is_map0 = base_map_id == 0;
  a.out = pairs_dev; a.cap = capacity;
  const int flip = h->flip_lsi;
  a.counter = h->d_counter + flip;
  a.work_counter = (unsigned int*) (h->d_counter + kSchedLsi + flip * kSchedBlockWords);
  a.next_counter = h->d_counter + (1 - flip);
  a.next_work_counter = (unsigned int*) (h->d_counter + kSchedLsi + (1 - flip) * kSchedBlockWords);
  a.chunk_groups = (uint32_t) h->chunk_groups;  // (0: the launch wrapper picks it by the size of the query set)
  a.group_lanes = (uint32_t) h->group_lanes;
  a.stack_cap = h->debug_stack_cap;
  a.stats = h->stats_on ? h->d_stats : nullptr;
  // "pip_concurrent": the caller pairs every asynchronous LSI query with a PIP query (the step of a
  // join).  k_lsi is latency-bound and k_pip VALU-bound, so the two SHARE the chip instead of taking
  // turns: k_lsi runs on 1.25 blocks per CU (more when its side of the step is the heavier one: co_ratio), k_pip (second stream) on 5 -- per SIMD 5 x 80 + 1 x 72
  // VGPRs and per CU 5 x 26 + 1..2 x 17 KiB of LDS fit together.  Measured, whole step of USCounty x
  // BlockGroup: 1.39 ms instead of 1.52 taking turns (1.60 with both on full grids); 1/2, 1/4, 1/8
  // shards 0.76 / 0.43 / 0.28 instead of 0.89 / 0.51 / 0.30.  Not for re-ordered or instrumented
  // queries (shared scratch), and not for the synchronous rj_lsi_query (nothing can run beside it).
  // "pip_concurrent" 2 decides per workload (co_pick above).
  int max_blocks = h->max_blocks;
  const bool pairable = async_call && h->pip_concurrent != 0 && !(order && h->order_fresh) && !h->stats_on && qe > qb;
  h->co_mode = pairable ? (h->pip_concurrent == 2 ? co_pick(h, qe - qb) : 1) : 0;
  h->lsi_inflight = async_call && qe > qb;
  h->co_measure = h->co_points = false;
  h->lsi_shared = pairable && h->co_mode == 1;
  h->co_wpc = pip_walk_blocks_per_cu(h->bvh[base_map_id].top);
  if (h->lsi_shared && h->lsi_share_blocks() < max_blocks) max_blocks = h->lsi_share_blocks();
  tic(h, RJ_T_LSI_KERNEL);  // (after co_pick, which reads the previous pair's events)
  if (qe > qb) {
    RJ_HIP(h, launch_lsi(h->stream, a, h->stats_on, max_blocks, h->lsi_segments, &h->last_lsi_segments));
    h->flip_lsi = 1 - flip;
    rj_handle_s::PlanRec::Lsi& pl = h->plan.lsi;
    pl.ran = true; pl.epoch = h->plan.epoch; pl.n = qe - qb; pl.k = last_launch();
    pl.order = order ? (h->order_fresh ? 2 : 1) : 0;
    pl.paired = pairable; pl.co_mode = h->co_mode; pl.shared = h->lsi_shared;
  } else {
    RJ_HIP(h, hipMemsetAsync(a.counter, 0, 8, h->stream));  // (an empty query: nothing ran that could have counted)
  }
  h->count_word = (size_t) flip;
  toc(h, RJ_T_LSI_KERNEL);
  return RJ_OK;
}

int rj_lsi_query_async(rj_handle h, int base_map_id, int query_map_id, uint64_t qb, uint64_t qe,
                       uint64_t capacity, uint32_t* pairs_dev) {
  RJ_CHECK_H(h);
  return lsi_launch(h, base_map_id, query_map_id, qb, qe, capacity, pairs_dev, true);
}

int rj_lsi_query_finish(rj_handle h, uint64_t capacity, uint64_t* n_found) {
  RJ_CHECK_H(h);
  if (int r = set_device(h)) return r;
  RJ_HIP(h, hipMemcpyAsync(h->h_pinned, h->d_counter + h->count_word, 8, hipMemcpyDeviceToHost, h->stream));
  if (h->stats_on) RJ_HIP(h, hipMemcpyAsync(h->h_pinned + 1, h->d_stats, 128, hipMemcpyDeviceToHost, h->stream));
  RJ_HIP(h, hipStreamSynchronize(h->stream));
  h->lsi_shared = h->lsi_inflight = false;
  uint64_t n = h->h_pinned[0];
  h->h_rest[2] = n;  // (how many records a query like this one asks for: lsi_points_on_stream)
  if (h->stats_on) for (int i = 0; i < 16; i++) h->last_stats[i] = h->h_pinned[1 + i];
  if (n_found) *n_found = n;
  if (int r = check_fault(h)) return r;
  if (n > capacity)
    return fail(h, RJ_E_OVERFLOW, "intersection queue overflow: %llu found, capacity %llu",
                (unsigned long long) n, (unsigned long long) capacity);
  return RJ_OK;
}

int rj_lsi_count_async(rj_handle h, int slot) {
  RJ_CHECK_H(h);
  if (slot < 0 || slot > 1) return fail(h, RJ_E_INVALID, "rj_lsi_count_async: slot is 0 or 1");
  if (int r = set_device(h)) return r;
  RJ_HIP(h, hipMemcpyAsync(h->h_pinned + 28 + slot, h->d_counter + h->count_word, 8, hipMemcpyDeviceToHost, h->stream));
  RJ_HIP(h, hipEventRecord(h->ev_count[slot], h->stream));
  h->count_pending[slot] = true;
  return RJ_OK;
}

int rj_lsi_count_wait(rj_handle h, int slot, uint64_t capacity, uint64_t* n_found) {
  RJ_CHECK_H(h);
  if (slot < 0 || slot > 1 || !h->count_pending[slot]) return fail(h, RJ_E_INVALID, "rj_lsi_count_wait: no count in flight in slot %d", slot);
  if (int r = set_device(h)) return r;
  RJ_HIP(h, hipEventSynchronize(h->ev_count[slot]));
  h->count_pending[slot] = false;
  const uint64_t n = h->h_pinned[28 + slot];
  h->h_rest[2] = n;
  if (n_found) *n_found = n;
  if (int r = check_fault(h)) return r;
  if (n > capacity)
    return fail(h, RJ_E_OVERFLOW, "intersection queue overflow: %llu found, capacity %llu", (unsigned long long) n, (unsigned long long) capacity);
  return RJ_OK;
}

int rj_lsi_count_to(rj_handle h, uint64_t* n_found_dev) {
  RJ_CHECK_H(h);
  if (!n_found_dev) return fail(h, RJ_E_INVALID, "rj_lsi_count_to: null destination");
  if (int r = set_device(h)) return r;
  RJ_HIP(h, hipMemcpyAsync(n_found_dev, h->d_counter + h->count_word, 8, hipMemcpyDeviceToDevice, h->stream));
  return RJ_OK;
}

int rj_lsi_query(rj_handle h, int base_map_id, int query_map_id, uint64_t qb, uint64_t qe,
                 uint64_t capacity, uint32_t* pairs_dev, uint64_t* n_found) {
  RJ_CHECK_H(h);
  if (int r = lsi_launch(h, base_map_id, query_map_id, qb, qe, capacity, pairs_dev)) return r;
  return rj_lsi_query_finish(h, capacity, n_found);
}

// The records of n pairs (or of the queue's capacity, with the count on the device) on the main stream.  Two forms:
// k_lsi_points (gcd-free) + k_lsi_points_gcd over the pairs it declines -- 3.5x faster on 2 M pairs -- or
// k_lsi_points_gcd alone, which is ONE latency chain instead of two: better below ~0.4 M pairs (headline pair, 0.19 M:
// 53 us against 75 alone on the chip, and the step ends on this chain).  With the count on the device the host goes by
// the count of the last query, which the kernels leave in mapped host memory (unknown yet: two kernels).
constexpr uint64_t kPointsSplitAbove = 384 * 1024;
static hipError_t lsi_points_on_stream(rj_handle h, const uint32_t* pairs_dev, uint64_t n, const unsigned long long* n_dev, XsectRec* out) {
  const uint64_t seen = n_dev ? (uint64_t) h->h_rest[2] : n;
  const bool split = n < (1ull << 32) && (h->points_split >= 0 ? h->points_split == 1 : (seen == ~0ull || seen >= kPointsSplitAbove));
  if (split && h->slow_cap < n) {
    hipError_t e = hipStreamSynchronize(h->stream);  // (the list may be in use by records still being produced)
    if (e != hipSuccess) return e;
    (void) hipFree(h->slow_list);
    h->slow_list = nullptr; h->slow_cap = 0;
    if (hipMalloc((void**) &h->slow_list, n * 4) == hipSuccess) {
      h->slow_cap = n;
      (void) hipMemsetAsync(h->slow_list, 0, n * 4, h->stream);  // first touch outside the first timed use
    } else {
      (void) hipGetLastError();  // no room for the list: the one-kernel form below
    }
  }
  uint32_t* list = split && h->slow_cap >= n ? h->slow_list : nullptr;
  const int f = h->flip_slow;
  hipError_t e = launch_lsi_points(h->stream, h->map[0].seg, h->map[1].seg, pairs_dev, n, n_dev, out, list,
                                   h->d_counter + kSlowCountWord + f, h->d_counter + kSlowCountWord + (1 - f), h->d_rest + 2);
  if (list) h->flip_slow = 1 - f;
  h->last_points_split = list ? 1 : 0;
  h->plan.rec.ran = true; h->plan.rec.epoch = h->plan.epoch; h->plan.rec.two_kernels = list != nullptr;
  h->plan.rec.count_on_device = n_dev != nullptr; h->plan.rec.seen = seen;
  return e;
}

int rj_lsi_points(rj_handle h, const uint32_t* pairs_dev, uint64_t n, rj_xsect* out_dev) {
  RJ_CHECK_H(h);
  if (!h->map[0].present || !h->map[1].present) return fail(h, RJ_E_INVALID, "rj_lsi_points: both maps must be uploaded");
  if (n && (!pairs_dev || !out_dev)) return fail(h, RJ_E_INVALID, "rj_lsi_points: null buffer");
  if (int r = set_device(h)) return r;
  h->co_measure = false;  // (the pair-span events are re-recorded outside a pair)
  tic(h, RJ_T_LSI_POINTS);
  RJ_HIP(h, lsi_points_on_stream(h, pairs_dev, n, nullptr, (XsectRec*) out_dev));
  toc(h, RJ_T_LSI_POINTS);
  RJ_HIP(h, hipStreamSynchronize(h->stream));
  return RJ_OK;
}

int rj_lsi_points_async(rj_handle h, const uint32_t* pairs_dev, uint64_t capacity, rj_xsect* out_dev) {
  RJ_CHECK_H(h);
  if (!h->map[0].present || !h->map[1].present) return fail(h, RJ_E_INVALID, "rj_lsi_points_async: both maps must be uploaded");
  if (capacity && (!pairs_dev || !out_dev)) return fail(h, RJ_E_INVALID, "rj_lsi_points_async: null buffer");
  if (int r = set_device(h)) return r;
  if (h->lsi_inflight) h->co_points = true;
  tic(h, RJ_T_LSI_POINTS);
  RJ_HIP(h, lsi_points_on_stream(h, pairs_dev, capacity, h->d_counter + h->count_word, (XsectRec*) out_dev));
  toc(h, RJ_T_LSI_POINTS);
  return RJ_OK;
}

int rj_sort_pairs(rj_handle h, uint32_t* pairs_dev, uint64_t n) {
  RJ_CHECK_H(h);
  if (n == 0) return RJ_OK;
  if (!pairs_dev) return fail(h, RJ_E_INVALID, "rj_sort_pairs: null buffer");
  if (int r = set_device(h)) return r;
  uint64_t* keys = (uint64_t*) pairs_dev;  // little-endian (eid0, eid1) -> swap so eid0 is the high word
  uint64_t* tmp = nullptr;
  void* temp = nullptr;
  size_t temp_bytes = 0;
  if (int r = dev_alloc(h, &tmp, n)) return r;
  tic(h, RJ_T_SORT);
  hipError_t e = launch_swap_halves(h->stream, keys, n);
  if (e == hipSuccess) e = sort_keys_u64(h->stream, nullptr, temp_bytes, keys, tmp, n);
  if (e == hipSuccess) e = hipMalloc(&temp, temp_bytes ? temp_bytes : 1);
  if (e == hipSuccess) e = sort_keys_u64(h->stream, temp, temp_bytes, keys, tmp, n);
  if (e == hipSuccess) e = hipMemcpyAsync(keys, tmp, 8 * n, hipMemcpyDeviceToDevice, h->stream);
  if (e == hipSuccess) e = launch_swap_halves(h->stream, keys, n);
  toc(h, RJ_T_SORT);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  (void) hipFree(tmp); (void) hipFree(temp);
  RJ_HIP(h, e);
  return RJ_OK;
}

int rj_pip_query_async(rj_handle h, int base_map_id, int query_map_id, const int64_t* pts_dev,
                       uint64_t pt_begin, uint64_t n, uint32_t* closest_eid_dev, int32_t* face_id_dev) {
  RJ_CHECK_H(h);
  if (base_map_id < 0 || base_map_id > 1 || query_map_id != 1 - base_map_id)
    return fail(h, RJ_E_INVALID, "rj_pip_query: base/query map ids must be {0,1} and differ");
  if (!h->bvh[base_map_id].built) return fail(h, RJ_E_INVALID, "rj_pip_query: call rj_build_lbvh(base map) first");
  if (n && !closest_eid_dev) return fail(h, RJ_E_INVALID, "rj_pip_query: null output");
  const int64_t* pts = pts_dev;
  uint64_t coh_begin = 0;
  if (!pts) {
    coh_begin = pt_begin;
    const MapState& q = h->map[query_map_id];
    if (!q.present || pt_begin + n > q.np) return fail(h, RJ_E_INVALID, "rj_pip_query: bad point range of the query map");
    pts = q.pts + 2 * pt_begin;
  }
  if (int r = set_device(h)) return r;
  if (h->stats_on) RJ_HIP(h, hipMemsetAsync(h->d_stats, 0, 128, h->stream));
  const uint32_t* order = nullptr;
  // (a base map with a column index answers every point on its own -- nothing is shared between the points of a wave,
  //  so a scattered query set needs no re-ordering there)
  bool by_columns = h->bvh[base_map_id].strips_built && h->pip_walk != 0 && (!h->stats_on || h->pip_walk == 2);
  h->order_strip_shift = by_columns && h->debug_query_key_strips ? h->bvh[base_map_id].strip_shift : 0;
  if (by_columns && h->query_order != 2) { h->last_ordered = false; h->order_fresh = false; h->cur_caller = -1; h->cur_order = nullptr; }
  else {
    // (an incoherent set of enough points over a base map without a column index, "pip_columns" auto: build the index now
    //  instead of sorting the points -- see lazy_columns_ok)
    const uint64_t lazy_min = h->debug_lazy_columns_min ? (uint64_t) h->debug_lazy_columns_min : (1ull << 22);
    h->lazy_columns_ok = !by_columns && !h->bvh[base_map_id].strips_built && h->pip_columns < 0 && h->query_order == 1 && h->pip_walk != 0 &&
                         !h->stats_on && n >= lazy_min;
    h->lazy_columns_want = false;
    if (int r = maybe_order_queries(h, true, pts, nullptr, 0, n, &order, pts_dev ? -1 : query_map_id, coh_begin)) { h->lazy_columns_ok = false; return r; }
    h->lazy_columns_ok = false;
    if (h->lazy_columns_want) {
      BvhState& bb = h->bvh[base_map_id];
      RJ_HIP(h, hipStreamSynchronize(h->stream));  // (build_strips works on the main stream with the build's scratch words)
      RJ_HIP(h, join_aux(h));
      if (int r = build_strips(h, bb, false)) return r;
      if (bb.strips_built) {
        bb.columns_why = "built at the first PIP query over a spatially incoherent point set (uniform random points: every point on its own)";
        by_columns = true;
        co_reset(h);
        h->plan.epoch++;
        h->h_rest[0] = h->h_rest[1] = ~0ull;
        h->walk_n[0] = h->walk_n[1] = 0;
        h->last_ordered = false; h->order_fresh = false; h->cur_caller = -1; h->cur_order = nullptr;
      } else if (int r = maybe_order_queries(h, true, pts, nullptr, 0, n, &order, pts_dev ? -1 : query_map_id, coh_begin)) {
        return r;  // (no index after all -- a segment too wide, no memory: the permutation, as before)
      }
    }
  }
  // "pip_concurrent": the kernel goes to the handle's second stream and runs BESIDE the LSI kernel
  // of the same step on the main stream (both only read the maps and the tree): with an LSI query in
  // flight on its reduced grid (lsi_launch), on 5 blocks per CU; otherwise on the full grid.  Not
  // when the query went through the re-ordering pass or the instrumented build (shared scratch).
  // ("auto", taking turns: still the second stream, behind everything the main stream holds so far -- whatever a
  //  stream costs the first time it is used then lands in the first, cold pair and not in another schedule's trial)
  const bool aux = !(order && h->order_fresh) && !h->stats_on && (h->pip_concurrent == 1 || (h->pip_concurrent == 2 && h->lsi_inflight));
  if (aux && h->pip_concurrent == 2 && h->co_mode == 0) {
    RJ_HIP(h, hipEventRecord(h->ev_order, h->stream));
    RJ_HIP(h, hipStreamWaitEvent(h->aux_stream, h->ev_order, 0));
  }
  // (this pair's PIP share: derived NOW, from the split its LSI side -- already in flight -- was launched with; the reset below
  //  clears what that split was derived from)
  const int max_blocks = aux && h->lsi_shared && h->pip_share_blocks() < h->max_blocks ? h->pip_share_blocks() : h->max_blocks;
  const int walk_share_of_this_pair = h->lsi_share_blocks();
  // (auto mode: this PIP query completes a pair whose span the next pair's launch reads)
  h->co_measure = h->pip_concurrent == 2 && h->lsi_inflight && !(order && h->order_fresh) && !h->stats_on && n > 0;
  if (h->co_measure) {
    // (a pair with another point count is another workload: the schedule is decided again from the next pair on --
    //  the pair in flight keeps the grids its LSI side was launched with.  A size that differs ONCE -- the odd query of a
    //  stream of equal ones -- does not start the trials again: only when the new size is seen a second time)
    if (h->co_np && (n > h->co_np + h->co_np / 4 || n + n / 4 < h->co_np)) {
      h->co_measure = false;  // (its span belongs to neither workload's trials)
      if (h->co_np_other && !(n > h->co_np_other + h->co_np_other / 4 || n + n / 4 < h->co_np_other)) {
        const int keep_mode = h->co_mode;
        co_reset(h);
        h->co_mode = keep_mode;
        h->co_np = n;
        h->co_np_other = 0;
      } else {
        h->co_np_other = n;
      }
    } else {
      h->co_np = n;
      h->co_np_other = 0;
    }
  }
  hipStream_t st = aux ? h->aux_stream : h->stream;
  // each stream has its own scheduler block: a PIP on the aux stream and one on the main stream may be
  // in flight together (calls on ONE stream are ordered by the stream)
  const int pflip = h->flip_pip[aux ? 1 : 0];  // (cleared by the previous launch on this stream, see kSchedLsi)
  unsigned long long* sched = h->d_counter + (aux ? kSchedPipAux : kSchedPipMain) + pflip * kSchedBlockWords;
  PipArgs a;
  a.bvh = bvh_view(h->bvh[base_map_id]);
  a.base = map_view(h->map[base_map_id]);
  a.pts = pts; a.n = n;
  a.order = order;
  a.query_map_id = query_map_id;
  a.closest = closest_eid_dev; a.face = face_id_dev;
  a.work_counter = (unsigned int*) sched;
  a.next_work_counter = (unsigned int*) (h->d_counter + (aux ? kSchedPipAux : kSchedPipMain) + (1 - pflip) * kSchedBlockWords);
  a.chunk_groups = (uint32_t) h->chunk_groups;  // (0: the launch wrappers pick it by the size of the query set)
  a.group_lanes = (uint32_t) h->group_lanes;
  a.stack_cap = h->debug_stack_cap;
  a.walk_stack = h->debug_walk_stack;
  a.stats = h->stats_on ? h->d_stats : nullptr;
  a.rest = nullptr; a.rest_count = nullptr; a.next_rest_count = nullptr; a.n_dev = nullptr;
  a.todo = nullptr; a.todo_mask = nullptr;
  // (k_pip_walk keeps 8 blocks per CU resident where k_pip keeps 6: the shared schedule leaves k_lsi the same room)
  const int walk_full = h->cus * pip_walk_blocks_per_cu(h->bvh[base_map_id].top);
  int walk_blocks = h->max_blocks;
  if (aux && h->lsi_shared) {
    // (k_lsi on up to two blocks per CU: the walk leaves exactly that room -- 6 + 2 resident blocks per CU, nothing
    //  waits for a slot; a larger LSI share: the walk keeps all but one, the LSI side's later blocks fill in as it drains.
    //  tools/share_probe.py on the headline pair: 512 + 1536 blocks 0.897 ms, 512 + 1792 0.907, 448 + 1792 0.917)
    const int share = h->pip_share_set ? h->pip_share_set : walk_full - (walk_share_of_this_pair <= 2 * h->cus ? 2 : 1) * h->cus;
    walk_blocks = share < h->max_blocks ? share : h->max_blocks;
  }
  if (aux && h->lsi_shared) h->last_pip_share = max_blocks;  // (overwritten below when the walk runs)
  // Two passes unless instrumented: k_pip_walk (integer tests only, 8 waves per SIMD) settles every point whose
  // answer is a single certain hit and lists the others; k_pip, the exact kernel, locates those from scratch
  // right behind it, reading the count on the device.  "auto" drops the first pass for a query size whose
  // last run left more than 30 % of its points over (the hint arrives through mapped host memory).
  const int si = aux ? 1 : 0;
  // (instrumented: k_pip alone, unless "pip_walk" 2 asks for the walk's own counters)
  bool walk = h->pip_walk != 0 && (!h->stats_on || h->pip_walk == 2) && n > 0 && n < (1ull << 32);
  // how many points the last two-pass query of this size left to k_pip (either stream: the count belongs to the query)
  uint64_t seen = ~0ull;
  for (int k : {si, 1 - si})
    if (seen == ~0ull && h->walk_n[k] == n && h->h_rest[k] != ~0ull) {
      seen = h->h_rest[k];
    }
  // "auto" drops the first pass where it does not pay: most points left over, or many overflowed lists in absolute
  // terms -- k_pip locates those one scattered handful per wave (the list is in no useful order), which on the gaussian
  // polygons (25 k of 8 M) costs more than the walk saves
  const char* why = !h->pip_walk ? "\"pip_walk\" 0: the exact kernel alone" : (walk ? "" : "instrumented (\"stats\"), no points, or more than 2^32 of them: the exact kernel alone");
  if (walk && h->pip_walk == 1 && seen != ~0ull && (seen * 10 > n * 3 || seen > 16384)) {
    walk = false;
    why = "the last two-pass query of this size left too many points to the exact kernel (> 30 %, or > 16384): first pass dropped";
  }
  // the exact kernel on its own stream ("pip_exact_stream"): list set 1 or 2, see the handle
  const bool own = aux && walk && h->exact_own_stream;
  const int li = own ? 1 + h->exact_buf : si;
  if (walk && (h->rest_cap[si] < n || h->rest_cap[li] < n)) {
    // (both streams' lists at once, the first time a size is seen: a later query on the other stream -- the shared
    //  schedule's trial pair -- must not pay for an allocation inside its measured span)
    RJ_HIP(h, hipStreamSynchronize(h->stream));  // (the lists may still be in use by a query in flight)
    RJ_HIP(h, join_aux(h));
    for (int k = 0; k < (h->exact_own_stream ? 3 : 2); k++) {
      if (h->rest_cap[k] >= n) continue;
      (void) hipFree(h->rest[k]); (void) hipFree(h->todo[k]); (void) hipFree(h->todo_mask[k]);
      h->rest[k] = nullptr; h->rest_cap[k] = 0; h->todo[k] = nullptr; h->todo_mask[k] = nullptr;
      if (int r = dev_alloc(h, &h->rest[k], n)) return r;
      if (int r = dev_alloc(h, &h->todo[k], n * (uint64_t) pip_walk_list_slots())) return r;
      if (int r = dev_alloc(h, &h->todo_mask[k], n / 4 + 1)) return r;  // (groups of >= 4 points)
      h->rest_cap[k] = n;
      // first touch now, not inside the first query that uses them (fresh device memory is slow to touch: the
      // shared schedule's trial pair would look 0.15 ms worse than it is)
      RJ_HIP(h, hipMemsetAsync(h->todo[k], 0xFF, n * (uint64_t) pip_walk_list_slots() * 4, h->stream));
      RJ_HIP(h, hipMemsetAsync(h->todo_mask[k], 0, (n / 4 + 1) * 8, h->stream));
      RJ_HIP(h, hipMemsetAsync(h->rest[k], 0, n * 4, h->stream));
    }
    RJ_HIP(h, hipStreamSynchronize(h->stream));
  }
  h->last_passes = walk ? 3 : 1;
  h->last_walk_points = 1;
  tic(h, RJ_T_PIP_KERNEL, st);
  if (n && walk) {
    const int wflip = h->flip_walk[si];
    PipArgs w = a;
    w.work_counter = (unsigned int*) (h->d_counter + (aux ? kSchedWalkAux : kSchedWalkMain) + wflip * kSchedBlockWords);
    w.next_work_counter = (unsigned int*) (h->d_counter + (aux ? kSchedWalkAux : kSchedWalkMain) + (1 - wflip) * kSchedBlockWords);
    w.rest = h->rest[li];
    w.rest_count = h->d_counter + kRestCountWord + 2 * si + wflip;
    w.next_rest_count = h->d_counter + kRestCountWord + 2 * si + (1 - wflip);
    if (own) {
      w.rest_count = h->d_counter + kExactRestWord[h->exact_rot];
      w.next_rest_count = h->d_counter + kExactRestWord[(h->exact_rot + 1) % 3];
      h->exact_rot = (h->exact_rot + 1) % 3;
      // the exact kernel two queries back: the last reader of this list set, and of the count word this walk clears for the next
      if (h->exact_recorded[h->exact_buf]) RJ_HIP(h, hipStreamWaitEvent(st, h->ev_exact_done[h->exact_buf], 0));
      // ... and the LAST query's, if this one writes the same output arrays: its exact kernel still stores into them while this
      // walk would (two queries in flight over one array is the caller's rule to keep -- every host wrapper here reuses one array:
      // such a pair simply runs one after the other, as without the option)
      if (h->exact_last >= 0 && h->exact_recorded[h->exact_last] &&
          (h->exact_out_closest == (const void*) closest_eid_dev || (face_id_dev && h->exact_out_face == (const void*) face_id_dev)))
        RJ_HIP(h, hipStreamWaitEvent(st, h->ev_exact_done[h->exact_last], 0));
    }
    w.todo = h->todo[li]; w.todo_mask = h->todo_mask[li];
    if (!w.group_lanes) w.group_lanes = pip_walk_group_lanes(n, w.bvh.top, h->cus);
    // Two points per lane (k_pip_walk2: one traversal per 128 positions) where the query set is large enough for full
    // 64-position groups and nobody is counting visits: headline step -5.5 %, Zipcode -6 %, nested -5 %.  (The walk's
    // stack is cut at kWalkStack entries whatever the tree's height, so eight blocks per CU fit on every tree.)
    // From four 128-position groups per resident wave on: below that -- a 1/8 shard of the headline's query map -- the
    // one-point kernel's smaller groups fill the waves better (1/8 shard, pipelined step: 0.203 -> 0.190 ms).
    const int wp = h->walk_points == 4 ? 4 : 2;  // (four per lane: 256 positions per wave, half the blocks per CU)
    const bool two = h->walk_points >= 2 && w.group_lanes == 64 && !h->chunk_groups &&
                     pip_walk2_blocks_per_cu(w.bvh.top, wp) >= (wp == 4 ? 3 : 6) &&
                     n >= (uint64_t) wp * 256 * 4 * h->cus * pip_walk2_blocks_per_cu(w.bvh.top, wp);  // (four groups per resident wave)
    tic(h, RJ_T_PIP_WALK, st);
    // a base map with a column index (isolated rings): the first pass reads the point's strip instead of walking the tree
    const bool columns = w.bvh.strips.ytab != nullptr;  // (instrumented too: k_pip_strip counts its scans)
    h->last_columns = columns ? 1 : 0;
    if (columns) {
      w.group_lanes = 64;  // (one todo mask per 64 positions)
      RJ_HIP(h, launch_pip_strip(st, w, walk_blocks, h->cus));
    } else if (two) {
      if (aux && h->lsi_shared && !h->pip_share_set)
        walk_blocks = h->cus * pip_walk2_blocks_beside(w.bvh.top, walk_share_of_this_pair / h->cus < 1 ? 1 : walk_share_of_this_pair / h->cus, wp);
      RJ_HIP(h, launch_pip_walk2(st, w, walk_blocks, h->cus, h->stats_on, wp));
      h->last_walk_points = wp;
    } else {
      RJ_HIP(h, launch_pip_walk(st, w, h->stats_on, walk_blocks));
    }
    if (aux && h->lsi_shared) h->last_pip_share = walk_blocks;
    h->plan.pip.first = columns ? last_strip_launch() : last_launch();
    toc(h, RJ_T_PIP_WALK, st);
    h->flip_walk[si] = 1 - wflip;
    // second pass: the exact predicate over the candidate lists, and -- the kernel's first blocks -- k_pip's traversal
    // over the points whose list overflowed; that part's grid follows the last count seen for this query size
    // (the list is appended group by group all over the map: the fewer points it holds, the less a wave's points have
    //  to do with each other -- a handful of overflowed lists are unrelated traversals, one wave each)
    const uint64_t left = seen != ~0ull ? seen : 8192;
    PipRestArgs r;
    r.order = h->rest[li];
    r.n_dev = w.rest_count;
    r.rest_count = h->d_rest + si;  // (the count goes to the host)
    r.work_counter = a.work_counter;
    r.next_work_counter = a.next_work_counter;
    r.group_lanes = left < 8192 ? 1 : (left * 100 < n ? 4 : (left * 10 < n ? 8 : 16));
    r.chunk_groups = 1;
    int rest_blocks = h->cus * 4;
    if (seen != ~0ull) {
      const uint64_t want = seen / (r.group_lanes * 4 * 2) + 1;  // about two groups per wave
      const uint64_t lo = (uint64_t) h->cus / 4, hi = (uint64_t) h->cus * 4;  // (at least a block per four CUs: the count is a hint)
      rest_blocks = (int) (want < lo ? lo : (want > hi ? hi : want));
    }
    r.blocks = (uint32_t) rest_blocks;
    h->walk_n[si] = n;
    if (own) {
      RJ_HIP(h, hipEventRecord(h->ev_walk_done, st));
      RJ_HIP(h, hipStreamWaitEvent(h->exact_stream, h->ev_walk_done, 0));
      RJ_HIP(h, launch_pip_exact(h->exact_stream, w, h->cus * 8, r));
      RJ_HIP(h, hipEventRecord(h->ev_exact_done[h->exact_buf], h->exact_stream));
      h->exact_recorded[h->exact_buf] = true;
      h->exact_last = h->exact_buf;
      h->exact_out_closest = closest_eid_dev; h->exact_out_face = face_id_dev;
      h->exact_buf ^= 1;
      h->exact_pending = true;
    } else {
      RJ_HIP(h, launch_pip_exact(st, w, h->cus * 8, r));
    }
    h->flip_pip[si] = 1 - pflip;
    h->plan.pip.exact_blocks = h->cus * 8; h->plan.pip.locate_blocks = rest_blocks;
    if (columns) why = "the base map has a column index (closed rings or short chains: index[].columns_why): the first pass reads the point's strip";
    else if (!two && h->walk_points >= 2) why = "one point per lane: a query set too small for full groups on every resident wave (or a debug knob set)";
  } else if (n) {
    // (k_pip uses the scheduler block the last exact kernel of this stream used or cleared)
    if (aux && h->exact_pending && h->exact_last >= 0) RJ_HIP(h, hipStreamWaitEvent(st, h->ev_exact_done[h->exact_last], 0));
    RJ_HIP(h, launch_pip(st, a, h->stats_on, max_blocks));
    h->flip_pip[aux ? 1 : 0] = 1 - pflip;
    h->plan.pip.first = last_launch();
    h->plan.pip.exact_blocks = h->plan.pip.locate_blocks = 0;
  }
  {
    rj_handle_s::PlanRec::Pip& pp = h->plan.pip;
    pp.ran = n > 0; pp.epoch = h->plan.epoch; pp.n = n; pp.aux = aux; pp.caller = pts_dev != nullptr;
    pp.shared = aux && h->lsi_shared; pp.passes = h->last_passes; pp.rest_hint = seen; pp.why = why;
    pp.order = order ? (h->order_fresh ? 2 : 1) : 0;
  }
  toc(h, RJ_T_PIP_KERNEL, own ? h->exact_stream : st);
  // a caller-owned array: the estimate the next query over it will go by, behind this query's kernels on their stream
  // (the first queries after a sort or a first sight, then every fourth: one block, a few microseconds)
  if (h->cur_caller >= 0 && n) {
    const rj_handle_s::CallerSet& e = h->caller[h->cur_caller];
    if (e.queries <= 2 || e.queries % 4 == 0) RJ_HIP(h, launch_group_extent_tail(st, pts, h->cur_order, n, h->d_est + h->cur_caller));
  }
  if (aux) h->aux_pending = true;
  return RJ_OK;
}

int rj_pip_query(rj_handle h, int base_map_id, int query_map_id, const int64_t* pts_dev,
                 uint64_t pt_begin, uint64_t n, uint32_t* closest_eid_dev, int32_t* face_id_dev) {
  RJ_CHECK_H(h);
  if (int r = rj_pip_query_async(h, base_map_id, query_map_id, pts_dev, pt_begin, n, closest_eid_dev, face_id_dev)) return r;
  if (h->stats_on) RJ_HIP(h, hipMemcpyAsync(h->h_pinned + 1, h->d_stats, 128, hipMemcpyDeviceToHost, h->stream));
  RJ_HIP(h, hipStreamSynchronize(h->stream));
  RJ_HIP(h, join_aux(h));
  if (h->stats_on) for (int i = 0; i < 16; i++) h->last_stats[i] = h->h_pinned[1 + i];
  return check_fault(h);
}


// ---- -mode=grid on the device ------------------------------------------------------------------
int rj_build_grid(rj_handle h, int map_id, int grid_size) {
  RJ_CHECK_H(h);
  if (map_id < 0 || map_id > 1 || !h->map[map_id].present) return fail(h, RJ_E_INVALID, "rj_build_grid: map %d not uploaded", map_id);
  if (grid_size < 1 || grid_size > 32768) return fail(h, RJ_E_INVALID, "rj_build_grid: grid_size must be in [1, 32768]");
  if (int r = set_device(h)) return r;
  RJ_HIP(h, join_aux(h));
  const MapState& m = h->map[map_id];
  GridState& gr = h->grid[map_id];
  free_grid(gr);
  const uint64_t ncells = (uint64_t) grid_size * grid_size;
  // cell.h:16-22: scale = grid_size / (INTERNAL_MAX - INTERNAL_MIN) * 0.999, in double, once
  const double scale = (double) grid_size / (double) ((((int64_t) 1 << 46) - 1) + ((int64_t) 1 << 46)) * 0.999;
  tic(h, RJ_T_BUILD);
  uint32_t* counts = nullptr;
  uint64_t *keys = nullptr, *keys2 = nullptr;
  void* temp = nullptr;
  int rc = RJ_OK;
  hipError_t e = hipSuccess;
  do {
    if ((rc = dev_alloc(h, &counts, ncells + 1))) break;
    if ((rc = dev_alloc(h, &gr.begin, ncells + 1))) break;
    if ((e = hipMemsetAsync(counts, 0, (ncells + 1) * 4, h->stream)) != hipSuccess) break;
    if ((e = hipMemsetAsync(h->d_counter + 2, 0, 16, h->stream)) != hipSuccess) break;
    if ((e = launch_grid_count(h->stream, m.seg, m.ne, grid_size, scale, counts, h->d_counter + 2)) != hipSuccess) break;
    if ((e = hipMemcpyAsync(h->h_pinned + 26, h->d_counter + 2, 8, hipMemcpyDeviceToHost, h->stream)) != hipSuccess) break;
    if ((e = hipStreamSynchronize(h->stream)) != hipSuccess) break;
    gr.total = h->h_pinned[26];
    if (gr.total >= 0xFFFFFFFFull) {
      rc = fail(h, RJ_E_INVALID, "rj_build_grid: %llu (cell, edge) incidences do not fit 32-bit offsets: use a coarser grid",
                (unsigned long long) gr.total);
      break;
    }
    size_t scan_bytes = 0, sort_bytes = 0;
    if ((e = scan_cell_counts(h->stream, nullptr, scan_bytes, counts, gr.begin, ncells + 1)) != hipSuccess) break;
    if (gr.total && (e = sort_keys_u64(h->stream, nullptr, sort_bytes, (const uint64_t*) nullptr, (uint64_t*) nullptr, gr.total)) != hipSuccess) break;
    const size_t temp_bytes = scan_bytes > sort_bytes ? scan_bytes : sort_bytes;
    if ((e = hipMalloc(&temp, temp_bytes ? temp_bytes : 1)) != hipSuccess) break;
    size_t tb = temp_bytes;
    if ((e = scan_cell_counts(h->stream, temp, tb, counts, gr.begin, ncells + 1)) != hipSuccess) break;
    if ((rc = dev_alloc(h, &gr.eids, gr.total ? gr.total : 1))) break;
    if (gr.total) {
      if ((rc = dev_alloc(h, &keys, gr.total))) break;
      if ((rc = dev_alloc(h, &keys2, gr.total))) break;
      if ((e = launch_grid_emit(h->stream, m.seg, m.ne, grid_size, scale, keys, h->d_counter + 3)) != hipSuccess) break;
      tb = temp_bytes;
      if ((e = sort_keys_u64(h->stream, temp, tb, keys, keys2, gr.total)) != hipSuccess) break;
      if ((e = launch_grid_unpack(h->stream, keys2, gr.total, gr.eids)) != hipSuccess) break;
    }
    toc(h, RJ_T_BUILD);
    e = hipStreamSynchronize(h->stream);
  } while (0);
  (void) hipFree(counts); (void) hipFree(keys); (void) hipFree(keys2); (void) hipFree(temp);
  if (rc) { free_grid(gr); return rc; }
  if (e != hipSuccess) free_grid(gr);
  RJ_HIP(h, e);
  gr.g = grid_size;
  gr.scale = scale;
  gr.built = true;
  return RJ_OK;
}

int rj_lsi_query_grid(rj_handle h, uint64_t capacity, uint32_t* pairs_dev, uint64_t* n_found) {
  RJ_CHECK_H(h);
  const GridState &g0 = h->grid[0], &g1 = h->grid[1];
  if (!g0.built || !g1.built || g0.g != g1.g)
    return fail(h, RJ_E_INVALID, "rj_lsi_query_grid: call rj_build_grid for both maps with the same grid_size first");
  if (capacity && !pairs_dev) return fail(h, RJ_E_INVALID, "rj_lsi_query_grid: null output");
  if (int r = set_device(h)) return r;
  RJ_HIP(h, hipMemsetAsync(h->d_counter + kGridLsiCountWord, 0, 8, h->stream));  // Queue::Clear
  h->count_word = kGridLsiCountWord;
  GridLsiArgs a;
  a.g = g0.g; a.scale = g0.scale;
  a.begin0 = g0.begin; a.eids0 = g0.eids; a.begin1 = g1.begin; a.eids1 = g1.eids;
  a.seg0 = h->map[0].seg; a.seg1 = h->map[1].seg;
  a.out = pairs_dev; a.cap = capacity; a.counter = h->d_counter + kGridLsiCountWord;
  h->co_measure = false;  // (the pair-span events are re-recorded outside a pair)
  tic(h, RJ_T_LSI_KERNEL);
  RJ_HIP(h, launch_lsi_grid(h->stream, a));
  toc(h, RJ_T_LSI_KERNEL);
  return rj_lsi_query_finish(h, capacity, n_found);
}

int rj_pip_query_grid(rj_handle h, int base_map_id, int query_map_id, const int64_t* pts_dev, uint64_t pt_begin,
                      uint64_t n, uint32_t* closest_eid_dev, int32_t* face_id_dev) {
  RJ_CHECK_H(h);
  if (base_map_id < 0 || base_map_id > 1 || query_map_id != 1 - base_map_id)
    return fail(h, RJ_E_INVALID, "rj_pip_query_grid: base/query map ids must be {0,1} and differ");
  const GridState& gr = h->grid[base_map_id];
  if (!gr.built) return fail(h, RJ_E_INVALID, "rj_pip_query_grid: call rj_build_grid(base map) first");
  if (n && !closest_eid_dev) return fail(h, RJ_E_INVALID, "rj_pip_query_grid: null output");
  const int64_t* pts = pts_dev;
  if (!pts) {
    const MapState& q = h->map[query_map_id];
    if (!q.present || pt_begin + n > q.np) return fail(h, RJ_E_INVALID, "rj_pip_query_grid: bad point range of the query map");
    pts = q.pts + 2 * pt_begin;
  }
  if (int r = set_device(h)) return r;
  GridPipArgs a;
  a.g = gr.g; a.scale = gr.scale;
  a.begin = gr.begin; a.eids = gr.eids;
  a.base = map_view(h->map[base_map_id]);
  a.pts = pts; a.n = n;
  a.query_map_id = query_map_id;
  a.closest = closest_eid_dev; a.face = face_id_dev;
  h->co_measure = false;  // (the pair-span events are re-recorded outside a pair)
  tic(h, RJ_T_PIP_KERNEL);
  RJ_HIP(h, launch_pip_grid(h->stream, a));
  toc(h, RJ_T_PIP_KERNEL);
  RJ_HIP(h, hipStreamSynchronize(h->stream));
  return RJ_OK;
}


#define RJ_NCCL(h, expr)                                                                          \
  do {                                                                                            \
    ncclResult_t _r = (expr);                                                                     \
    if (_r != ncclSuccess)                                                                        \
      return fail(h, RJ_E_HIP, "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(_r), __FILE__, __LINE__); \
  } while (0)

int rj_comm_unique_id(uint8_t id[RJ_COMM_ID_BYTES]) {
  static_assert(RJ_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
  if (!id) return RJ_E_INVALID;
  ncclUniqueId u;
  if (ncclGetUniqueId(&u) != ncclSuccess) return RJ_E_HIP;
  memcpy(id, u.internal, RJ_COMM_ID_BYTES);
  return RJ_OK;
}

static void free_exchange(rj_handle h) {
  for (int k = 0; k < 2; k++) {
    (void) hipFree(h->ex[k].recv);  // (the send buffers are the caller's)
    h->ex[k] = rj_handle_s::Exch();
  }
  h->ex_capacity = h->ex_slot = 0;
}

int rj_comm_init(rj_handle h, int nranks, int rank, const uint8_t id[RJ_COMM_ID_BYTES]) {
  RJ_CHECK_H(h);
  if (!id || nranks < 1 || rank < 0 || rank >= nranks) return fail(h, RJ_E_INVALID, "rj_comm_init: bad arguments");
  if (h->comm) return fail(h, RJ_E_INVALID, "rj_comm_init: communicator already initialised");
  if (int r = set_device(h)) return r;
  ncclUniqueId u;
  memcpy(u.internal, id, RJ_COMM_ID_BYTES);
  RJ_NCCL(h, ncclCommInitRank(&h->comm, nranks, u, rank));
  // the second communicator (point queues): same ranks, its own stream -- collective on every rank, like the first
  // (should a RCCL build refuse the split, the point queues share the first communicator AND its stream: serialised
  //  behind the pair exchange, never concurrent with it)
  if (ncclCommSplit(h->comm, 0, rank, &h->comm2, nullptr) != ncclSuccess) h->comm2 = nullptr;
  h->nranks = nranks;
  h->rank = rank;
  if (int r = dev_alloc(h, &h->d_counts, 2 * (uint64_t) nranks)) return r;
  RJ_HIP(h, hipHostMalloc((void**) &h->h_heads, 3 * 16 * (size_t) nranks));
  RJ_HIP(h, hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
  RJ_HIP(h, hipStreamCreateWithFlags(&h->comm_stream2, hipStreamNonBlocking));
  RJ_HIP(h, hipEventCreateWithFlags(&h->ev_comm, hipEventDisableTiming));
  RJ_HIP(h, hipEventCreateWithFlags(&h->ev_comm2, hipEventDisableTiming));
  RJ_HIP(h, hipEventCreateWithFlags(&h->ev_comm2_aux, hipEventDisableTiming));
  return RJ_OK;
}

int rj_comm_destroy(rj_handle h) {
  RJ_CHECK_H(h);
  if (h->comm) {
    if (int r = set_device(h)) return r;
    RJ_HIP(h, hipStreamSynchronize(h->stream));
    if (h->comm_stream) RJ_HIP(h, hipStreamSynchronize(h->comm_stream));
    if (h->comm_stream2) RJ_HIP(h, hipStreamSynchronize(h->comm_stream2));
    if (h->comm2) RJ_NCCL(h, ncclCommDestroy(h->comm2));
    RJ_NCCL(h, ncclCommDestroy(h->comm));
    h->comm = h->comm2 = nullptr;
    free_exchange(h);
    (void) hipFree(h->d_counts); (void) hipFree(h->ag_send); (void) hipFree(h->ag_recv);
    (void) hipHostFree(h->h_heads);
    h->d_counts = nullptr; h->ag_send = h->ag_recv = nullptr; h->ag_send_words = h->ag_recv_words = 0; h->h_heads = nullptr;
    if (h->comm_stream) (void) hipStreamDestroy(h->comm_stream);
    if (h->comm_stream2) (void) hipStreamDestroy(h->comm_stream2);
    if (h->ev_comm) (void) hipEventDestroy(h->ev_comm);
    if (h->ev_comm2) (void) hipEventDestroy(h->ev_comm2);
    if (h->ev_comm2_aux) (void) hipEventDestroy(h->ev_comm2_aux);
    h->comm_stream = h->comm_stream2 = nullptr;
    h->ev_comm = h->ev_comm2 = h->ev_comm2_aux = nullptr;
    h->nranks = 1;
    h->rank = 0;
  }
  return RJ_OK;
}

int rj_allgatherv_plan(const uint64_t* counts, int nranks, uint64_t capacity, uint64_t* offsets, uint64_t* total) {
  if (!counts || !offsets || !total || nranks < 1) return RJ_E_INVALID;
  uint64_t sum = 0;
  for (int r = 0; r < nranks; r++) {
    offsets[r] = sum;
    if (counts[r] > UINT64_MAX - sum) return RJ_E_INVALID;
    sum += counts[r];
  }
  *total = sum;
  return sum > capacity ? RJ_E_OVERFLOW : RJ_OK;
}

int rj_exchange_verdict(const uint64_t* counts, const uint64_t* capacities, int nranks, uint64_t* max_count, int* first_bad) {
  if (!counts || !capacities || nranks < 1) return RJ_E_INVALID;
  uint64_t mx = 0;
  int bad = -1;
  for (int r = 0; r < nranks; r++) {
    if (counts[r] > mx) mx = counts[r];
    if (bad < 0 && counts[r] > capacities[r]) bad = r;
  }
  if (max_count) *max_count = mx;
  if (first_bad) *first_bad = bad;
  return bad >= 0 ? RJ_E_OVERFLOW : RJ_OK;
}

// all-gather-v of n_local elements of `elt_words` 32-bit words each, synchronous, exact slices at their offsets.
// Two symmetric collectives: (count, output capacity) of every rank, then every rank's slice padded to the largest
// count.  The verdict between them is a function of the gathered words alone, so every rank takes the same branch:
// if the total does not fit SOME rank's output, every rank returns RJ_E_OVERFLOW and no rank enters the second
// collective (round 3's form compared the total with the rank's own capacity only: ranks called with different
// capacities parted ways, and the others hung in ncclRecv).
static int allgatherv_words(rj_handle h, const uint32_t* src_dev, uint64_t n_local, uint32_t* out_dev,
                            uint64_t out_capacity, uint64_t* counts_out, uint64_t* n_total, int elt_words) {
  if (!h->comm) return fail(h, RJ_E_INVALID, "call rj_comm_init first");
  if ((n_local && !src_dev) || (out_capacity && !out_dev)) return fail(h, RJ_E_INVALID, "null buffer");
  if (int r = set_device(h)) return r;
  const int P = h->nranks;
  const size_t w = (size_t) elt_words;
  // 1. counts and capacities
  unsigned long long mine[2] = {n_local, out_capacity};
  RJ_HIP(h, hipMemcpyAsync(h->d_counts + 2 * h->rank, mine, 16, hipMemcpyHostToDevice, h->stream));
  RJ_NCCL(h, ncclAllGather(h->d_counts + 2 * h->rank, h->d_counts, 2, ncclUint64, h->comm, h->stream));
  unsigned long long* heads = h->h_heads + 4 * (size_t) P;
  RJ_HIP(h, hipMemcpyAsync(heads, h->d_counts, 16 * (size_t) P, hipMemcpyDeviceToHost, h->stream));
  RJ_HIP(h, hipStreamSynchronize(h->stream));
  std::vector<uint64_t> cnt(P), off(P);
  uint64_t total = 0, min_cap = UINT64_MAX, mx = 0;
  for (int r = 0; r < P; r++) {
    cnt[r] = heads[2 * r];
    if (heads[2 * r + 1] < min_cap) min_cap = heads[2 * r + 1];
    if (cnt[r] > mx) mx = cnt[r];
    if (counts_out) counts_out[r] = cnt[r];
  }
  const int plan = rj_allgatherv_plan(cnt.data(), P, min_cap, off.data(), &total);
  if (n_total) *n_total = total;
  if (plan == RJ_E_OVERFLOW)
    return fail(h, RJ_E_OVERFLOW, "all-gather-v: %llu elements in total, the smallest output capacity of a rank is %llu (this rank: %llu)",
                (unsigned long long) total, (unsigned long long) min_cap, (unsigned long long) out_capacity);
  if (plan != RJ_OK) return fail(h, plan, "all-gather-v: bad counts");
  if (mx == 0) return RJ_OK;
  // 2. every rank's slice, padded to the largest: one ncclAllGather, then the slices go to their offsets
  if (h->ag_send_words < mx * w) {
    (void) hipFree(h->ag_send); h->ag_send = nullptr; h->ag_send_words = 0;
    if (int r = dev_alloc(h, &h->ag_send, mx * w)) return r;
    h->ag_send_words = mx * w;
  }
  if (h->ag_recv_words < (uint64_t) P * mx * w) {
    (void) hipFree(h->ag_recv); h->ag_recv = nullptr; h->ag_recv_words = 0;
    if (int r = dev_alloc(h, &h->ag_recv, (uint64_t) P * mx * w)) return r;
    h->ag_recv_words = (uint64_t) P * mx * w;
  }
  if (n_local) RJ_HIP(h, hipMemcpyAsync(h->ag_send, src_dev, n_local * w * 4, hipMemcpyDeviceToDevice, h->stream));
  RJ_NCCL(h, ncclAllGather(h->ag_send, h->ag_recv, mx * w, ncclUint32, h->comm, h->stream));
  for (int r = 0; r < P; r++)
    if (cnt[r]) RJ_HIP(h, hipMemcpyAsync(out_dev + off[r] * w, h->ag_recv + (size_t) r * mx * w, cnt[r] * w * 4, hipMemcpyDeviceToDevice, h->stream));
  RJ_HIP(h, hipStreamSynchronize(h->stream));
  return RJ_OK;
}

int rj_allgather_pairs(rj_handle h, const uint32_t* pairs_dev, uint64_t n_local, uint32_t* out_dev,
                       uint64_t out_capacity, uint64_t* counts_out, uint64_t* n_total) {
  RJ_CHECK_H(h);
  return allgatherv_words(h, pairs_dev, n_local, out_dev, out_capacity, counts_out, n_total, 2);
}

int rj_allgather_u32(rj_handle h, const uint32_t* src_dev, uint64_t n_local, uint32_t* out_dev,
                     uint64_t out_capacity, uint64_t* counts_out, uint64_t* n_total) {
  RJ_CHECK_H(h);
  return allgatherv_words(h, src_dev, n_local, out_dev, out_capacity, counts_out, n_total, 1);
}

// ---- the overlapped exchange of a step --------------------------------------------------------
int rj_exchange_init(rj_handle h, uint64_t capacity, uint64_t slot, uint32_t* buf0_dev, uint32_t* buf1_dev) {
  RJ_CHECK_H(h);
  if (!h->comm) return fail(h, RJ_E_INVALID, "rj_exchange_init: call rj_comm_init first");
  if (capacity == 0 || capacity >= (1ull << 31) || !buf0_dev) return fail(h, RJ_E_INVALID, "rj_exchange_init: capacity out of range or null buffer");
  if (int r = set_device(h)) return r;
  RJ_HIP(h, hipStreamSynchronize(h->stream));
  RJ_HIP(h, hipStreamSynchronize(h->comm_stream));
  free_exchange(h);
  h->ex_capacity = capacity;
  h->ex_slot = slot < 1 ? 1 : (slot > capacity ? capacity : slot);
  const unsigned long long head[2] = {0, capacity};
  h->ex[0].send = buf0_dev;
  h->ex[1].send = buf1_dev;
  for (int k = 0; k < 2; k++)
    if (h->ex[k].send) RJ_HIP(h, hipMemcpy(h->ex[k].send, head, 16, hipMemcpyHostToDevice));
  return RJ_OK;
}

static int exchange_gather(rj_handle h, int buf, uint64_t slot, hipStream_t st) {
  rj_handle_s::Exch& x = h->ex[buf];
  const uint64_t per_rank = kExchHead + 2 * slot, need = (uint64_t) h->nranks * per_rank;
  if (x.recv_words < need) {
    RJ_HIP(h, hipStreamSynchronize(h->comm_stream));  // (the old buffer may be the target of a collective in flight)
    (void) hipFree(x.recv); x.recv = nullptr; x.recv_words = 0;
    if (int r = dev_alloc(h, &x.recv, need)) return r;
    x.recv_words = need;
  }
  x.slot = slot;
  RJ_NCCL(h, ncclAllGather(x.send, x.recv, per_rank, ncclUint32, h->comm, st));
  // the heads, packed, into pinned memory behind the collective
  RJ_HIP(h, hipMemcpy2DAsync(h->h_heads + 2 * (size_t) h->nranks * buf, 16, x.recv, per_rank * 4, 16, (size_t) h->nranks, hipMemcpyDeviceToHost, st));
  return RJ_OK;
}

int rj_exchange_pairs_begin(rj_handle h, int buf) {
  RJ_CHECK_H(h);
  if (buf < 0 || buf > 1 || !h->ex[buf].send) return fail(h, RJ_E_INVALID, "rj_exchange_pairs_begin: rj_exchange_init first (with this buffer); buf is 0 or 1");
  if (h->ex[buf].pending) return fail(h, RJ_E_INVALID, "rj_exchange_pairs_begin: buffer %d has an exchange in flight (rj_exchange_pairs_finish first)", buf);
  if (int r = set_device(h)) return r;
  // the device-side count of the LSI query that was just launched into this buffer goes into its head, on the
  // handle's stream; the collective waits for that point of the stream and no further
  RJ_HIP(h, hipMemcpyAsync(h->ex[buf].send, h->d_counter + h->count_word, 8, hipMemcpyDeviceToDevice, h->stream));
  RJ_HIP(h, hipEventRecord(h->ev_comm, h->stream));
  RJ_HIP(h, hipStreamWaitEvent(h->comm_stream, h->ev_comm, 0));
  if (int r = exchange_gather(h, buf, h->ex_slot, h->comm_stream)) return r;
  h->ex[buf].pending = true;
  h->ex_last_begin = buf;
  return RJ_OK;
}

int rj_exchange_pairs_finish(rj_handle h, int buf, uint64_t* counts_out, const uint32_t** slices_dev, uint64_t* n_total) {
  RJ_CHECK_H(h);
  if (buf < 0 || buf > 1 || !h->ex[buf].pending) return fail(h, RJ_E_INVALID, "rj_exchange_pairs_finish: no exchange in flight on buffer %d", buf);
  if (int r = set_device(h)) return r;
  rj_handle_s::Exch& x = h->ex[buf];
  const int P = h->nranks;
  RJ_HIP(h, hipStreamSynchronize(h->comm_stream));  // the step's one host sync (on the LSI side)
  x.pending = false;
  if (buf == h->ex_last_begin) h->lsi_shared = h->lsi_inflight = false;  // (not when a later step's LSI query is already in flight)
  std::vector<uint64_t> cnt(P), cap(P);
  const unsigned long long* heads = h->h_heads + 2 * (size_t) P * buf;
  for (int r = 0; r < P; r++) { cnt[r] = heads[2 * r]; cap[r] = heads[2 * r + 1]; }
  uint64_t mx = 0, total = 0;
  int bad = -1;
  const int verdict = rj_exchange_verdict(cnt.data(), cap.data(), P, &mx, &bad);
  for (int r = 0; r < P; r++) { total += cnt[r]; if (counts_out) counts_out[r] = cnt[r]; }
  if (n_total) *n_total = total;
  h->h_rest[2] = cnt[h->rank];  // (how many records a query like this one asks for: lsi_points_on_stream)
  if (int r = check_fault(h)) return r;
  if (verdict == RJ_E_OVERFLOW)  // the same words on every rank: every rank returns here
    return fail(h, RJ_E_OVERFLOW, "intersection queue overflow on rank %d: %llu found, capacity %llu", bad,
                (unsigned long long) cnt[bad], (unsigned long long) cap[bad]);
  if (mx > x.slot) {
    // (rare) some rank found more than a slot ships: every rank sees that, every rank gathers again with the same
    // larger slot -- the send buffer still holds this step's queue
    h->ex_slot = 2 * mx > h->ex_capacity ? h->ex_capacity : 2 * mx;
    if (int r = exchange_gather(h, buf, h->ex_slot, h->comm_stream)) return r;
    RJ_HIP(h, hipStreamSynchronize(h->comm_stream));
  } else if (4 * mx < h->ex_slot && h->ex_slot > 4096) {
    h->ex_slot = 2 * mx > 4096 ? 2 * mx : 4096;  // shrink for the next step
  }
  for (int r = 0; r < P && slices_dev; r++) slices_dev[r] = x.recv + (size_t) r * (kExchHead + 2 * x.slot) + kExchHead;
  return RJ_OK;
}

int rj_exchange_u32_begin(rj_handle h, const uint32_t* src_dev, uint64_t n_per_rank, uint32_t* recv_dev) {
  RJ_CHECK_H(h);
  if (!h->comm) return fail(h, RJ_E_INVALID, "rj_exchange_u32_begin: call rj_comm_init first");
  if (n_per_rank && (!src_dev || !recv_dev)) return fail(h, RJ_E_INVALID, "rj_exchange_u32_begin: null buffer");
  if (int r = set_device(h)) return r;
  ncclComm_t c = h->comm2 ? h->comm2 : h->comm;
  hipStream_t cs = h->comm2 ? h->comm_stream2 : h->comm_stream;
  // behind everything enqueued so far on BOTH of the handle's streams (the PIP kernels run on either)
  RJ_HIP(h, hipEventRecord(h->ev_comm2, h->stream));
  RJ_HIP(h, hipStreamWaitEvent(cs, h->ev_comm2, 0));
  RJ_HIP(h, hipEventRecord(h->ev_comm2_aux, h->aux_stream));
  RJ_HIP(h, hipStreamWaitEvent(cs, h->ev_comm2_aux, 0));
  if (h->exact_pending && h->exact_last >= 0) RJ_HIP(h, hipStreamWaitEvent(cs, h->ev_exact_done[h->exact_last], 0));  // ("pip_exact_stream")
  if (n_per_rank) RJ_NCCL(h, ncclAllGather(src_dev, recv_dev, n_per_rank, ncclUint32, c, cs));
  return RJ_OK;
}

int rj_exchange_u32_finish(rj_handle h) {
  RJ_CHECK_H(h);
  if (!h->comm) return fail(h, RJ_E_INVALID, "rj_exchange_u32_finish: call rj_comm_init first");
  if (int r = set_device(h)) return r;
  RJ_HIP(h, hipStreamSynchronize(h->comm2 ? h->comm_stream2 : h->comm_stream));
  return RJ_OK;
}

int rj_overlay_edge_xsects(rj_handle h, int im, const uint32_t* pairs_dev, uint64_t n, rj_xsect* xsects_dev) {
  RJ_CHECK_H(h);
  if (im < 0 || im > 1) return fail(h, RJ_E_INVALID, "rj_overlay_edge_xsects: im must be 0 or 1");
  if (!h->map[0].present || !h->map[1].present) return fail(h, RJ_E_INVALID, "rj_overlay_edge_xsects: both maps must be uploaded");
  if (!h->bvh[1 - im].built && !h->grid[1 - im].built)
    return fail(h, RJ_E_INVALID, "rj_overlay_edge_xsects: call rj_build_lbvh(%d) or rj_build_grid(%d, g) first (mid-points are located in the other map)", 1 - im, 1 - im);
  if (n && (!pairs_dev || !xsects_dev)) return fail(h, RJ_E_INVALID, "rj_overlay_edge_xsects: null buffer");
  if (n == 0) return RJ_OK;
  if (n >= (1ull << 32)) return fail(h, RJ_E_INVALID, "rj_overlay_edge_xsects: too many intersections");
  if (int r = set_device(h)) return r;
  // carve all temporaries out of one grow-only arena
  auto up = [](size_t v) { return (v + 255) & ~(size_t) 255; };
  size_t sort_bytes = 0;
  RJ_HIP(h, sort_pairs_u64_u32(h->stream, nullptr, sort_bytes, (const uint64_t*) nullptr, (uint64_t*) nullptr,
                              (const uint32_t*) nullptr, (uint32_t*) nullptr, n));
  const size_t need = up(48 * n) + 2 * up(8 * n) + 4 * up(4 * n) + up(16 * n) + up(sort_bytes);
  if (need > h->arena_bytes) {
    (void) hipFree(h->arena);
    h->arena = nullptr; h->arena_bytes = 0;
    RJ_HIP(h, hipMalloc((void**) &h->arena, need));
    h->arena_bytes = need;
  }
  char* p = h->arena;
  auto take = [&](size_t bytes) { char* r = p; p += up(bytes); return r; };
  XsectRec* tmp = (XsectRec*) take(48 * n);
  uint64_t* kin = (uint64_t*) take(8 * n);
  uint64_t* kout = (uint64_t*) take(8 * n);
  uint32_t* vin = (uint32_t*) take(4 * n);
  uint32_t* vout = (uint32_t*) take(4 * n);
  uint32_t* closest = (uint32_t*) take(4 * n);
  int32_t* face = (int32_t*) take(4 * n);
  int64_t* mid = (int64_t*) take(16 * n);
  void* temp = take(sort_bytes);
  size_t temp_bytes = sort_bytes;
  int rc = RJ_OK;
  hipError_t e = hipSuccess;
  do {
    // 1. the 48-byte records  2. order by (eid[im], eid[1-im])  3. per-edge order by distance, mid-points
    if ((e = lsi_points_on_stream(h, pairs_dev, n, nullptr, tmp)) != hipSuccess) break;
    if ((e = launch_xsect_keys(h->stream, tmp, n, im, kin, vin)) != hipSuccess) break;
    if ((e = sort_pairs_u64_u32(h->stream, temp, temp_bytes, kin, kout, vin, vout, n)) != hipSuccess) break;
    if ((e = launch_xsect_gather(h->stream, tmp, vout, n, (XsectRec*) xsects_dev)) != hipSuccess) break;
    if ((e = launch_xsect_order_runs(h->stream, (XsectRec*) xsects_dev, n, im, h->map[im].seg, mid)) != hipSuccess) break;
    // 4. locate the mid-points in the other map (query map id = im, map_overlay_lbvh.h:232-236)
    // (through the LBVH of the other map when there is one, else through its grid: MapOverlayGrid)
    if (h->bvh[1 - im].built) {
      // (the mid-points were produced on the main stream and the faces are consumed there)
      if ((e = hipStreamSynchronize(h->stream)) != hipSuccess) break;
      rc = rj_pip_query_async(h, 1 - im, im, mid, 0, n, closest, face);
      if (!rc && (e = join_aux(h)) != hipSuccess) break;
    } else {
      rc = rj_pip_query_grid(h, 1 - im, im, mid, 0, n, closest, face);
    }
    if (rc) break;
    if ((e = launch_xsect_set_mid(h->stream, (XsectRec*) xsects_dev, n, im, face)) != hipSuccess) break;
    e = hipStreamSynchronize(h->stream);
  } while (0);
  if (rc) return rc;
  RJ_HIP(h, e);
  return RJ_OK;
}

int rj_last_ms(rj_handle h, int which, float* ms) {
  RJ_CHECK_H(h);
  if (which < 0 || which >= kNumTimers || !ms) return fail(h, RJ_E_INVALID, "rj_last_ms: bad timer");
  if (!h->ev_valid[which]) return fail(h, RJ_E_INVALID, "rj_last_ms: stage %d has not run", which);
  if (int r = set_device(h)) return r;
  if (timers_off(h)) {
    // ("timers" 0: the last query recorded nothing -- the time is an earlier query's; what the header promises of this
    //  call, that the stage's outputs are complete when it returns, is kept by waiting for the streams themselves)
    RJ_HIP(h, hipStreamSynchronize(h->stream));
    RJ_HIP(h, join_aux(h));
  }
  RJ_HIP(h, hipEventSynchronize(h->ev[which][1]));
  RJ_HIP(h, hipEventElapsedTime(ms, h->ev[which][0], h->ev[which][1]));
  return RJ_OK;
}

int rj_last_ms_all(rj_handle h, float* ms, int n) {
  RJ_CHECK_H(h);
  if (!ms || n < 0) return fail(h, RJ_E_INVALID, "rj_last_ms_all: bad arguments");
  if (int r = set_device(h)) return r;
  for (int t = 0; t < n; t++) {
    ms[t] = -1.0f;
    if (t >= kNumTimers || !h->ev_valid[t]) continue;
    RJ_HIP(h, hipEventSynchronize(h->ev[t][1]));
    RJ_HIP(h, hipEventElapsedTime(&ms[t], h->ev[t][0], h->ev[t][1]));
  }
  return RJ_OK;
}

int rj_last_stats(rj_handle h, uint64_t stats[16]) {
  RJ_CHECK_H(h);
  if (!stats) return fail(h, RJ_E_INVALID, "null stats");
  for (int i = 0; i < 16; i++) stats[i] = h->last_stats[i];
  return RJ_OK;
}

int rj_dev_alloc(rj_handle h, size_t bytes, void** out_dev) {
  RJ_CHECK_H(h);
  if (!out_dev) return fail(h, RJ_E_INVALID, "null out pointer");
  if (int r = set_device(h)) return r;
  RJ_HIP(h, hipMalloc(out_dev, bytes ? bytes : 1));
  return RJ_OK;
}

int rj_dev_free(rj_handle h, void* dev) {
  RJ_CHECK_H(h);
  if (int r = set_device(h)) return r;
  RJ_HIP(h, hipFree(dev));
  return RJ_OK;
}

int rj_memcpy_h2d(rj_handle h, void* dst_dev, const void* src, size_t bytes) {
  RJ_CHECK_H(h);
  if (int r = set_device(h)) return r;
  if (bytes) RJ_HIP(h, hipMemcpyAsync(dst_dev, src, bytes, hipMemcpyHostToDevice, h->stream));
  RJ_HIP(h, hipStreamSynchronize(h->stream));
  return RJ_OK;
}

int rj_memcpy_d2h(rj_handle h, void* dst, const void* src_dev, size_t bytes) {
  RJ_CHECK_H(h);
  if (int r = set_device(h)) return r;
  if (bytes) RJ_HIP(h, hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, h->stream));
  RJ_HIP(h, hipStreamSynchronize(h->stream));
  return RJ_OK;
}

}  // extern "C"
