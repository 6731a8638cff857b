// rj_device.h -- device-side layout of maps and of the wave64-native implicit LBVH,
// plus the wave-level helpers every kernel uses.  gfx950 only (wave = 64 lanes).
//
// LBVH layout (replaces deps/lbvh's pointer-based binary Karras tree, deps/lbvh/lbvh/bvh.cuh:
// tree shape does not affect results, only the exact predicate does -- SURVEY fact 8):
//   * base segments are sorted by a 2-D Morton key of their midpoint (32 bits kept; y is the
//     most significant interleaved bit so the first children of a node are its low-y half);
//   * level 0 = the sorted segments, level l = groups of 64 consecutive level-(l-1) nodes:
//     a 64-ary, pointer-free, implicit tree.  Node i of level l covers children
//     [64 i, 64 i + 64) of level l-1, so ONE wave tests all children of a node with one
//     coalesced 1 KiB load and one __ballot;
//   * boxes are conservative 31-bit integer boxes: q(v) = (v + 2^46) >> 16.  floor() is monotone,
//     so real closed-interval overlap implies quantised overlap; the exact predicate decides.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rj_predicates.h"

namespace rj {

constexpr int kWave = 64;
constexpr int kMaxLevels = 6;          // levels 1..5 suffice for 2^32 segments (64^5 = 2^30 leaf blocks)
constexpr int kMaxTop = kMaxLevels - 1;  // highest level a tree accepted by rj_build_lbvh can have
// Traversal stacks (per wave, LDS).  Worst case = every child of every node overlaps the query
// group (one domain-spanning query is enough), depth-first from <= 64 top-level entries:
//   * popping ONE entry and pushing its <= 64 children grows the stack by <= 63 per level
//     descended: 64 + 63 (top - 1) entries (k_pip);
//   * popping TWO entries per step (k_lsi keeps two node loads in flight) grows it by <= 126 per
//     step, and by induction over the level of the entries on top (a batch of pushed children is
//     consumed completely before anything below it is touched) by <= 126 (top - 1) in all:
//     64 + 126 (top - 1) entries.
// Both bounds are checked against a simulation of the stack discipline in
// tests/test_stack_bounds.py; the kernels still test every push against the capacity and raise
// the handle's fault word (-> RJ_E_INTERNAL) instead of writing past the stack.
constexpr int kStackEntries = 64 + 126 * (kMaxTop - 1);  // 568
constexpr int kPipStack = 64 + 63 * (kMaxTop - 1) + 4;   // 320 (316 needed)
constexpr uint32_t kFaultLsiStack = 0, kFaultPipStack = 1;  // index of the kernel's word in the handle's fault words
// A scheduler block = 8 chunk counters 128 B apart (zeroed before every launch) followed by one
// line that is written once: the device-visible address of the handle's fault words.
constexpr int kSchedFaultPtrWord = 8 * 32;  // in 32-bit words from the block's start
constexpr int kPairBuf = 128;          // >= 64 (carry) + 64 (one append step)
constexpr int kQuantShift = 16;
constexpr int64_t kCoordOffset = (int64_t) 1 << 46;
constexpr int kOccShift = 19;                  // 31-bit quantised coordinate -> 12-bit cell
constexpr int kOccDim = 1 << (31 - kOccShift);  // 4096 x 4096 cells = 2 MiB of bits
constexpr int kOccRowWords = kOccDim / 32;
constexpr int kOccMaxCellsPerSeg = 4096;
constexpr uint32_t kOccDensePermille = 300;  // from here on launch_lsi treats the map as one whose every query group traverses (measured on the stand-ins, set cells per thousand: USCounty 60, Zipcode 147, Gaussian5M 149, WaterBodiesLike 243, LakesLike 268 | BlockGroup 358, LakesNA 483, WaterBodies 571)
// The SKYLINE of an indexed map (round 4): per x-bucket of 2^kSkyShift quanta, 1 + the highest quantised y of any
// segment whose box touches the bucket (0: none does).  A query point that lies above it has NO edge above itself --
// a certain miss of the upward ray, answered without a traversal.  Proving a miss is the one thing a box hierarchy is
// bad at (every box over the column has to be opened: 35 % of a lattice's vertices against a lake-shaped base map,
// 240 leaf blocks each); here it is one 4-byte load.  Word [kSkyBuckets] = 1 when some segment was too wide to
// register (the table is then not exhaustive and the kernels ignore it).
constexpr int kSkyShift = 13;
constexpr int kSkyBuckets = 1 << (31 - kSkyShift);  // 262 144 words = 1 MiB
constexpr int kSkyMaxSpan = 2048;                   // buckets one segment may raise  // larger boxes are not rasterised; word [kOccDim*kOccRowWords] flags that
// Morton keys keep only their top 32 bits (16 per axis over the scaled +-2^46 domain; the reference's
// codes have 10 per axis, deps/lbvh/lbvh/morton_code.cuh:23-35): finer bits do not change tree
// quality -- ties keep eid order = chain order, the sort is stable -- and a 32-bit key sorts in 4
// radix passes instead of 8.  The keys are stored already shifted and sorted from bit 0: rocPRIM
// 4.2's merge-sort path (10^4..10^6 items) returns a non-permutation for begin_bit > 0
// (tools/sort_probe.hip), begin_bit = 0 is fine at every size.
constexpr unsigned kMortonDropBits = 32;
using MortonKey = uint32_t;  // 64 - kMortonDropBits bits
constexpr int32_t kEmptyMin = 0x7FFFFFFF;
constexpr int32_t kEmptyMax = -1;

struct QBox {  // 16 bytes: one dwordx4 load
  int32_t x0, y0, x1, y1;
};

struct DeviceMap {
  const int64_t* pts;         // [np][2]
  const Seg* seg;             // [ne] in eid order
  const uint32_t* edge_chain; // [ne]
  const uint32_t* left;       // [nc]
  const uint32_t* right;      // [nc]
  uint64_t np, ne, nc;
};

// The COLUMN index of an indexed map (round 4, maps of isolated rings): the domain cut into vertical strips of
// 2^shift quanta; per strip the boxes of the sorted slots that touch it, ascending by y0, and a table of where
// every one of 1024 height buckets starts in that list.  An upward ray lives in ONE strip: one table read finds the
// entries at its height, a short scan upwards finds the edges above it -- O(1) per point wherever the point lies, where
// the box hierarchy opens every leaf block over the column whose x-extent contains the point (20 per point on the
// lake-shaped stand-in, 19 of them with nothing at that x).  The walk's job on such maps, with the walk's hand-over
// (k_pip_strip, rj_strip.hip).
// The strip width follows the map: what a point's scan wastes is entries that lie in its strip BESIDE it (5.0 of the
// 6.6 entries a point of the lake-shaped pair reads, 8.1 of 10.5 on lakes x parks), fewer the narrower the strip, while
// every strip a segment touches is one more entry.  Widest power of two below 2.3 x the mean x-extent of a segment,
// within [2^15, 2^17] (measured at 2^14 / 2^15 / 2^16, first pass: lake-shaped 2.55 / 2.19 / 2.13 ms, lakes x parks
// 2.08 / 1.98 / 3.72, gaussian polygons 0.84 / 0.63 / 0.67).
constexpr int kStripShiftMin = 15, kStripShiftMax = 17;
constexpr int kStripYBits = 10;                   // height buckets per strip: 1024 of 2^21 quanta (256: 2.42 / 3.98 / 1.10 ms first pass on the three ring pairs; 1024: 2.26 / 3.89 / 0.72; 2048: 2.27 / 3.91 / 0.66 at twice the table)
constexpr int kStripYShift = 31 - kStripYBits;
constexpr int kStripMaxSpan = 1024;               // strips one segment may touch (more: the index is not built)
__host__ __device__ __forceinline__ uint32_t strip_count(int shift) { return 1u << (31 - shift); }
struct DeviceStrips {
  const uint32_t* ytab;   // [strips * 1024 + 1] first entry at or above (strip, height bucket); nullptr: no index
  const QBox* ebox;       // [entries] the slot's box, entries ascending by (strip, y0)
  const uint4* einfo;     // [entries] {sorted slot, its edge id, its face id, 0}: what a candidate and a settled point need, one line
  const uint2* tall;      // [strips] {largest box height (y1 - y0) among the strip's entries, where the strip's entries end}
  int shift;              // a strip is 2^shift quanta wide
};

struct DeviceBvh {
  const Seg* sseg;        // [n0p] segments in Morton order (padding = zero segments)
  const uint32_t* seid;   // [n0p] original eid of each sorted slot
  const int32_t* sface;   // [n0p] face below each segment (get_face_id, src/map/map.h:79-87)
  const QBox* box0;       // [n0p] per-segment boxes (padding = empty); sorted by x0 inside each 64-block
  const int32_t* pmx1;    // [n0p] prefix max of box x1 inside each 64-block
  const uint2* xtab;      // [n0p] per 64-block: 256 x-buckets (leaf_bucket_shift), two bytes each -- lane l holds buckets 4l..4l+3:
                          // .x byte k = slots of the block whose x0 lies in a bucket <= 4l+k (the scan of a point in that
                          // bucket starts below this slot), .y byte k = slots whose prefix-max x1 ends before the bucket
                          // (where the scan stops): the candidates of a point, without a search and without a stop test
  const uint2* ytab2;     // [n0p] the same table on y over the order by y0, for blocks taller than wide (LSI only; nullable) -- leaf_is_steep
  const uint32_t* occ;    // occupancy bitmap, kOccDim x kOccDim cells (bit set = some segment box touches the cell)
  const uint32_t* sky;    // skyline, kSkyBuckets + 1 words (see kSkyShift); nullable
  DeviceStrips strips;    // column index (ytab == nullptr: none)
  const QBox* lvl[kMaxLevels];  // lvl[l] for l = 1..top, each padded to a multiple of 64
  // Behind the boxes of every level l (at lvl[l] + pad64(nlvl[l])) sits one 64-bit word per node:
  // bit k set = sibling k (same 64-entry group) lies HIGHER (box centre, ties by index) -- the
  // precomputed front-to-back order of an upward ray's traversal; see sibling_order().
  const uint64_t* ord[kMaxLevels];  // where that word array of level l starts
  uint32_t nlvl[kMaxLevels];    // real node count per level
  int top;                // top level: nlvl[top] <= 64
  // The level the LSI traversals START at (round 6): `top`, or top - 1 when the top level holds at most 4 nodes and the level
  // below at most 128 -- a 64-ary tree over 27 M slots ends in a top level of TWO nodes, and expanding it is a dependent round
  // trip to memory per query group for nothing: the <= 128 boxes of the level below are two loads side by side.
  int lsi_root;
  // set bits of the occupancy bitmap per thousand cells (k_occ_count at the build): how little the LSI pre-filter can dismiss
  uint32_t occ_permille;
  uint64_t n0;            // real segment count
};

// x-buckets of a leaf block (k_pip_walk): 256 buckets of 2^shift quanta from the block's x0, the smallest
// power of two that covers its x-extent.  Build (k_build_leaves) and query derive the shift from the same
// level-1 box, so a point's bucket is comparable with the segments' buckets: monotone in x.
__host__ __device__ __forceinline__ int leaf_bucket_shift(uint32_t extent_minus_1) {
  if (extent_minus_1 < 256u) return 0;
  return 24 - __builtin_clz(extent_minus_1);  // (extent_minus_1 >> shift) < 256
}
// A leaf block's SECOND order (k_build_leaves, round 6): blocks lie in x order (x0 ascending, prefix max of x1, bucket table on
// x: xtab).  A block TALLER than wide -- by its level-1 box, so whoever pushes it can tell -- also has ytab2: the same table on
// y over the order by y0, for the LSI kernels only; the x-order slot of y-rank `lane` sits in the table's spare bits (every
// count is <= 64: bit 7 of the four bytes of .x and of the low two bytes of .y).
// (taller than 1.25 x its width: a step of the y order costs one cross-lane read more than one of the x order, and a block
//  of packed rings -- about as tall as wide -- gains nothing from it: measured +4 % on k_lsi2 over the lake-shaped base map)
__host__ __device__ __forceinline__ bool leaf_is_steep(int32_t x0, int32_t y0, int32_t x1, int32_t y1) {
  const uint32_t w = (uint32_t) (x1 - x0), h = (uint32_t) (y1 - y0);
  return h > w && h - w > (w >> 2);
}
__device__ __forceinline__ uint32_t leaf_perm_bits_lo(uint32_t slot) {  // bits 0..3 of the slot -> bit 7 of bytes 0..3
  return ((slot & 1u) << 7) | ((slot & 2u) << 14) | ((slot & 4u) << 21) | ((slot & 8u) << 28);
}
__device__ __forceinline__ uint32_t leaf_perm_bits_hi(uint32_t slot) {  // bits 4, 5 -> bit 7 of bytes 0, 1
  return ((slot & 16u) << 3) | ((slot & 32u) << 10);
}
__device__ __forceinline__ uint32_t leaf_perm_of(const uint2& tab) {
  return ((tab.x >> 7) & 1u) | ((tab.x >> 14) & 2u) | ((tab.x >> 21) & 4u) | ((tab.x >> 28) & 8u) | ((tab.y >> 3) & 16u) | ((tab.y >> 10) & 32u);
}
__device__ __forceinline__ const uint64_t* sibling_order(const DeviceBvh& T, int l) {
  return T.ord[l];  // (= lvl[l] + pad64(nlvl[l]): one scalar load instead of a dozen scalar instructions per node expansion)
}
__device__ __forceinline__ int lane_id() {
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0));
}
// number of set bits of m below this lane
__device__ __forceinline__ int rank_below(uint64_t m) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0));
}
__device__ __forceinline__ int32_t quant(int64_t v) { return (int32_t) ((v + kCoordOffset) >> kQuantShift); }

// Wave64 reductions on the VALU's DPP path (no LDS crossbar round trips): quad swaps, row rotates,
// then row_bcast:15 / row_bcast:31 fold the four 16-lane rows; the total lands in lane 63 and is
// returned wave-uniform.  min/max are idempotent, so lanes outside a row_mask keeping their own
// value is harmless.
// (The DPP operand rides inside the min / max / or itself: six VALU instructions per reduction.  Through
// __builtin_amdgcn_update_dpp the compiler emits a copy, a v_mov_b32_dpp and the operation -- 18.  A VGPR written by a
// VALU instruction needs two wait states before a DPP instruction reads it: s_nop 1 between the dependent steps.)
#define RJ_DPP_REDUCE(OP, v)                                                                           \
  asm volatile("s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"  \
               OP " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"                \
               OP " %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"                          \
               OP " %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"                          \
               OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"                       \
               OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"                                       \
               : "+v"(v))
__device__ __forceinline__ int32_t wave_min(int32_t v) {
  RJ_DPP_REDUCE("v_min_i32_dpp", v);
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int32_t wave_max(int32_t v) {
  RJ_DPP_REDUCE("v_max_i32_dpp", v);
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ uint32_t wave_or(uint32_t v) {
  RJ_DPP_REDUCE("v_or_b32_dpp", v);
  return (uint32_t) __builtin_amdgcn_readlane((int32_t) v, 63);
}
// three reductions side by side (min, max, min): each step's three instructions are independent, so they are
// one another's wait states -- 18 VALU instructions and no s_nop for what a query group's head needs
__device__ __forceinline__ void wave_min_max_min(int32_t& a, int32_t& b, int32_t& c) {
#define RJ_DPP3(ctrl)                                       \
  "v_min_i32_dpp %0, %0, %0 " ctrl " bank_mask:0xf\n\t"      \
  "v_max_i32_dpp %1, %1, %1 " ctrl " bank_mask:0xf\n\t"      \
  "v_min_i32_dpp %2, %2, %2 " ctrl " bank_mask:0xf\n\t"
  asm volatile("s_nop 1\n\t" RJ_DPP3("quad_perm:[1,0,3,2] row_mask:0xf") RJ_DPP3("quad_perm:[2,3,0,1] row_mask:0xf")
               RJ_DPP3("row_ror:4 row_mask:0xf") RJ_DPP3("row_ror:8 row_mask:0xf") RJ_DPP3("row_bcast:15 row_mask:0xa")
               RJ_DPP3("row_bcast:31 row_mask:0xc")
               : "+v"(a), "+v"(b), "+v"(c));
#undef RJ_DPP3
  a = __builtin_amdgcn_readlane(a, 63);
  b = __builtin_amdgcn_readlane(b, 63);
  c = __builtin_amdgcn_readlane(c, 63);
}
#undef RJ_DPP_REDUCE
__device__ __forceinline__ int32_t bcast(int32_t v, int src_lane_uniform) {
  return __builtin_amdgcn_readlane(v, src_lane_uniform);
}
// compiler-level ordering of LDS traffic between lanes of one wave (hardware executes a wave's
// DS instructions in order; this only stops the compiler from reordering/caching them)
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
// candidate slots a PIP walk keeps per query point (the todo record handed to k_pip_exact holds as many); a point
// that needs more goes to the rest list
#ifndef RJ_WALK_LIST
#define RJ_WALK_LIST 6
#endif
constexpr int kWalkList = RJ_WALK_LIST;
// the upward ray from the quantised point (qx, qy) can meet something: some segment over its x-bucket reaches its height
// (see kSkyShift; `sky` null or not exhaustive: cannot tell)
__device__ __forceinline__ bool ray_has_sky(const uint32_t* __restrict__ sky, int32_t qx, int32_t qy) {
  return !sky || sky[(uint32_t) qx >> kSkyShift] > (uint32_t) qy;
}
// Closed-interval box overlap as ONE sign test: every difference below is non-negative exactly
// when the corresponding inequality holds, and no difference can overflow (coordinates are 31-bit
// quantised values, empty boxes are {INT_MAX, INT_MAX, -1, -1}).  Three ORs and one compare on
// the VALU replace four compares and three scalar ANDs -- the scalar unit (one per CU, shared by
// four SIMDs) is the scarcer issue port in the traversal loops.
__device__ __forceinline__ bool boxes_overlap(int32_t ax0, int32_t ay0, int32_t ax1, int32_t ay1, int32_t bx0,
                                              int32_t by0, int32_t bx1, int32_t by1) {
  return ((bx1 - ax0) | (ax1 - bx0) | (by1 - ay0) | (ay1 - by0)) >= 0;
}
__device__ __forceinline__ bool overlap(const QBox& a, int32_t bx0, int32_t by0, int32_t bx1, int32_t by1) {
  return boxes_overlap(a.x0, a.y0, a.x1, a.y1, bx0, by0, bx1, by1);
}
// PIP relevance of a box for a point: x0 <= qx <= x1, y1 >= qy - 1, y0 <= qbest
// (qym1 = max(qy - 1, 0) keeps every difference inside 32 bits; qbest = -1 marks an idle lane)
__device__ __forceinline__ bool ray_can_hit(int32_t qx, int32_t qym1, int32_t qbest, int32_t x0, int32_t y0,
                                            int32_t x1, int32_t y1) {
  return ((qx - x0) | (x1 - qx) | (y1 - qym1) | (qbest - y0)) >= 0;
}

}  // namespace rj
