// rj_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X): map upload, LBVH build,
// LSI and PIP traversal + exact predicates.  See rj_device.h for the tree layout and DESIGN.md
// for the roofline accounting.  No thrust/OptiX/CUDA anywhere; rocPRIM is used only for the
// radix sorts of the build/sort steps.
#include "rj_kernels.h"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

namespace rj {

// =============================================================================================
// Map upload: chain layout -> per-edge segments (src/map/map.h:187-230 restated: eid = p - c)
// =============================================================================================
// What the LSI pre-filter needs of a segment, in 4 bytes: the occupancy-bitmap cell of the low corner
// of its quantised box (12 + 12 bits) and how many cells the box spans per axis (0, 1, "more").
// k_lsi streams these instead of the 32-byte segments and only fetches the segments of groups the
// bitmap does not clear (a third of them in a sparse join).
__device__ __forceinline__ uint32_t cell_code(const Seg& s) {
  const int cx0 = quant(s.x1 < s.x2 ? s.x1 : s.x2) >> kOccShift, cx1 = quant(s.x1 < s.x2 ? s.x2 : s.x1) >> kOccShift;
  const int cy0 = quant(s.y1 < s.y2 ? s.y1 : s.y2) >> kOccShift, cy1 = quant(s.y1 < s.y2 ? s.y2 : s.y1) >> kOccShift;
  const int dx = cx1 - cx0 > 1 ? 2 : cx1 - cx0, dy = cy1 - cy0 > 1 ? 2 : cy1 - cy0;
  return (uint32_t) cx0 | ((uint32_t) cy0 << 12) | ((uint32_t) dx << 24) | ((uint32_t) dy << 26);
}
// edge_begin[c] = row_index[c] - c = first eid of chain c (strictly increasing), [nc+1]
__global__ __launch_bounds__(256) void k_build_segs(const int64_t* __restrict__ pts,
                                                    const uint32_t* __restrict__ edge_begin,
                                                    uint32_t nc, uint64_t ne, Seg* __restrict__ seg,
                                                    uint32_t* __restrict__ edge_chain, uint32_t* __restrict__ ccode) {
  for (uint64_t e = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; e < ne;
       e += (uint64_t) gridDim.x * blockDim.x) {
    // largest c with edge_begin[c] <= e
    uint32_t lo = 0, hi = nc;  // invariant: edge_begin[lo] <= e < edge_begin[hi]
    while (hi - lo > 1) {
      uint32_t mid = lo + ((hi - lo) >> 1);
      if (edge_begin[mid] <= e) lo = mid; else hi = mid;
    }
    uint64_t p = e + lo;
    const longlong2* P = reinterpret_cast<const longlong2*>(pts);
    longlong2 a = P[p], b = P[p + 1];
    Seg s;
    s.x1 = a.x; s.y1 = a.y; s.x2 = b.x; s.y2 = b.y;
    seg[e] = s;
    edge_chain[e] = lo;
    ccode[e] = cell_code(s);
  }
}

// =============================================================================================
// LBVH build
// =============================================================================================
__device__ __forceinline__ uint64_t spread32(uint32_t v) {  // insert a 0 bit between bits
  uint64_t x = v;
  x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
  x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
  x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
  x = (x | (x << 2)) & 0x3333333333333333ull;
  x = (x | (x << 1)) & 0x5555555555555555ull;
  return x;
}

// Position of the cell (x, y), 16 bits per axis, along the Hilbert curve (the classic rotate-and-
// descend form).  The base map is sorted along this curve rather than the Z-curve: consecutive cells
// of a Hilbert curve are always neighbours, so a run of 64 consecutive segments -- one node of the
// implicit 64-ary tree -- never straddles one of the Z-curve's long jumps, and its box stays tight.
// Measured on the stand-ins (boxes a 64-point PIP group needs, perfect pruning): leaf blocks 6.7 ->
// 5.3 (USCounty) and 8.5 -> 5.6 (WaterBodies), level-2 nodes 3.6 -> 1.9.
__device__ __forceinline__ uint32_t hilbert16(uint32_t x, uint32_t y) {
  uint32_t d = 0;
#pragma unroll
  for (uint32_t s = 0x8000u; s > 0; s >>= 1) {
    const uint32_t rx = (x & s) ? 1u : 0u, ry = (y & s) ? 1u : 0u;
    d += s * s * ((3u * rx) ^ ry);
    if (!ry) {
      if (rx) {
        x = 0xFFFFu - x;
        y = 0xFFFFu - y;
      }
      const uint32_t t = x;
      x = y;
      y = t;
    }
  }
  return d;
}

// Sort key of a base segment straight from the int64 midpoint (no float): 16 bits per axis over the
// scaled +-2^46 domain (the reference's Morton codes have 10 per axis, deps/lbvh/lbvh/morton_code.cuh:23-35)
__global__ __launch_bounds__(256) void k_morton(const Seg* __restrict__ seg, uint64_t ne,
                                                MortonKey* __restrict__ keys,
                                                uint32_t* __restrict__ vals) {
  for (uint64_t e = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; e < ne;
       e += (uint64_t) gridDim.x * blockDim.x) {
    Seg s = seg[e];
    uint64_t mx = (uint64_t) (((s.x1 + s.x2) >> 1) + kCoordOffset);  // 47 bits
    uint64_t my = (uint64_t) (((s.y1 + s.y2) >> 1) + kCoordOffset);
    keys[e] = (MortonKey) hilbert16((uint32_t) (mx >> 31), (uint32_t) (my >> 31));
    vals[e] = (uint32_t) e;
  }
}

// Polyline-run leaves: the sort key of a run = the Hilbert key of the midpoint of the middle edge of its middle piece
__global__ __launch_bounds__(256) void k_run_keys(const Seg* __restrict__ seg, const uint32_t* __restrict__ piece_begin,
                                                  const uint32_t* __restrict__ piece_len, const uint32_t* __restrict__ run_first,
                                                  uint64_t nruns, MortonKey* __restrict__ keys, uint32_t* __restrict__ vals,
                                                  uint32_t* __restrict__ run_len, QBox* __restrict__ run_box, uint32_t box_upto) {
  for (uint64_t r = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; r < nruns; r += (uint64_t) gridDim.x * blockDim.x) {
    const uint32_t p0 = run_first[r], p1 = run_first[r + 1];
    const uint32_t p = (p0 + p1) >> 1;
    const Seg s = seg[piece_begin[p] + (piece_len[p] >> 1)];
    const uint64_t mx = (uint64_t) (((s.x1 + s.x2) >> 1) + kCoordOffset), my = (uint64_t) (((s.y1 + s.y2) >> 1) + kCoordOffset);
    keys[r] = (MortonKey) hilbert16((uint32_t) (mx >> 31), (uint32_t) (my >> 31));
    vals[r] = (uint32_t) r;
    uint32_t len = 0;
    for (uint32_t q = p0; q < p1; q++) len += piece_len[q];
    run_len[r] = len;
    // the box of a SHORT run (one that may share a leaf, k_pack_runs): the vertices of its pieces
    QBox b = {kEmptyMin, kEmptyMin, kEmptyMax, kEmptyMax};
    if (len <= box_upto) {
      for (uint32_t q = p0; q < p1; q++) {
        const uint32_t e0 = piece_begin[q], n = piece_len[q];
        for (uint32_t k = 0; k < n; k++) {
          const Seg t = seg[e0 + k];
          const int32_t ax = quant(t.x1), ay = quant(t.y1), bx = quant(t.x2), by = quant(t.y2);
          b.x0 = min(b.x0, min(ax, bx)); b.y0 = min(b.y0, min(ay, by));
          b.x1 = max(b.x1, max(ax, bx)); b.y1 = max(b.y1, max(ay, by));
        }
      }
    }
    run_box[r] = b;
  }
}

// Leaves of SEVERAL runs (round 4).  A map of isolated rings -- lakes, parks: ten edges per chain, nothing to stitch --
// has runs far shorter than a leaf; one leaf per run would be mostly padding (> 2.5 slots per segment: the build used to
// fall back to Hilbert leaves there).  Consecutive runs of the Hilbert-sorted order share a leaf while they fit its 64
// slots; a run is never split, and a run longer than `solo_above` edges keeps a leaf to itself (a strip of one
// polyline is what makes a leaf thin).  Greedy along the sorted order, in independent chunks of kPackChunk runs (a
// chunk starts a new leaf: one part-filled leaf per 256 runs is the price of not being sequential): pass 1 counts
// the leaves of every chunk, a scan places the chunks, pass 2 writes where every leaf starts.
constexpr uint32_t kPackChunk = 256;
// (one WAVE per chunk: the lanes fetch 64 runs' lengths and boxes at once, the greedy decision then walks them in
//  registers -- a thread per chunk chased two dependent loads per run through memory, 0.3 ms for a 111 k-run map)
template <bool WRITE>
__global__ __launch_bounds__(256) void k_pack_runs(const uint32_t* __restrict__ order, const uint32_t* __restrict__ run_len,
                                                   const QBox* __restrict__ run_box, uint64_t nruns,
                                                   uint32_t solo_above, uint32_t spread, uint32_t* __restrict__ chunk_leaves,
                                                   const uint32_t* __restrict__ chunk_base, uint32_t* __restrict__ leaf_first) {
  const int lane = lane_id();
  const uint64_t nchunks = (nruns + kPackChunk - 1) / kPackChunk;
  const uint64_t wave = (blockIdx.x * (uint64_t) blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
  for (uint64_t c = wave; c < nchunks; c += nwaves) {
    const uint64_t j0 = c * kPackChunk, j1 = j0 + kPackChunk < nruns ? j0 + kPackChunk : nruns;
    uint32_t cur = 0, leaves = 0;
    bool closed = false;
    int32_t ux0 = kEmptyMin, uy0 = kEmptyMin, ux1 = kEmptyMax, uy1 = kEmptyMax;
    uint64_t own = 0;  // sum of the half-perimeters of the runs in the leaf
    const uint32_t base = WRITE ? chunk_base[c] : 0;
    for (uint64_t jb = j0; jb < j1; jb += 64) {
      const uint64_t j = jb + lane;
      uint32_t len = 0;
      QBox b = {kEmptyMin, kEmptyMin, kEmptyMax, kEmptyMax};
      if (j < j1) {
        const uint32_t r = order[j];
        len = run_len[r];
        if (len <= solo_above) b = run_box[r];
      }
      const int n = (int) (j1 - jb < 64 ? j1 - jb : 64);
      uint64_t starts = 0;
      for (int k = 0; k < n; k++) {
        const uint32_t lk = (uint32_t) bcast((int32_t) len, k);
        const bool solo = lk > solo_above;
        const int32_t bx0 = bcast(b.x0, k), by0 = bcast(b.y0, k), bx1 = bcast(b.x1, k), by1 = bcast(b.y1, k);
        const uint64_t hp = solo ? 0 : (uint64_t) (bx1 - bx0) + (uint64_t) (by1 - by0) + 2;
        bool join = cur != 0 && !closed && !solo && cur + lk <= 64;
        const int32_t vx0 = ux0 < bx0 ? ux0 : bx0, vy0 = uy0 < by0 ? uy0 : by0, vx1 = ux1 > bx1 ? ux1 : bx1, vy1 = uy1 > by1 ? uy1 : by1;
        // ... and only while the leaf stays about as large as what it holds: where the sorted order jumps -- the curve
        // leaving one cluster of rings for the next -- a shared leaf would be a box across the gap that every ray
        // through the gap has to open (measured on the lake-shaped stand-in: 650 leaf visits per 64 points)
        if (join) join = (uint64_t) (vx1 - vx0) + (uint64_t) (vy1 - vy0) <= (uint64_t) spread * (own + hp);
        if (join) {
          cur += lk;
          ux0 = vx0; uy0 = vy0; ux1 = vx1; uy1 = vy1;
          own += hp;
        } else {
          starts |= 1ull << k;
          cur = lk;
          closed = solo;
          ux0 = bx0; uy0 = by0; ux1 = bx1; uy1 = by1;
          own = hp;
        }
      }
      if (WRITE && ((starts >> lane) & 1)) leaf_first[base + leaves + (uint32_t) rank_below(starts)] = (uint32_t) j;
      leaves += (uint32_t) __popcll(starts);
    }
    if (!WRITE && lane == 0) chunk_leaves[c] = leaves;
  }
}
// exclusive scan of n counts by ONE block (n = runs / 256: tens of thousands at most), the total behind the last entry
// and, for the host, in *total_out (mapped or device memory)
__global__ __launch_bounds__(1024) void k_scan_counts(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint64_t n,
                                                      unsigned long long* __restrict__ total_out) {
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t carry;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint64_t b = 0; b < n; b += 1024) {
    const uint64_t i = b + threadIdx.x;
    const uint32_t v = i < n ? in[i] : 0;
    uint32_t inc = v;
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t t = __shfl_up(inc, d, 64);
      if (lane >= d) inc += t;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t before = carry;
    for (int k = 0; k < w; k++) before += wsum[k];
    if (i < n) out[i] = before + inc - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry = before + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[n] = carry;
    *total_out = carry;
  }
}
__global__ void k_pack_end(uint32_t* __restrict__ leaf_first, const uint32_t* __restrict__ chunk_base, uint64_t nchunks, uint32_t nruns) {
  leaf_first[chunk_base[nchunks]] = nruns;  // the sentinel behind the last leaf
}

// Occupancy bitmap of the indexed map: every cell a segment's quantised box touches is set.
// Two boxes that overlap share a point, hence a cell, so "no set bit under the query box" proves
// the query can have no candidate: the LSI kernel drops such lanes before the traversal (most of
// a sparse join) and, when a whole group is clear, skips the tree with ONE gather.
__device__ __forceinline__ void occ_or(uint32_t* word, uint32_t mask) {
  // neighbouring segments set the same bits: look (past the L1) before paying an atomic
  if ((__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & mask) != mask) atomicOr(word, mask);
}
// one wave marks its 64 boxes (any order inside the wave; `valid` = not a padding slot)
__device__ __forceinline__ void mark_occupancy_wave(const QBox& b, bool valid, uint32_t* __restrict__ occ, int lane) {
  const int cx0 = b.x0 >> kOccShift, cx1 = b.x1 >> kOccShift;
  const int cy0 = b.y0 >> kOccShift, cy1 = b.y1 >> kOccShift;
  // Common case: the box lies inside one 32-cell word horizontally and spans at most two rows.
  // The 64 segments of a leaf block are Morton neighbours, so most of the wave targets the same
  // few words: OR the masks per distinct word across the wave (DPP), then the leader lanes fire
  // one atomicOr each -- no return value, so nothing waits on the L2 round trip.
  const bool simple = valid && cy1 - cy0 <= 1 && (cx0 >> 5) == (cx1 >> 5);
  const int lo = cx0 & 31, hi = cx1 & 31;
  const uint32_t mask = (hi == 31 ? 0xFFFFFFFFu : ((1u << (hi + 1)) - 1u)) & ~((1u << lo) - 1u);
  for (int row = 0; row < 2; row++) {
    const bool act = simple && (row == 0 || cy1 > cy0);
    const uint32_t widx = (uint32_t) (cy0 + row) * kOccRowWords + (uint32_t) (cx0 >> 5);
    uint64_t todo = __ballot(act);
    uint32_t lead_mask = 0;
    while (todo) {
      const int leader = __builtin_ctzll(todo);
      const uint32_t w = (uint32_t) bcast((int32_t) widx, leader);
      const bool same = act && widx == w;
      const uint32_t m = wave_or(same ? mask : 0u);
      if (lane == leader) lead_mask = m;
      todo &= ~__ballot(same);
    }
    if (lead_mask) atomicOr(&occ[widx], lead_mask);
  }
  if (!valid || simple) return;
  if ((int64_t) (cx1 - cx0 + 1) * (cy1 - cy0 + 1) > kOccMaxCellsPerSeg) {
    // a segment whose box covers a large part of the map: rasterising its box would cost up to
    // 512 k atomics per segment; raise the "bitmap not exhaustive" word instead (the LSI kernel
    // then skips the pre-filter -- a performance hint, never a correctness input)
    atomicOr(&occ[(size_t) kOccDim * kOccRowWords], 1u);
    return;
  }
  for (int cy = cy0; cy <= cy1; cy++)
    for (int w = cx0 >> 5; w <= (cx1 >> 5); w++) {
      const int l = w == (cx0 >> 5) ? (cx0 & 31) : 0;
      const int h = w == (cx1 >> 5) ? (cx1 & 31) : 31;
      occ_or(&occ[(size_t) cy * kOccRowWords + w], (h == 31 ? 0xFFFFFFFFu : ((1u << (h + 1)) - 1u)) & ~((1u << l) - 1u));
    }
}

// One 8-byte load per row instead of four 4-byte gathers: the (at most two) cells of a row start at
// bit (cx0 & 31) of word cx0 >> 5 and never reach past the following word, and most query boxes
// lie in one row.  The scattered bitmap gathers were the largest consumer of the kernel's L1/TA
// bandwidth (4 x 64 lane-addresses per group against 2 x 64 for the segments themselves).
typedef uint64_t __attribute__((aligned(4))) occ_window_t;  // two consecutive words, word-aligned
__device__ __forceinline__ bool occ_any(const uint32_t* __restrict__ occ, int32_t x0, int32_t y0, int32_t x1, int32_t y1) {
  const int cx0 = x0 >> kOccShift, cx1 = x1 >> kOccShift;
  const int cy0 = y0 >> kOccShift, cy1 = y1 >> kOccShift;
  if (cx1 - cx0 > 1 || cy1 - cy0 > 1) return true;  // large box: let the tree decide
  const uint64_t want = (uint64_t) (cx1 > cx0 ? 3u : 1u) << (cx0 & 31);
  // (the word after the last one of the bitmap is the "not exhaustive" flag word: allocated)
  uint64_t bits = *reinterpret_cast<const occ_window_t*>(occ + (size_t) cy0 * kOccRowWords + (cx0 >> 5));
  if (cy1 != cy0) bits |= *reinterpret_cast<const occ_window_t*>(occ + (size_t) cy1 * kOccRowWords + (cx0 >> 5));
  return (bits & want) != 0;
}

// the same test from a segment's 4-byte cell code (cell_code), in two halves: the bitmap windows are REQUESTED for every
// lane (no branch around a load, so a kernel can have several sets' windows in flight at once), the verdict comes later
struct OccWindows { uint64_t a, b; };
__device__ __forceinline__ OccWindows occ_fetch_code(const uint32_t* __restrict__ occ, uint32_t code) {
  const int cx0 = code & 0xFFF, cy0 = (code >> 12) & 0xFFF, dy = (code >> 26) & 3;
  OccWindows w;
  w.a = *reinterpret_cast<const occ_window_t*>(occ + (size_t) cy0 * kOccRowWords + (cx0 >> 5));
  w.b = *reinterpret_cast<const occ_window_t*>(occ + (size_t) (cy0 + (dy ? 1 : 0)) * kOccRowWords + (cx0 >> 5));  // (row 4095 + 1: the flag word's row, allocated)
  return w;
}
__device__ __forceinline__ bool occ_verdict_code(const OccWindows& w, uint32_t code) {
  const int cx0 = code & 0xFFF, dx = (code >> 24) & 3, dy = (code >> 26) & 3;
  const uint64_t want = (uint64_t) (dx ? 3u : 1u) << (cx0 & 31);
  return dx > 1 || dy > 1 || ((w.a | w.b) & want) != 0;  // (large box: let the tree decide)
}
__device__ __forceinline__ bool occ_any_code(const uint32_t* __restrict__ occ, uint32_t code) {
  const int cx0 = code & 0xFFF, cy0 = (code >> 12) & 0xFFF, dx = (code >> 24) & 3, dy = (code >> 26) & 3;
  if (dx > 1 || dy > 1) return true;  // large box: let the tree decide
  const uint64_t want = (uint64_t) (dx ? 3u : 1u) << (cx0 & 31);
  uint64_t bits = *reinterpret_cast<const occ_window_t*>(occ + (size_t) cy0 * kOccRowWords + (cx0 >> 5));
  if (dy) bits |= *reinterpret_cast<const occ_window_t*>(occ + (size_t) (cy0 + 1) * kOccRowWords + (cx0 >> 5));
  return (bits & want) != 0;
}

// Leaf construction, one wave per 64-segment block, one pass over the data: gather the block's
// segments through the sorted order, derive face id (get_face_id, map.h:79-87) and quantised box,
// mark the occupancy bitmap, order the block, write it once, and emit the block's level-1 box.
// Inside a leaf block the order of the 64 segments is free (upper levels only see the union),
// so each block is sorted by box x0 and gets pmx1[j] = max(x1[0..j]).  A query lane then finds its
// candidates with a cross-lane binary search (segments with x0 <= key form a prefix) and a
// backward scan that stops as soon as pmx1 drops below the query's x -- one or two steps for
// an x-monotone run of a polyline instead of a 64-iteration uniform loop.
__global__ __launch_bounds__(256) void k_build_leaves(const Seg* __restrict__ seg, const uint32_t* __restrict__ order,
                                                      const uint32_t* __restrict__ edge_chain,
                                                      const uint32_t* __restrict__ left,
                                                      const uint32_t* __restrict__ right, uint64_t ne,
                                                      const uint32_t* __restrict__ piece_begin,
                                                      const uint32_t* __restrict__ piece_len,
                                                      const uint32_t* __restrict__ run_first,
                                                      const uint32_t* __restrict__ run_len,
                                                      const uint32_t* __restrict__ leaf_first,
                                                      uint64_t nblocks, uint64_t n_parent_alloc,
                                                      Seg* __restrict__ sseg, uint32_t* __restrict__ seid,
                                                      int32_t* __restrict__ sface, QBox* __restrict__ box0,
                                                      int32_t* __restrict__ pmx1, uint2* __restrict__ xtab,
                                                      QBox* __restrict__ lvl1, uint32_t* __restrict__ occ, uint2* __restrict__ ytab2) {
  __shared__ int32_t sx1[4][64];
  __shared__ uint4 hist[4][64];  // 256 x-bucket counters per wave
  const int lane = lane_id();
  const int wib = threadIdx.x >> 6;
  const uint64_t wave = (blockIdx.x * (uint64_t) blockDim.x + threadIdx.x) >> 6;
  const uint64_t nwaves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
  for (uint64_t blk = wave; blk < n_parent_alloc; blk += nwaves) {
    if (blk >= nblocks) {  // padding of level 1
      if (lane == 0) lvl1[blk] = QBox{kEmptyMin, kEmptyMin, kEmptyMax, kEmptyMax};
      continue;
    }
    const uint64_t i = blk * 64 + lane;
    Seg s = {0, 0, 0, 0};
    QBox b = {kEmptyMin, kEmptyMin, kEmptyMax, kEmptyMax};
    uint32_t id = 0xFFFFFFFFu;
    int32_t fc = 0;
    // Hilbert leaves: the block's segments are 64 neighbours of the sorted order.  Polyline-run leaves ("leaf_order" 1):
    // the block is one run (<= 64 edges in a few pieces: eid ranges of the chains the polyline crosses) or a few short runs
    // that follow each other in the sorted order (`order` sorts the runs, leaf_first[blk] is the block's first).
    bool valid = i < ne;
    if (piece_begin) {
      // The leaf's runs (k_pack_runs): one, or a few short ones -- at most 64, a run has an edge.  Lane l fetches run l
      // (its place in the sorted order, its length, its first piece), a prefix sum places the runs in the block, and the
      // lane of slot s then walks the pieces of ITS run only: four dependent loads deep whatever the number of runs.
      // (Round 4's form walked runs and pieces one after the other with wave-uniform loads: a dozen dependent round
      //  trips per block on a map of ten-edge rings -- 2.5 of the 9.2 ms of the lake-shaped map's first index build.)
      const uint32_t j0 = leaf_first[blk], nr = leaf_first[blk + 1] - j0;
      uint32_t rlen = 0, rf = 0;
      if ((uint32_t) lane < nr) {
        const uint32_t r = order[j0 + (uint32_t) lane];
        rlen = run_len[r];
        rf = run_first[r];
      }
      uint32_t inc = rlen;  // where run `lane` ends in the block
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = (uint32_t) __shfl_up((int) inc, d, 64);
        if (lane >= d) inc += t;
      }
      uint32_t k = 0;  // the run slot `lane` lies in: how many runs end at or before it
      for (uint32_t q = 0; q < nr; q++) k += (uint32_t) bcast((int32_t) inc, (int) q) <= (uint32_t) lane ? 1u : 0u;
      valid = k < nr;
      const uint32_t kk = valid ? k : 0u;
      const uint32_t run_end = (uint32_t) __shfl((int) inc, (int) kk, 64), run_n = (uint32_t) __shfl((int) rlen, (int) kk, 64);
      uint32_t p = (uint32_t) __shfl((int) rf, (int) kk, 64);
      if (valid) {
        uint32_t off = (uint32_t) lane - (run_end - run_n);
        uint32_t len = piece_len[p];
        while (off >= len) { off -= len; p++; len = piece_len[p]; }  // (a handful of pieces)
        id = piece_begin[p] + off;
      }
    } else if (valid) {
      id = __builtin_nontemporal_load(&order[i]);
    }
    if (valid) {
      s = seg[id];
      const uint32_t c = edge_chain[id];
      fc = (int32_t) (s.x1 < s.x2 ? right[c] : left[c]);
      b.x0 = quant(s.x1 < s.x2 ? s.x1 : s.x2);
      b.x1 = quant(s.x1 < s.x2 ? s.x2 : s.x1);
      b.y0 = quant(s.y1 < s.y2 ? s.y1 : s.y2);
      b.y1 = quant(s.y1 < s.y2 ? s.y2 : s.y1);
    }
    mark_occupancy_wave(b, valid, occ, lane);
    const int32_t ux0 = wave_min(b.x0), uy0 = wave_min(b.y0), ux1 = wave_max(b.x1), uy1 = wave_max(b.y1);
    if (lane == 0) lvl1[blk] = QBox{ux0, uy0, ux1, uy1};
    // The block's order: by x0, with the prefix max of x1 and a 256-bucket table on x -- what an upward ray needs (the
    // slots over its x) and what a query SEGMENT needs of a block that is wider than tall (the slots over its x-range).
    // Round 6: a block TALLER than wide -- a steep run of a polyline: its edges fold back and forth in x, most of them lie
    // over every query's x-range -- gets a SECOND order for the segment queries: by y0, table on y (ytab2), the slots of
    // that order packed into the table's spare bits (the counts are <= 64: bit 7 of six of a lane's eight bytes hold the
    // x-order slot of y-rank `lane`).  The block itself stays in x order: the PIP traversals never see the difference.
    // One pass per axis: rank by a0 (ties by lane), prefix max of a1 along that order, two histograms + two prefix sums:
    //   hi_b = how many slots START in a bucket <= b, lo_b = how many have a prefix max that ENDS before bucket b.
    auto order_on = [&](const int32_t a0, const int32_t a1, const int32_t ua0, const int32_t ua1, int& rank, int32_t& pmax, uint32_t (&packed)[2]) {
      rank = 0;
      for (int k = 0; k < 64; k++) {
        const int32_t xk = bcast(a0, k);
        rank += (xk < a0 || (xk == a0 && k < lane)) ? 1 : 0;
      }
      sx1[wib][rank] = a1;
      wave_lds_fence();
      int32_t m = sx1[wib][lane];
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        int32_t t = __shfl_up(m, d, 64);
        if (lane >= d) m = t > m ? t : m;
      }
      pmax = m;
      wave_lds_fence();
      const int sh = leaf_bucket_shift((uint32_t) (ua1 - ua0));
      const int nvalid = __popcll(__ballot(valid));
      uint32_t* hw = reinterpret_cast<uint32_t*>(&hist[wib][0]);
#pragma unroll
      for (int pass = 0; pass < 2; pass++) {
        hist[wib][lane] = make_uint4(0, 0, 0, 0);
        wave_lds_fence();
        // pass 0: by the bucket of a0 (this lane's own segment); pass 1: by the bucket of the prefix max of slot `lane`
        // (padding slots sort last and count in neither: a query never scans them)
        const bool use = pass == 0 ? valid : lane < nvalid;
        const int32_t v = pass == 0 ? a0 : m;
        if (use) atomicAdd(&hw[(uint32_t) (v - ua0) >> sh], 1u);
        wave_lds_fence();
        const uint4 c = hist[wib][lane];
        const uint32_t p0 = c.x, p1 = p0 + c.y, p2 = p1 + c.z, p3 = p2 + c.w;
        uint32_t inc = p3;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const uint32_t t = __shfl_up(inc, d, 64);
          if (lane >= d) inc += t;
        }
        const uint32_t exc = inc - p3;
        // pass 0: inclusive counts (bucket <= b); pass 1: exclusive counts (bucket < b)
        packed[pass] = pass == 0 ? (exc + p0) | ((exc + p1) << 8) | ((exc + p2) << 16) | ((exc + p3) << 24)
                                 : exc | ((exc + p0) << 8) | ((exc + p1) << 16) | ((exc + p2) << 24);
        wave_lds_fence();
      }
    };
    int rank = 0;
    int32_t m = 0;
    uint32_t packed[2];
    order_on(b.x0, b.x1, ux0, ux1, rank, m, packed);
    const uint64_t o = blk * 64 + rank;
    sseg[o] = s;
    seid[o] = id;
    sface[o] = fc;
    box0[o] = b;
    pmx1[blk * 64 + lane] = m;
    xtab[blk * 64 + lane] = make_uint2(packed[0], packed[1]);
    if (ytab2 && leaf_is_steep(ux0, uy0, ux1, uy1)) {
      int ranky = 0;
      int32_t my = 0;
      uint32_t py[2];
      order_on(b.y0, b.y1, uy0, uy1, ranky, my, py);
      sx1[wib][ranky] = rank;  // the x-order slot of y-rank `ranky`
      wave_lds_fence();
      const uint32_t xs = (uint32_t) sx1[wib][lane];
      wave_lds_fence();
      ytab2[blk * 64 + lane] = make_uint2(py[0] | leaf_perm_bits_lo(xs), py[1] | leaf_perm_bits_hi(xs));
    }
  }
}

// How dense the occupancy bitmap is (round 6): one block counts its set bits into a mapped host word.  What the LSI
// pre-filter can dismiss depends on it -- a third of the headline's query groups pass over USCounty's bitmap, nearly all
// over a dense lattice's -- and launch_lsi sizes the grid of a SMALL query set by it (DeviceBvh::occ_permille).
__global__ __launch_bounds__(1024) void k_occ_count(const uint32_t* __restrict__ occ, unsigned long long* __restrict__ part) {
  // (sixteen blocks, a partial sum each: one block over the whole 2 MiB was 0.23 ms -- half of USCounty's rebuild)
  __shared__ unsigned long long ws[16];
  unsigned long long c = 0;
  const uint32_t words = (uint32_t) kOccDim * kOccRowWords;
  for (uint32_t i = blockIdx.x * 1024u + threadIdx.x; i < words; i += gridDim.x * 1024u) c += (unsigned long long) __popc(occ[i]);
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int k = 0; k < 16; k++) t += ws[k];
    part[blockIdx.x] = t;
  }
}
__global__ void k_occ_sum(const unsigned long long* __restrict__ part, int n, unsigned long long* __restrict__ out_mapped) {
  unsigned long long t = 0;
  for (int k = 0; k < n; k++) t += part[k];
  __hip_atomic_store(out_mapped, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The skyline (maps of isolated rings that have no column index): every x-bucket a segment's box touches is at least as
// high as the box.  A pass of its own over the built leaves' boxes -- inside k_build_leaves it was 0.96 of that kernel's
// 2.19 ms on the lake-shaped map (an L2 look and sometimes an atomic per segment and bucket), paid by every build of a
// ring map although the default PIP path of such a map, the column index, never reads the table.
__global__ __launch_bounds__(256) void k_build_sky(const QBox* __restrict__ box0, const uint32_t* __restrict__ seid, uint64_t n0p,
                                                   uint32_t* __restrict__ sky) {
  for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n0p; i += (uint64_t) gridDim.x * blockDim.x) {
    if (seid[i] == 0xFFFFFFFFu) continue;
    const QBox b = box0[i];
    const int k0 = b.x0 >> kSkyShift, k1 = b.x1 >> kSkyShift;
    const uint32_t top = (uint32_t) b.y1 + 1u;
    if (k1 - k0 >= kSkyMaxSpan) {
      sky[kSkyBuckets] = 1u;
    } else {
      for (int k = k0; k <= k1; k++)  // (neighbours in a leaf raise the same buckets: look before paying an atomic)
        if (__hip_atomic_load(&sky[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < top) atomicMax(&sky[k], top);
    }
  }
}

// one wave per parent node: union of its 64 children
__global__ __launch_bounds__(256) void k_reduce_level(const QBox* __restrict__ child,
                                                      uint64_t n_child_alloc,
                                                      QBox* __restrict__ parent,
                                                      uint64_t n_parent_alloc) {
  const int lane = lane_id();
  uint64_t wave = (blockIdx.x * (uint64_t) blockDim.x + threadIdx.x) >> 6;
  uint64_t nwaves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
  for (uint64_t p = wave; p < n_parent_alloc; p += nwaves) {
    uint64_t c = p * 64 + lane;
    QBox b = {kEmptyMin, kEmptyMin, kEmptyMax, kEmptyMax};
    if (c < n_child_alloc) b = child[c];
    b.x0 = wave_min(b.x0);
    b.y0 = wave_min(b.y0);
    b.x1 = wave_max(b.x1);
    b.y1 = wave_max(b.y1);
    if (lane == 0) parent[p] = b;
  }
}


// Front-to-back order of siblings for upward rays (k_pip), precomputed: one wave per 64-entry
// group of a level; lane i gets the set of siblings whose box centre lies higher than its own (ties
// by index).  At push time "how many of the pushed siblings pop after me" is then one AND + popcount
// instead of a rank loop over the pushed children.  (Along the Hilbert curve the lane order says
// nothing about y, and an unordered push costs 3-7x in visits.)
__global__ __launch_bounds__(256) void k_sibling_order(const QBox* __restrict__ box, uint64_t n_alloc,
                                                       uint64_t* __restrict__ higher) {
  const int lane = lane_id();
  const uint64_t wave = (blockIdx.x * (uint64_t) blockDim.x + threadIdx.x) >> 6;
  const uint64_t nwaves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
  for (uint64_t g = wave; g * 64 < n_alloc; g += nwaves) {
    const QBox b = box[g * 64 + lane];
    const int32_t key = (int32_t) (((int64_t) b.y0 + b.y1) >> 1);
    uint64_t m = 0;
    for (int k = 0; k < 64; k++) {
      const int32_t kk = bcast(key, k);
      if (kk > key || (kk == key && k > lane)) m |= 1ull << k;
    }
    higher[g * 64 + lane] = m;
  }
}

// number of lanes j with v[j] <= key, for v non-decreasing over the 64 lanes (lane j holds v[j]).
// Two stages: 7 wave-uniform pivots (v at lanes 7, 15, ... 55; v_readlane -> SGPR, no latency chain)
// pick the 8-lane bucket, then 3 dependent cross-lane probes + 1 finish inside it -- instead of 7
// dependent ds_bpermute round trips.
__device__ __forceinline__ int wave_upper_bound(int32_t v, int32_t key) {
  int k = 0;
#pragma unroll
  for (int p = 7; p < 63; p += 8) k += (bcast(v, p) <= key) ? 8 : 0;  // v[p] <= key => lanes 0..p all count
  // now the answer lies in [k, k + 8]; v[k-1] <= key (or k == 0)
#pragma unroll
  for (int step = 4; step >= 1; step >>= 1) {
    const int32_t probe = __shfl(v, k + step - 1, 64);
    if (probe <= key) k += step;
  }
  const int32_t last = __shfl(v, k < 63 ? k : 63, 64);
  if (k < 64 && last <= key) k++;
  return k;
}

// =============================================================================================
// Query-side ordering for spatially incoherent query sets (GenerateLSIQueries/GeneratePIPQueries,
// run_query.cu:102-167, or any unsorted point array).  A wave works on 64 consecutive queries; if
// those are scattered over the map the wave serialises 64 unrelated traversals.  k_group_extent
// samples the average box half-perimeter of such groups; when it is large the host sorts the
// queries by Morton key once (rocPRIM) and the kernels run through the permutation.
// =============================================================================================
// boxes: per query a quantised box (points: x0==x1, y0==y1).  extent_sum += w + h of sampled groups
template <bool POINTS>
__global__ __launch_bounds__(256) void k_group_extent(const int64_t* __restrict__ pts, const Seg* __restrict__ segs,
                                                      uint64_t begin, uint64_t n, uint64_t group_stride,
                                                      unsigned long long* __restrict__ out /* [2] sum, groups */) {
  const int lane = lane_id();
  const uint64_t wave = (blockIdx.x * (uint64_t) blockDim.x + threadIdx.x) >> 6;
  const uint64_t nwaves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
  const uint64_t ngroups = (n + 63) >> 6;
  unsigned long long sum = 0, cnt = 0;
  for (uint64_t g = wave * group_stride; g < ngroups; g += nwaves * group_stride) {
    const uint64_t i = g * 64 + lane;
    int32_t x0 = kEmptyMin, y0 = kEmptyMin, x1 = kEmptyMax, y1 = kEmptyMax;
    if (i < n) {
      if (POINTS) {
        x0 = x1 = quant(pts[2 * (begin + i)]);
        y0 = y1 = quant(pts[2 * (begin + i) + 1]);
      } else {
        const Seg s = segs[begin + i];
        x0 = quant(s.x1 < s.x2 ? s.x1 : s.x2); x1 = quant(s.x1 < s.x2 ? s.x2 : s.x1);
        y0 = quant(s.y1 < s.y2 ? s.y1 : s.y2); y1 = quant(s.y1 < s.y2 ? s.y2 : s.y1);
      }
    }
    x0 = wave_min(x0); y0 = wave_min(y0); x1 = wave_max(x1); y1 = wave_max(y1);
    sum += (unsigned long long) (x1 - x0) + (unsigned long long) (y1 - y0);
    cnt++;
  }
  if (lane == 0 && cnt) {
    atomicAdd(&out[0], sum);
    atomicAdd(&out[1], cnt);
  }
}

// The same estimate for a caller-owned point array, ONE block, through the permutation the query runs with (if any),
// the result -- mean extent of <= 256 sampled groups, + 1 so that 0 means "nothing yet" -- stored as one word into
// mapped host memory: launched behind a query's kernels, read by the host when the NEXT query over that array is
// issued.  No synchronisation, and no 2048 same-address atomics (what made k_group_extent 55 us).
__global__ __launch_bounds__(1024) void k_group_extent_tail(const int64_t* __restrict__ pts, const uint32_t* __restrict__ order,
                                                            uint64_t n, unsigned long long* __restrict__ out) {
  __shared__ unsigned long long s_sum[16];
  __shared__ uint32_t s_cnt[16];
  const int lane = lane_id(), w = threadIdx.x >> 6;
  const uint64_t ngroups = (n + 63) >> 6;
  const uint64_t samples = ngroups < 256 ? ngroups : 256;  // (16 per wave: 1024 took 49 us behind every fourth query)
  const uint64_t stride = samples ? ngroups / samples : 1;
  unsigned long long sum = 0;
  uint32_t cnt = 0;
  for (uint64_t k = w; k < samples; k += 16) {
    const uint64_t i = k * stride * 64 + lane;
    int32_t x0 = kEmptyMin, y0 = kEmptyMin, x1 = kEmptyMax, y1 = kEmptyMax;
    if (i < n) {
      const uint64_t idx = order ? order[i] : i;
      x0 = x1 = quant(pts[2 * idx]);
      y0 = y1 = quant(pts[2 * idx + 1]);
    }
    x0 = wave_min(x0); y0 = wave_min(y0); x1 = wave_max(x1); y1 = wave_max(y1);
    sum += (unsigned long long) (x1 - x0) + (unsigned long long) (y1 - y0);
    cnt++;
  }
  if (lane == 0) { s_sum[w] = sum; s_cnt[w] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    uint32_t c = 0;
    for (int k = 0; k < 16; k++) { t += s_sum[k]; c += s_cnt[k]; }
    __hip_atomic_store(out, (c ? t / c : 0) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

template <bool POINTS>
__global__ __launch_bounds__(256) void k_query_keys(const int64_t* __restrict__ pts, const Seg* __restrict__ segs,
                                                    uint64_t begin, uint64_t n, MortonKey* __restrict__ keys,
                                                    uint32_t* __restrict__ vals, int strip_shift) {
  for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
    int64_t mx, my;
    if (POINTS) {
      mx = pts[2 * (begin + i)]; my = pts[2 * (begin + i) + 1];
    } else {
      const Seg s = segs[begin + i];
      mx = (s.x1 + s.x2) >> 1; my = (s.y1 + s.y2) >> 1;
    }
    if (strip_shift) {
      // STRIP-MAJOR order (a PIP query over a column index, rj_strip.hip): the point's strip above its height -- consecutive
      // positions then read consecutive places of ONE strip's table and list
      const int nb = 31 - strip_shift;  // bits of strip
      keys[i] = (MortonKey) ((((uint32_t) quant(mx) >> strip_shift) << (32 - nb)) | ((uint32_t) quant(my) >> (nb - 1)));
      vals[i] = (uint32_t) i;
      continue;
    }
    const uint32_t ux = (uint32_t) ((uint64_t) (mx + kCoordOffset) >> 15), uy = (uint32_t) ((uint64_t) (my + kCoordOffset) >> 15);
    keys[i] = (MortonKey) (((spread32(uy) << 1) | spread32(ux)) >> kMortonDropBits);
    vals[i] = (uint32_t) i;  // index relative to `begin`
  }
}

// XCD-aware dynamic scheduler.  The chunk range is cut into 8 contiguous parts, one per XCD
// (blocks b and b+8 share an XCD under the observed round-robin placement -- speed only, never
// correctness), each with its own counter on its own cache line.  Waves of one XCD therefore
// work on neighbouring chunks = neighbouring map regions = the same tree nodes, which stay in
// that XCD's private 4 MiB L2.  A wave whose part is exhausted steals from the next XCD's part.
__device__ __forceinline__ bool next_chunk(unsigned int* counters, uint32_t nchunks, int& part, int& tried,
                                           int lane, uint32_t& chunk_out) {
  while (tried < 8) {
    const uint32_t lo = (uint32_t) (((uint64_t) nchunks * part) >> 3);
    const uint32_t hi = (uint32_t) (((uint64_t) nchunks * (part + 1)) >> 3);
    uint32_t c = 0;
    if (lane == 0) c = atomicAdd(&counters[part * 32], 1u);
    c = __builtin_amdgcn_readfirstlane(c);
    if (lo + c < hi) {
      chunk_out = lo + c;
      return true;
    }
    part = (part + 1) & 7;
    tried++;
  }
  return false;
}

// Group-by-group hand-out on top of next_chunk, with stealing inside the block (k_pip).  A wave
// walks its chunk in order (the next group re-hits what the last one loaded), but the chunk's
// unstarted rest is visible to the block's other waves in LDS -- {end : next} per wave, a
// group is claimed with one ds_add_rtn_u64 by owner and thief alike.  When the global queue is dry,
// a wave without work takes single groups from a sibling that still holds some: the kernel's tail
// is a group, not a chunk, per block.  Measured with tools/timeline: with
// whole chunks committed to a wave, half the waves of a 1/8 shard had left 80 us before the last
// one (slot occupancy 0.57); with the stealing and chunks of 8, k_pip takes 0.207 instead of
// 0.240 ms there and 1.02 instead of 1.10 ms on the whole query map.
__device__ __forceinline__ bool take_from(unsigned long long* range, int lane, uint32_t& g) {
  unsigned long long w = 0;
  if (lane == 0) w = atomicAdd(range, 1ull);
  const uint32_t nx = __builtin_amdgcn_readfirstlane((uint32_t) w), en = __builtin_amdgcn_readfirstlane((uint32_t) (w >> 32));
  g = nx;
  return nx < en;
}
template <int NWAVES>
__device__ __forceinline__ bool next_group(unsigned long long* ranges, int wib, unsigned int* counters, uint32_t nchunks,
                                           uint32_t chunk_groups, uint64_t ngroups, int& part, int& tried, int lane, uint32_t& g) {
  if (take_from(&ranges[wib], lane, g)) return true;
  uint32_t chunk = 0;
  if (tried < 8 && next_chunk(counters, nchunks, part, tried, lane, chunk)) {
    const uint64_t b = (uint64_t) chunk * chunk_groups;
    const uint64_t e = b + chunk_groups < ngroups ? b + chunk_groups : ngroups;
    // (a sibling's failed claim on the old, used-up word may land before or after this store: either way it fails or
    // sees the new range whole -- LDS operations on one address are atomic and ordered)
    if (lane == 0) __hip_atomic_store(&ranges[wib], ((unsigned long long) e << 32) | (unsigned long long) (b + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    g = (uint32_t) b;
    return true;
  }
  // the global queue is dry for good (tried == 8): help a sibling out
  for (int s = 1; s < NWAVES; s++) {
    // (this is the kernel's tail: the siblings' LDS addresses are formed here, when wanted -- left to itself the compiler
    //  keeps all three in VGPRs through the whole kernel, and k_pip_walk2 has none to spare)
    int me = wib;
    asm volatile("" : "+v"(me));
    const int w = me + s < NWAVES ? me + s : me + s - NWAVES;
    if (take_from(&ranges[w], lane, g)) return true;
  }
  return false;
}

// Every push onto a traversal stack is checked against its capacity (wave-uniform scalars: three
// scalar instructions).  The capacities cover the worst case of every tree rj_build_lbvh accepts
// (rj_device.h), so this never fires; if it did, the children are dropped and the handle's fault
// word -- pinned host memory the device writes directly -- turns the query into RJ_E_INTERNAL
// instead of a silently corrupted stack (the reference: a fixed 64-entry per-thread stack with no
// check at all, deps/lbvh/lbvh/query.cuh:16).  The fault word's address sits behind the kernel's
// scheduler counters (kSchedFaultPtrWord), so the cold path costs the hot loops no register.
__device__ __noinline__ void raise_fault(unsigned int* work_counter, uint32_t which) {
  uint32_t* fault = *reinterpret_cast<uint32_t* const*>(work_counter + kSchedFaultPtrWord);
  // a plain store (idempotent; one word per kernel kind), not an atomic: no PCIe atomics needed
  __hip_atomic_store(fault + which, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ bool stack_has_room(int sp, int n, int cap, unsigned int* work_counter, uint32_t which, int lane) {
  if (sp + n <= cap) return true;
  if (lane == 0) raise_fault(work_counter, which);
  return false;
}

// =============================================================================================
// LSI: wave-cooperative traversal, LDS stack, ballot-compacted candidate pairs, dense predicate
// =============================================================================================
struct LsiWaveLds {
  uint32_t stack[kStackEntries];
  uint2 pairs[kPairBuf];  // (query eid, sorted base slot)
  uint2 hits[kPairBuf];   // (eid map 0, eid map 1)
};


// a stack entry of the LSI traversals: level above the index; a level-1 child (a leaf block) that is taller than wide and has
// its second order goes as level 0 -- the kernels then read its table on y (k_build_leaves, rj_device.h leaf_is_steep)
// (YS = false: the instantiation for trees without a second order -- maps of closed rings, "leaf_ysort" 0 -- in which nothing
//  of it is left: those kernels are the round-5 ones)
template <bool YS>
__device__ __forceinline__ uint32_t lsi_entry(const DeviceBvh& T, int lvl, const QBox& b, uint32_t index) {
  const bool by_y = YS && lvl == 1 && T.ytab2 && leaf_is_steep(b.x0, b.y0, b.x1, b.y1);
  return ((uint32_t) (by_y ? 0 : lvl) << 28) | index;
}

template <bool STATS>
__device__ __forceinline__ void lsi_flush_hits(LsiWaveLds& L, int& nh, int n, const LsiArgs& A, int lane) {
  // write the top n (<= 64) hits of the wave's LDS buffer with ONE atomic
  unsigned long long base = 0;
  if (lane == 0) base = atomicAdd(A.counter, (unsigned long long) n);
  base = ((unsigned long long) __builtin_amdgcn_readfirstlane((uint32_t) (base >> 32)) << 32) |
         __builtin_amdgcn_readfirstlane((uint32_t) base);
  if (lane < n) {
    uint2 h = L.hits[nh - n + lane];
    unsigned long long pos = base + lane;
    if (pos < A.cap) reinterpret_cast<uint2*>(A.out)[pos] = h;
  }
  nh -= n;
  wave_lds_fence();
}

template <bool STATS>
__device__ __forceinline__ void lsi_drain(LsiWaveLds& L, int& np, int& nh, int n, const LsiArgs& A,
                                          int lane, unsigned long long& st_tests) {
  // exact predicate on the top n (<= 64) candidate pairs, one pair per lane
  bool hit = false;
  uint2 h = {0, 0};
  if (lane < n) {
    uint2 pr = L.pairs[np - n + lane];
    Seg qs = A.qseg[pr.x];
    Seg bs = A.bvh.sseg[pr.y];
    uint32_t beid = A.bvh.seid[pr.y];
    if (A.base_is_map0) {
      hit = lsi_test(bs, qs);
      h.x = beid; h.y = pr.x;
    } else {
      hit = lsi_test(qs, bs);
      h.x = pr.x; h.y = beid;
    }
  }
  np -= n;
  if (STATS) st_tests += n;
  uint64_t hm = __ballot(hit);
  if (hm) {
    if (hit) L.hits[nh + rank_below(hm)] = h;
    nh += __popcll(hm);
    wave_lds_fence();
    if (nh >= 64) lsi_flush_hits<STATS>(L, nh, 64, A, lane);
  }
}

// (96 SGPRs: above that the hardware admits one block per CU fewer than the occupancy query says --
// MI355X_MICROARCH.md, "Residency" -- and this kernel lives on its resident waves)
template <bool STATS, bool YS>
__device__ __forceinline__ void lsi_body(const LsiArgs& A) {
  __shared__ LsiWaveLds lds[4];
  const int lane = lane_id();
  const int wib = threadIdx.x >> 6;
  LsiWaveLds& L = lds[wib];
  const uint64_t nq = A.qend - A.qbeg;
  const uint32_t GL = A.group_lanes;  // queries per wave: 64, or fewer for small query sets (more waves, shorter chains)
  const uint64_t ngroups = (nq + GL - 1) / GL;
  const DeviceBvh& T = A.bvh;
  // (this launch's counters were cleared by the previous launch on the stream; it clears the next one's)
  if (blockIdx.x == 0 && threadIdx.x < 8) A.next_work_counter[threadIdx.x * 32] = 0;
  if (blockIdx.x == 0 && threadIdx.x == 8) *A.next_counter = 0;
  const bool occ_usable = T.occ[(size_t) kOccDim * kOccRowWords] == 0;  // every base segment was rasterised
  const int stack_cap = STATS && A.stack_cap < kStackEntries ? A.stack_cap : kStackEntries;  // (lowered only by tests of the fault path, instrumented kernel)
  int np = 0, nh = 0;  // wave-uniform fill of L.pairs / L.hits
  unsigned long long st_leaf = 0, st_tests = 0, st_nodes = 0, st_box = 0, st_refined = 0, st_kept = 0;
  long long tk_node = 0, tk_leaf = 0, tk_head = 0, tk_sched = 0;  // STATS: cycle stamps
  const long long tk_begin = STATS ? clock64() : 0;

  // Dynamic scheduling in chunks of consecutive groups: consecutive groups of a chain walk almost
  // the same tree nodes, so a chunk re-hits them in this CU's L1/L2 instead of the Infinity Cache;
  // the atomic hands the next chunk to whichever wave is free (no tail).
  const uint32_t nchunks = (uint32_t) ((ngroups + A.chunk_groups - 1) / A.chunk_groups);
  int part = blockIdx.x & 7, tried = 0;
  for (;;) {
  uint32_t chunk = 0;
  const long long tks = STATS ? clock64() : 0;
  if (!next_chunk(A.work_counter, nchunks, part, tried, lane, chunk)) break;
  if (STATS) tk_sched += clock64() - tks;
  const uint64_t g_begin = (uint64_t) chunk * A.chunk_groups;
  const uint64_t g_end = g_begin + A.chunk_groups < ngroups ? g_begin + A.chunk_groups : ngroups;
  for (uint64_t g = g_begin; g < g_end; g++) {
    const long long tkg = STATS ? clock64() : 0;
    const uint64_t qi = g * GL + lane;  // position in the (possibly Morton-sorted) query order
    const bool valid = (uint32_t) lane < GL && qi < nq;
    const uint64_t qc = valid ? qi : nq - 1;  // (an idle lane reads the last query's data: no load stands under a per-lane branch)
    const uint64_t q = A.qbeg + (A.order ? A.order[qc] : qc);
    int32_t qx0 = kEmptyMin, qy0 = kEmptyMin, qx1 = kEmptyMax, qy1 = kEmptyMax;
    // Pre-filter on the 4-byte cell codes (streamed once): nothing of the base map is near a segment
    // whose cells are clear in the occupancy bitmap.  (Requesting the codes and bitmap windows of
    // four groups together -- two round trips per four groups -- measured +-0: the kernel's time is
    // in the groups that do traverse.)
    bool near = valid;
    if (occ_usable) {
      const uint32_t code = __builtin_nontemporal_load(A.qcode + q);
      const OccWindows win = occ_fetch_code(T.occ, code);
      near = valid && occ_verdict_code(win, code);
    }
    if (!__ballot(near)) {  // the whole group is clear of the base map: its segments are never read
      if (STATS) tk_head += clock64() - tkg;
      continue;
    }
    const int rl = T.lsi_root;                 // (the level the traversal starts at: DeviceBvh::lsi_root)
    const uint32_t rn = T.nlvl[rl];            // (<= 128 nodes)
    const QBox root_box = T.lvl[rl][lane];     // (requested beside the segments: one round trip, not two)
    const QBox root_box2 = T.lvl[rl][(rn > 64 ? 64 : 0) + lane];
    {
      Seg s = Seg{0, 0, 0, 0};
      if (near) s = A.qseg[q];  // (a masked load: a lane the pre-filter cleared reads nothing)
      if (near) {
        qx0 = quant(s.x1 < s.x2 ? s.x1 : s.x2);
        qx1 = quant(s.x1 < s.x2 ? s.x2 : s.x1);
        qy0 = quant(s.y1 < s.y2 ? s.y1 : s.y2);
        qy1 = quant(s.y1 < s.y2 ? s.y2 : s.y1);
      }
    }
    const int32_t gx0 = wave_min(qx0), gy0 = wave_min(qy0);
    const int32_t gx1 = wave_max(qx1), gy1 = wave_max(qy1);

    // A child is pushed only if SOME lane's own query box overlaps it: the union box is just
    // a cheap first filter.  The wave therefore visits exactly the union of the nodes its 64
    // queries need, whatever the spatial coherence of the group.
    auto refine = [&](const QBox& b, uint64_t um) -> uint64_t {
      uint64_t keep = 0;
      if (STATS) st_refined += (unsigned long long) __popcll(um);
      while (um) {
        const int c = __builtin_ctzll(um);
        um &= um - 1;
        const int32_t cx0 = bcast(b.x0, c), cy0 = bcast(b.y0, c);
        const int32_t cx1 = bcast(b.x1, c), cy1 = bcast(b.y1, c);
        if (__ballot(boxes_overlap(qx0, qy0, qx1, qy1, cx0, cy0, cx1, cy1))) keep |= 1ull << c;
      }
      if (STATS) st_kept += (unsigned long long) __popcll(keep);
      return keep;
    };
    int sp = 0;
    {  // the starting level: <= 64 nodes, one per lane -- or <= 128, the second 64 likewise
      QBox b = root_box;
      uint64_t m = refine(b, __ballot(overlap(b, gx0, gy0, gx1, gy1)));
      if (!stack_has_room(0, __popcll(m), stack_cap, A.work_counter, kFaultLsiStack, lane)) m = 0;
      if ((m >> lane) & 1) L.stack[rank_below(m)] = lsi_entry<YS>(T, rl, b, (uint32_t) lane);
      sp = __popcll(m);
      if (rn > 64) {
        b = root_box2;
        m = refine(b, __ballot(overlap(b, gx0, gy0, gx1, gy1)));
        if (!stack_has_room(sp, __popcll(m), stack_cap, A.work_counter, kFaultLsiStack, lane)) m = 0;
        if ((m >> lane) & 1) L.stack[sp + rank_below(m)] = lsi_entry<YS>(T, rl, b, 64u + (uint32_t) lane);
        sp += __popcll(m);
      }
      wave_lds_fence();
    }
    if (STATS) tk_head += clock64() - tkg;
    // Order is irrelevant for LSI (nothing prunes), so the stack is consumed two entries at a time
    // with both entries' boxes requested before either is processed: two dependent-load chains in
    // flight per wave instead of one (the kernel is bound by the latency of these loads).
    auto process = [&](uint32_t e, const QBox& b, const uint2& tab) {
      const int lvl = (int) (e >> 28);
      const uint32_t idx = e & 0x0FFFFFFFu;
      const long long tk0 = STATS ? clock64() : 0;
      if (lvl > 1) {
        uint64_t m = refine(b, __ballot(overlap(b, gx0, gy0, gx1, gy1)));
        // (cannot happen for a tree rj_build_lbvh accepted: kStackEntries covers the worst case, rj_device.h)
        if (!stack_has_room(sp, __popcll(m), stack_cap, A.work_counter, kFaultLsiStack, lane)) m = 0;
        if ((m >> lane) & 1) L.stack[sp + rank_below(m)] = lsi_entry<YS>(T, lvl - 1, b, idx * 64 + lane);
        sp += __popcll(m);
        if (STATS) st_nodes++;
        wave_lds_fence();
        if (STATS) tk_node += clock64() - tk0;
      } else {
        // leaf block: 64 base segments sorted by x0, one per lane.  The block's x-bucket table (k_build_leaves) gives
        // the slots [lo, hi) that can overlap the query's x-range -- hi from the bucket of its right end, lo from the
        // bucket of its left end -- instead of a 64-wide cross-lane binary search and a prefix-max fetch per step: four
        // dependent LDS round trips less per visit in a kernel that is bound by exactly such chains.  The block's
        // x-extent (what the table was built on) is re-derived from its boxes: padding slots are empty boxes.
        const uint32_t slot0 = idx * 64;
        if (STATS) st_leaf++;
        // (the order the block is taken by: x, or -- a block taller than wide, pushed as level 0 -- its second one, y)
        const bool ysort = YS && lvl == 0;
        const uint32_t yslot = leaf_perm_of(tab);  // (ysort: the x-order slot of y-rank `lane`)
        const int32_t lx0 = wave_min(ysort ? b.y0 : b.x0), lx1 = wave_max(ysort ? b.y1 : b.x1);
        const int sh = leaf_bucket_shift((uint32_t) (lx1 - lx0));
        const int32_t qa0 = ysort ? qy0 : qx0, qa1 = ysort ? qy1 : qx1;
        const int32_t ca = qa0 > lx0 ? qa0 : lx0, cz = qa1 < lx1 ? qa1 : lx1;  // the query's range on that axis inside the block's
        const bool some = ca <= cz;                                           // (idle lanes: qx0 > qx1)
        const uint32_t bhi = some ? (uint32_t) (cz - lx0) >> sh : 0u, blo = some ? (uint32_t) (ca - lx0) >> sh : 0u;
        const uint32_t hi = ((uint32_t) __builtin_amdgcn_ds_bpermute((int) (bhi >> 2) << 2, (int) tab.x) >> ((bhi & 3u) * 8u)) & 0x7Fu;
        const uint32_t lo = ((uint32_t) __builtin_amdgcn_ds_bpermute((int) (blo >> 2) << 2, (int) tab.y) >> ((blo & 3u) * 8u)) & 0x7Fu;
        int j = some ? (int) hi - 1 : -1;
        const int jlo = some ? (int) lo : 0;
        while (__ballot(j >= jlo)) {
          // (ds_bpermute takes the lane from bits 7:2 of the address; a lane past its range reads some slot, harmlessly)
          const int ja = ysort ? __builtin_amdgcn_ds_bpermute(j << 2, (int) yslot) << 2 : j << 2;
          const int32_t sx0 = __builtin_amdgcn_ds_bpermute(ja, b.x0), sx1 = __builtin_amdgcn_ds_bpermute(ja, b.x1);
          const int32_t sy0 = __builtin_amdgcn_ds_bpermute(ja, b.y0), sy1 = __builtin_amdgcn_ds_bpermute(ja, b.y1);
          // box overlap and "still inside my range" as one sign test
          const bool c = ((qx1 - sx0) | (sx1 - qx0) | (qy1 - sy0) | (sy1 - qy0) | (j - jlo)) >= 0;
          const uint64_t cm = __ballot(c);
          if (STATS) st_box++;
          if (cm) {
            if (c) L.pairs[np + rank_below(cm)] = make_uint2((uint32_t) q, slot0 + (uint32_t) ((ja >> 2) & 63));
            np += __popcll(cm);
            wave_lds_fence();
            if (np >= 64) lsi_drain<STATS>(L, np, nh, 64, A, lane, st_tests);
          }
          j--;
        }
        if (STATS) tk_leaf += clock64() - tk0;
      }
    };
    auto fetch = [&](uint32_t e, QBox& b, uint2& tab) {
      const int lvl = (int) (e >> 28);
      const uint64_t c = (uint64_t) (e & 0x0FFFFFFFu) * 64 + lane;
      // one address for both kinds of entry keeps the loads branch-free (the bucket table is only meaningful for leaves)
      const QBox* src = lvl > 1 ? T.lvl[lvl - 1] : T.box0;
      b = src[c];
      const uint2* tsrc = YS && lvl == 0 ? T.ytab2 : T.xtab;  // (level 0 = a leaf block taken by its y order: lsi_entry)
      tab = tsrc[lvl > 1 ? (uint64_t) lane : c];
    };
    while (sp > 0) {
      const bool two = sp > 1;
      const uint32_t ea = __builtin_amdgcn_readfirstlane(L.stack[sp - 1]);
      const uint32_t eb = __builtin_amdgcn_readfirstlane(L.stack[two ? sp - 2 : sp - 1]);
      sp -= two ? 2 : 1;
      QBox ba, bb2;
      uint2 ta, tb;
      fetch(ea, ba, ta);
      fetch(eb, bb2, tb);
      process(ea, ba, ta);
      if (two) process(eb, bb2, tb);
    }
  }
  }
  if (np > 0) lsi_drain<STATS>(L, np, nh, np, A, lane, st_tests);
  if (nh >= 64) lsi_flush_hits<STATS>(L, nh, 64, A, lane);
  if (nh > 0) lsi_flush_hits<STATS>(L, nh, nh, A, lane);
  if (STATS && lane == 0 && A.stats) {
    const long long tk_total = clock64() - tk_begin;
    atomicAdd(&A.stats[0], st_leaf);
    atomicAdd(&A.stats[1], st_tests);
    atomicAdd(&A.stats[2], st_nodes);
    atomicAdd(&A.stats[3], st_box);
    atomicAdd(&A.stats[4], (unsigned long long) tk_total);
    atomicAdd(&A.stats[5], (unsigned long long) tk_node);
    atomicAdd(&A.stats[6], (unsigned long long) tk_leaf);  // includes the dense predicate phase
    atomicAdd(&A.stats[7], (unsigned long long) tk_head);  // group load + union box + top level
    atomicAdd(&A.stats[10], (unsigned long long) tk_sched);
    atomicAdd(&A.stats[11], st_refined);  // children that overlap the group's box and were tested against every query ...
    atomicAdd(&A.stats[12], st_kept);     // ... and those some query overlaps (pushed)
    atomicMax(&A.stats[9], (unsigned long long) tk_total);
  }
}

// k_lsi with TWO query segments per lane: a wave takes 128 consecutive query positions (lane l: positions l and 64 + l
// of the group) through one traversal -- the pre-filter's verdict, the union box, node expansions, stack traffic and
// the leaf blocks' loads are shared; the in-leaf scan runs per segment set, and a set with nothing in the block's
// x-range skips it.  The kernel is bound by its chain of dependent node fetches per group, so twice the queries per
// chain is most of twice the throughput per wave.  Candidate pairs and hits go through the same LDS buffers
// (they carry the query eid), so everything behind the traversal is k_lsi's.  Requires group_lanes == 64.
template <bool YS>
__device__ __forceinline__ void lsi2_body(const LsiArgs& A) {
  __shared__ LsiWaveLds lds[4];
  const int lane = lane_id();
  const int wib = threadIdx.x >> 6;
  LsiWaveLds& L = lds[wib];
  const uint64_t nq = A.qend - A.qbeg;
  const uint64_t ngroups = (nq + 127) / 128;
  const DeviceBvh& T = A.bvh;
  if (blockIdx.x == 0 && threadIdx.x < 8) A.next_work_counter[threadIdx.x * 32] = 0;
  if (blockIdx.x == 0 && threadIdx.x == 8) *A.next_counter = 0;
  const bool occ_usable = T.occ[(size_t) kOccDim * kOccRowWords] == 0;  // every base segment was rasterised
  const int stack_cap = kStackEntries;
  int np = 0, nh = 0;  // wave-uniform fill of L.pairs / L.hits
  unsigned long long st_tests = 0;
  const uint32_t nchunks = (uint32_t) ((ngroups + A.chunk_groups - 1) / A.chunk_groups);
  int part = blockIdx.x & 7, tried = 0;
  for (;;) {
  uint32_t chunk = 0;
  if (!next_chunk(A.work_counter, nchunks, part, tried, lane, chunk)) break;
  const uint64_t g_begin = (uint64_t) chunk * A.chunk_groups;
  const uint64_t g_end = g_begin + A.chunk_groups < ngroups ? g_begin + A.chunk_groups : ngroups;
  for (uint64_t g = g_begin; g < g_end; g++) {
    uint64_t q[2];
    bool near[2];
    int32_t qx0[2], qy0[2], qx1[2], qy1[2];
    // The loads a group starts with go out in three batches, not seven: both sets' cell codes; both sets' bitmap windows;
    // then -- only where the pre-filter lets the group through -- both sets' segments beside the root's boxes.  Nothing is
    // loaded under a per-lane branch (a position past the end reads the last one's): left in `if (valid)` regions every
    // load was waited for on its own, and this kernel is bound by exactly such chains (SQ_WAIT_ANY 73 % of its wave-cycles).
    uint32_t code[2];
#pragma unroll
    for (int p = 0; p < 2; p++) {
      const uint64_t qi = g * 128 + (uint64_t) p * 64 + lane;
      near[p] = qi < nq;
      const uint64_t qc = near[p] ? qi : nq - 1;
      q[p] = A.qbeg + (A.order ? A.order[qc] : qc);
    }
    if (occ_usable) {
#pragma unroll
      for (int p = 0; p < 2; p++) code[p] = __builtin_nontemporal_load(A.qcode + q[p]);
      asm volatile("" : "+v"(code[0]), "+v"(code[1]));  // (both codes requested before either is used: the scheduler would otherwise put set 1's behind set 0's windows)
      OccWindows win[2];
#pragma unroll
      for (int p = 0; p < 2; p++) win[p] = occ_fetch_code(T.occ, code[p]);
      asm volatile("" : "+v"(win[0].a), "+v"(win[0].b), "+v"(win[1].a), "+v"(win[1].b));
#pragma unroll
      for (int p = 0; p < 2; p++) near[p] = near[p] && occ_verdict_code(win[p], code[p]);
    }
    if (!__ballot(near[0] || near[1])) continue;  // both halves of the group are clear of the base map
    const int rl = T.lsi_root;                 // (the level the traversal starts at: DeviceBvh::lsi_root)
    const uint32_t rn = T.nlvl[rl];            // (<= 128 nodes)
    const QBox root_box = T.lvl[rl][lane];
    const QBox root_box2 = T.lvl[rl][(rn > 64 ? 64 : 0) + lane];
    {
      // (only the lanes the pre-filter let through read their segment -- two thirds of a passing group's lanes are clear of
      //  the base map too, and unmasked these loads doubled the kernel's traffic, 0.46 -> 0.93 GB -- but both sets' loads
      //  stand before the first use of either: one round trip)
      Seg sg[2];
#pragma unroll
      for (int p = 0; p < 2; p++) {
        sg[p] = Seg{0, 0, 0, 0};
        if (near[p]) sg[p] = A.qseg[q[p]];
      }
#pragma unroll
      for (int p = 0; p < 2; p++) {
        const Seg& s = sg[p];
        qx0[p] = near[p] ? quant(s.x1 < s.x2 ? s.x1 : s.x2) : kEmptyMin;
        qx1[p] = near[p] ? quant(s.x1 < s.x2 ? s.x2 : s.x1) : kEmptyMax;
        qy0[p] = near[p] ? quant(s.y1 < s.y2 ? s.y1 : s.y2) : kEmptyMin;
        qy1[p] = near[p] ? quant(s.y1 < s.y2 ? s.y2 : s.y1) : kEmptyMax;
      }
    }
    const int32_t gx0 = wave_min(qx0[0] < qx0[1] ? qx0[0] : qx0[1]), gy0 = wave_min(qy0[0] < qy0[1] ? qy0[0] : qy0[1]);
    const int32_t gx1 = wave_max(qx1[0] > qx1[1] ? qx1[0] : qx1[1]), gy1 = wave_max(qy1[0] > qy1[1] ? qy1[0] : qy1[1]);
    auto refine = [&](const QBox& b, uint64_t um) -> uint64_t {
      uint64_t keep = 0;
      while (um) {
        const int c = __builtin_ctzll(um);
        um &= um - 1;
        const int32_t cx0 = bcast(b.x0, c), cy0 = bcast(b.y0, c);
        const int32_t cx1 = bcast(b.x1, c), cy1 = bcast(b.y1, c);
        if (__ballot(boxes_overlap(qx0[0], qy0[0], qx1[0], qy1[0], cx0, cy0, cx1, cy1) ||
                     boxes_overlap(qx0[1], qy0[1], qx1[1], qy1[1], cx0, cy0, cx1, cy1)))
          keep |= 1ull << c;
      }
      return keep;
    };
    int sp = 0;
    {  // the starting level: <= 64 nodes, one per lane -- or <= 128, the second 64 likewise
      QBox b = root_box;
      uint64_t m = refine(b, __ballot(overlap(b, gx0, gy0, gx1, gy1)));
      if (!stack_has_room(0, __popcll(m), stack_cap, A.work_counter, kFaultLsiStack, lane)) m = 0;
      if ((m >> lane) & 1) L.stack[rank_below(m)] = lsi_entry<YS>(T, rl, b, (uint32_t) lane);
      sp = __popcll(m);
      if (rn > 64) {
        b = root_box2;
        m = refine(b, __ballot(overlap(b, gx0, gy0, gx1, gy1)));
        if (!stack_has_room(sp, __popcll(m), stack_cap, A.work_counter, kFaultLsiStack, lane)) m = 0;
        if ((m >> lane) & 1) L.stack[sp + rank_below(m)] = lsi_entry<YS>(T, rl, b, 64u + (uint32_t) lane);
        sp += __popcll(m);
      }
      wave_lds_fence();
    }
    auto process = [&](uint32_t e, const QBox& b, const uint2& tab) {
      const int lvl = (int) (e >> 28);
      const uint32_t idx = e & 0x0FFFFFFFu;
      if (lvl > 1) {
        uint64_t m = refine(b, __ballot(overlap(b, gx0, gy0, gx1, gy1)));
        if (!stack_has_room(sp, __popcll(m), stack_cap, A.work_counter, kFaultLsiStack, lane)) m = 0;
        if ((m >> lane) & 1) L.stack[sp + rank_below(m)] = lsi_entry<YS>(T, lvl - 1, b, idx * 64 + lane);
        sp += __popcll(m);
        wave_lds_fence();
      } else {
        const uint32_t slot0 = idx * 64;
        const bool ysort = YS && lvl == 0;  // (the order the block is taken by: x, or -- taller than wide, pushed as level 0 -- y)
        const uint32_t yslot = leaf_perm_of(tab);
        const int32_t lx0 = wave_min(ysort ? b.y0 : b.x0), lx1 = wave_max(ysort ? b.y1 : b.x1);
        const int sh = leaf_bucket_shift((uint32_t) (lx1 - lx0));
#pragma unroll
        for (int p = 0; p < 2; p++) {
          const int32_t qa0 = ysort ? qy0[p] : qx0[p], qa1 = ysort ? qy1[p] : qx1[p];
          const int32_t ca = qa0 > lx0 ? qa0 : lx0, cz = qa1 < lx1 ? qa1 : lx1;
          const bool some = ca <= cz;
          if (!__ballot(some)) continue;  // none of this set's segments reaches into the block's range on its sort axis
          const uint32_t bhi = some ? (uint32_t) (cz - lx0) >> sh : 0u, blo = some ? (uint32_t) (ca - lx0) >> sh : 0u;
          const uint32_t hi = ((uint32_t) __builtin_amdgcn_ds_bpermute((int) (bhi >> 2) << 2, (int) tab.x) >> ((bhi & 3u) * 8u)) & 0x7Fu;
          const uint32_t lo = ((uint32_t) __builtin_amdgcn_ds_bpermute((int) (blo >> 2) << 2, (int) tab.y) >> ((blo & 3u) * 8u)) & 0x7Fu;
          int j = some ? (int) hi - 1 : -1;
          const int jlo = some ? (int) lo : 0;
          while (__ballot(j >= jlo)) {
            const int ja = ysort ? __builtin_amdgcn_ds_bpermute(j << 2, (int) yslot) << 2 : j << 2;
            const int32_t sx0 = __builtin_amdgcn_ds_bpermute(ja, b.x0), sx1 = __builtin_amdgcn_ds_bpermute(ja, b.x1);
            const int32_t sy0 = __builtin_amdgcn_ds_bpermute(ja, b.y0), sy1 = __builtin_amdgcn_ds_bpermute(ja, b.y1);
            const bool c = ((qx1[p] - sx0) | (sx1 - qx0[p]) | (qy1[p] - sy0) | (sy1 - qy0[p]) | (j - jlo)) >= 0;
            const uint64_t cm = __ballot(c);
            if (cm) {
              if (c) L.pairs[np + rank_below(cm)] = make_uint2((uint32_t) q[p], slot0 + (uint32_t) ((ja >> 2) & 63));
              np += __popcll(cm);
              wave_lds_fence();
              if (np >= 64) lsi_drain<false>(L, np, nh, 64, A, lane, st_tests);
            }
            j--;
          }
        }
      }
    };
    auto fetch = [&](uint32_t e, QBox& b, uint2& tab) {
      const int lvl = (int) (e >> 28);
      const uint64_t c = (uint64_t) (e & 0x0FFFFFFFu) * 64 + lane;
      const QBox* src = lvl > 1 ? T.lvl[lvl - 1] : T.box0;
      b = src[c];
      const uint2* tsrc = YS && lvl == 0 ? T.ytab2 : T.xtab;  // (level 0 = a leaf block taken by its y order: lsi_entry)
      tab = tsrc[lvl > 1 ? (uint64_t) lane : c];
    };
    while (sp > 0) {
      const bool two = sp > 1;
      const uint32_t ea = __builtin_amdgcn_readfirstlane(L.stack[sp - 1]);
      const uint32_t eb = __builtin_amdgcn_readfirstlane(L.stack[two ? sp - 2 : sp - 1]);
      sp -= two ? 2 : 1;
      QBox ba, bb2;
      uint2 ta, tb;
      fetch(ea, ba, ta);
      fetch(eb, bb2, tb);
      process(ea, ba, ta);
      if (two) process(eb, bb2, tb);
    }
  }
  }
  if (np > 0) lsi_drain<false>(L, np, nh, np, A, lane, st_tests);
  if (nh >= 64) lsi_flush_hits<false>(L, nh, 64, A, lane);
  if (nh > 0) lsi_flush_hits<false>(L, nh, nh, A, lane);
}

// The kernels of the two bodies above: with the second order of steep leaf blocks (trees that have ytab2), and without --
// maps of closed rings, "leaf_ysort" 0 -- where the compiler drops every trace of it.
template <bool STATS>
__global__ __launch_bounds__(256, 6) __attribute__((amdgpu_num_sgpr(96))) void k_lsi(LsiArgs A) { lsi_body<STATS, true>(A); }
template <bool STATS>
__global__ __launch_bounds__(256, 6) __attribute__((amdgpu_num_sgpr(96))) void k_lsix(LsiArgs A) { lsi_body<STATS, false>(A); }
__global__ __launch_bounds__(256, 6) __attribute__((amdgpu_num_sgpr(96))) void k_lsi2(LsiArgs A) { lsi2_body<true>(A); }
__global__ __launch_bounds__(256, 6) __attribute__((amdgpu_num_sgpr(96))) void k_lsi2x(LsiArgs A) { lsi2_body<false>(A); }

// =============================================================================================
// LSI intersection points (per hit only): rational point, clamp, narrowing store
// =============================================================================================
// n_dev (nullable): the result count as the LSI kernel left it on the device -- the records of a
// query are then produced on the stream with no host round trip in between (min(*n_dev, n) pairs).
// Two kernels: k_lsi_points decides a pair's stored coordinates from one exact floor division per coordinate
// (lsi_stored_fast: 98-99 % of the pairs at map-like magnitudes, every exact hit on a shared vertex among them) and
// appends the pairs it declines -- a coordinate within |v| 2^-51 of an integer without being one -- to a list;
// k_lsi_points_gcd, right behind it, gives those the simplified rational itself (128-bit gcd: ~20x the instructions,
// divergent loops, 145 VGPRs) with every lane busy.  One kernel doing both held 62 lanes of nearly every wave up for
// the one or two that needed the gcd (a wave sees too few pairs to fill a queue of its own), and kept the fast leg at
// 3 waves per SIMD.  `slow_list` == nullptr: k_lsi_points_gcd alone over all pairs (n >= 2^32, or no list memory).
__global__ __launch_bounds__(256) void k_lsi_points_gcd(const Seg* __restrict__ seg0, const Seg* __restrict__ seg1,
                                                        const uint32_t* __restrict__ pairs, uint64_t n,
                                                        const unsigned long long* __restrict__ n_dev,
                                                        const uint32_t* __restrict__ slow_list,
                                                        const unsigned long long* __restrict__ slow_count,
                                                        unsigned long long* __restrict__ count_hint,
                                                        XsectRec* __restrict__ out) {
  if (n_dev) {
    const unsigned long long found = *n_dev;
    n = found < n ? found : n;
  }
  // (the host picks the one- or two-kernel form for the next query by this count: mapped host memory, a plain store)
  if (count_hint && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(count_hint, (unsigned long long) n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  uint64_t m = n;
  if (slow_list) {
    const unsigned long long listed = *slow_count;
    m = listed < n ? listed : n;
  }
  for (uint64_t j = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; j < m; j += (uint64_t) gridDim.x * blockDim.x) {
    const uint64_t i = slow_list ? slow_list[j] : j;
    const uint32_t e0 = pairs[2 * i], e1 = pairs[2 * i + 1];
    const Seg s1 = seg0[e0], s2 = seg1[e1];
    Rat x, y;
    lsi_point(s1, make_eqn(s1), s2, make_eqn(s2), &x, &y);
    XsectRec r;
    r.x_num = (int64_t) rat_to_double(x);
    r.x_den = 1;
    r.y_num = (int64_t) rat_to_double(y);
    r.y_den = 1;
    r.eid0 = e0;
    r.eid1 = e1;
    r.mid = -1;
    r.pad = 0;
    out[i] = r;
  }
}

__global__ __launch_bounds__(256) void k_lsi_points(const Seg* __restrict__ seg0,
                                                    const Seg* __restrict__ seg1,
                                                    const uint32_t* __restrict__ pairs, uint64_t n,
                                                    const unsigned long long* __restrict__ n_dev,
                                                    uint32_t* __restrict__ slow_list,
                                                    unsigned long long* __restrict__ slow_count,
                                                    unsigned long long* __restrict__ next_slow_count,
                                                    unsigned long long* __restrict__ count_hint,
                                                    XsectRec* __restrict__ out) {
  __shared__ uint32_t slowq[4][128];
  if (blockIdx.x == 0 && threadIdx.x == 0) *next_slow_count = 0;  // (the next query's list; see k_lsi's counters)
  if (n_dev) {
    const unsigned long long found = *n_dev;
    n = found < n ? found : n;
  }
  if (count_hint && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(count_hint, (unsigned long long) n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const int lane = lane_id();
  const int wib = threadIdx.x >> 6;
  int nq = 0;  // wave-uniform fill of this wave's queue
  auto flush = [&](int count) {  // the first `count` entries of the queue go to the list: one atomic per flush
    unsigned long long at = 0;
    if (lane == 0) at = atomicAdd(slow_count, (unsigned long long) count);
    at = ((unsigned long long) __builtin_amdgcn_readfirstlane((uint32_t) (at >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t) at);
    if (lane < count) slow_list[at + lane] = slowq[wib][lane];
  };
  for (uint64_t base = ((uint64_t) blockIdx.x * 4 + wib) * 64; base < n; base += (uint64_t) gridDim.x * 256) {
    const uint64_t i = base + lane;
    bool declined = false;
    if (i < n) {
      const uint2 pr = reinterpret_cast<const uint2*>(pairs)[i];
      const Seg s1 = seg0[pr.x], s2 = seg1[pr.y];
      XsectRec r;
      if (lsi_stored_fast(s1, s2, &r.x_num, &r.y_num)) {
        r.x_den = 1;
        r.y_den = 1;
        r.eid0 = pr.x;
        r.eid1 = pr.y;
        r.mid = -1;
        r.pad = 0;
        out[i] = r;
      } else {
        declined = true;
      }
    }
    const uint64_t dm = __ballot(declined);
    if (dm) {
      if (declined) slowq[wib][nq + rank_below(dm)] = (uint32_t) i;  // (n < 2^32 on this path)
      nq += __popcll(dm);
      wave_lds_fence();
      if (nq >= 64) {
        flush(64);
        wave_lds_fence();
        if (lane < nq - 64) slowq[wib][lane] = slowq[wib][64 + lane];
        nq -= 64;
        wave_lds_fence();
      }
    }
  }
  if (nq) flush(nq);
}


// =============================================================================================
// Overlay support (src/app/map_overlay_lbvh.h:109-265, ComputeOutputPolygons): order the
// intersections of every edge of map `im` along the edge, take the mid-points of consecutive
// intersections (they are located in the other map by an ordinary PIP query) and store the face
// found for each mid-point in the record that precedes it.
// =============================================================================================
__global__ __launch_bounds__(256) void k_xsect_keys(const XsectRec* __restrict__ rec, uint64_t n, int im,
                                                    uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
    const uint32_t mine = im ? rec[i].eid1 : rec[i].eid0, other = im ? rec[i].eid0 : rec[i].eid1;
    keys[i] = ((uint64_t) mine << 32) | other;
    vals[i] = (uint32_t) i;
  }
}

__global__ __launch_bounds__(256) void k_xsect_gather(const XsectRec* __restrict__ in, const uint32_t* __restrict__ order,
                                                      uint64_t n, XsectRec* __restrict__ out) {
  for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x)
    out[i] = in[order[i]];
}

// squared distance of the stored (truncated) intersection point from p1, exact in int128
// (map_overlay_lbvh.h:204-213: SQ(x - p1.x) + SQ(y - p1.y) on rational<__int128> with denominators 1)
__device__ __forceinline__ u128 xsect_dist2(const XsectRec& r, int64_t px, int64_t py) {
  const i128 dx = (i128) r.x_num - px, dy = (i128) r.y_num - py;
  return (u128) (dx * dx) + (u128) (dy * dy);
}

// one thread per run of equal eid[im] (run starts are detected in place): stable insertion sort by
// distance from the edge's p1 (runs are tiny; ties keep the other map's eid ascending), then the
// mid-points.  Run ends get their own point as a dummy mid-point and keep mid = DONTKNOW.
__global__ __launch_bounds__(256) void k_xsect_order_runs(XsectRec* __restrict__ rec, uint64_t n, int im,
                                                          const Seg* __restrict__ seg_im,
                                                          int64_t* __restrict__ midpts) {
  for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
    const uint32_t e = im ? rec[i].eid1 : rec[i].eid0;
    if (i > 0 && (im ? rec[i - 1].eid1 : rec[i - 1].eid0) == e) continue;  // not a run start
    uint64_t end = i + 1;
    while (end < n && (im ? rec[end].eid1 : rec[end].eid0) == e) end++;
    const Seg s = seg_im[e];
    for (uint64_t a = i + 1; a < end; a++) {  // insertion sort of [i, end)
      const XsectRec cur = rec[a];
      const u128 dc = xsect_dist2(cur, s.x1, s.y1);
      uint64_t b = a;
      while (b > i && xsect_dist2(rec[b - 1], s.x1, s.y1) > dc) {
        rec[b] = rec[b - 1];
        b--;
      }
      rec[b] = cur;
    }
    for (uint64_t a = i; a < end; a++) {
      int64_t mx = rec[a].x_num, my = rec[a].y_num;
      if (a + 1 < end) {  // x1 + (x2 - x1) / 2 as a rational, then the narrowing store: trunc((x1 + x2) / 2)
        mx = (int64_t) (((i128) rec[a].x_num + rec[a + 1].x_num) / 2);
        my = (int64_t) (((i128) rec[a].y_num + rec[a + 1].y_num) / 2);
      }
      midpts[2 * a] = mx;
      midpts[2 * a + 1] = my;
      rec[a].mid = -1;  // DONTKNOW
    }
  }
}

__global__ __launch_bounds__(256) void k_xsect_set_mid(XsectRec* __restrict__ rec, uint64_t n, int im,
                                                       const int32_t* __restrict__ face) {
  for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i + 1 < n; i += (uint64_t) gridDim.x * blockDim.x) {
    const uint32_t a = im ? rec[i].eid1 : rec[i].eid0, b = im ? rec[i + 1].eid1 : rec[i + 1].eid0;
    if (a == b) rec[i].mid = face[i];
  }
}

__global__ __launch_bounds__(256) void k_swap_halves(uint64_t* __restrict__ v, uint64_t n) {
  for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n;
       i += (uint64_t) gridDim.x * blockDim.x) {
    uint64_t x = v[i];
    v[i] = (x << 32) | (x >> 32);
  }
}

// =============================================================================================
// PIP: upward ray through the same tree.
//   * pruning is integer-only and sound: a base segment whose quantised box strictly contains the
//     point's x and lies strictly above the point is a CERTAIN hit (exact x-range test passes,
//     diff_y < 0 with margin), so its box top bounds the lane's best from above;
//   * candidates go to per-lane lists in LDS ([k][lane]: bank = lane, conflict-free) and are
//     evaluated exactly by their own lane -- no cross-lane merge;
//   * leaf blocks are x0-sorted: a lane finds its segments by a cross-lane binary search and a
//     backward scan bounded by the prefix max of x1 (k_build_leaves);
//   * the kernel is VALU-issue bound: the exact test computes two 128-bit
//     products, and the slope only on ties.
// =============================================================================================
constexpr int kPipList = 6;     // candidate slots per lane between two exact-evaluation rounds
#ifndef RJ_PIP_WAVES
#define RJ_PIP_WAVES 4
#endif
constexpr int kPipWaves = RJ_PIP_WAVES;  // waves per block = the waves that share their chunks' rests (next_group)
constexpr int kPipRefineAbove = 16;  // per-lane check at push time only when more children than this pass the group test
// (kPipStack: rj_device.h -- 16-byte entries; with the lists, 6656 B per wave = 6 blocks per CU)

// A stack entry carries the node's y0 and x-range, so a stale entry (every lane under it has since
// found something lower) is dropped at pop time without touching memory.
struct PipWaveLds {
  uint4 stack[kPipStack];  // {level<<28 | index, y0, x0, x1}
  uint32_t cand[kPipList][64];
};

// (a device function: k_pip runs it on every block, k_pip_exact on its first few -- the overflowed lists of the walk)
template <bool STATS>
__device__ __forceinline__ void pip_locate(const PipArgs& A, const uint32_t bid, const PipRestArgs& Q, unsigned long long* const stats) {
  // (Q: the query set and the scheduler block -- k_pip passes its own arguments' -- so that k_pip_exact's first blocks
  //  need no private copy of the arguments: the tree's level arrays are indexed dynamically and would live in scratch)
  __shared__ PipWaveLds lds[kPipWaves];
  const int lane = lane_id();
  const int wib = threadIdx.x >> 6;
  PipWaveLds& L = lds[wib];
  const uint32_t GL = Q.group_lanes;  // points per wave: 64, or fewer for small query sets
  // (behind k_pip_walk: the queries are the points it left over, counted on the device)
  uint64_t nq = A.n;
  if (Q.n_dev) {
    const unsigned long long left = __hip_atomic_load(Q.n_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    nq = left < nq ? left : nq;
    // (the host sizes the next launch's grid by this count: mapped host memory, a plain store)
    if (Q.rest_count && bid == 0 && threadIdx.x == 0) __hip_atomic_store(Q.rest_count, left, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const uint64_t ngroups = (nq + GL - 1) / GL;
  const DeviceBvh& T = A.bvh;
  const uint32_t* const sky = (T.sky && T.sky[kSkyBuckets] == 0u) ? T.sky : nullptr;  // (exhaustive, or not used)
  const int qm = A.query_map_id;
  const int stack_cap = STATS && A.stack_cap < kPipStack ? A.stack_cap : kPipStack;  // (lowered only by tests of the fault path, instrumented kernel)
  unsigned long long st_leaf = 0, st_tests = 0, st_nodes = 0, st_box = 0;
  unsigned long long st_stale = 0, st_leaf_nocand = 0, st_leaf_lanes = 0;
  long long tk_drain = 0, tk_leaf = 0, tk_node = 0, tk_rounds = 0, tk_sched = 0, tk_head = 0, tk_tail = 0;  // STATS: cycle stamps
  const long long tk_begin = STATS ? clock64() : 0;

  const uint32_t nchunks = (uint32_t) ((ngroups + Q.chunk_groups - 1) / Q.chunk_groups);
  int part = bid & 7, tried = 0;
  __shared__ unsigned long long ranges[kPipWaves];  // per wave {end : next}: the unstarted rest of its chunk
  if (threadIdx.x < kPipWaves) ranges[threadIdx.x] = 0;
  if (bid == 0 && threadIdx.x < 8) Q.next_work_counter[threadIdx.x * 32] = 0;  // (see k_lsi; any block size has these threads)
  __syncthreads();
  for (;;) {  // XCD-aware dynamic chunked scheduling, see next_chunk / next_group
  {
    uint32_t g32 = 0;
    const long long tks = STATS ? clock64() : 0;
    if (!next_group<kPipWaves>(ranges, wib, Q.work_counter, nchunks, Q.chunk_groups, ngroups, part, tried, lane, g32)) break;
    if (STATS) tk_sched += clock64() - tks;
    const uint64_t g = g32;
    const long long tkg = STATS ? clock64() : 0;
    const uint64_t ipos = g * GL + lane;  // position in the (possibly Morton-sorted) query order
    const bool valid = (uint32_t) lane < GL && ipos < nq;
    const uint64_t ip = Q.order ? (valid ? Q.order[ipos] : 0) : ipos;
    // The traversal works on the quantised point; the exact coordinates are read again (an L2 hit)
    // by the few lanes whose candidates need the exact arithmetic -- 0.2 per group on the headline
    // workload -- instead of occupying four of the kernel's 80 registers throughout.
    int32_t qx = 0, qy = 0;
    if (valid) {
      qx = quant(A.pts[2 * ip]);
      qy = quant(A.pts[2 * ip + 1]);
    }
    // (a point above the skyline of the map has nothing above it: a miss, without a traversal -- it sits the group out)
    const bool live = valid && ray_has_sky(sky, qx, qy);
    const int32_t qym1 = qy > 0 ? qy - 1 : 0;
    const int32_t gx0 = wave_min(live ? qx : kEmptyMin);
    const int32_t gx1 = wave_max(live ? qx : kEmptyMax);
    const int32_t gy0 = wave_min(live ? qy : kEmptyMin);
    double best_yy = __builtin_inf();
    uint32_t best_slot = 0xFFFFFFFFu;  // sorted slot of the best edge so far (eid/face are looked up at the end)
    int32_t qbest = live ? 0x7FFFFFFF : -1;  // sound quantised upper bound of this lane's answer
    int32_t gbest = 0x7FFFFFFF;               // wave max of qbest
    int cnt = 0;                              // this lane's candidate-list fill
    bool sure1 = false;  // the list holds exactly one candidate and it is a certain hit
    int32_t sure_y0 = 0;  // ... whose box starts here

    // every lane evaluates its own candidate list exactly (pip.h:36-95), then clears it
    auto evaluate = [&](bool final_round) {
      const long long tk0 = STATS ? clock64() : 0;
      // A lane whose ONLY candidate of the whole traversal is a certain hit needs no arithmetic:
      // every other edge over this x was pruned because it starts above that candidate's box top,
      // so the candidate is the answer (the common case for an x-monotone polyline overhead).
      if (final_round && cnt == 1 && sure1 && best_slot == 0xFFFFFFFFu) {
        best_slot = L.cand[0][lane];
        cnt = 0;
      }
      const int maxc = wave_max(cnt);
      int64_t px = 0, py = 0;
      if (cnt > 0) {
        px = A.pts[2 * ip];
        py = A.pts[2 * ip + 1];
      }
      for (int r = 0; r < maxc; r++) {
        if (r < cnt) {
          const uint32_t slot = L.cand[r][lane];
          const Seg bs = T.sseg[slot];
          double yy;
          if (pip_eval_y(bs, px, py, qm, &yy)) {
            bool better = yy < best_yy;
            if (yy == best_yy && best_slot != 0xFFFFFFFFu) {  // tie: slope rule, then eid (rare: slopes and eids are looked up now)
              const Seg cur = T.sseg[best_slot];
              better = pip_better(yy, pip_slope(bs), T.seid[slot], best_yy, pip_slope(cur), T.seid[best_slot], qm);
            }
            if (better) {
              best_yy = yy; best_slot = slot;
            }
          }
        }
        if (STATS) tk_rounds++;
      }
      if (STATS) st_tests += (unsigned long long) __popcll(__ballot(cnt > 0));
      cnt = 0;
      if (final_round) {  // nothing left to prune
        if (STATS) tk_drain += clock64() - tk0;
        return;
      }
      if (best_slot != 0xFFFFFFFFu && best_yy < __builtin_inf()) {
        // exact best known: tighten the integer bound (conservative: +1 quantum)
        double t = (best_yy + (double) kCoordOffset) * (1.0 / 65536.0);
        int32_t qb = t < 2147483000.0 ? (t < -1.0 ? -1 : (int32_t) t + 1) : 0x7FFFFFFF;
        qbest = qb < qbest ? qb : qbest;
      }
      gbest = wave_max(qbest);
      if (STATS) tk_drain += clock64() - tk0;
    };

    // Children are normally pushed on the cheap group-level test alone: the per-lane test happens
    // when an entry is popped, and by then many have gone stale and die in the bulk sweep without
    // costing a pop.  Only when the group test lets MANY children through (a sparse or scattered
    // group whose box covers far more than its points) is each one checked against the lanes first.
    auto refine_if_many = [&](const QBox& b, uint64_t um) -> uint64_t {
      if (__popcll(um) <= kPipRefineAbove) return um;
      uint64_t keep = 0;
      while (um) {
        const int c = __builtin_ctzll(um);
        um &= um - 1;
        const int32_t cx0 = bcast(b.x0, c), cy0 = bcast(b.y0, c);
        const int32_t cx1 = bcast(b.x1, c), cy1 = bcast(b.y1, c);
        if (__ballot(ray_can_hit(qx, qym1, qbest, cx0, cy0, cx1, cy1))) keep |= 1ull << c;
      }
      return keep;
    };
    // Children are pushed so that the lowest one pops first (front to back for an upward ray):
    // position = number of pushed siblings that lie higher, from the precomputed sibling order.
    int sp = 0;
    {
      QBox b = T.lvl[T.top][lane];
      const uint64_t higher = sibling_order(T, T.top)[lane];  // (requested with the box: one latency, not two)
      uint64_t m = refine_if_many(b, __ballot(b.x0 <= gx1 && gx0 <= b.x1 && b.y1 >= gy0 - 1));
      if (!stack_has_room(0, __popcll(m), stack_cap, Q.work_counter, kFaultPipStack, lane)) m = 0;
      const int n = __popcll(m);
      const int at = __popcll(m & higher);
      if ((m >> lane) & 1)
        L.stack[at] = make_uint4(((uint32_t) T.top << 28) | (uint32_t) lane, (uint32_t) b.y0, (uint32_t) b.x0, (uint32_t) b.x1);
      sp = n;
      wave_lds_fence();
    }
    if (STATS) tk_head += clock64() - tkg;
    while (sp > 0) {
      const uint4 ent = L.stack[sp - 1];
      --sp;
      const uint32_t e = __builtin_amdgcn_readfirstlane(ent.x);
      const int32_t ey0 = (int32_t) __builtin_amdgcn_readfirstlane(ent.y);
      const int32_t ex0 = (int32_t) __builtin_amdgcn_readfirstlane(ent.z), ex1 = (int32_t) __builtin_amdgcn_readfirstlane(ent.w);
      // stale?  (every lane under this node's x-range has a bound below it by now)
      const bool want = ((qx - ex0) | (ex1 - qx) | (qbest - ey0)) >= 0;
      if (!__ballot(want)) {
        if (STATS) st_stale++;
        continue;
      }
      const int lvl = (int) (e >> 28);
      const uint32_t idx = e & 0x0FFFFFFFu;
      if (lvl > 1) {
        const long long tk0 = STATS ? clock64() : 0;
        QBox b = T.lvl[lvl - 1][(uint64_t) idx * 64 + lane];
        const uint64_t higher = sibling_order(T, lvl - 1)[(uint64_t) idx * 64 + lane];  // (both loads in flight together)
        uint64_t m = refine_if_many(b, __ballot(b.x0 <= gx1 && gx0 <= b.x1 && b.y1 >= gy0 - 1 && b.y0 <= gbest));
        if (!stack_has_room(sp, __popcll(m), stack_cap, Q.work_counter, kFaultPipStack, lane)) m = 0;
        const int n = __popcll(m);
        const int at = __popcll(m & higher);
        if ((m >> lane) & 1)
          L.stack[sp + at] =
              make_uint4(((uint32_t) (lvl - 1) << 28) | (idx * 64 + lane), (uint32_t) b.y0, (uint32_t) b.x0, (uint32_t) b.x1);
        sp += n;
        if (STATS) st_nodes++;
        wave_lds_fence();
        if (STATS) tk_node += clock64() - tk0;
      } else {
        const long long tk0 = STATS ? clock64() : 0;
        const long long tkd0 = tk_drain;
        const uint32_t slot0 = idx * 64;
        const QBox bb = T.box0[(uint64_t) slot0 + lane];  // one base segment per lane, sorted by x0
        const int32_t pm = T.pmx1[(uint64_t) slot0 + lane];
        if (STATS) st_leaf++;
        // (lanes whose ray cannot use this block any more, or never could, sit the visit out: `want`)
        const int32_t qbest_before = qbest;
        const int ub = wave_upper_bound(bb.x0, qx);  // (shuffles: executed by every lane)
        int j = want ? ub - 1 : -1;
        const int cnt_before = cnt;
        if (STATS) st_leaf_lanes += (unsigned long long) __popcll(__ballot(want));
        for (;;) {
          const int jj = j < 0 ? 0 : j;
          const int32_t pmj = __shfl(pm, jj, 64);
          const bool act = j >= 0 && pmj >= qx;
          if (!__ballot(act)) break;
          const int32_t sx0 = __shfl(bb.x0, jj, 64), sx1 = __shfl(bb.x1, jj, 64);
          const int32_t sy0 = __shfl(bb.y0, jj, 64), sy1 = __shfl(bb.y1, jj, 64);
          if (act && ray_can_hit(qx, qym1, qbest, sx0, sy0, sx1, sy1)) {
            // certain hit (strictly inside in x, strictly above) => its box top bounds the answer
            const bool certain = sx0 < qx && qx < sx1 && sy0 > qy;
            // A certain hit that ends below the start of the one certain hit held so far replaces it
            // (that one is certainly higher): the lane keeps ONE candidate and skips the arithmetic
            // at the end, whatever order the blocks were visited in.
            const bool replace = certain && sure1 && cnt == 1 && sy1 < sure_y0;  // (sure1 may be stale once an evaluation has emptied the list)
            L.cand[replace ? 0 : cnt][lane] = slot0 + (uint32_t) jj;
            sure1 = replace || (cnt == 0 && certain);
            sure_y0 = sure1 ? sy0 : sure_y0;
            cnt += replace ? 0 : 1;
            if (certain && sy1 < qbest - 1) qbest = sy1 + 1;
          }
          if (STATS) st_box++;
          if (__ballot(cnt >= kPipList)) evaluate(false);
          j--;
        }
        if (STATS && !__ballot(cnt != cnt_before)) st_leaf_nocand++;
        const int32_t gbest_before = gbest;
        if (__ballot(qbest != qbest_before)) gbest = wave_max(qbest);  // (a visit without a certain hit changes no bound)
        if (gbest < gbest_before && sp > 1) {
          // The bound of the whole group dropped: sweep the stack once, 64 entries per pass, and
          // drop every entry that starts above it (order preserved).  One pass replaces a dozen
          // one-at-a-time stale pops.
          int kept = 0;
          for (int base = 0; base < sp; base += 64) {
            const int i = base + lane;
            const bool have = i < sp;
            uint4 en = make_uint4(0, 0, 0, 0);
            if (have) en = L.stack[i];
            const bool alive = have && (int32_t) en.y <= gbest;
            const uint64_t am = __ballot(alive);
            wave_lds_fence();
            if (alive) L.stack[kept + rank_below(am)] = en;
            kept += __popcll(am);
          }
          if (STATS) st_stale += (unsigned long long) (sp - kept);
          sp = kept;
          wave_lds_fence();
        }
        if (STATS) tk_leaf += (clock64() - tk0) - (tk_drain - tkd0);
      }
    }
    const long long tkt = STATS ? clock64() : 0;
    evaluate(true);
    if (valid) {
      const bool hit = best_slot != 0xFFFFFFFFu;
      __builtin_nontemporal_store(hit ? T.seid[best_slot] : 0xFFFFFFFFu, A.closest + ip);
      // face below the hit edge (precomputed per sorted slot at build time), EXTERIOR_FACE_ID on a miss
      if (A.face) __builtin_nontemporal_store(hit ? T.sface[best_slot] : 0, A.face + ip);
    }
    if (STATS) tk_tail += clock64() - tkt;
  }
  }
  if (STATS && lane == 0 && stats) {
    const long long tk_total = clock64() - tk_begin;
    atomicAdd(&stats[0], st_leaf);
    atomicAdd(&stats[1], st_tests);
    atomicAdd(&stats[2], st_nodes);
    atomicAdd(&stats[3], st_box);
    atomicAdd(&stats[4], (unsigned long long) tk_total);
    atomicAdd(&stats[5], (unsigned long long) tk_node);
    atomicAdd(&stats[6], (unsigned long long) tk_leaf);
    atomicAdd(&stats[7], (unsigned long long) tk_drain);
    atomicAdd(&stats[8], (unsigned long long) tk_rounds);
    atomicMax(&stats[9], (unsigned long long) tk_total);
    atomicAdd(&stats[10], (unsigned long long) tk_sched);
    atomicAdd(&stats[11], (unsigned long long) tk_head);
    atomicAdd(&stats[12], (unsigned long long) tk_tail);
    atomicAdd(&stats[13], st_stale);
    atomicAdd(&stats[14], st_leaf_nocand);
    atomicAdd(&stats[15], st_leaf_lanes);
  }
}

template <bool STATS>
__global__ __launch_bounds__(64 * kPipWaves, 6) void k_pip(PipArgs A) {
  PipRestArgs Q;
  Q.order = A.order; Q.n_dev = A.n_dev; Q.rest_count = A.rest_count;
  Q.work_counter = A.work_counter; Q.next_work_counter = A.next_work_counter;
  Q.group_lanes = A.group_lanes; Q.chunk_groups = A.chunk_groups; Q.blocks = 0;
  pip_locate<STATS>(A, blockIdx.x, Q, A.stats);
}


// =============================================================================================
// PIP, first pass: k_pip_walk -- the same front-to-back traversal as k_pip with everything that is
// not integer work taken out, so that it fits 64 VGPRs and (stack sized by the tree's height) 20 KiB of
// LDS per block: 8 waves per SIMD instead of 6.  k_pip's time is t = 0.39 ms + 3.83 ms / (waves per SIMD)
// on the headline pair (tools/sweep.py --opt max_blocks): per-wave latency, not VALU throughput
// (tools/issue_probe.hip: a SIMD issues a wave-instruction every 0.7-2 cycles with 6-8 waves; k_pip used
// one in 4.9), so both more waves and fewer instructions / LDS round trips per leaf visit pay.
//   * A point whose candidate list ends up holding exactly one CERTAIN hit (or nothing) is settled here:
//     99.7 % of the headline's points.  Every other point -- ties, points on base vertices, boxes that
//     touch the ray's x or start below the point -- leaves with its complete candidate list (WalkTodo) and
//     k_pip_exact evaluates that list with the exact arithmetic, no second traversal; only a point whose
//     list overflowed goes to the `rest` list, which k_pip locates from scratch (order = rest, n from the device).
//   * Leaf search without a search: the block's 256-entry x-bucket table (k_build_leaves) gives the slot
//     the backward scan starts at -- one cross-lane read instead of 7 pivots + 4 dependent probes.
// Pruning is exactly k_pip's (certain hits bound a lane; a certain hit below the held one replaces it),
// so whatever is settled here is what k_pip would have answered.
// =============================================================================================
// The walk's stack: the worst case of a 64-ary tree is 64 + 63 (top - 1) entries; measured on every stand-in pair a 64-point
// group never holds more than 94 (USCounty 53, WaterBodies 68, LakesNA 77, the gaussian polygons 94).  The stack is cut at
// kWalkStack entries whatever the height, and a group that would need more LEAVES THE WALK: its points go to the rest
// list, which k_pip_exact's first blocks locate with k_pip's traversal (worst-case stack).  That is what lets eight
// blocks of k_pip_walk2 share a CU's LDS with six candidate slots per point on a tree of any height.
constexpr int kWalkStack = 124;
__host__ __device__ __forceinline__ int walk_stack_entries(int top) {
  const int worst = 64 + 63 * (top > 1 ? top - 1 : 0) + 3;
  return worst < kWalkStack ? worst : kWalkStack;
}
__host__ __device__ __forceinline__ size_t walk_wave_lds(int top) { return (size_t) 16 * walk_stack_entries(top) + (size_t) kWalkList * 256; }

template <bool STATS>
__global__ __launch_bounds__(256, 8) void k_pip_walk(PipArgs A) {
  extern __shared__ uint4 walk_smem[];
  const int lane = lane_id();
  const int wib = threadIdx.x >> 6;
  const DeviceBvh& T = A.bvh;
  const uint32_t* const sky = (T.sky && T.sky[kSkyBuckets] == 0u) ? T.sky : nullptr;  // (exhaustive, or not used)
  const int stack_entries = walk_stack_entries(T.top);
  const int stack_cap = A.walk_stack > 0 && A.walk_stack < stack_entries ? A.walk_stack : stack_entries;
  uint4* const stack = walk_smem + (size_t) wib * (walk_wave_lds(T.top) / 16);
  uint32_t* const cand = reinterpret_cast<uint32_t*>(stack + stack_entries);  // [kWalkList][64], bank = lane
  const uint32_t stack_lds = (uint32_t) (uintptr_t) stack;  // the stack's LDS byte address (the low half of the generic pointer)
  const uint32_t GL = A.group_lanes;
  const uint64_t ngroups = (A.n + GL - 1) / GL;
  const uint32_t nchunks = (uint32_t) ((ngroups + A.chunk_groups - 1) / A.chunk_groups);
  int part = blockIdx.x & 7, tried = 0;
  unsigned long long st_leaf = 0, st_nodes = 0, st_box = 0, st_stale = 0, st_rest = 0, st_leaf_lanes = 0;
  long long tk_leaf = 0, tk_node = 0, tk_sched = 0, tk_head = 0, tk_tail = 0;  // STATS: cycle stamps
  const long long tk_begin = STATS ? clock64() : 0;
  __shared__ unsigned long long ranges[4];
  if (threadIdx.x < 4) ranges[threadIdx.x] = 0;
  if (blockIdx.x == 0 && threadIdx.x < 8) A.next_work_counter[threadIdx.x * 32] = 0;  // (see k_lsi)
  if (blockIdx.x == 0 && threadIdx.x == 8) *A.next_rest_count = 0;
  __syncthreads();
  for (;;) {
    uint32_t g32 = 0;
    const long long tks = STATS ? clock64() : 0;
    if (!next_group<4>(ranges, wib, A.work_counter, nchunks, A.chunk_groups, ngroups, part, tried, lane, g32)) break;
    if (STATS) tk_sched += clock64() - tks;
    const long long tkg = STATS ? clock64() : 0;
    const uint64_t ipos = (uint64_t) g32 * GL + lane;
    const bool valid = (uint32_t) lane < GL && ipos < A.n;
    // (an idle lane reads the last point: no load under a per-lane branch, so the point and the root's boxes -- which do
    //  not depend on it -- are one memory round trip, not two)
    const uint64_t ipc = valid ? ipos : A.n - 1;
    const uint32_t ip = A.order ? A.order[ipc] : (uint32_t) ipc;
    typedef long long ll2_t __attribute__((ext_vector_type(2)));
    const ll2_t pt_raw = __builtin_nontemporal_load(reinterpret_cast<const ll2_t*>(A.pts) + ip);
    const QBox root_box = T.lvl[T.top][lane];
    const uint64_t root_higher = sibling_order(T, T.top)[lane];
    const int32_t qx = valid ? quant(pt_raw.x) : 0, qy = valid ? quant(pt_raw.y) : 0;
    const bool live = valid && ray_has_sky(sky, qx, qy);  // (above the map's skyline: a certain miss, no traversal)
    const int32_t qym1 = qy > 0 ? qy - 1 : 0;
    int32_t gx0 = live ? qx : kEmptyMin, gx1 = live ? qx : kEmptyMax, gy0 = live ? qy : kEmptyMin;
    wave_min_max_min(gx0, gx1, gy0);
    int32_t qbest = live ? 0x7FFFFFFF : -1;  // sound quantised upper bound of this lane's answer (-1: the lane sits out)
    int32_t gbest = 0x7FFFFFFF;               // wave max of qbest
    // This lane's candidate list is cand[lane + 64 k] (bank = lane); `cand_at` = where the next one goes, so the fill
    // is (cand_at - lane) / 64 and kWalkList + 1 fills mean "overflowed: the rest list takes the point".
    const uint32_t cand_base = (uint32_t) lane;
    uint32_t cand_at = cand_base;
    int32_t sure_y0 = INT32_MIN;  // the list holds exactly one candidate, a certain hit whose box starts here (INT32_MIN: it does not)

    auto refine_if_many = [&](const QBox& b, uint64_t um) -> uint64_t {
      if (__popcll(um) <= kPipRefineAbove) return um;
      uint64_t keep = 0;
      while (um) {
        const int c = __builtin_ctzll(um);
        um &= um - 1;
        const int32_t cx0 = bcast(b.x0, c), cy0 = bcast(b.y0, c);
        const int32_t cx1 = bcast(b.x1, c), cy1 = bcast(b.y1, c);
        if (__ballot(ray_can_hit(qx, qym1, qbest, cx0, cy0, cx1, cy1))) keep |= 1ull << c;
      }
      return keep;
    };
    int sp = 0, sp_max = 0;
    bool ovf = false;  // wave-uniform: the stack would not hold this group's traversal
    {
      const QBox b = root_box;
      const uint64_t higher = root_higher;
      uint64_t m = refine_if_many(b, __ballot(b.x0 <= gx1 && gx0 <= b.x1 && b.y1 >= gy0 - 1));
      if (__popcll(m) > stack_cap) { m = 0; ovf = true; }
      if ((m >> lane) & 1)
        stack[__popcll(m & higher)] = make_uint4(((uint32_t) T.top << 28) | (uint32_t) lane, (uint32_t) b.y0, (uint32_t) b.x0, (uint32_t) b.x1);
      sp = __popcll(m);
      sp_max = sp;
      wave_lds_fence();
    }
    if (STATS) tk_head += clock64() - tkg;
    while (sp > 0) {
      // every lane reads the same entry (an LDS broadcast): the staleness test runs on those registers as they
      // are, only the node id moves to the scalar side
      // (one ds_read_b128 at one computed address: left to itself the compiler reads the entry in three pieces
      //  behind three address computations -- 4 of the pop's 10 VALU instructions, and a group pops 19 entries)
      --sp;
      uint4 ent;
      asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(ent) : "v"(stack_lds + (uint32_t) sp * 16u) : "memory");
      const int32_t ey0 = (int32_t) ent.y, ex0 = (int32_t) ent.z, ex1 = (int32_t) ent.w;
      const bool want = ((uint32_t) (qx - ex0) <= (uint32_t) (ex1 - ex0)) & (qbest >= ey0);  // stale entries die here, untouched
      if (!__ballot(want)) {
        if (STATS) st_stale++;
        continue;
      }
      const uint32_t e = __builtin_amdgcn_readfirstlane(ent.x);
      const long long tk0 = STATS ? clock64() : 0;
      const int lvl = (int) (e >> 28);
      const uint32_t idx = e & 0x0FFFFFFFu;
      if (lvl > 1) {
        const QBox b = T.lvl[lvl - 1][(uint64_t) idx * 64 + lane];
        const uint64_t higher = sibling_order(T, lvl - 1)[(uint64_t) idx * 64 + lane];
        uint64_t m = refine_if_many(b, __ballot(b.x0 <= gx1 && gx0 <= b.x1 && b.y1 >= gy0 - 1 && b.y0 <= gbest));
        if (sp + __popcll(m) > stack_cap) { ovf = true; break; }  // (the group leaves the walk: see kWalkStack)
        if ((m >> lane) & 1)
          stack[sp + __popcll(m & higher)] =
              make_uint4(((uint32_t) (lvl - 1) << 28) | (idx * 64 + lane), (uint32_t) b.y0, (uint32_t) b.x0, (uint32_t) b.x1);
        sp += __popcll(m);
        wave_lds_fence();
        if (STATS) {
          sp_max = sp > sp_max ? sp : sp_max;
          st_nodes++;
          tk_node += clock64() - tk0;
        }
      } else {
        const uint32_t slot0 = idx * 64;
        const QBox bb = T.box0[(uint64_t) slot0 + lane];  // one base segment per lane, sorted by x0
        const uint2 tab = T.xtab[(uint64_t) slot0 + lane];
        // The block's bucket table gives the slots [lo, hi) that can contain this point's x: no search for the
        // first one, no prefix-max fetch to know the last (the entry carries the block's x-extent, which is what
        // the table was built on).
        const uint32_t sx0s = __builtin_amdgcn_readfirstlane((uint32_t) ex0);
        const int sh = leaf_bucket_shift(__builtin_amdgcn_readfirstlane((uint32_t) ex1) - sx0s);
        const uint32_t bk = want ? ((uint32_t) qx - sx0s) >> sh : 0u;
        const uint32_t bsh = (bk & 3u) * 8u;
        const uint32_t hi = ((uint32_t) __shfl((int) tab.x, (int) (bk >> 2), 64) >> bsh) & 0xFFu;
        const uint32_t lo = ((uint32_t) __shfl((int) tab.y, (int) (bk >> 2), 64) >> bsh) & 0xFFu;
        // One correction per end (both fetches in one round trip): the range is exact to a bucket, and with ~28 lanes
        // looking, some lane's bucket nearly always holds a vertex -- a slot that starts behind the point, or one
        // that ends before it -- which would cost the whole wave an iteration each.
        int j = (int) hi - 1, jlo = (int) lo;
        {
          const int32_t top_x0 = __builtin_amdgcn_ds_bpermute(j << 2, bb.x0);
          const int32_t low_x1 = __builtin_amdgcn_ds_bpermute(jlo << 2, bb.x1);
          j -= top_x0 > qx ? 1 : 0;
          jlo += low_x1 < qx ? 1 : 0;  // (slots below lo end before the bucket, so slot lo's prefix max is its own x1)
        }
        j = want ? j : -1;
        jlo = want ? jlo : 0;
        const int32_t qbest_before = qbest;
        if (STATS) {
          st_leaf++;
          st_leaf_lanes += (unsigned long long) __popcll(__ballot(want));
        }
        while (__ballot(j >= jlo)) {
          const int jj = j & 63;  // (a lane past its range reads some slot: harmless, the test below fails on j - jlo)
          const int ja = j << 2;  // (ds_bpermute takes the lane from bits 7:2 of the address)
          const int32_t sx0 = __builtin_amdgcn_ds_bpermute(ja, bb.x0), sx1 = __builtin_amdgcn_ds_bpermute(ja, bb.x1);
          const int32_t sy0 = __builtin_amdgcn_ds_bpermute(ja, bb.y0), sy1 = __builtin_amdgcn_ds_bpermute(ja, bb.y1);
          if (STATS) st_box++;
          // ray_can_hit and "still inside my range" (compares joined on the scalar side: fewer VALU than one sign test)
          if ((sx0 <= qx) & (qx <= sx1) & (sy1 >= qym1) & (qbest >= sy0) & (j >= jlo)) {
            // certain hit (strictly inside in x, strictly above) => its box top bounds the answer; one that ends below
            // the start of the one certain hit held so far replaces it (that one is certainly higher)
            const bool certain = sx0 < qx && qx < sx1 && sy0 > qy;
            const bool replace = certain && sy1 < sure_y0;
            const bool first = cand_at == cand_base;
            // (a fifth candidate overwrites slot 0 of a list that is abandoned anyway: fill 5 = "the rest list takes
            // this point", and the lane sits the rest of the traversal out)
            const bool over = !replace && cand_at == cand_base + kWalkList * 64;
            cand[(replace || over) ? cand_base : cand_at] = slot0 + (uint32_t) jj;
            sure_y0 = (replace || (first && certain)) ? sy0 : INT32_MIN;
            cand_at += replace ? 0u : 64u;
            const int32_t top = certain ? sy1 + 1 : 0x7FFFFFFF;
            qbest = over ? -1 : (top < qbest ? top : qbest);
          }
          j--;
        }
        if (__ballot(qbest != qbest_before)) {
          const int32_t gbest_before = gbest;
          gbest = wave_max(qbest);
          if (gbest < gbest_before && sp > 1) {  // sweep the stack once: drop every entry that starts above the group's bound
            int kept = 0;
            for (int base = 0; base < sp; base += 64) {
              const int i = base + lane;
              const bool have = i < sp;
              uint4 en = make_uint4(0, 0, 0, 0);
              if (have) en = stack[i];
              const bool alive = have && (int32_t) en.y <= gbest;
              const uint64_t am = __ballot(alive);
              wave_lds_fence();
              if (alive) stack[kept + rank_below(am)] = en;
              kept += __popcll(am);
            }
            if (STATS) st_stale += (unsigned long long) (sp - kept);
            sp = kept;
            wave_lds_fence();
          }
        }
        if (STATS) tk_leaf += clock64() - tk0;
      }
    }
    const long long tkt = STATS ? clock64() : 0;
    // settled: nothing above the point, or exactly one candidate and it is a certain hit (every other
    // edge over this x was pruned because it starts above that candidate's box top)
    const bool done = valid && !ovf && (cand_at == cand_base || sure_y0 != INT32_MIN);
    {  // (both gathers requested before either is stored; a lane without a hit reads slot 0)
      const bool hit = done && cand_at != cand_base;
      const uint32_t slot = hit ? cand[lane] : 0u;
      const uint32_t e = T.seid[slot];
      const int32_t f = A.face ? T.sface[slot] : 0;
      if (done) {
        __builtin_nontemporal_store(hit ? e : 0xFFFFFFFFu, A.closest + ip);
        if (A.face) __builtin_nontemporal_store(hit ? f : 0, A.face + ip);
      }
    }
    // the others leave with their candidate list (the exact kernel needs nothing else: every edge that could be the
    // answer is on it, pruning only ever used certain hits) -- or, with an overflowed list, as a point for k_pip.
    // One slot per query position and one mask per group: nothing to contend for (an append counter shared by
    // 450 k groups, most of which have something to hand over on a map pair with shared vertices, stalls them all).
    const uint32_t fill = (cand_at - cand_base) >> 6;
    const bool listed = valid && !done && !ovf && fill <= (uint32_t) kWalkList;
    const bool rest = valid && !done && !listed;
    // (the group's records lie side by side at the head of its todo region, in the order of the mask's bits: a group
    //  with a dozen listed points writes -- and k_pip_exact reads -- three lines, not a dozen)
    const uint64_t lm = __ballot(listed);
    if (listed) {
      const uint64_t rec = (uint64_t) g32 * GL + rank_below(lm);
#pragma unroll
      for (int k = 0; k < kWalkList; k++) A.todo[rec * kWalkList + k] = (uint32_t) k < fill ? cand[lane + 64 * k] : 0xFFFFFFFFu;
    }
    if (lane == 0) A.todo_mask[g32] = lm;
    const uint64_t rm = __ballot(rest);
    if (rm) {
      unsigned long long base = 0;
      if (lane == 0) base = atomicAdd(A.rest_count, (unsigned long long) __popcll(rm));
      base = ((unsigned long long) __builtin_amdgcn_readfirstlane((uint32_t) (base >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t) base);
      if (rest) A.rest[base + rank_below(rm)] = ip;
    }
    wave_lds_fence();  // (the lists are reused by the next group)
    if (STATS) {
      if (lane == 0 && A.stats) atomicMax(&A.stats[8], (unsigned long long) sp_max);  // (the deepest stack: what kWalkStack is sized by)
      st_rest += (unsigned long long) __popcll(rm);
      tk_tail += clock64() - tkt;
    }
  }
  if (STATS && lane == 0 && A.stats) {  // (same slots as k_pip's where the meaning is the same)
    const long long tk_total = clock64() - tk_begin;
    atomicAdd(&A.stats[0], st_leaf);
    atomicAdd(&A.stats[1], st_rest);
    atomicAdd(&A.stats[2], st_nodes);
    atomicAdd(&A.stats[3], st_box);
    atomicAdd(&A.stats[4], (unsigned long long) tk_total);
    atomicAdd(&A.stats[5], (unsigned long long) tk_node);
    atomicAdd(&A.stats[6], (unsigned long long) tk_leaf);
    atomicMax(&A.stats[9], (unsigned long long) tk_total);
    atomicAdd(&A.stats[10], (unsigned long long) tk_sched);
    atomicAdd(&A.stats[11], (unsigned long long) tk_head);
    atomicAdd(&A.stats[12], (unsigned long long) tk_tail);
    atomicAdd(&A.stats[13], st_stale);
    atomicAdd(&A.stats[15], st_leaf_lanes);
  }
}


// k_pip_walk with TWO points per lane: a wave takes 128 consecutive query positions (lane l: positions l and 64 + l of
// the group) through ONE traversal -- node expansions, pops, the leaf blocks' loads and bucket-table reads are shared,
// the per-point work (pop-time test, bucket lookup, scan, candidate bookkeeping) is done per point set, and a set none
// of whose lanes wants a leaf block skips it (wave-uniform).  Same stack, twice the candidate lists; the hand-over
// (one todo slot per position, one mask per 64 positions, the rest list) is exactly k_pip_walk's, so k_pip_exact
// cannot tell the two apart.  Requires group_lanes == 64 (a large query set).
// (P points per lane: 2, or 4 -- 256 positions per wave, the traversal's own costs shared by twice the points)
__host__ __device__ __forceinline__ size_t walk2_wave_lds(int top, int P = 2) { return (size_t) 16 * walk_stack_entries(top) + (size_t) kWalkList * 256 * P; }

// (96 SGPRs: above that the hardware admits one block per CU fewer than the occupancy query reports, and the shared
//  schedule's walk + k_lsi2 blocks per CU no longer fit -- the skyline pointer took it to 100: 0.785 -> 0.852 ms.
//  256 threads x 9 blocks: the compiler then aims below 57 VGPRs -- at 56, six blocks fit beside two of k_lsi2's 80)
constexpr int kWalkCycleStamps = 7777;  // "stack_cap" value that turns the instrumented walk's counters into cycle stamps (tools/walk_stats_probe.py --cycles)
template <bool STATS, int P>
__device__ __forceinline__ void pip_walk_many(const PipArgs& A) {
  extern __shared__ uint4 walk_smem[];
  const int lane = lane_id();
  const int wib = threadIdx.x >> 6;
  const DeviceBvh& T = A.bvh;
  const uint32_t* const sky = (T.sky && T.sky[kSkyBuckets] == 0u) ? T.sky : nullptr;  // (exhaustive, or not used)
  const int stack_entries = walk_stack_entries(T.top);
  const int stack_cap = A.walk_stack > 0 && A.walk_stack < stack_entries ? A.walk_stack : stack_entries;
  uint4* const stack = walk_smem + (size_t) wib * (walk2_wave_lds(T.top, P) / 16);
  uint32_t* const cand = reinterpret_cast<uint32_t*>(stack + stack_entries);  // [P][kWalkList][64], bank = lane
  const uint32_t stack_lds = (uint32_t) (uintptr_t) stack;
  constexpr uint32_t GP = 64u * P;  // positions per wave-group
  const uint64_t ngroups = (A.n + GP - 1) / GP;
  const uint32_t nchunks = (uint32_t) ((ngroups + A.chunk_groups - 1) / A.chunk_groups);
  int part = blockIdx.x & 7, tried = 0;
  __shared__ unsigned long long ranges[4];
  if (threadIdx.x < 4) ranges[threadIdx.x] = 0;
  if (blockIdx.x == 0 && threadIdx.x < 8) A.next_work_counter[threadIdx.x * 32] = 0;  // (see k_lsi)
  if (blockIdx.x == 0 && threadIdx.x == 8) *A.next_rest_count = 0;
  __syncthreads();
  // STATS (the instrumented build, rj_set_option "stats" 1 + "pip_walk" 2): what a 128-position group costs, per wave
  unsigned long long st_pops = 0, st_stale = 0, st_nodes = 0, st_leaf = 0, st_setvisits = 0, st_steps = 0, st_pushed = 0, st_sweeps = 0, st_swept = 0,
                     st_groups = 0, st_want = 0, st_hits = 0, st_rest = 0;
  // ... and, with "stack_cap" set to kWalkCycleStamps, where a wave's cycles go instead (s_memtime stamps around the
  // phases; the waits for memory are made explicit so that they are charged to the phase that needs the data)
  const bool cyc = STATS && A.stack_cap == kWalkCycleStamps;
  long long ck_sched = 0, ck_load = 0, ck_head = 0, ck_pop = 0, ck_node_wait = 0, ck_node = 0, ck_leaf_wait = 0, ck_leaf = 0, ck_bound = 0, ck_tail = 0;
  const long long ck_begin = STATS ? clock64() : 0;
#define RJ_CK(var, since) do { if (STATS && cyc) { const long long _n = clock64(); var += _n - since; since = _n; } } while (0)
  for (;;) {
    uint32_t g32 = 0;
    long long ck = STATS && cyc ? clock64() : 0;
    if (!next_group<4>(ranges, wib, A.work_counter, nchunks, A.chunk_groups, ngroups, part, tried, lane, g32)) break;
    RJ_CK(ck_sched, ck);
    typedef long long ll2_t __attribute__((ext_vector_type(2)));
    int32_t qx[P], qy[P], qbest[P], sure_y0[P];
    uint32_t cand_base[P], cand_at[P];
    bool valid[P];
    // Every load the group starts with is requested before the first one is waited for: the P points (positions past the
    // end read the last point: no branch around a load, so the compiler batches them) and the root's boxes, which do not
    // depend on the points at all.  Left in `if (valid)` regions each load was waited for inside its own region: three
    // memory round trips before the first test, of the 23 k cycles a group's critical path is long.
    uint32_t ipl[P];
#pragma unroll
    for (int p = 0; p < P; p++) {
      const uint64_t ipos = (uint64_t) g32 * GP + (uint64_t) p * 64 + lane;
      valid[p] = ipos < A.n;
      const uint64_t ipc = valid[p] ? ipos : A.n - 1;
      ipl[p] = A.order ? A.order[ipc] : (uint32_t) ipc;
    }
    ll2_t ptl[P];
#pragma unroll
    for (int p = 0; p < P; p++) ptl[p] = __builtin_nontemporal_load(reinterpret_cast<const ll2_t*>(A.pts) + ipl[p]);
    const QBox root_box = T.lvl[T.top][lane];
    const uint64_t root_higher = sibling_order(T, T.top)[lane];
    if (STATS && cyc) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); RJ_CK(ck_load, ck); }
#pragma unroll
    for (int p = 0; p < P; p++) {
      qx[p] = valid[p] ? quant(ptl[p].x) : 0;
      qy[p] = valid[p] ? quant(ptl[p].y) : 0;
      qbest[p] = valid[p] ? 0x7FFFFFFF : -1;
      cand_base[p] = (uint32_t) lane + (uint32_t) p * (kWalkList * 64);
      cand_at[p] = cand_base[p];
      sure_y0[p] = INT32_MIN;
    }
    // (above the map's skyline: a certain miss -- the point sits the traversal out like a position past the end)
    if (sky) {
#pragma unroll
      for (int p = 0; p < P; p++)
        if (sky[(uint32_t) qx[p] >> kSkyShift] <= (uint32_t) qy[p]) qbest[p] = -1;
    }
    int32_t gx0, gx1, gy0;
    {
      const bool l0 = qbest[0] >= 0, l1 = qbest[1] >= 0;
      const int32_t a0 = l0 ? qx[0] : kEmptyMin, a1 = l1 ? qx[1] : kEmptyMin;
      const int32_t b0 = l0 ? qx[0] : kEmptyMax, b1 = l1 ? qx[1] : kEmptyMax;
      const int32_t c0 = l0 ? qy[0] : kEmptyMin, c1 = l1 ? qy[1] : kEmptyMin;
      gx0 = a0 < a1 ? a0 : a1; gx1 = b0 > b1 ? b0 : b1; gy0 = c0 < c1 ? c0 : c1;
#pragma unroll
      for (int p = 2; p < P; p++) {
        const bool lp = qbest[p] >= 0;
        const int32_t ap = lp ? qx[p] : kEmptyMin, bp = lp ? qx[p] : kEmptyMax, cp = lp ? qy[p] : kEmptyMin;
        gx0 = gx0 < ap ? gx0 : ap; gx1 = gx1 > bp ? gx1 : bp; gy0 = gy0 < cp ? gy0 : cp;
      }
    }
    wave_min_max_min(gx0, gx1, gy0);
    int32_t gbest = 0x7FFFFFFF;  // wave max of both sets' qbest
    bool gbest_stale = false;    // ... some lane's bound has dropped since it was last reduced (wave-uniform)

    auto refine_if_many = [&](const QBox& b, uint64_t um) -> uint64_t {
      if (__popcll(um) <= kPipRefineAbove) return um;
      uint64_t keep = 0;
      while (um) {
        const int c = __builtin_ctzll(um);
        um &= um - 1;
        const int32_t cx0 = bcast(b.x0, c), cy0 = bcast(b.y0, c);
        const int32_t cx1 = bcast(b.x1, c), cy1 = bcast(b.y1, c);
        bool any = ray_can_hit(qx[0], qy[0] > 0 ? qy[0] - 1 : 0, qbest[0], cx0, cy0, cx1, cy1) || ray_can_hit(qx[1], qy[1] > 0 ? qy[1] - 1 : 0, qbest[1], cx0, cy0, cx1, cy1);
#pragma unroll
        for (int p = 2; p < P; p++) any = any || ray_can_hit(qx[p], qy[p] > 0 ? qy[p] - 1 : 0, qbest[p], cx0, cy0, cx1, cy1);
        if (__ballot(any)) keep |= 1ull << c;
      }
      return keep;
    };
    int sp = 0;
    bool ovf = false;  // wave-uniform: the stack would not hold this group's traversal
    {
      const QBox b = root_box;
      const uint64_t higher = root_higher;
      uint64_t m = refine_if_many(b, __ballot(b.x0 <= gx1 && gx0 <= b.x1 && b.y1 >= gy0 - 1));
      if (__popcll(m) > stack_cap) { m = 0; ovf = true; }
      if ((m >> lane) & 1)
        stack[__popcll(m & higher)] = make_uint4(((uint32_t) T.top << 28) | (uint32_t) lane, (uint32_t) b.y0, (uint32_t) b.x0, (uint32_t) b.x1);
      sp = __popcll(m);
      wave_lds_fence();
      if (STATS) { st_groups++; st_pushed += (unsigned long long) sp; }
    }
    RJ_CK(ck_head, ck);
    while (sp > 0) {
      --sp;
      uint4 ent;
      if (STATS) st_pops++;
      asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(ent) : "v"(stack_lds + (uint32_t) sp * 16u) : "memory");
      const int32_t ey0 = (int32_t) ent.y, ex0 = (int32_t) ent.z, ex1 = (int32_t) ent.w;
      // (compares, not sign bits: 3 VALU + 1 SALU per set where the sign test took 5 VALU, and 15 of a group's 25 pops
      //  end right here)
      const uint32_t ew = (uint32_t) (ex1 - ex0);
      bool want[P];
      bool want_any = false;
#pragma unroll
      for (int p = 0; p < P; p++) {
        want[p] = ((uint32_t) (qx[p] - ex0) <= ew) & (qbest[p] >= ey0);
        want_any = want_any || want[p];
      }
      if (!__ballot(want_any)) {  // stale: untouched
        if (STATS) st_stale++;
        RJ_CK(ck_pop, ck);
        continue;
      }
      RJ_CK(ck_pop, ck);
      const uint32_t e = __builtin_amdgcn_readfirstlane(ent.x);
      const int lvl = (int) (e >> 28);
      const uint32_t idx = e & 0x0FFFFFFFu;
      if (lvl > 1) {
        uint32_t lane_here = (uint32_t) lane;  // (see the leaf branch)
        asm volatile("" : "+v"(lane_here));
        const QBox b = T.lvl[lvl - 1][(uint64_t) idx * 64 + lane_here];
        const uint64_t higher = sibling_order(T, lvl - 1)[(uint64_t) idx * 64 + lane_here];
        if (STATS && cyc) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); RJ_CK(ck_node_wait, ck); }
        if (gbest_stale) {  // (while the boxes are on their way)
          gbest_stale = false;
          int32_t lane_best = qbest[0] > qbest[1] ? qbest[0] : qbest[1];
#pragma unroll
          for (int p = 2; p < P; p++) lane_best = lane_best > qbest[p] ? lane_best : qbest[p];
          gbest = wave_max(lane_best);
        }
        uint64_t m = refine_if_many(b, __ballot(b.x0 <= gx1 && gx0 <= b.x1 && b.y1 >= gy0 - 1 && b.y0 <= gbest));
        if (sp + __popcll(m) > stack_cap) { ovf = true; break; }  // (the group leaves the walk: see kWalkStack)
        if ((m >> lane) & 1)
          stack[sp + __popcll(m & higher)] =
              make_uint4(((uint32_t) (lvl - 1) << 28) | (idx * 64 + lane), (uint32_t) b.y0, (uint32_t) b.x0, (uint32_t) b.x1);
        sp += __popcll(m);
        wave_lds_fence();
        if (STATS) { st_nodes++; st_pushed += (unsigned long long) __popcll(m); }
        RJ_CK(ck_node, ck);
      } else {
        if (STATS) st_leaf++;
        const uint32_t slot0 = idx * 64;
        // (the lane offset passes through an empty asm: left alone the compiler keeps base + 16 lane and base + 8 lane of
        //  the two arrays in four VGPRs through the whole traversal, and the kernel needs its 56)
        uint32_t lane_here = (uint32_t) lane;
        asm volatile("" : "+v"(lane_here));
        const QBox bb = T.box0[(uint64_t) slot0 + lane_here];  // one base segment per lane, sorted by x0
        const uint2 tab = T.xtab[(uint64_t) slot0 + lane_here];
        if (STATS && cyc) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); RJ_CK(ck_leaf_wait, ck); }
        const uint32_t sx0s = __builtin_amdgcn_readfirstlane((uint32_t) ex0);
        const int sh = leaf_bucket_shift(__builtin_amdgcn_readfirstlane((uint32_t) ex1) - sx0s);
        bool changed = false;
#pragma unroll
        for (int p = 0; p < P; p++) {
          if (!__ballot(want[p])) continue;  // none of this set's points is under this block
          if (STATS) { st_setvisits++; st_want += (unsigned long long) __popcll(__ballot(want[p])); }
          // (a lane that does not want the block computes on whatever its x gives: every cross-lane read masks its
          //  index, such a lane's j is set to -1 below, and no box of the block can pass its test -- its x lies outside
          //  the block or its bound below the block: two selects less per visit)
          const uint32_t bk = ((uint32_t) qx[p] - sx0s) >> sh;
          const uint32_t bsh = (bk & 3u) * 8u;
          const uint32_t hi = ((uint32_t) __shfl((int) tab.x, (int) (bk >> 2), 64) >> bsh) & 0xFFu;
          const uint32_t lo = ((uint32_t) __shfl((int) tab.y, (int) (bk >> 2), 64) >> bsh) & 0xFFu;
          int j = (int) hi - 1, jlo = (int) lo;
          {
            const int32_t top_x0 = __builtin_amdgcn_ds_bpermute(j << 2, bb.x0);
            const int32_t low_x1 = __builtin_amdgcn_ds_bpermute(jlo << 2, bb.x1);
            j -= top_x0 > qx[p] ? 1 : 0;
            jlo += low_x1 < qx[p] ? 1 : 0;
          }
          j = want[p] ? j : -1;
          const int32_t qbest_before = qbest[p];
          auto scan_step = [&]() {
                  if (STATS) st_steps++;
            const int jj = j & 63;
            const int ja = j << 2;
            const int32_t sx0 = __builtin_amdgcn_ds_bpermute(ja, bb.x0), sx1 = __builtin_amdgcn_ds_bpermute(ja, bb.x1);
            const int32_t sy0 = __builtin_amdgcn_ds_bpermute(ja, bb.y0), sy1 = __builtin_amdgcn_ds_bpermute(ja, bb.y1);
            if ((sx0 <= qx[p]) & (qx[p] <= sx1) & (sy1 >= qy[p] - 1) & (qbest[p] >= sy0) & (j >= jlo)) {
              const bool certain = sx0 < qx[p] && qx[p] < sx1 && sy0 > qy[p];
              const bool replace = certain && sy1 < sure_y0[p];
              const bool first = cand_at[p] == cand_base[p];
              const bool over = !replace && cand_at[p] == cand_base[p] + kWalkList * 64;
              if (STATS && lane == __builtin_ctzll(__ballot(true))) st_hits++;
              cand[(replace || over) ? cand_base[p] : cand_at[p]] = slot0 + (uint32_t) jj;
              sure_y0[p] = (replace || (first && certain)) ? sy0 : INT32_MIN;
              cand_at[p] += replace ? 0u : 64u;
              const int32_t top = certain ? sy1 + 1 : 0x7FFFFFFF;
              qbest[p] = over ? -1 : (top < qbest[p] ? top : qbest[p]);
            }
            j--;
          };
          // (a visit takes 1.1 steps: the first one stands outside the loop, where it carries no copies of the three
          //  values the loop hands from step to step)
          if (__ballot(j >= jlo)) {
            scan_step();
            while (__ballot(j >= jlo)) scan_step();
          }
          changed = changed || qbest[p] != qbest_before;
        }
        RJ_CK(ck_leaf, ck);
        // (the group's bound is what a node expansion filters its pushes by and what a sweep drops entries by: with at
        //  most one entry left neither may ever happen again -- the reduction waits until an expansion asks for it)
        if (__ballot(changed)) gbest_stale = true;
        if (gbest_stale && sp > 1) {
          gbest_stale = false;
          const int32_t gbest_before = gbest;
          int32_t lane_best = qbest[0] > qbest[1] ? qbest[0] : qbest[1];
#pragma unroll
          for (int p = 2; p < P; p++) lane_best = lane_best > qbest[p] ? lane_best : qbest[p];
          gbest = wave_max(lane_best);
          if (gbest < gbest_before && sp > 1) {  // sweep the stack once: drop every entry that starts above the group's bound
            int kept = 0;
            if (STATS) st_sweeps++;
            for (int base = 0; base < sp; base += 64) {
              const int i = base + lane;
              const bool have = i < sp;
              uint4 en = make_uint4(0, 0, 0, 0);
              if (have) en = stack[i];
              const bool alive = have && (int32_t) en.y <= gbest;
              const uint64_t am = __ballot(alive);
              wave_lds_fence();
              if (alive) stack[kept + rank_below(am)] = en;
              kept += __popcll(am);
            }
            if (STATS) st_swept += (unsigned long long) (sp - kept);
            sp = kept;
            wave_lds_fence();
          }
        }
        RJ_CK(ck_bound, ck);
      }
    }
    RJ_CK(ck_bound, ck);  // (what is left of the last iteration: the bound's reduction and sweeps are charged here and below)
    // hand-over, per set: exactly k_pip_walk's.  The gathers of the settled points -- edge id and face id under the one
    // candidate -- are all requested first (a lane without a hit reads slot 0: no branch around a load) and stored last,
    // behind the lists and masks: one memory round trip at the end of a group instead of 2 P.
    uint32_t out_e[P], ipv[P];
    int32_t out_f[P];
    bool donev[P];
#pragma unroll
    for (int p = 0; p < P; p++) {
      const uint64_t ipos = (uint64_t) g32 * GP + (uint64_t) p * 64 + lane;
      // (the point's index is read again rather than held in a register through the traversal: the kernel keeps to 56
      //  VGPRs, six of its blocks fit beside two of k_lsi2)
      ipv[p] = A.order ? A.order[valid[p] ? ipos : A.n - 1] : (uint32_t) ipos;
      donev[p] = valid[p] && !ovf && (cand_at[p] == cand_base[p] || sure_y0[p] != INT32_MIN);
      const bool hit = donev[p] && cand_at[p] != cand_base[p];
      const uint32_t slot = hit ? cand[cand_base[p]] : 0u;
      const uint32_t e = T.seid[slot];
      const int32_t f = A.face ? T.sface[slot] : 0;
      out_e[p] = hit ? e : 0xFFFFFFFFu;
      out_f[p] = hit ? f : 0;
    }
#pragma unroll
    for (int p = 0; p < P; p++) {
      const uint32_t fill = (cand_at[p] - cand_base[p]) >> 6;
      const bool listed = valid[p] && !donev[p] && !ovf && fill <= (uint32_t) kWalkList;
      const bool rest = valid[p] && !donev[p] && !listed;
      const uint64_t lm = __ballot(listed);
      const uint64_t g64 = (uint64_t) g32 * P + p;  // the 64-position group this set is
      if (listed) {
        const uint64_t rec = g64 * 64 + rank_below(lm);  // (records side by side at the head of the group's region: k_pip_walk)
#pragma unroll
        for (int k = 0; k < kWalkList; k++) A.todo[rec * kWalkList + k] = (uint32_t) k < fill ? cand[cand_base[p] + 64 * k] : 0xFFFFFFFFu;
      }
      if (lane == 0 && g64 * 64 < A.n) A.todo_mask[g64] = lm;
      const uint64_t rm = __ballot(rest);
      if (rm) {
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(A.rest_count, (unsigned long long) __popcll(rm));
        base = ((unsigned long long) __builtin_amdgcn_readfirstlane((uint32_t) (base >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t) base);
        if (rest) A.rest[base + rank_below(rm)] = ipv[p];
      }
      if (STATS) st_rest += (unsigned long long) __popcll(rm);
    }
#pragma unroll
    for (int p = 0; p < P; p++) {
      if (donev[p]) {
        __builtin_nontemporal_store(out_e[p], A.closest + ipv[p]);
        if (A.face) __builtin_nontemporal_store(out_f[p], A.face + ipv[p]);
      }
    }
    wave_lds_fence();  // (the lists are reused by the next group)
    if (STATS && cyc) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); RJ_CK(ck_tail, ck); }
  }
#undef RJ_CK
  if (STATS && A.stats && cyc) {
    if (lane == 0) {
      const long long total = clock64() - ck_begin;
      atomicAdd(&A.stats[0], (unsigned long long) total);
      atomicAdd(&A.stats[1], (unsigned long long) ck_sched);
      atomicAdd(&A.stats[2], (unsigned long long) ck_load);
      atomicAdd(&A.stats[3], (unsigned long long) ck_head);
      atomicAdd(&A.stats[4], st_groups);
      atomicAdd(&A.stats[5], (unsigned long long) ck_pop);
      atomicAdd(&A.stats[6], (unsigned long long) ck_node_wait);
      atomicAdd(&A.stats[7], (unsigned long long) ck_node);
      atomicAdd(&A.stats[8], (unsigned long long) ck_leaf_wait);
      atomicAdd(&A.stats[9], (unsigned long long) ck_leaf);
      atomicAdd(&A.stats[10], (unsigned long long) ck_bound);
      atomicAdd(&A.stats[11], (unsigned long long) ck_tail);
      atomicAdd(&A.stats[12], 1ull);  // waves
    }
    return;
  }
  if (STATS && A.stats) {
    // (per lane: st_hits is counted by one lane of each hit; everything else is wave-uniform and reported by lane 0)
    const unsigned long long hits = st_hits;
    if (hits) atomicAdd(&A.stats[6], hits);
    if (lane == 0) {
      atomicAdd(&A.stats[0], st_leaf);        // leaf blocks opened (per 128-position group traversal)
      atomicAdd(&A.stats[1], st_rest);
      atomicAdd(&A.stats[2], st_nodes);       // nodes expanded
      atomicAdd(&A.stats[3], st_steps);       // scan steps (per set)
      atomicAdd(&A.stats[4], st_groups);      // 128-position groups
      atomicAdd(&A.stats[5], st_setvisits);   // (leaf, set) visits with a wanting lane
      atomicAdd(&A.stats[7], st_sweeps);
      atomicAdd(&A.stats[8], st_swept);       // entries the sweeps dropped
      atomicAdd(&A.stats[10], st_pops);
      atomicAdd(&A.stats[11], st_pushed);
      atomicAdd(&A.stats[13], st_stale);      // pops no lane could use
      atomicAdd(&A.stats[15], st_want);       // lanes that wanted their (leaf, set) visit
    }
  }
}


// The kernels of the body above.  k_pip_walk2 is held to 56 VGPRs (the attribute takes a literal, hence no template): six
// of its blocks fit beside two of k_lsi2's 80 on a SIMD's 512.
__global__ __launch_bounds__(256, 8) __attribute__((amdgpu_num_sgpr(96))) __attribute__((amdgpu_num_vgpr(28))) void k_pip_walk2(PipArgs A) { pip_walk_many<false, 2>(A); }
__global__ __launch_bounds__(256, 4) __attribute__((amdgpu_num_sgpr(96))) void k_pip_walk2_stats(PipArgs A) { pip_walk_many<true, 2>(A); }
__global__ __launch_bounds__(256, 4) __attribute__((amdgpu_num_sgpr(96))) void k_pip_walk4(PipArgs A) { pip_walk_many<false, 4>(A); }
__global__ __launch_bounds__(256, 4) __attribute__((amdgpu_num_sgpr(96))) void k_pip_walk4_stats(PipArgs A) { pip_walk_many<true, 4>(A); }


// The walk's leftovers: the exact predicate (pip.h:36-95, as in k_pip's evaluate) over each listed point's complete
// candidate list.  A wave reads the masks of its share of the groups, compacts the listed positions into an LDS
// queue (__ballot / mbcnt, like k_lsi's pair buffer) and evaluates 64 of them at a time, one point per lane: on a
// map pair with shared vertices a quarter of the points arrive here, on the headline pair 0.3 %, and either way
// the lanes that do the 128-bit arithmetic are all busy.
#ifndef RJ_EXACT_WAVES
#define RJ_EXACT_WAVES 5
#endif
// The kernel's first R.blocks blocks do something else: they locate the points whose list overflowed (the walk's `rest`
// list, a handful to a few thousand) from scratch with k_pip's traversal.  That used to be a launch of its own behind
// this kernel -- 30-75 us of cold, unrelated traversals, one wave each, at the very end of the step; as the first
// blocks of this launch they start with it and end inside it.
static_assert(kPipWaves == 4, "k_pip_exact's first blocks run pip_locate: same block size");
__global__ __launch_bounds__(256, RJ_EXACT_WAVES) void k_pip_exact(PipArgs A, PipRestArgs R) {
  if (blockIdx.x < R.blocks) {
    pip_locate<false>(A, blockIdx.x, R, nullptr);
    return;
  }
  __shared__ uint32_t queue[4][128];   // query positions ...
  __shared__ uint32_t qrec[4][128];    // ... and where their records lie (group base + rank among the group's listed points)
  const DeviceBvh& T = A.bvh;
  const int lane = lane_id();
  const int wib = threadIdx.x >> 6;
  const uint32_t gl_shift = 31 - __builtin_clz(A.group_lanes);  // (group_lanes is a power of two)
  const uint64_t wave = ((blockIdx.x - R.blocks) * (uint64_t) blockDim.x + threadIdx.x) >> 6;
  const uint64_t nwaves = ((uint64_t) (gridDim.x - R.blocks) * blockDim.x) >> 6;
  auto evaluate = [&](uint32_t i, uint32_t rec) {
    const uint64_t ip = A.order ? A.order[i] : i;
    uint32_t slot[kWalkList];
#pragma unroll
    for (int r = 0; r < kWalkList; r++) slot[r] = A.todo[(uint64_t) rec * kWalkList + r];  // (one round trip for the record)
    typedef long long ll2_t __attribute__((ext_vector_type(2)));
    const ll2_t pxy = reinterpret_cast<const ll2_t*>(A.pts)[ip];  // (one 16-byte request)
    const int64_t px = pxy.x, py = pxy.y;
    double best_yy = __builtin_inf();
    uint32_t best_slot = 0xFFFFFFFFu;
    Seg best_seg = {0, 0, 0, 0};
    // The candidates one after the other, the next one's segment requested while this one is evaluated: a handful of
    // live registers (the kernel is bound by the latency of these gathers, so resident waves count: 8 per SIMD, where
    // holding all six segments at once allowed 4), and integer triage before the 128-bit arithmetic --
    // xsect_y is the double image of a value inside the edge's exact y-range [ylo, yhi], off by less than 2^-6 (two
    // roundings of a value below 2^46), so an edge with yhi <= py - 1 passes below the point (rejected, diff_y > 0) and
    // one with ylo >= py + 1 above it (a hit, no substitution); such a sure hit only needs the arithmetic when its
    // y-range overlaps the best so far.
    int64_t best_lo = INT64_MAX, best_hi = INT64_MAX;  // exact y-range of the best edge (its xsect_y lies inside, to 2^-6)
    bool best_exact = false;                           // best_yy has been computed
    Seg cur = {0, 0, 0, 0}, nxt = {0, 0, 0, 0};  // (two requests in flight: most listed points hold two or three candidates)
    if (slot[0] != 0xFFFFFFFFu) cur = T.sseg[slot[0]];
    if (slot[1] != 0xFFFFFFFFu) nxt = T.sseg[slot[1]];
#pragma unroll
    for (int r = 0; r < kWalkList; r++) {
      if (slot[r] == 0xFFFFFFFFu) break;
      const Seg e = cur;
      cur = nxt;
      if (r + 2 < kWalkList && slot[r + 2] != 0xFFFFFFFFu) nxt = T.sseg[slot[r + 2]];
      const int64_t x_min = e.x1 < e.x2 ? e.x1 : e.x2, x_max = e.x1 < e.x2 ? e.x2 : e.x1;
      if (px < x_min || px > x_max || px == (A.query_map_id == 0 ? x_min : x_max)) continue;  // pip.h:44-46, in integers
      const int64_t ylo = e.y1 < e.y2 ? e.y1 : e.y2, yhi = e.y1 < e.y2 ? e.y2 : e.y1;
      if (yhi <= py - 1) continue;              // passes below the point
      if (ylo >= py + 1 && ylo > best_hi) continue;  // a sure hit, but certainly higher than the best so far
      if (ylo >= py + 1 && yhi < best_lo) {     // a sure hit certainly lower than the best so far: no arithmetic yet
        best_slot = slot[r]; best_seg = e; best_lo = ylo; best_hi = yhi; best_exact = false;
        continue;
      }
      // overlapping y-ranges, or an edge that touches the point's y: the exact predicate decides
      if (best_slot != 0xFFFFFFFFu && !best_exact) {
        double byy;
        (void) pip_eval_y(best_seg, px, py, A.query_map_id, &byy);  // (a sure hit: always accepted)
        best_yy = byy; best_exact = true;
      }
      double yy;
      if (pip_eval_y(e, px, py, A.query_map_id, &yy)) {
        bool better = yy < best_yy;
        if (yy == best_yy && best_slot != 0xFFFFFFFFu)  // tie: slope rule, then eid
          better = pip_better(yy, pip_slope(e), T.seid[slot[r]], best_yy, pip_slope(best_seg), T.seid[best_slot], A.query_map_id);
        if (better) {
          best_yy = yy; best_slot = slot[r]; best_seg = e; best_lo = ylo; best_hi = yhi; best_exact = true;
        }
      }
    }
    const bool hit = best_slot != 0xFFFFFFFFu;
    A.closest[ip] = hit ? T.seid[best_slot] : 0xFFFFFFFFu;
    if (A.face) A.face[ip] = hit ? T.sface[best_slot] : 0;
  };
  int nq = 0;  // wave-uniform fill of the queue
  // 16 groups per step: lane l < 16 fetches the mask of group g0 + l (one load, not one round trip per group)
  const uint64_t ngroups = (A.n + A.group_lanes - 1) >> gl_shift;
  for (uint64_t g0 = wave * 16; g0 < ngroups; g0 += nwaves * 16) {
    unsigned long long mine = 0;
    if (lane < 16 && g0 + lane < ngroups) mine = A.todo_mask[g0 + lane];
    uint64_t any = __ballot(mine != 0);
    while (any) {
      const int k = __builtin_ctzll(any);
      any &= any - 1;
      const uint64_t m = ((uint64_t) (uint32_t) __builtin_amdgcn_readlane((int) (mine >> 32), k) << 32) |
                         (uint32_t) __builtin_amdgcn_readlane((int) mine, k);
      if ((m >> lane) & 1) {
        const uint32_t r = rank_below(m);
        queue[wib][nq + r] = (uint32_t) (((g0 + k) << gl_shift) + lane);
        qrec[wib][nq + r] = (uint32_t) ((g0 + k) << gl_shift) + r;
      }
      nq += __popcll(m);
      wave_lds_fence();
      if (nq >= 64) {
        evaluate(queue[wib][nq - 64 + lane], qrec[wib][nq - 64 + lane]);
        nq -= 64;
        wave_lds_fence();
      }
    }
  }
  if (lane < nq) evaluate(queue[wib][lane], qrec[wib][lane]);
}

// =============================================================================================
// launch wrappers
// =============================================================================================
__global__ void k_noop() {}
hipError_t warm_query_kernels(hipStream_t st) {  // (see warm_stitch_kernels)
  hipLaunchKernelGGL(k_noop, dim3(1), dim3(1), 0, st);
  return hipGetLastError();
}

static thread_local LaunchNote g_note = {"", 0, 0};
LaunchNote last_launch() { return g_note; }
static inline void note(const char* kernel, int grid, int per_lane) { g_note.kernel = kernel; g_note.grid = grid; g_note.per_lane = per_lane; }

static inline int grid_for(uint64_t work_items, int per_block, int max_blocks) {
  uint64_t b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > (uint64_t) max_blocks) b = max_blocks;
  return (int) b;
}

hipError_t launch_build_segs(hipStream_t st, const int64_t* pts, const uint32_t* edge_begin,
                             uint32_t nc, uint64_t ne, Seg* seg, uint32_t* edge_chain, uint32_t* ccode) {
  if (ne == 0) return hipSuccess;
  hipLaunchKernelGGL(k_build_segs, dim3(grid_for(ne, 256, 8192)), dim3(256), 0, st, pts, edge_begin,
                     nc, ne, seg, edge_chain, ccode);
  return hipGetLastError();
}

hipError_t launch_morton(hipStream_t st, const Seg* seg, uint64_t ne, MortonKey* keys, uint32_t* vals) {
  if (ne == 0) return hipSuccess;
  hipLaunchKernelGGL(k_morton, dim3(grid_for(ne, 256, 8192)), dim3(256), 0, st, seg, ne, keys, vals);
  return hipGetLastError();
}

hipError_t sort_pairs_u64_u32(hipStream_t st, void* temp, size_t& temp_bytes, const uint64_t* kin,
                              uint64_t* kout, const uint32_t* vin, uint32_t* vout, uint64_t n, unsigned begin_bit,
                              unsigned end_bit) {
  return rocprim::radix_sort_pairs(temp, temp_bytes, kin, kout, vin, vout, (size_t) n, begin_bit, end_bit, st);
}

hipError_t sort_morton_pairs(hipStream_t st, void* temp, size_t& temp_bytes, const MortonKey* kin, MortonKey* kout,
                             const uint32_t* vin, uint32_t* vout, uint64_t n) {
  return rocprim::radix_sort_pairs(temp, temp_bytes, kin, kout, vin, vout, (size_t) n, 0, 64 - kMortonDropBits, st);
}

hipError_t sort_keys_u64(hipStream_t st, void* temp, size_t& temp_bytes, const uint64_t* kin,
                         uint64_t* kout, uint64_t n) {
  return rocprim::radix_sort_keys(temp, temp_bytes, kin, kout, (size_t) n, 0, 64, st);
}

hipError_t launch_xsect_keys(hipStream_t st, const XsectRec* rec, uint64_t n, int im, uint64_t* keys, uint32_t* vals) {
  if (n) hipLaunchKernelGGL(k_xsect_keys, dim3(grid_for(n, 256, 4096)), dim3(256), 0, st, rec, n, im, keys, vals);
  return hipGetLastError();
}
hipError_t launch_xsect_gather(hipStream_t st, const XsectRec* in, const uint32_t* order, uint64_t n, XsectRec* out) {
  if (n) hipLaunchKernelGGL(k_xsect_gather, dim3(grid_for(n, 256, 4096)), dim3(256), 0, st, in, order, n, out);
  return hipGetLastError();
}
hipError_t launch_xsect_order_runs(hipStream_t st, XsectRec* rec, uint64_t n, int im, const Seg* seg_im, int64_t* midpts) {
  if (n) hipLaunchKernelGGL(k_xsect_order_runs, dim3(grid_for(n, 256, 4096)), dim3(256), 0, st, rec, n, im, seg_im, midpts);
  return hipGetLastError();
}
hipError_t launch_xsect_set_mid(hipStream_t st, XsectRec* rec, uint64_t n, int im, const int32_t* face) {
  if (n) hipLaunchKernelGGL(k_xsect_set_mid, dim3(grid_for(n, 256, 4096)), dim3(256), 0, st, rec, n, im, face);
  return hipGetLastError();
}

hipError_t launch_swap_halves(hipStream_t st, uint64_t* v, uint64_t n) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_swap_halves, dim3(grid_for(n, 256, 8192)), dim3(256), 0, st, v, n);
  return hipGetLastError();
}

hipError_t launch_run_keys(hipStream_t st, const Seg* seg, const uint32_t* piece_begin, const uint32_t* piece_len, const uint32_t* run_first,
                           uint64_t nruns, MortonKey* keys, uint32_t* vals, uint32_t* run_len, QBox* run_box, uint32_t box_upto) {
  if (nruns == 0) return hipSuccess;
  hipLaunchKernelGGL(k_run_keys, dim3(grid_for(nruns, 256, 8192)), dim3(256), 0, st, seg, piece_begin, piece_len, run_first, nruns, keys, vals,
                     run_len, run_box, box_upto);
  return hipGetLastError();
}

uint64_t pack_runs_chunks(uint64_t nruns) { return (nruns + kPackChunk - 1) / kPackChunk; }
// order[nruns] (sorted run ids), run_len / run_box[nruns] -> leaf_first[leaves + 1]; chunk_leaves / chunk_base: scratch of
// pack_runs_chunks(nruns) + 1 words each; *total_out = the number of leaves (written on the stream)
hipError_t launch_pack_runs(hipStream_t st, const uint32_t* order, const uint32_t* run_len, const QBox* run_box, uint64_t nruns,
                            uint32_t solo_above, uint32_t spread, uint32_t* chunk_leaves, uint32_t* chunk_base, uint32_t* leaf_first,
                            unsigned long long* total_out) {
  if (nruns == 0) return hipSuccess;
  const uint64_t nchunks = pack_runs_chunks(nruns);
  const int grid = grid_for(nchunks, 4, 8192);  // (a wave per chunk)
  hipLaunchKernelGGL(k_pack_runs<false>, dim3(grid), dim3(256), 0, st, order, run_len, run_box, nruns, solo_above, spread, chunk_leaves,
                     (const uint32_t*) nullptr, (uint32_t*) nullptr);
  hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, st, chunk_leaves, chunk_base, nchunks, total_out);
  hipLaunchKernelGGL(k_pack_runs<true>, dim3(grid), dim3(256), 0, st, order, run_len, run_box, nruns, solo_above, spread, (uint32_t*) nullptr,
                     chunk_base, leaf_first);
  hipLaunchKernelGGL(k_pack_end, dim3(1), dim3(1), 0, st, leaf_first, chunk_base, nchunks, (uint32_t) nruns);
  return hipGetLastError();
}

hipError_t launch_build_leaves(hipStream_t st, const Seg* seg, const uint32_t* order, const uint32_t* edge_chain,
                               const uint32_t* left, const uint32_t* right, uint64_t ne, const uint32_t* piece_begin,
                               const uint32_t* piece_len, const uint32_t* run_first, const uint32_t* run_len, const uint32_t* leaf_first, uint64_t nblocks,
                               uint64_t n_parent_alloc, Seg* sseg, uint32_t* seid, int32_t* sface, QBox* box0,
                               int32_t* pmx1, uint2* xtab, QBox* lvl1, uint32_t* occ, uint2* ytab2) {
  hipLaunchKernelGGL(k_build_leaves, dim3(grid_for(n_parent_alloc, 4, 16384)), dim3(256), 0, st, seg, order, edge_chain,
                     left, right, ne, piece_begin, piece_len, run_first, run_len, leaf_first, nblocks, n_parent_alloc, sseg, seid, sface, box0, pmx1, xtab, lvl1, occ,
                     ytab2);
  return hipGetLastError();
}

hipError_t launch_occ_count(hipStream_t st, const uint32_t* occ, unsigned long long* part16, unsigned long long* out_mapped) {
  hipLaunchKernelGGL(k_occ_count, dim3(16), dim3(1024), 0, st, occ, part16);
  hipLaunchKernelGGL(k_occ_sum, dim3(1), dim3(1), 0, st, part16, 16, out_mapped);
  return hipGetLastError();
}

hipError_t launch_build_sky(hipStream_t st, const QBox* box0, const uint32_t* seid, uint64_t n0p, uint32_t* sky) {
  hipLaunchKernelGGL(k_build_sky, dim3(grid_for(n0p, 256, 8192)), dim3(256), 0, st, box0, seid, n0p, sky);
  return hipGetLastError();
}

hipError_t launch_sibling_order(hipStream_t st, const QBox* box, uint64_t n_alloc, uint64_t* higher) {
  hipLaunchKernelGGL(k_sibling_order, dim3(grid_for(n_alloc / 64, 4, 8192)), dim3(256), 0, st, box, n_alloc, higher);
  return hipGetLastError();
}

hipError_t launch_reduce_level(hipStream_t st, const QBox* child, uint64_t n_child_alloc, QBox* parent,
                               uint64_t n_parent_alloc) {
  hipLaunchKernelGGL(k_reduce_level, dim3(grid_for(n_parent_alloc, 4, 8192)), dim3(256), 0, st, child,
                     n_child_alloc, parent, n_parent_alloc);
  return hipGetLastError();
}

static int resident_blocks(const void* kernel, int max_blocks, int block_threads = 256) {
  int dev = 0, cus = 256, per_cu = 4;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block_threads, 0) != hipSuccess || per_cu < 1) per_cu = 4;
  int b = cus * per_cu;
  return b < max_blocks ? b : max_blocks;
}

// Small query sets: with 64 queries per wave there are fewer groups than resident waves and each
// wave walks a long serial chain of node visits.  Fewer queries per wave spread the same visits
// over more waves (the kernels are latency-bound, idle lanes cost nothing).
static uint32_t pick_group_lanes(uint64_t nqueries, int resident_blocks_, int block_waves = 4) {
  const uint64_t waves = (uint64_t) resident_blocks_ * block_waves;
  uint32_t gl = 64;
  while (gl > 4 && (nqueries + gl - 1) / gl < 2 * waves) gl >>= 1;
  return gl;
}

// Consecutive groups handed to a wave at a time: k_lsi 8, the PIP kernels 6 (measured optima; a chunk's groups share
// their tree nodes in the caches and, in the PIP kernels, a block's waves share a chunk's rest).  Smaller chunks for small
// query sets -- so that the last chunk handed out is a smaller part of a wave's work -- were tried on the 1/8 shard of
// the headline pair (9 groups per walk wave): 1 group per chunk, walk 169 -> 232 us.
static uint32_t pick_chunk_groups(uint32_t requested, uint32_t optimum) { return requested ? requested : optimum; }

hipError_t launch_lsi(hipStream_t st, const LsiArgs& a_in, bool stats, int max_blocks, int segs_per_lane, int* segs_used) {
  LsiArgs a = a_in;
  if (segs_used) *segs_used = 1;
  const bool ys = a.bvh.ytab2 != nullptr;  // (the tree has a second order of its steep blocks: the kernels that use it)
  const void* k = stats ? (const void*) k_lsi<true> : (const void*) k_lsi<false>;
  static int res[2] = {0, 0};
  if (!res[stats]) res[stats] = resident_blocks(k, 1 << 20);
  const bool auto_lanes = !a.group_lanes;
  if (!a.group_lanes) a.group_lanes = pick_group_lanes(a.qend - a.qbeg, res[stats]);
  // two segments per lane (k_lsi2) where the query set fills 64-segment groups twice over and nobody is counting visits
  if (segs_per_lane == 2 && !stats && auto_lanes && a.group_lanes == 64 && !a.chunk_groups &&
      pick_group_lanes((a.qend - a.qbeg) / 2, res[0]) == 64) {
    static int res2 = 0;
    if (!res2) res2 = resident_blocks((const void*) k_lsi2, 1 << 20);
    const uint64_t ngroups = (a.qend - a.qbeg + 127) / 128;
    a.chunk_groups = 4;  // (128-segment groups: the same 8 x 64 positions per chunk)
    const uint64_t nchunks = (ngroups + a.chunk_groups - 1) / a.chunk_groups;
    int grid = grid_for(nchunks, 4, res2 < max_blocks ? res2 : max_blocks);
    // (the small-query rule below, for groups of twice the size: ten groups per resident wave -- tuned on the headline, where
    //  two thirds of the groups are dismissed by the pre-filter.  Over a DENSE occupancy bitmap nearly every group traverses and
    //  that is too few waves: five -- the 1/8 shard of WaterBodies x BlockGroup, k_lsi2 0.248 -> 0.200 ms, the shard's
    //  pipelined step 0.354 -> 0.323)
    const uint64_t per_wave = a.bvh.occ_permille >= kOccDensePermille ? 5 : 10;
    const uint64_t by_work = ngroups / (4 * per_wave);
    const int floor_blocks = 512 < grid ? 512 : grid;
    if (by_work < (uint64_t) grid) grid = by_work > (uint64_t) floor_blocks ? (int) by_work : floor_blocks;
    if (ys) { note("k_lsi2", grid, 2); hipLaunchKernelGGL(k_lsi2, dim3(grid), dim3(256), 0, st, a); }
    else { note("k_lsi2x", grid, 2); hipLaunchKernelGGL(k_lsi2x, dim3(grid), dim3(256), 0, st, a); }
    if (segs_used) *segs_used = 2;
    return hipGetLastError();
  }
  uint64_t ngroups = (a.qend - a.qbeg + a.group_lanes - 1) / a.group_lanes;
  a.chunk_groups = pick_chunk_groups(a.chunk_groups, 8);
  uint64_t nchunks = (ngroups + a.chunk_groups - 1) / a.chunk_groups;
  int grid = grid_for(nchunks, 4, res[stats] < max_blocks ? res[stats] : max_blocks);
  // A small query set (a shard of an 8-GPU run) is faster on FEWER resident waves: with under ~20
  // groups per wave the kernel is all ramp -- every wave's first groups miss the caches together --
  // and two thirds of the groups are dismissed by the pre-filter anyway.  Measured (1/16, 1/8, 1/4
  // of the headline query map): 768-1024 blocks beat the full grid by 22 / 18 / 7 %; from 1/2 up the
  // full grid wins.  (k_pip is fastest on the full grid at every size.)
  const uint64_t by_work = ngroups / (4 * 20);
  const int floor_blocks = 512 < grid ? 512 : grid;
  if (by_work < (uint64_t) grid) grid = by_work > (uint64_t) floor_blocks ? (int) by_work : floor_blocks;
  if (stats)
    { note("k_lsi (instrumented)", grid, 1); if (ys) hipLaunchKernelGGL(k_lsi<true>, dim3(grid), dim3(256), 0, st, a); else hipLaunchKernelGGL(k_lsix<true>, dim3(grid), dim3(256), 0, st, a); }
  else if (ys)
    { note("k_lsi", grid, 1); hipLaunchKernelGGL(k_lsi<false>, dim3(grid), dim3(256), 0, st, a); }
  else
    { note("k_lsix", grid, 1); hipLaunchKernelGGL(k_lsix<false>, dim3(grid), dim3(256), 0, st, a); }
  return hipGetLastError();
}

hipError_t launch_group_extent(hipStream_t st, bool points, const int64_t* pts, const Seg* segs, uint64_t begin,
                               uint64_t n, unsigned long long* out2) {
  const uint64_t ngroups = (n + 63) / 64;
  const uint64_t stride = ngroups > 8192 ? ngroups / 8192 : 1;  // sample <= ~8192 groups
  const int grid = grid_for((ngroups + stride - 1) / stride, 4, 512);
  if (points)
    hipLaunchKernelGGL(k_group_extent<true>, dim3(grid), dim3(256), 0, st, pts, segs, begin, n, stride, out2);
  else
    hipLaunchKernelGGL(k_group_extent<false>, dim3(grid), dim3(256), 0, st, pts, segs, begin, n, stride, out2);
  return hipGetLastError();
}

hipError_t launch_group_extent_tail(hipStream_t st, const int64_t* pts, const uint32_t* order, uint64_t n, unsigned long long* out_mapped) {
  hipLaunchKernelGGL(k_group_extent_tail, dim3(1), dim3(1024), 0, st, pts, order, n, out_mapped);
  return hipGetLastError();
}

hipError_t launch_query_keys(hipStream_t st, bool points, const int64_t* pts, const Seg* segs, uint64_t begin,
                             uint64_t n, MortonKey* keys, uint32_t* vals, int strip_shift) {
  if (n == 0) return hipSuccess;
  if (points)
    hipLaunchKernelGGL(k_query_keys<true>, dim3(grid_for(n, 256, 8192)), dim3(256), 0, st, pts, segs, begin, n, keys, vals, strip_shift);
  else
    hipLaunchKernelGGL(k_query_keys<false>, dim3(grid_for(n, 256, 8192)), dim3(256), 0, st, pts, segs, begin, n, keys, vals, 0);
  return hipGetLastError();
}

hipError_t launch_lsi_points(hipStream_t st, const Seg* seg0, const Seg* seg1, const uint32_t* pairs,
                             uint64_t n, const unsigned long long* n_dev, XsectRec* out, uint32_t* slow_list,
                             unsigned long long* slow_count, unsigned long long* next_slow_count, unsigned long long* count_hint) {
  if (n == 0) return hipSuccess;
  // with a device-side count the grid is sized for a typical result, the loop is grid-stride anyway
  const uint64_t expect = n_dev ? (n < (1u << 20) ? n : (1u << 20)) : n;
  if (!slow_list || n >= (1ull << 32)) {
    hipLaunchKernelGGL(k_lsi_points_gcd, dim3(grid_for(expect, 256, 4096)), dim3(256), 0, st, seg0, seg1, pairs, n, n_dev,
                       (const uint32_t*) nullptr, (const unsigned long long*) nullptr, count_hint, out);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(k_lsi_points, dim3(grid_for(expect, 256, 2048)), dim3(256), 0, st, seg0, seg1, pairs, n, n_dev, slow_list,
                     slow_count, next_slow_count, count_hint, out);
  // (a few per cent of the pairs at most on map-like data; sized for that, grid-stride for the rest)
  hipLaunchKernelGGL(k_lsi_points_gcd, dim3(grid_for(expect / 16 + 1, 256, 1024)), dim3(256), 0, st, seg0, seg1, pairs, n, n_dev,
                     (const uint32_t*) slow_list, (const unsigned long long*) slow_count, (unsigned long long*) nullptr, out);
  return hipGetLastError();
}

int pip_walk_list_slots() { return kWalkList; }

int pip_walk_blocks_per_cu(int top) {
  const size_t block = 4 * walk_wave_lds(top) + 64;
  const size_t by_lds = (size_t) 160 * 1024 / block;
  return (int) (by_lds < 8 ? by_lds : 8);
}

uint32_t pip_walk_group_lanes(uint64_t n, int top, int cus) { return pick_group_lanes(n, cus * pip_walk_blocks_per_cu(top), 4); }

hipError_t launch_pip_walk(hipStream_t st, const PipArgs& a_in, bool stats, int max_blocks) {
  PipArgs a = a_in;
  const size_t lds = 4 * walk_wave_lds(a.bvh.top);
  static int cus = 0;  // (asked once: the query is not free, and this sits in every step)
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
  }
  const int res = cus * pip_walk_blocks_per_cu(a.bvh.top);
  if (!a.group_lanes) a.group_lanes = pick_group_lanes(a.n, res, 4);
  const uint64_t ngroups = (a.n + a.group_lanes - 1) / a.group_lanes;
  a.chunk_groups = pick_chunk_groups(a.chunk_groups, 6);
  const uint64_t nchunks = (ngroups + a.chunk_groups - 1) / a.chunk_groups;
  const int grid = grid_for(nchunks, 4, res < max_blocks ? res : max_blocks);
  if (stats)
    { note("k_pip_walk (instrumented)", grid, 1); hipLaunchKernelGGL(k_pip_walk<true>, dim3(grid), dim3(256), lds, st, a); }
  else
    { note("k_pip_walk", grid, 1); hipLaunchKernelGGL(k_pip_walk<false>, dim3(grid), dim3(256), lds, st, a); }
  return hipGetLastError();
}

int pip_walk2_blocks_per_cu(int top, int points) {
  const size_t block = 4 * walk2_wave_lds(top, points == 4 ? 4 : 2) + 64;
  const size_t by_lds = (size_t) 160 * 1024 / block;
  const size_t cap = points == 4 ? 4 : 8;  // (k_pip_walk2<*, 4> is built for four blocks per CU: its registers)
  return (int) (by_lds < cap ? by_lds : cap);
}

// ... beside `lsi_blocks_per_cu` resident blocks of k_lsi (17.5 KiB of LDS and one wave slot per SIMD each)
int pip_walk2_blocks_beside(int top, int lsi_blocks_per_cu, int points) {
  const int P = points == 4 ? 4 : 2;
  const size_t block = 4 * walk2_wave_lds(top, P) + 64;
  const size_t left = (size_t) 160 * 1024 > (size_t) lsi_blocks_per_cu * 17920 ? (size_t) 160 * 1024 - (size_t) lsi_blocks_per_cu * 17920 : 0;
  int n = (int) (left / block);
  if (n > 8 - lsi_blocks_per_cu) n = 8 - lsi_blocks_per_cu;
  // ... and of a SIMD's 512 VGPRs 80 per wave of k_lsi2 and 56 per wave of the walk (it is kept to that: the point
  // indices are read again at the end, base + lane addresses are not held): 6 beside 2.  (At 64 it was 5 beside 2.
  // Tried: k_lsi2 held to 64 VGPRs instead -- 14 spilled registers make it 48 % slower.)
  static int walk_regs[2] = {0, 0}, lsi_regs = 0;
  if (!walk_regs[0]) {
    hipFuncAttributes fa;
    walk_regs[0] = hipFuncGetAttributes(&fa, (const void*) k_pip_walk2) == hipSuccess && fa.numRegs > 0 ? (fa.numRegs + 7) / 8 * 8 : 64;
    walk_regs[1] = hipFuncGetAttributes(&fa, (const void*) k_pip_walk4) == hipSuccess && fa.numRegs > 0 ? (fa.numRegs + 7) / 8 * 8 : 96;
    lsi_regs = hipFuncGetAttributes(&fa, (const void*) k_lsi2) == hipSuccess && fa.numRegs > 0 ? (fa.numRegs + 7) / 8 * 8 : 80;
  }
  const int by_vgpr = (512 - lsi_blocks_per_cu * lsi_regs) / walk_regs[P == 4];
  if (n > by_vgpr) n = by_vgpr;
  if (n > pip_walk2_blocks_per_cu(top, P)) n = pip_walk2_blocks_per_cu(top, P);
  return n < 1 ? 1 : n;
}

hipError_t launch_pip_walk2(hipStream_t st, const PipArgs& a_in, int max_blocks, int cus, bool stats, int points) {
  PipArgs a = a_in;
  const int P = points == 4 ? 4 : 2;
  const size_t lds = 4 * walk2_wave_lds(a.bvh.top, P);
  const int res = cus * pip_walk2_blocks_per_cu(a.bvh.top, P);
  a.group_lanes = 64;  // (positions per mask; a wave takes P of them)
  const uint64_t ngroups = (a.n + 64 * P - 1) / (64 * P);
  a.chunk_groups = a.chunk_groups ? a.chunk_groups : (P == 2 ? 3 : 2);  // (128-point groups: the same 6 x 64 positions per chunk; 256-point groups: 8 x 64)
  const uint64_t nchunks = (ngroups + a.chunk_groups - 1) / a.chunk_groups;
  const int grid = grid_for(nchunks, 4, res < max_blocks ? res : max_blocks);
  note(P == 4 ? "k_pip_walk4" : "k_pip_walk2", grid, P);
  if (P == 4) {
    if (stats)
      hipLaunchKernelGGL(k_pip_walk4_stats, dim3(grid), dim3(256), lds, st, a);
    else
      hipLaunchKernelGGL(k_pip_walk4, dim3(grid), dim3(256), lds, st, a);
  } else if (stats) {
    hipLaunchKernelGGL(k_pip_walk2_stats, dim3(grid), dim3(256), lds, st, a);
  } else {
    hipLaunchKernelGGL(k_pip_walk2, dim3(grid), dim3(256), lds, st, a);
  }
  return hipGetLastError();
}

hipError_t launch_pip_exact(hipStream_t st, const PipArgs& a, int blocks, const PipRestArgs& r) {
  hipLaunchKernelGGL(k_pip_exact, dim3((blocks < 1 ? 1 : blocks) + r.blocks), dim3(256), 0, st, a, r);
  return hipGetLastError();
}

hipError_t launch_pip(hipStream_t st, const PipArgs& a_in, bool stats, int max_blocks) {
  PipArgs a = a_in;
  const void* k = stats ? (const void*) k_pip<true> : (const void*) k_pip<false>;
  static int res[2] = {0, 0};
  if (!res[stats]) res[stats] = resident_blocks(k, 1 << 20, 64 * kPipWaves);
  if (!a.group_lanes) a.group_lanes = pick_group_lanes(a.n, res[stats], kPipWaves);
  uint64_t ngroups = (a.n + a.group_lanes - 1) / a.group_lanes;
  a.chunk_groups = pick_chunk_groups(a.chunk_groups, 6);
  uint64_t nchunks = (ngroups + a.chunk_groups - 1) / a.chunk_groups;
  int grid = grid_for(nchunks, kPipWaves, res[stats] < max_blocks ? res[stats] : max_blocks);
  if (stats)
    { note("k_pip (instrumented)", grid, 1); hipLaunchKernelGGL(k_pip<true>, dim3(grid), dim3(64 * kPipWaves), 0, st, a); }
  else
    { note("k_pip", grid, 1); hipLaunchKernelGGL(k_pip<false>, dim3(grid), dim3(64 * kPipWaves), 0, st, a); }
  return hipGetLastError();
}

}  // namespace rj
