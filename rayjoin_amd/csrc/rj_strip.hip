// rj_strip.hip -- the column index of an indexed map and the PIP pass that runs on it (rj_device.h DeviceStrips).
//
// Built for maps of isolated rings, where the box hierarchy is at its worst for upward rays: a leaf block of a few
// rings is mostly gaps in x, and a point opens every block over its column whose x-extent contains it until one
// holds an edge at its x -- measured on the lake-shaped stand-in: 20 leaf blocks opened per point, one segment box
// tested.  Here a point reads its strip's list: one table read for its height (1024 buckets per strip), then the
// entries' boxes upwards -- consecutive 16-byte reads -- until a certain hit bounds the answer.  (First version: a
// binary search over 8-byte keys, then key -> slot -> box per entry: 4.9 ms for the 29.7 M lattice vertices.)  The pass is k_pip_walk's in every other respect -- integer tests only, the same
// certain-hit pruning, the same hand-over (settled points written, candidate lists in `todo`, overflowed lists in
// `rest`) -- so k_pip_exact follows it unchanged and the results are the walk's (tests/test_gpu_strip.py).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/reverse_iterator.hpp>

#include "rj_kernels.h"

namespace rj {

namespace {

#define RJ_GRID_STRIDE(i, n) \
  for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < (n); i += (uint64_t) gridDim.x * blockDim.x)

inline int blocks_for(uint64_t n, int per_block = 256, int cap = 16384) {
  uint64_t b = (n + per_block - 1) / per_block;
  return (int) (b < 1 ? 1 : (b > (uint64_t) cap ? cap : b));
}

// how many strips the box of sorted slot i touches (0: a padding slot); flag[0] = 1 when one is too wide to register
__global__ __launch_bounds__(256) void k_strip_count(const QBox* __restrict__ box0, const uint32_t* __restrict__ seid, uint64_t n0p,
                                                     int shift, uint32_t* __restrict__ cnt, uint32_t* __restrict__ flag) {
  RJ_GRID_STRIDE(i, n0p) {
    uint32_t c = 0;
    if (seid[i] != 0xFFFFFFFFu) {
      const QBox b = box0[i];
      c = (uint32_t) ((b.x1 >> shift) - (b.x0 >> shift) + 1);
      if (c > (uint32_t) kStripMaxSpan) { c = 0; flag[0] = 1u; }
    }
    cnt[i] = c;
  }
}
// The sort key of an entry, 32 bits: its strip above the BAND of its y0 (2^15 quanta).  The column's entries ascend by
// band, not by y0 itself: the scan stops at the first band above the point's bound and tests every entry of the bands
// before -- a handful more than the exact order would -- while the key is half as wide and two radix passes shorter
// (round 4: 64-bit keys of 47 bits, six passes -- most of the index's 5-11 ms).
constexpr int kStripBandShift = 15, kStripBandBits = 31 - kStripBandShift;
__device__ __forceinline__ uint32_t strip_key(uint32_t strip, int32_t y0) { return (strip << kStripBandBits) | ((uint32_t) y0 >> kStripBandShift); }
__global__ __launch_bounds__(256) void k_strip_emit(const QBox* __restrict__ box0, const uint32_t* __restrict__ cnt,
                                                    const uint32_t* __restrict__ offs, uint64_t n0p, int shift, uint32_t* __restrict__ key,
                                                    uint32_t* __restrict__ slot) {
  RJ_GRID_STRIDE(i, n0p) {
    const uint32_t c = cnt[i];
    if (!c) continue;
    const QBox b = box0[i];
    const uint32_t s0 = (uint32_t) (b.x0 >> shift);
    const uint32_t o = offs[i];
    for (uint32_t k = 0; k < c; k++) {
      key[o + k] = strip_key(s0 + k, b.y0);
      slot[o + k] = (uint32_t) i;
    }
  }
}
// the sorted entries: their boxes beside them, and where every (strip, height bucket) starts.  The first entry of a
// bucket writes its index at the bucket (the table is pre-filled with "no entry": 0xFFFFFFFF, the slot behind the last
// bucket with the entry count); a suffix minimum over the table then gives every empty bucket the first entry behind
// it.  (Filling the gaps from the entries themselves left one thread writing millions of buckets where the map is a
// small cluster in a large domain: +7 ms on the gaussian polygons.)
__device__ __forceinline__ uint32_t strip_bucket(uint32_t key) {
  return ((key >> kStripBandBits) << kStripYBits) | ((key & ((1u << kStripBandBits) - 1u)) >> (kStripYShift - kStripBandShift));
}
// ... and every strip's tallest box (tall[s]: what a point's scan must start below itself by) and, where asked for, the
// map's SKYLINE (rj_device.h kSkyShift: per 2^13-quanta bucket 1 + the highest y1 of any box over it), from data this
// pass holds anyway.  The entries of a strip are consecutive: the maxima are taken over each wave's runs of equal strip
// first -- one atomicMax per run and bucket, not per entry.  (Round 4 raised `tall` from k_strip_emit, a look
// at the L2 per entry, and the skyline from k_build_leaves, one per segment and bucket: 0.96 ms of the lake-shaped
// map's 9.2 ms first build.)
__global__ __launch_bounds__(256) void k_strip_finish(const uint32_t* __restrict__ key, const uint32_t* __restrict__ slot, uint64_t n,
                                                      const QBox* __restrict__ box0, const uint32_t* __restrict__ seid,
                                                      const int32_t* __restrict__ sface, QBox* __restrict__ ebox,
                                                      uint4* __restrict__ einfo, uint32_t* __restrict__ ytab, uint32_t strips,
                                                      uint32_t* __restrict__ tall, uint32_t* __restrict__ sky, int shift) {
  const int lane = (int) (threadIdx.x & 63);
  const int nsub = 1 << (shift - kSkyShift);  // skyline buckets per strip: 4, 8 or 16
  for (uint64_t base = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x - (uint64_t) lane; base < n; base += (uint64_t) gridDim.x * blockDim.x) {
    const uint64_t j = base + (uint64_t) lane;
    uint32_t strip = 0xFFFFFFFFu, hgt = 0, tp = 0;
    int k0 = 0, k1 = -1;  // the skyline buckets of the strip this entry's box lies over
    if (j < n) {
      const uint32_t sl = slot[j];
      const QBox b = box0[sl];
      ebox[j] = b;
      einfo[j] = make_uint4(sl, seid[sl], (uint32_t) sface[sl], 0u);
      const uint32_t kj = key[j];
      const uint32_t g = strip_bucket(kj);
      if (j == 0 || strip_bucket(key[j - 1]) != g) ytab[g] = (uint32_t) j;
      if (j == n - 1) ytab[strips << kStripYBits] = (uint32_t) n;
      strip = kj >> kStripBandBits;
      hgt = (uint32_t) (b.y1 - b.y0);
      tp = (uint32_t) b.y1 + 1u;
      const int first = (int) (strip << (shift - kSkyShift));
      k0 = (b.x0 >> kSkyShift) - first; k1 = (b.x1 >> kSkyShift) - first;
      k0 = k0 < 0 ? 0 : k0; k1 = k1 >= nsub ? nsub - 1 : k1;
    }
    const uint32_t before = (uint32_t) __shfl_up((int) strip, 1, 64);
    const bool head = j < n && (lane == 0 || before != strip);
    bool same[6];  // lane + 2^i holds an entry of the same strip
#pragma unroll
    for (int i = 0; i < 6; i++) {
      const uint32_t os = (uint32_t) __shfl_down((int) strip, 1 << i, 64);  // (read by ALL lanes: not behind the && below)
      same[i] = lane + (1 << i) < 64 && os == strip;
    }
#pragma unroll
    for (int i = 0; i < 6; i++) {
      const uint32_t oh = (uint32_t) __shfl_down((int) hgt, 1 << i, 64);
      if (same[i]) hgt = oh > hgt ? oh : hgt;
    }
    if (head) atomicMax(&tall[strip], hgt);
    if (sky) {
      for (int k = 0; k < nsub; k++) {
        uint32_t v = (k0 <= k && k <= k1) ? tp : 0u;
#pragma unroll
        for (int i = 0; i < 6; i++) {
          const uint32_t ov = (uint32_t) __shfl_down((int) v, 1 << i, 64);
          if (same[i]) v = ov > v ? ov : v;
        }
        // (no look before the atomic here: a wave would wait for nsub dependent round trips to the L2 per 64 entries --
        //  1.8 ms of the lake-shaped map's build -- where an atomic without a return value is not waited for at all)
        if (head && v) atomicMax(&sky[(strip << (shift - kSkyShift)) + (uint32_t) k], v);
      }
    }
  }
}

// every strip's tallest box and, beside it, where its entries end (the table entry of the next strip's first bucket,
// after the suffix minimum): one 8-byte read per query point instead of two lines
__global__ __launch_bounds__(256) void k_strip_ends(const uint32_t* __restrict__ tall, const uint32_t* __restrict__ ytab, uint32_t strips,
                                                    uint2* __restrict__ out) {
  RJ_GRID_STRIDE(s, (uint64_t) strips) out[s] = make_uint2(tall[s], ytab[(s + 1) << kStripYBits]);
}
// the sum of the x-extents of one slot of every 64-slot block, and how many were summed (out[0], out[1]): what the
// strip width is chosen by.  (Which slot varies from block to block: the blocks are sorted by x0 inside.  Every 8th slot,
// as in round 4, touched every line of the boxes: 0.28 ms of a 9 ms build for a mean of 5 M samples where 0.6 M do.)
__global__ __launch_bounds__(256) void k_strip_width(const QBox* __restrict__ box0, const uint32_t* __restrict__ seid, uint64_t n0p,
                                                     unsigned long long* __restrict__ out) {
  unsigned long long w = 0, c = 0;
  RJ_GRID_STRIDE(i, n0p / 64) {
    const uint64_t k = i * 64 + ((i * 37) & 63);
    if (seid[k] != 0xFFFFFFFFu) { const QBox b = box0[k]; w += (unsigned long long) (b.x1 - b.x0); c++; }
  }
  for (int o = 32; o > 0; o >>= 1) { w += __shfl_down(w, o, 64); c += __shfl_down(c, o, 64); }
  // (one pair of atomics per BLOCK, on a grid of a block per CU or two: 16 k atomics on one address -- a pair per wave of
  //  2048 blocks -- were 0.19 of the kernel's 0.21 ms)
  __shared__ unsigned long long part[2][4];
  if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = w; part[1][threadIdx.x >> 6] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long ws = part[0][0] + part[0][1] + part[0][2] + part[0][3], cs = part[1][0] + part[1][1] + part[1][2] + part[1][3];
    if (cs) { atomicAdd(&out[0], ws); atomicAdd(&out[1], cs); }
  }
}

// PTS query points per lane: a wave takes PTS x 64 consecutive positions (point set p = positions 64 p + lane of the
// group; one todo mask per 64 positions, as the walk writes them).  The scans of a lane's points are independent chains
// of dependent reads -- table, box, box, ... -- so running PTS of them side by side multiplies the reads in flight per
// wave (the kernel has all 8 waves per SIMD already and is bound by the latency of those reads).
// (Round 5, tried and dropped: the walk kernels' cure -- no load under a per-lane branch, the sets' loads in batches, U
//  consecutive entries per scan step walked in registers, {slot, edge, face} looked up once at the hand-over.  Bit-exact,
//  and 1.5-1.7x SLOWER: first pass alone 2.33 -> 3.98 / 3.59 ms at U = 1 / 4 on the lake-shaped pair, 1.96 -> 3.05 / 2.70
//  lakes x parks.  A lane that is done must stop reading, and it can only do so behind a branch: the pass is bound by
//  the number of lane-reads the texture path serves, not by how long a wave waits for them.)
// (Round 5, tried and dropped: every XCD on one contiguous eighth of the query set instead of the plain stride over the
//  grid, so that neighbouring groups meet in ONE L2 -- lake-shaped pair 1.79 -> 2.04 ms in the step, lakes x parks 1.58 ->
//  1.59: the eighths are unequal and nothing was being fetched eight times.)
// (entries a point reads on its own before the wave takes its scan over.  First pass, ms, at 8 / 12 / 16 / 24 / 32 / 48 / 64 /
//  never: lake-shaped base x lattice vertices 2.70 / 2.32 / 2.14 / 1.92 / 1.80 / 1.86 / 1.95 / 2.13; lakes x parks 2.40 / 1.87 /
//  1.68 / 1.61 / 1.62 / 1.70 / 1.77 / 1.94; gaussian polygons 1.27 / 0.93 / 0.75 / 0.62 / 0.59 / 0.58 / 0.59 / 0.58: a
//  cooperative trip costs about three solo ones, so it pays for the 3 % of the points that need more than 32 entries --
//  which nearly every wave has -- and not for the 10 % beyond 16.  With two entries per solo trip and the records looked
//  up at the hand-over, at 16 / 24 / 32 / 48 / 64: 1.84 / 1.65 / 1.51 / 1.45 / 1.50; 1.40 / 1.30 / 1.28 / 1.30 / 1.34;
//  0.59 / 0.46 / 0.44 / 0.43 / 0.43: 48)
#ifndef RJ_STRIP_SOLO
#define RJ_STRIP_SOLO 48
#endif
constexpr int kStripSolo = RJ_STRIP_SOLO;
#ifndef RJ_STRIP_PER_TRIP
#define RJ_STRIP_PER_TRIP 2
#endif
constexpr int kStripPerTrip = RJ_STRIP_PER_TRIP;  // entries of a point read per solo trip
template <int PTS, bool STATS = false>
__global__ __launch_bounds__(256, 8) void k_pip_strip(PipArgs A) {
  __shared__ uint32_t cand_all[4][PTS * kWalkList * 64];
  const int lane = lane_id();
  const int wib = threadIdx.x >> 6;
  const DeviceBvh& T = A.bvh;
  const DeviceStrips& S = T.strips;
  const uint32_t* const sky = (T.sky && T.sky[kSkyBuckets] == 0u) ? T.sky : nullptr;
  uint32_t* const cand = cand_all[wib];  // [PTS][kWalkList][64], bank = lane
  const uint64_t per_group = (uint64_t) PTS * 64;
  const uint64_t ngroups = (A.n + per_group - 1) / per_group;
  const uint64_t wave = (blockIdx.x * (uint64_t) blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
  // (the counters of the next launch of this kind on this stream are cleared here, as every walk does: k_lsi)
  if (blockIdx.x == 0 && threadIdx.x < 8) A.next_work_counter[threadIdx.x * 32] = 0;
  if (blockIdx.x == 0 && threadIdx.x == 8) *A.next_rest_count = 0;
  for (uint64_t g = wave; g < ngroups; g += nwaves) {
    bool valid[PTS];
    uint32_t ip[PTS], j[PTS], jend[PTS], cand_base[PTS], cand_at[PTS];
    int32_t qx[PTS], qy[PTS], qym1[PTS], qbest[PTS], sure_y0[PTS];
#pragma unroll
    for (int p = 0; p < PTS; p++) {
      const uint64_t ipos = g * per_group + (uint64_t) p * 64 + lane;
      valid[p] = ipos < A.n;
      ip[p] = A.order ? (valid[p] ? A.order[ipos] : 0u) : (uint32_t) ipos;
      qx[p] = 0; qy[p] = 0;
      if (valid[p]) {
        typedef long long ll2_t __attribute__((ext_vector_type(2)));
        const ll2_t pt = __builtin_nontemporal_load(reinterpret_cast<const ll2_t*>(A.pts) + ip[p]);
        qx[p] = quant(pt.x);
        qy[p] = quant(pt.y);
      }
      qym1[p] = qy[p] > 0 ? qy[p] - 1 : 0;
      qbest[p] = 0x7FFFFFFF;
      cand_base[p] = (uint32_t) lane + (uint32_t) p * (kWalkList * 64);
      cand_at[p] = cand_base[p];
      sure_y0[p] = INT32_MIN;
      j[p] = jend[p] = 0;
    }
    // the strip's entries from the height bucket of the lowest y0 that can still reach up to the point
    // (y0 >= qy - 1 - the tallest box of the strip); what lies below the point inside that bucket fails the test.
    // (A sentinel entry behind every strip instead of reading where the strip ends was tried: a lane that is done
    //  cannot stop reading without a branch around the loads, and the pass is bound by the number of random reads:
    //  2.61 / 4.58 ms against 2.42 / 4.01.)
    // (Round 5, tried and dropped: the skyline's word, the strip's {tallest box, end} and the table's entries for the point's
    //  own bucket and the one below requested at once, without the branches -- one round trip instead of three per group,
    //  and 17-27 % SLOWER (1.51 -> 1.77, 1.25 -> 1.59, 0.44 -> 0.49 ms): the reads a miss and a position past the end do
    //  not make, and the second table word, cost this pass more than the two round trips.)
#pragma unroll
    for (int p = 0; p < PTS; p++) {
      if (valid[p] && ray_has_sky(sky, qx[p], qy[p])) {  // (above the map's skyline: a certain miss)
        const uint32_t s = (uint32_t) qx[p] >> S.shift;
        const uint2 te = S.tall[s];
        const int64_t from = (int64_t) qym1[p] - (int64_t) te.x;
        const uint32_t want = from > 0 ? (uint32_t) from : 0u;
        j[p] = S.ytab[(s << kStripYBits) | (want >> kStripYShift)];
        jend[p] = te.y;
      }
    }
    uint32_t st_coop = 0;  // STATS: trips of the cooperative phase
    uint32_t st_len[PTS], st_below[PTS], st_iter = 0;  // STATS: entries this lane's point read, those that end below it, the wave's trips
#pragma unroll
    for (int p = 0; p < PTS; p++) st_len[p] = st_below[p] = 0;
    // Every point scans its column on its own for at most kStripSolo entries (a wave's trip = one read per point set);
    // whoever is not done by then -- a tenth of the points of the lake-shaped pair, but nearly every wave has some, and a
    // wave went on until its slowest lane was done: ~60 trips for points that need 7.5 entries on average -- is finished
    // by the whole wave, 64 consecutive entries per trip (below).
    const int solo = A.walk_stack > 0 ? A.walk_stack : kStripSolo;  // ("walk_stack", a debug knob: experiments with the hand-over point)
    for (int trip = 0; trip < solo; trip += kStripPerTrip) {
      bool more = false;
      QBox bb[PTS][kStripPerTrip];
      if (STATS) st_iter++;
      // (the reads of all points first: they are what the lane waits for.  kStripPerTrip consecutive entries of a point per
      //  trip: the second is in the first one's line seven times out of eight, and the wave's trips -- the round trips to
      //  memory its groups are made of -- halve)
#pragma unroll
      for (int p = 0; p < PTS; p++)
#pragma unroll
        for (int u = 0; u < kStripPerTrip; u++)
          if (j[p] + (uint32_t) u < jend[p]) bb[p][u] = S.ebox[j[p] + (uint32_t) u];
#pragma unroll
      for (int p = 0; p < PTS; p++) {
#pragma unroll
       for (int u = 0; u < kStripPerTrip; u++) {
        if (j[p] >= jend[p]) continue;
        const QBox be = bb[p][u];
        if (STATS) { st_len[p]++; st_below[p] += be.y1 < qym1[p] ? 1u : 0u; }
        // (the entries ascend by BAND of y0, 2^15 quanta: inside a band their order is the build's)
        if ((be.y0 >> kStripBandShift) > (qbest[p] >> kStripBandShift)) { j[p] = jend[p]; continue; }  // everything further starts above the bound
        if (((qx[p] - be.x0) | (be.x1 - qx[p]) | (be.y1 - qym1[p]) | (qbest[p] - be.y0)) >= 0) {
          // k_pip_walk's bookkeeping: a certain hit (strictly inside in x, strictly above) bounds the answer; one that
          // ends below the start of the one certain hit held so far replaces it
          // (the list takes the ENTRY's index: what the entry stands for -- slot, edge id, face id -- is looked up at the
          //  hand-over, once per settled point and list slot and all at once.  Read here, inside the scan, every candidate
          //  put a memory round trip into its wave's trip: 18 % of the pass, measured with the read left out)
          const bool certain = be.x0 < qx[p] && qx[p] < be.x1 && be.y0 > qy[p];
          const bool replace = certain && be.y1 < sure_y0[p];
          const bool first = cand_at[p] == cand_base[p];
          const bool over = !replace && cand_at[p] == cand_base[p] + kWalkList * 64;
          cand[(replace || over) ? cand_base[p] : cand_at[p]] = j[p];
          sure_y0[p] = (replace || (first && certain)) ? be.y0 : INT32_MIN;
          cand_at[p] += replace ? 0u : 64u;
          const int32_t top = certain ? be.y1 + 1 : 0x7FFFFFFF;
          qbest[p] = over ? -1 : (top < qbest[p] ? top : qbest[p]);
        }
        j[p]++;
       }
       more = more || j[p] < jend[p];
      }
      if (!__ballot(more)) break;
    }
    // The long scans, one point at a time, by the whole wave: its state goes to scalar registers, lane k reads entry j + k
    // (one coalesced kilobyte), the tests run on all 64 at once.  The bookkeeping is the solo scan's in set form: the
    // certain hits of the batch lower the bound first, a candidate is an entry over the point's x that starts at or below
    // it, the one certain hit held so far is dropped if it now starts above; exactly one candidate in all, a certain one,
    // settles the point (its edge and face come from that lane), more go to the list, too many to the rest list.
#pragma unroll
    for (int p = 0; p < PTS; p++) {
      uint64_t todo_m = __ballot(j[p] < jend[p]);
      while (todo_m) {
        const int o = __builtin_ctzll(todo_m);
        todo_m &= todo_m - 1;
        const int32_t oqx = bcast(qx[p], o), oqy = bcast(qy[p], o), oqym1 = bcast(qym1[p], o);
        int32_t oqbest = bcast(qbest[p], o), osure = bcast(sure_y0[p], o);
        uint32_t oj = (uint32_t) bcast((int32_t) j[p], o);
        const uint32_t ojend = (uint32_t) bcast((int32_t) jend[p], o);
        const uint32_t obase = (uint32_t) o + (uint32_t) p * (kWalkList * 64);
        uint32_t oat = (uint32_t) bcast((int32_t) cand_at[p], o);
        bool stop = false;
        while (!stop && oj < ojend) {
          if (STATS) st_coop++;
          const uint32_t idx = oj + (uint32_t) lane;
          const bool in = idx < ojend;
          QBox e = {0, 0, 0, 0};
          if (in) e = S.ebox[idx];
          // (ascending by band: from the first entry of a band above the bound on, nothing counts)
          const uint64_t bm = __ballot(in && (e.y0 >> kStripBandShift) > (oqbest >> kStripBandShift));
          stop = bm != 0;
          const bool live = in && (bm == 0 || lane < __builtin_ctzll(bm));
          const bool pass = live && ((oqx - e.x0) | (e.x1 - oqx) | (e.y1 - oqym1)) >= 0;
          const bool certain = pass && e.x0 < oqx && oqx < e.x1 && e.y0 > oqy;
          const int32_t nb = wave_min(certain ? e.y1 + 1 : 0x7FFFFFFF);
          oqbest = nb < oqbest ? nb : oqbest;
          const bool is_cand = pass && e.y0 <= oqbest;
          const uint64_t cm = __ballot(is_cand);
          if (cm) {
            if (osure != INT32_MIN && osure > oqbest) { oat = obase; osure = INT32_MIN; }  // (the held hit lies above the new bound)
            const uint32_t nc = (uint32_t) __popcll(cm), fill = (oat - obase) >> 6;
            if (fill + nc > (uint32_t) kWalkList) {  // the list would overflow: the rest list takes the point
              oat = obase + (uint32_t) (kWalkList + 1) * 64;
              osure = INT32_MIN;
              oqbest = -1;
              stop = true;
            } else {
              if (is_cand) cand[oat + 64u * (uint32_t) rank_below(cm)] = idx;
              const int c0 = __builtin_ctzll(cm);
              const bool settles = fill == 0 && nc == 1 && ((__ballot(certain) >> c0) & 1);
              osure = settles ? bcast(e.y0, c0) : INT32_MIN;
              oat += 64u * nc;
            }
          }
          oj += 64;
        }
        if (lane == o) {
          qbest[p] = oqbest; sure_y0[p] = osure; cand_at[p] = oat;
          j[p] = jend[p];
        }
        wave_lds_fence();
      }
    }
    if (STATS && A.stats) {
      // [0] groups, [1] trips of the waves (a trip = one read per point set), [2] entries read, [3] of them ending below
      // the point, [4] the longest scan of any point, [5..12] points by scan length: 0, <= 2, <= 4, <= 8, <= 16, <= 32, <= 64, more
      if (lane == 0) { atomicAdd(&A.stats[0], 1ull); atomicAdd(&A.stats[1], (unsigned long long) st_iter); atomicAdd(&A.stats[13], (unsigned long long) st_coop); }
#pragma unroll
      for (int p = 0; p < PTS; p++) {
        if (!valid[p]) continue;
        atomicAdd(&A.stats[2], (unsigned long long) st_len[p]);
        atomicAdd(&A.stats[3], (unsigned long long) st_below[p]);
        atomicMax(&A.stats[4], (unsigned long long) st_len[p]);
        const uint32_t n = st_len[p];
        const int bin = n == 0 ? 0 : n <= 2 ? 1 : n <= 4 ? 2 : n <= 8 ? 3 : n <= 16 ? 4 : n <= 32 ? 5 : n <= 64 ? 6 : 7;
        atomicAdd(&A.stats[5 + bin], 1ull);
      }
    }
    // hand-over, per point set: exactly k_pip_walk's.  The lists hold entry indices: a settled point reads its one entry's
    // record {slot, edge id, face id}, a listed point the slots of its entries -- every read of the group requested before
    // the first is used.
    bool done[PTS];
    uint4 sure_inf[PTS];
    uint32_t lslot[PTS][kWalkList];
#pragma unroll
    for (int p = 0; p < PTS; p++) {
      done[p] = valid[p] && (cand_at[p] == cand_base[p] || sure_y0[p] != INT32_MIN);
      const uint32_t fill = (cand_at[p] - cand_base[p]) >> 6;
      sure_inf[p] = make_uint4(0u, 0xFFFFFFFFu, 0u, 0u);
      // (round 6, timing only: with this one record read of a settled point left out -- wrong answers -- the pass took 1.31
      //  instead of 1.44 ms on the lake-shaped pair, 1.11 / 1.30 lakes x parks, 0.81 / 0.96 on the WaterBodies lattice: the
      //  ceiling of any layout that brings the record into the box's own line, 9-16 %; not built)
      if (done[p] && fill) sure_inf[p] = S.einfo[cand[cand_base[p]]];
      const bool listed = valid[p] && !done[p] && fill <= (uint32_t) kWalkList;
#pragma unroll
      for (int k = 0; k < kWalkList; k++) {
        lslot[p][k] = 0xFFFFFFFFu;
        if (listed && (uint32_t) k < fill) lslot[p][k] = S.einfo[cand[cand_base[p] + 64 * k]].x;
      }
    }
#pragma unroll
    for (int p = 0; p < PTS; p++) {
      if (done[p]) {
        __builtin_nontemporal_store(sure_inf[p].y, A.closest + ip[p]);
        if (A.face) __builtin_nontemporal_store((int32_t) sure_inf[p].z, A.face + ip[p]);
      }
      const uint32_t fill = (cand_at[p] - cand_base[p]) >> 6;
      const bool listed = valid[p] && !done[p] && fill <= (uint32_t) kWalkList;
      const bool rest = valid[p] && !done[p] && !listed;
      const uint64_t lm = __ballot(listed);
      const uint64_t g64 = g * PTS + p;  // the 64-position group this set is
      if (listed) {
        const uint64_t rec = g64 * 64 + rank_below(lm);  // (records side by side at the head of the group's region: k_pip_walk)
#pragma unroll
        for (int k = 0; k < kWalkList; k++) A.todo[rec * kWalkList + k] = lslot[p][k];
      }
      if (lane == 0 && g64 * 64 < A.n) A.todo_mask[g64] = lm;
      const uint64_t rm = __ballot(rest);
      if (rm) {
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(A.rest_count, (unsigned long long) __popcll(rm));
        base = ((unsigned long long) __builtin_amdgcn_readfirstlane((uint32_t) (base >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t) base);
        if (rest) A.rest[base + rank_below(rm)] = ip[p];
      }
    }
    wave_lds_fence();  // (the lists are reused by the next group)
  }
}

__global__ void k_strip_noop() {}

}  // namespace

hipError_t warm_strip_kernels(hipStream_t st) {
  hipLaunchKernelGGL(k_strip_noop, dim3(1), dim3(1), 0, st);
  return hipGetLastError();
}

// pass 1 of the build: per-slot strip counts and their exclusive scan (cnt, offs: n0p + 1 words each); *total_out on
// the stream (mapped or device memory), flag[0] = 1 when the index cannot be built
hipError_t launch_strip_width(hipStream_t st, const QBox* box0, const uint32_t* seid, uint64_t n0p, unsigned long long* out2) {
  hipError_t e = hipMemsetAsync(out2, 0, 16, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_strip_width, dim3(blocks_for(n0p / 64 + 1, 256, 512)), dim3(256), 0, st, box0, seid, n0p, out2);
  return hipGetLastError();
}
hipError_t launch_strip_count(hipStream_t st, const QBox* box0, const uint32_t* seid, uint64_t n0p, int shift, uint32_t* cnt, uint32_t* offs,
                              void* temp, size_t& temp_bytes, uint32_t* flag) {
  if (!temp) return rocprim::exclusive_scan(nullptr, temp_bytes, (const uint32_t*) nullptr, (uint32_t*) nullptr, 0u, (size_t) n0p + 1,
                                            rocprim::plus<uint32_t>(), st);
  hipError_t e = hipMemsetAsync(cnt + n0p, 0, 4, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_strip_count, dim3(blocks_for(n0p)), dim3(256), 0, st, box0, seid, n0p, shift, cnt, flag);
  return rocprim::exclusive_scan(temp, temp_bytes, cnt, offs, 0u, (size_t) n0p + 1, rocprim::plus<uint32_t>(), st);  // offs[n0p] = the total
}
// pass 2: the entries, sorted by (strip, y0), with their boxes, their {slot, edge id, face id} and the height-bucket
// table; key / eslot / key_tmp / slot_tmp: temporaries of `entries` elements; tall[strips] (temporary) zeroed here
hipError_t launch_strip_fill(hipStream_t st, const QBox* box0, const uint32_t* seid, const int32_t* sface, const uint32_t* cnt,
                             const uint32_t* offs, uint64_t n0p, int shift, uint64_t entries, uint32_t* key, uint32_t* eslot, uint32_t* key_tmp,
                             uint32_t* slot_tmp, uint32_t* tall, uint32_t* ytab, QBox* ebox, uint4* einfo, uint2* tall_end, uint32_t* sky, void* temp,
                             size_t& temp_bytes) {
  const unsigned bits = (unsigned) kStripBandBits + (31 - shift);  // (<= 32: strips of 2^15 quanta or wider)
  const uint32_t strips = strip_count(shift);
  if (!temp) {
    size_t a = 0, b = 0;
    hipError_t q = rocprim::radix_sort_pairs(nullptr, a, (const uint32_t*) nullptr, (uint32_t*) nullptr, (const uint32_t*) nullptr,
                                             (uint32_t*) nullptr, (size_t) entries, 0, bits, st);
    if (q != hipSuccess) return q;
    uint32_t* z = nullptr;
    auto rz = rocprim::make_reverse_iterator(z);
    q = rocprim::inclusive_scan(nullptr, b, rz, rz, ((size_t) strips << kStripYBits) + 1, rocprim::minimum<uint32_t>(), st);
    temp_bytes = a > b ? a : b;
    return q;
  }
  hipError_t e = hipMemsetAsync(tall, 0, (size_t) strips * 4, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_strip_emit, dim3(blocks_for(n0p)), dim3(256), 0, st, box0, cnt, offs, n0p, shift, key_tmp, slot_tmp);
  if ((e = rocprim::radix_sort_pairs(temp, temp_bytes, key_tmp, key, slot_tmp, eslot, (size_t) entries, 0, bits, st)) != hipSuccess) return e;
  const size_t nt = ((size_t) strips << kStripYBits) + 1;
  if ((e = hipMemsetAsync(ytab, 0xFF, nt * 4, st)) != hipSuccess) return e;
  hipLaunchKernelGGL(k_strip_finish, dim3(blocks_for(entries)), dim3(256), 0, st, key, eslot, entries, box0, seid, sface, ebox, einfo, ytab, strips, tall, sky, shift);
  // suffix minimum, in place (the sort's temporary storage is free again and larger than a scan's)
  size_t need = 0;
  auto rb = rocprim::make_reverse_iterator(ytab + nt);
  if ((e = rocprim::inclusive_scan(nullptr, need, rb, rb, nt, rocprim::minimum<uint32_t>(), st)) != hipSuccess) return e;
  if (need > temp_bytes) return hipErrorInvalidValue;
  if ((e = rocprim::inclusive_scan(temp, need, rb, rb, nt, rocprim::minimum<uint32_t>(), st)) != hipSuccess) return e;
  hipLaunchKernelGGL(k_strip_ends, dim3(blocks_for(strips)), dim3(256), 0, st, tall, ytab, strips, tall_end);
  return hipGetLastError();
}

static thread_local int strip_note_grid = 0, strip_note_pts = 0;
LaunchNote last_strip_launch() { return LaunchNote{"k_pip_strip", strip_note_grid, strip_note_pts}; }
hipError_t launch_pip_strip(hipStream_t st, const PipArgs& a, int max_blocks, int cus) {
  // two points per lane where the query set keeps every resident wave busy with at least two 128-position groups
  // (measured, first pass alone: 2.68 -> 2.42 ms on the lake-shaped base, 4.66 -> 4.01 lakes x parks; four per lane --
  //  86 VGPRs, 5 waves per SIMD -- 3.27 / 5.66: the pass is bound by the number of random reads, not by their latency)
  const uint64_t resident_waves = (uint64_t) cus * 32;
  const int pts = a.stats || a.n >= resident_waves * 2 * 128 ? 2 : 1;  // (the instrumented build is the two-point one)
  const uint64_t ngroups = (a.n + (uint64_t) pts * 64 - 1) / ((uint64_t) pts * 64);
  int grid = blocks_for(ngroups, 4, cus * 8);
  if (grid > max_blocks) grid = max_blocks;
  strip_note_grid = grid; strip_note_pts = pts;
  if (a.stats) { hipLaunchKernelGGL((k_pip_strip<2, true>), dim3(grid), dim3(256), 0, st, a); return hipGetLastError(); }  // (the instrumented build: "stats" 1 + "pip_walk" 2)
  if (pts == 2) hipLaunchKernelGGL(k_pip_strip<2>, dim3(grid), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(k_pip_strip<1>, dim3(grid), dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace rj
