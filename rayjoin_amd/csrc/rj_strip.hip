// rj_strip.hip -- the column index of an indexed map and the PIP pass that runs on it (rj_device.h DeviceStrips).
//
// Built for maps of isolated rings, where the box hierarchy is at its worst for upward rays: a leaf block of a few
// rings is mostly gaps in x, and a point opens every block over its column whose x-extent contains it until one
// holds an edge at its x -- measured on the lake-shaped stand-in: 20 leaf blocks opened per point, one segment box
// tested.  Here a point reads its strip's list: a binary search for its height, then the entries upwards until a
// certain hit bounds the answer.  The pass is k_pip_walk's in every other respect -- integer tests only, the same
// certain-hit pruning, the same hand-over (settled points written, candidate lists in `todo`, overflowed lists in
// `rest`) -- so k_pip_exact follows it unchanged and the results are the walk's (tests/test_gpu_strip.py).
#include <hip/hip_runtime.h>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

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
                                                     uint32_t* __restrict__ cnt, uint32_t* __restrict__ flag) {
  RJ_GRID_STRIDE(i, n0p) {
    uint32_t c = 0;
    if (seid[i] != 0xFFFFFFFFu) {
      const QBox b = box0[i];
      c = (uint32_t) ((b.x1 >> kStripShift) - (b.x0 >> kStripShift) + 1);
      if (c > (uint32_t) kStripMaxSpan) { c = 0; flag[0] = 1u; }
    }
    cnt[i] = c;
  }
}
__global__ __launch_bounds__(256) void k_strip_emit(const QBox* __restrict__ box0, const uint32_t* __restrict__ cnt,
                                                    const uint32_t* __restrict__ offs, uint64_t n0p, uint64_t* __restrict__ key,
                                                    uint32_t* __restrict__ slot, uint32_t* __restrict__ tall) {
  RJ_GRID_STRIDE(i, n0p) {
    const uint32_t c = cnt[i];
    if (!c) continue;
    const QBox b = box0[i];
    const uint32_t s0 = (uint32_t) (b.x0 >> kStripShift), h = (uint32_t) (b.y1 - b.y0);
    const uint32_t o = offs[i];
    for (uint32_t k = 0; k < c; k++) {
      key[o + k] = ((uint64_t) (s0 + k) << 32) | (uint32_t) b.y0;
      slot[o + k] = (uint32_t) i;
      if (__hip_atomic_load(&tall[s0 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < h) atomicMax(&tall[s0 + k], h);
    }
  }
}
// begin[s] = first entry whose strip is >= s (the keys are sorted)
__global__ __launch_bounds__(256) void k_strip_begin(const uint64_t* __restrict__ key, uint64_t n, uint32_t* __restrict__ begin) {
  RJ_GRID_STRIDE(s, (uint64_t) kStrips + 1) {
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
      const uint64_t mid = (lo + hi) >> 1;
      if ((uint32_t) (key[mid] >> 32) < (uint32_t) s) lo = mid + 1; else hi = mid;
    }
    begin[s] = (uint32_t) lo;
  }
}

// One query point per lane, 64 consecutive positions per wave (one todo mask per wave, as the walk writes them).
__global__ __launch_bounds__(256, 8) void k_pip_strip(PipArgs A) {
  __shared__ uint32_t cand_all[4][kWalkList * 64];
  const int lane = lane_id();
  const int wib = threadIdx.x >> 6;
  const DeviceBvh& T = A.bvh;
  const DeviceStrips& S = T.strips;
  const uint32_t* const sky = (T.sky && T.sky[kSkyBuckets] == 0u) ? T.sky : nullptr;
  uint32_t* const cand = cand_all[wib];  // [kWalkList][64], bank = lane
  const uint64_t ngroups = (A.n + 63) / 64;
  const uint64_t wave = (blockIdx.x * (uint64_t) blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
  // (the counters of the next launch of this kind on this stream are cleared here, as every walk does: k_lsi)
  if (blockIdx.x == 0 && threadIdx.x < 8) A.next_work_counter[threadIdx.x * 32] = 0;
  if (blockIdx.x == 0 && threadIdx.x == 8) *A.next_rest_count = 0;
  for (uint64_t g = wave; g < ngroups; g += nwaves) {
    const uint64_t ipos = g * 64 + lane;
    const bool valid = ipos < A.n;
    const uint32_t ip = A.order ? (valid ? A.order[ipos] : 0u) : (uint32_t) ipos;
    int32_t qx = 0, qy = 0;
    if (valid) {
      typedef long long ll2_t __attribute__((ext_vector_type(2)));
      const ll2_t p = __builtin_nontemporal_load(reinterpret_cast<const ll2_t*>(A.pts) + ip);
      qx = quant(p.x);
      qy = quant(p.y);
    }
    const bool live = valid && ray_has_sky(sky, qx, qy);  // (above the map's skyline: a certain miss)
    const int32_t qym1 = qy > 0 ? qy - 1 : 0;
    int32_t qbest = 0x7FFFFFFF;
    const uint32_t cand_base = (uint32_t) lane;
    uint32_t cand_at = cand_base;
    int32_t sure_y0 = INT32_MIN;
    // the strip's entries from the first that can still reach up to the point: y0 >= qy - 1 - (tallest box of the strip)
    uint32_t j = 0, jend = 0;
    if (live) {
      const uint32_t s = (uint32_t) qx >> kStripShift;
      uint32_t lo = S.begin[s];
      const uint32_t hi = S.begin[s + 1];
      jend = hi;
      const int64_t from = (int64_t) qym1 - (int64_t) S.tall[s];
      const uint32_t want = from > 0 ? (uint32_t) from : 0u;
      uint32_t n = hi - lo;  // lower bound of `want` among the low words of key[lo, hi)
      while (n) {
        const uint32_t half = n >> 1;
        const bool below = (uint32_t) S.key[lo + half] < want;
        lo = below ? lo + half + 1 : lo;
        n = below ? n - half - 1 : half;
      }
      j = lo;
    }
    while (j < jend) {
      const int32_t sy0 = (int32_t) (uint32_t) S.key[j];
      if (sy0 > qbest) break;  // everything further starts above the bound
      const uint32_t slot = S.slot[j];
      const QBox b = T.box0[slot];
      if (((qx - b.x0) | (b.x1 - qx) | (b.y1 - qym1)) >= 0) {
        // k_pip_walk's bookkeeping: a certain hit (strictly inside in x, strictly above) bounds the answer; one that
        // ends below the start of the one certain hit held so far replaces it
        const bool certain = b.x0 < qx && qx < b.x1 && sy0 > qy;
        const bool replace = certain && b.y1 < sure_y0;
        const bool first = cand_at == cand_base;
        const bool over = !replace && cand_at == cand_base + kWalkList * 64;
        cand[(replace || over) ? cand_base : cand_at] = slot;
        sure_y0 = (replace || (first && certain)) ? sy0 : INT32_MIN;
        cand_at += replace ? 0u : 64u;
        const int32_t top = certain ? b.y1 + 1 : 0x7FFFFFFF;
        qbest = over ? -1 : (top < qbest ? top : qbest);
      }
      j++;
    }
    // hand-over: exactly k_pip_walk's
    const bool done = valid && (cand_at == cand_base || sure_y0 != INT32_MIN);
    if (done) {
      const bool hit = cand_at != cand_base;
      const uint32_t slot = hit ? cand[lane] : 0u;
      __builtin_nontemporal_store(hit ? T.seid[slot] : 0xFFFFFFFFu, A.closest + ip);
      if (A.face) __builtin_nontemporal_store(hit ? T.sface[slot] : 0, A.face + ip);
    }
    const uint32_t fill = (cand_at - cand_base) >> 6;
    const bool listed = valid && !done && fill <= (uint32_t) kWalkList;
    const bool rest = valid && !done && !listed;
    if (listed) {
#pragma unroll
      for (int k = 0; k < kWalkList; k++) A.todo[ipos * kWalkList + k] = (uint32_t) k < fill ? cand[lane + 64 * k] : 0xFFFFFFFFu;
    }
    const uint64_t lm = __ballot(listed);
    if (lane == 0) A.todo_mask[g] = lm;
    const uint64_t rm = __ballot(rest);
    if (rm) {
      unsigned long long base = 0;
      if (lane == 0) base = atomicAdd(A.rest_count, (unsigned long long) __popcll(rm));
      base = ((unsigned long long) __builtin_amdgcn_readfirstlane((uint32_t) (base >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t) base);
      if (rest) A.rest[base + rank_below(rm)] = ip;
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
hipError_t launch_strip_count(hipStream_t st, const QBox* box0, const uint32_t* seid, uint64_t n0p, uint32_t* cnt, uint32_t* offs,
                              void* temp, size_t& temp_bytes, uint32_t* flag) {
  if (!temp) return rocprim::exclusive_scan(nullptr, temp_bytes, (const uint32_t*) nullptr, (uint32_t*) nullptr, 0u, (size_t) n0p + 1,
                                            rocprim::plus<uint32_t>(), st);
  hipError_t e = hipMemsetAsync(cnt + n0p, 0, 4, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_strip_count, dim3(blocks_for(n0p)), dim3(256), 0, st, box0, seid, n0p, cnt, flag);
  return rocprim::exclusive_scan(temp, temp_bytes, cnt, offs, 0u, (size_t) n0p + 1, rocprim::plus<uint32_t>(), st);  // offs[n0p] = the total
}
// pass 2: the entries, sorted by (strip, y0); key_tmp / slot_tmp: sort buffers of the same size; tall[kStrips] zeroed here
hipError_t launch_strip_fill(hipStream_t st, const QBox* box0, const uint32_t* cnt, const uint32_t* offs, uint64_t n0p, uint64_t entries,
                             uint64_t* key, uint32_t* slot, uint64_t* key_tmp, uint32_t* slot_tmp, uint32_t* tall, uint32_t* begin,
                             void* temp, size_t& temp_bytes) {
  const unsigned bits = 32 + (31 - kStripShift);
  if (!temp) return rocprim::radix_sort_pairs(nullptr, temp_bytes, (const uint64_t*) nullptr, (uint64_t*) nullptr, (const uint32_t*) nullptr,
                                              (uint32_t*) nullptr, (size_t) entries, 0, bits, st);
  hipError_t e = hipMemsetAsync(tall, 0, (size_t) kStrips * 4, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_strip_emit, dim3(blocks_for(n0p)), dim3(256), 0, st, box0, cnt, offs, n0p, key_tmp, slot_tmp, tall);
  if ((e = rocprim::radix_sort_pairs(temp, temp_bytes, key_tmp, key, slot_tmp, slot, (size_t) entries, 0, bits, st)) != hipSuccess) return e;
  hipLaunchKernelGGL(k_strip_begin, dim3(blocks_for(kStrips + 1)), dim3(256), 0, st, key, entries, begin);
  return hipGetLastError();
}

hipError_t launch_pip_strip(hipStream_t st, const PipArgs& a, int max_blocks, int cus) {
  const uint64_t ngroups = (a.n + 63) / 64;
  int grid = blocks_for(ngroups, 4, cus * 8);
  if (grid > max_blocks) grid = max_blocks;
  hipLaunchKernelGGL(k_pip_strip, dim3(grid), dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace rj
