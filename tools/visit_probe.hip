// visit_probe.hip -- the walk's (leaf, set) visit in its two possible forms, timed on the walk's own data shapes.
//
// The round-5 review asked for the PIP walk's leaf visits to be queued and run DENSE: a 128-position group of k_pip_walk2
// makes 7.5 (leaf, set) visits in which 20.5 of 64 lanes want the block (profiles/r05_walk_stats.txt) -- 154 lane-wants that
// would fill 2.4 waves.  What the dense form has to give up is the wave-wide access to a block: today ONE coalesced load
// brings the block's 64 boxes and its bucket table (1.5 KiB) and every lane reads its slots through ds_bpermute; a dense wave
// holds 64 items of up to 64 DIFFERENT blocks, so every lane gathers its own table word and its own boxes from global memory.
// This program times exactly that trade on a tree-shaped array pair (64 x 16-byte boxes + 64 x 8-byte table words per block,
// 111 k blocks like the USCounty index), with the walk's locality (a wave's consecutive visits fall on neighbouring blocks):
//   wide    a wave per visit: coalesced load of the block, then PER LANE the lookup (two table reads + two corrections through
//           ds_bpermute) and one scan step (four ds_bpermute + the box test) -- with `want` of 64 lanes doing useful work;
//   dense   a wave per 64 items: per lane one 8-byte table gather, then two 16-byte box gathers (the corrections), then one
//           more (the scan step): the same arithmetic, every lane useful.
// Output: ns per visit and per useful lane-item for `wide` at want = 20 (the measured average) and for `dense`, at the walk's
// occupancy (8 waves per SIMD).  DESIGN.md section 4 quotes it.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct QBox { int32_t x0, y0, x1, y1; };

__device__ __forceinline__ uint32_t mix32(uint32_t z) {
  z ^= z >> 16; z *= 0x7feb352du; z ^= z >> 15; z *= 0x846ca68bu; z ^= z >> 16;
  return z;
}
__device__ __forceinline__ int lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0)); }

// the per-lane arithmetic both forms share: slots [lo, hi) from the table bytes, one correction per end, one scan step
__device__ __forceinline__ uint32_t body(uint32_t hi, uint32_t lo, int32_t top_x0, int32_t low_x1, const QBox& s, int32_t qx, int32_t qy) {
  int j = (int) hi - 1, jlo = (int) lo;
  j -= top_x0 > qx ? 1 : 0;
  jlo += low_x1 < qx ? 1 : 0;
  const bool hit = (s.x0 <= qx) & (qx <= s.x1) & (s.y1 >= qy - 1) & (j >= jlo);
  return hit ? (uint32_t) s.y1 : 0u;
}

// `want` lanes of 64 do useful work, as in the walk; the instructions are issued for the wave either way
__global__ __launch_bounds__(256, 8) void wide(const QBox* __restrict__ box0, const uint2* __restrict__ xtab, uint32_t nblocks, uint32_t visits_per_wave,
                                               int want, uint32_t* __restrict__ sink) {
  const int lane = lane_id();
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  uint32_t acc = 0;
  uint32_t blk = mix32(wave) % nblocks;
  for (uint32_t v = 0; v < visits_per_wave; v++) {
    blk = (blk + 1 + (mix32(wave * 7919u + v) & 7u)) % nblocks;  // (neighbouring blocks: the walk's locality)
    const QBox bb = box0[(uint64_t) blk * 64 + lane];
    const uint2 tab = xtab[(uint64_t) blk * 64 + lane];
    const int32_t qx = (int32_t) (mix32(v * 64u + lane) & 0x3FFFFFFF), qy = (int32_t) (mix32(v + lane * 977u) & 0x3FFFFFFF);
    const uint32_t bk = (uint32_t) qx >> 22;   // a bucket 0..255
    const uint32_t bsh = (bk & 3u) * 8u;
    const uint32_t hi = ((uint32_t) __shfl((int) tab.x, (int) (bk >> 2), 64) >> bsh) & 0x3Fu;
    const uint32_t lo = ((uint32_t) __shfl((int) tab.y, (int) (bk >> 2), 64) >> bsh) & 0x3Fu;
    const int32_t top_x0 = __builtin_amdgcn_ds_bpermute((int) hi << 2, bb.x0);
    const int32_t low_x1 = __builtin_amdgcn_ds_bpermute((int) lo << 2, bb.x1);
    const int ja = (int) ((hi - (top_x0 > qx ? 1u : 0u) + (low_x1 < qx ? 1u : 0u)) & 63u) << 2;  // (the scan starts where the corrections say)
    QBox s;
    s.x0 = __builtin_amdgcn_ds_bpermute(ja, bb.x0); s.x1 = __builtin_amdgcn_ds_bpermute(ja, bb.x1);
    s.y0 = __builtin_amdgcn_ds_bpermute(ja, bb.y0); s.y1 = __builtin_amdgcn_ds_bpermute(ja, bb.y1);
    const uint32_t r = body(hi, lo, top_x0, low_x1, s, qx, qy);
    acc += lane < want ? r : 0u;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

__global__ __launch_bounds__(256, 8) void dense(const QBox* __restrict__ box0, const uint2* __restrict__ xtab, uint32_t nblocks, uint32_t batches_per_wave,
                                                uint32_t* __restrict__ sink) {
  const int lane = lane_id();
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  uint32_t acc = 0;
  uint32_t base = mix32(wave) % nblocks;
  for (uint32_t v = 0; v < batches_per_wave; v++) {
    // 64 items = about three visits' worth of wanting lanes: they fall on three to four neighbouring blocks
    base = (base + 3 + (mix32(wave * 7919u + v) & 7u)) % nblocks;
    const uint32_t blk = (base + ((uint32_t) lane / 20u)) % nblocks;
    const int32_t qx = (int32_t) (mix32(v * 64u + lane) & 0x3FFFFFFF), qy = (int32_t) (mix32(v + lane * 977u) & 0x3FFFFFFF);
    const uint32_t bk = (uint32_t) qx >> 22;
    const uint32_t bsh = (bk & 3u) * 8u;
    const uint2 tab = xtab[(uint64_t) blk * 64 + (bk >> 2)];           // gather 1: the lane's own table word
    const uint32_t hi = (tab.x >> bsh) & 0x3Fu, lo = (tab.y >> bsh) & 0x3Fu;
    const QBox bt = box0[(uint64_t) blk * 64 + hi];                     // gathers 2, 3 (side by side): the two corrections
    const QBox bl = box0[(uint64_t) blk * 64 + lo];
    const QBox s = box0[(uint64_t) blk * 64 + ((hi - (bt.x0 > qx ? 1u : 0u) + (bl.x1 < qx ? 1u : 0u)) & 63u)];  // gather 4: depends on the corrections, as in the walk
    acc += body(hi, lo, bt.x0, bl.x1, s, qx, qy);
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

__global__ void fill(QBox* box0, uint2* xtab, uint64_t n) {
  for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
    const uint32_t a = mix32((uint32_t) i * 4u) & 0x3FFFFFFF, b = mix32((uint32_t) i * 4u + 1u) & 0x3FFFFFFF;
    box0[i] = QBox{(int32_t) a, (int32_t) b, (int32_t) (a + 4096), (int32_t) (b + 4096)};
    xtab[i] = make_uint2(mix32((uint32_t) i * 4u + 2u), mix32((uint32_t) i * 4u + 3u));
  }
}

int main(int argc, char** argv) {
  const uint32_t nblocks = argc > 1 ? (uint32_t) strtoul(argv[1], nullptr, 10) : 111604u;  // USCounty's leaf blocks
  QBox* box0 = nullptr; uint2* xtab = nullptr; uint32_t* sink = nullptr;
  CK(hipMalloc((void**) &box0, (size_t) nblocks * 64 * sizeof(QBox)));
  CK(hipMalloc((void**) &xtab, (size_t) nblocks * 64 * sizeof(uint2)));
  CK(hipMalloc((void**) &sink, 256));
  fill<<<2048, 256>>>(box0, xtab, (uint64_t) nblocks * 64);   // (every slot different: no lane reads what its neighbour reads)
  CK(hipDeviceSynchronize());
  CK(hipMemset(sink, 0, 256));
  const int grid = 256 * 8;  // eight blocks per CU: the walk's residency
  const uint64_t waves = (uint64_t) grid * 4;
  // a headline query: 231 764 groups x 7.5 visits = 1.74 M visits, x 20.5 wanting lanes = 35.6 M items = 557 k dense batches
  // (x 16: sixteen queries' worth per launch, so that a launch is milliseconds long)
  const uint32_t visits_per_wave = (uint32_t) (16 * 1738230ull / waves) + 1, batches_per_wave = (uint32_t) (16 * 556700ull / waves) + 1;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float ms_wide = 0, ms_dense = 0;
  for (int rep = 0; rep < 4; rep++) {
    CK(hipEventRecord(a, 0));
    wide<<<grid, 256>>>(box0, xtab, nblocks, visits_per_wave, 20, sink);
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms_wide, a, b));
    CK(hipEventRecord(a, 0));
    dense<<<grid, 256>>>(box0, xtab, nblocks, batches_per_wave, sink);
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms_dense, a, b));
  }
  CK(hipGetLastError());
  const double visits = (double) visits_per_wave * waves, items = (double) batches_per_wave * waves * 64.0;
  printf("{\"blocks\": %u, \"wide\": {\"visits\": %.0f, \"ms\": %.4f, \"ns_per_visit_per_wave_slot\": %.2f, \"useful_items\": %.0f}, "
         "\"dense\": {\"items\": %.0f, \"ms\": %.4f}, \"same_useful_work\": \"wide = 16 x the headline walk's 1.74 M visits at 20 wanting lanes; dense = the same lane-items in full waves\", "
         "\"dense_over_wide\": %.3f}\n",
         nblocks, visits, ms_wide, ms_wide * 1e6 * waves / visits, visits * 20.0, items, ms_dense, ms_dense / ms_wide);
  return 0;
}
