// fetch_calibrate.hip -- what rocprofv3's FETCH_SIZE says about reads of KNOWN byte counts in k_pip_strip's access pattern.
//
// /opt/skills/guides/MI355X_MICROARCH.md (HBM): "On gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced
// streaming read ... Other access widths are uncalibrated: calibrate on a known byte count in your own access pattern before
// trusting an absolute."  k_pip_strip (rj_strip.hip) reads scattered 16-byte boxes, two consecutive ones per trip, plus one
// 4-byte table word and one 8-byte strip record per point; bench.py doubled its FETCH_SIZE like every streaming kernel's and
// quoted 9.0 GB per launch -- "between 4.7 and 9.0 GB" (DESIGN.md, round 5).  This program settles the factor: every kernel
// below issues N reads of a known size at known places of a table far larger than the L2s and the Infinity Cache together,
// so that every read is a miss all the way to HBM, and is run under
//     rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -o g -- tools/fetch_calibrate
// (tools/fetch_calibrate.py reads the counter file and prints the table: counter bytes per read, per pattern).
//   stream        coalesced 16 B per lane over the whole table (the guide's case: expect counter = bytes / 2)
//   rand16        one 16-byte read per lane at offset 0 of a random 128-byte line
//   rand16_same64 two 16-byte reads per lane, offsets 0 and 16 of one random line  (same 64-byte half)
//   rand16_other64 two 16-byte reads per lane, offsets 0 and 64 of one random line (the OTHER half): one request or two?
//                 -- if two, the fabric moves 64-byte halves and a scattered read costs 64 real bytes; if one, a 128-byte line
//   rand_pair     the strip pass's trip: entries j and j + 1 (16 bytes each) at a random 16-byte-aligned j
//   rand4 / rand8 one 4- / 8-byte read per lane at a random place (the height table's word, the strip's record)
// The same kernels' durations (the trace's columns) give the rate at which the chip serves such reads -- the ceiling the
// column pass sits under, whatever a byte counter says.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z) {  // splitmix64
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

typedef uint32_t v4u __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stream(const v4u* __restrict__ t, uint64_t n16, uint32_t* __restrict__ sink) {
  uint32_t acc = 0;
  for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n16; i += (uint64_t) gridDim.x * blockDim.x) {
    const v4u v = __builtin_nontemporal_load(t + i);
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
// OFF2 < 0: one read per lane; else a second read OFF2 bytes behind the first.  ALIGN = what the random place is a multiple of.
template <int BYTES, int OFF2, int ALIGN>
__global__ __launch_bounds__(256) void scattered(const unsigned char* __restrict__ t, uint64_t table_bytes, uint64_t n, uint64_t seed,
                                                 uint32_t* __restrict__ sink) {
  uint32_t acc = 0;
  const uint64_t places = (table_bytes - 256) / ALIGN;
  for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
    const uint64_t at = (mix(i + seed) % places) * ALIGN;
    if (BYTES == 16) {
      const uint4 v = *reinterpret_cast<const uint4*>(t + at);
      acc ^= v.x ^ v.w;
      if (OFF2 >= 0) { const uint4 w = *reinterpret_cast<const uint4*>(t + at + (OFF2 >= 0 ? OFF2 : 0)); acc ^= w.x ^ w.w; }
    } else if (BYTES == 8) {
      const uint2 v = *reinterpret_cast<const uint2*>(t + at);
      acc ^= v.x ^ v.y;
    } else {
      acc ^= *reinterpret_cast<const uint32_t*>(t + at);
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char** argv) {
  const uint64_t table_bytes = (argc > 1 ? strtoull(argv[1], nullptr, 10) : 8ull) << 30;  // GiB, default 8
  const uint64_t n = (argc > 2 ? strtoull(argv[2], nullptr, 10) : 64ull) << 20;           // Mi reads per lane-pattern, default 64 Mi
  unsigned char* t = nullptr;
  uint32_t* sink = nullptr;
  CK(hipMalloc((void**) &t, table_bytes));
  CK(hipMalloc((void**) &sink, 256));
  CK(hipMemset(t, 1, table_bytes));
  CK(hipMemset(sink, 0, 256));
  CK(hipDeviceSynchronize());
  const int grid = 256 * 8;  // a block per CU and wave slot: the column pass's own grid
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  printf("{\"table_bytes\": %llu, \"reads\": %llu", (unsigned long long) table_bytes, (unsigned long long) n);
#define RUN(NAME, ...)                                                            \
  for (int rep = 0; rep < 3; rep++) {                                             \
    CK(hipEventRecord(a, 0));                                                     \
    __VA_ARGS__;                                                                  \
    CK(hipGetLastError());                                                        \
    CK(hipEventRecord(b, 0));                                                     \
    CK(hipEventSynchronize(b));                                                   \
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));                             \
    if (rep == 2) printf(", \"%s_ms\": %.4f", NAME, ms);                          \
  }
  auto k_rand16 = scattered<16, -1, 128>;
  auto k_same64 = scattered<16, 16, 128>;
  auto k_other64 = scattered<16, 64, 128>;
  auto k_pair = scattered<16, 16, 16>;
  auto k_rand8 = scattered<8, -1, 8>;
  auto k_rand4 = scattered<4, -1, 4>;
  RUN("stream", stream<<<grid, 256>>>((const v4u*) t, table_bytes / 16, sink));
  RUN("rand16", k_rand16<<<grid, 256>>>(t, table_bytes, n, 1000ull * rep, sink));
  RUN("rand16_same64", k_same64<<<grid, 256>>>(t, table_bytes, n, 7000ull + rep, sink));
  RUN("rand16_other64", k_other64<<<grid, 256>>>(t, table_bytes, n, 9000ull + rep, sink));
  RUN("rand_pair", k_pair<<<grid, 256>>>(t, table_bytes, n, 11000ull + rep, sink));
  RUN("rand8", k_rand8<<<grid, 256>>>(t, table_bytes, n, 13000ull + rep, sink));
  RUN("rand4", k_rand4<<<grid, 256>>>(t, table_bytes, n, 15000ull + rep, sink));
  printf("}\n");
  CK(hipFree(t)); CK(hipFree(sink));
  return 0;
}
