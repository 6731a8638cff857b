// rj_stitch.hip -- the polyline runs of a map, cut on the device (rj_stitch.h has the stages and what they compute).
// Every kernel is a grid-stride loop over one of rj_stitch.h's per-element functions; rocPRIM does the two stable radix
// sorts that bring the incidences of a junction together and the two prefix sums that lay paths and pieces out.
// No host pass, no read-back of the map: the host reads two words (closed loops left? -- then the totals).
#include <hip/hip_runtime.h>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "rj_kernels.h"
#include "rj_stitch.h"

namespace rj {

using namespace stitch;

namespace {

constexpr int kThreads = 256;
inline int blocks_for(uint64_t n) {
  uint64_t b = (n + kThreads - 1) / kThreads;
  return (int) (b < 1 ? 1 : (b > 16384 ? 16384 : b));
}
#define RJ_GRID_STRIDE(i, n) \
  for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < (n); i += (uint64_t) gridDim.x * blockDim.x)

__global__ __launch_bounds__(kThreads) void k_st_end_keys(uint32_t ni, const int64_t* __restrict__ pts, const uint32_t* __restrict__ eb,
                                                          uint64_t* __restrict__ kx, uint64_t* __restrict__ ky, Dir* __restrict__ dir,
                                                          uint32_t* __restrict__ iota, Meta* meta) {
  uint32_t rings = 0;
  RJ_GRID_STRIDE(i, ni) {
    rings += end_keys((uint32_t) i, pts, eb, kx, ky, dir) ? 1u : 0u;
    iota[i] = (uint32_t) i;
  }
  __shared__ uint32_t part[kThreads / 64];
  for (int d = 32; d >= 1; d >>= 1) rings += __shfl_down(rings, d, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = rings;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t sum = 0;
    for (int w = 0; w < kThreads / 64; w++) sum += part[w];
    if (sum) atomicAdd(&meta->closed_chains, sum);
  }
}
__global__ __launch_bounds__(kThreads) void k_st_gather_keys(uint32_t ni, const uint64_t* __restrict__ kx, const uint32_t* __restrict__ sv,
                                                             uint64_t* __restrict__ out) {
  RJ_GRID_STRIDE(j, ni) out[j] = kx[sv[j]];
}
__global__ __launch_bounds__(kThreads) void k_st_pair(uint32_t ni, const uint64_t* __restrict__ skx, const uint32_t* __restrict__ sv,
                                                      const uint64_t* __restrict__ ky, const Dir* __restrict__ dir, uint32_t* __restrict__ partner) {
  RJ_GRID_STRIDE(j, ni) pair_node(j, ni, skx, sv, ky, dir, partner);
}
__global__ __launch_bounds__(kThreads) void k_st_rank_init(uint32_t ni, const uint32_t* __restrict__ partner, const uint32_t* __restrict__ eb,
                                                           Node* __restrict__ a, Node* __restrict__ b) {
  RJ_GRID_STRIDE(i, ni) rank_init((uint32_t) i, partner, eb, a, b);
}
// one round of pointer jumping; a round with nothing left to do returns at once and leaves its number behind
__global__ __launch_bounds__(kThreads) void k_st_rank_round(uint32_t ni, const Node* __restrict__ in, Node* __restrict__ out, Meta* meta, int r,
                                                            int second) {
  uint32_t* act = second ? meta->act2 : meta->act;
  uint32_t* done = second ? &meta->done_round2 : &meta->done_round;
  if (!rank_round_needed(act, r)) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && !*done) *done = (uint32_t) r;
    return;
  }
  uint32_t mine = 0;
  RJ_GRID_STRIDE(i, ni) mine += rank_round((uint32_t) i, in, out) ? 1u : 0u;
  // one atomic per block (27 k same-address atomics, one per wave of a 1.7 M-incidence map, were 90 % of this kernel)
  __shared__ uint32_t part[kThreads / 64];
  for (int d = 32; d >= 1; d >>= 1) mine += __shfl_down(mine, d, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t sum = 0;
    for (int w = 0; w < kThreads / 64; w++) sum += part[w];
    if (sum) atomicAdd(&act[r], sum);
  }
}
__global__ void k_st_rank_done(Meta* meta, int rounds, int second) {
  uint32_t* done = second ? &meta->done_round2 : &meta->done_round;
  if (!*done) *done = (uint32_t) rounds;
}
__global__ __launch_bounds__(kThreads) void k_st_cyc_init(uint32_t ni, const Node* __restrict__ F0, const Node* __restrict__ F1, const Meta* meta,
                                                          const uint32_t* __restrict__ partner, Link* __restrict__ a, Link* __restrict__ b) {
  const Node* F = (meta->done_round & 1) ? F1 : F0;
  RJ_GRID_STRIDE(i, ni) cyc_init((uint32_t) i, F, partner, a, b);
}
__global__ __launch_bounds__(kThreads) void k_st_cyc_round(uint32_t ni, const Link* __restrict__ in, Link* __restrict__ out) {
  RJ_GRID_STRIDE(i, ni) cyc_round((uint32_t) i, in, out);
}
__global__ __launch_bounds__(kThreads) void k_st_cyc_break(uint32_t ni, const Link* __restrict__ S, uint32_t* partner, uint8_t* in_loop) {
  RJ_GRID_STRIDE(i, ni) cyc_break((uint32_t) i, S, partner, in_loop);
}
__global__ __launch_bounds__(kThreads) void k_st_rerank_init(uint32_t ni, Node* a, Node* b, const Meta* meta, const uint32_t* __restrict__ partner,
                                                             const uint32_t* __restrict__ eb) {
  const Node* F = (meta->done_round & 1) ? b : a;
  RJ_GRID_STRIDE(i, ni) rerank_init((uint32_t) i, F, partner, eb, a, b);
}
__global__ __launch_bounds__(kThreads) void k_st_orient(uint32_t nc, const Node* __restrict__ F0, const Node* __restrict__ F1, const Meta* meta, int second,
                                                        const uint32_t* __restrict__ eb, const uint8_t* __restrict__ in_loop, uint32_t cap,
                                                        uint32_t* __restrict__ ch_key, uint32_t* __restrict__ ch_off, uint32_t* __restrict__ ch_total,
                                                        uint32_t* __restrict__ ch_rank, uint8_t* __restrict__ ch_back, uint64_t* __restrict__ head) {
  const Node* F = ((second ? meta->done_round2 : meta->done_round) & 1) ? F1 : F0;
  RJ_GRID_STRIDE(c, nc) chain_orient((uint32_t) c, F, eb, in_loop, cap, 2 * nc, ch_key, ch_off, ch_total, ch_rank, ch_back, head);
}
__global__ __launch_bounds__(kThreads) void k_st_place(uint32_t nc, const uint32_t* __restrict__ eb, uint32_t cap, const uint32_t* __restrict__ ch_key,
                                                       const uint32_t* __restrict__ ch_off, const uint32_t* __restrict__ ch_total,
                                                       const uint32_t* __restrict__ ch_rank, const uint8_t* __restrict__ ch_back,
                                                       const uint64_t* __restrict__ base, uint32_t* __restrict__ slot_chain, uint32_t* __restrict__ slot_pieces) {
  RJ_GRID_STRIDE(c, nc) chain_place((uint32_t) c, eb, cap, ch_key, ch_off, ch_total, ch_rank, ch_back, base, slot_chain, slot_pieces);
}
__global__ __launch_bounds__(kThreads) void k_st_emit(uint32_t nc, const uint32_t* __restrict__ eb, uint32_t cap, const uint32_t* __restrict__ ch_key,
                                                      const uint32_t* __restrict__ ch_off, const uint32_t* __restrict__ ch_total,
                                                      const uint64_t* __restrict__ base, const uint32_t* __restrict__ slot_chain,
                                                      const uint32_t* __restrict__ slot_pieces, const uint32_t* __restrict__ pbase,
                                                      uint32_t* __restrict__ piece_begin, uint32_t* __restrict__ piece_len, uint32_t* __restrict__ run_first,
                                                      Meta* meta) {
  RJ_GRID_STRIDE(s, nc)
    chain_emit((uint32_t) s, nc, eb, cap, ch_key, ch_off, ch_total, base, slot_chain, slot_pieces, pbase, piece_begin, piece_len, run_first, meta);
}

struct Arena {
  char* base = nullptr;
  size_t used = 0, size = 0;
  template <typename T>
  T* take(uint64_t count) {
    used = (used + 255) & ~(size_t) 255;
    T* p = base ? reinterpret_cast<T*>(base + used) : nullptr;
    used += count * sizeof(T);
    return p;
  }
};

}  // namespace

// first launch of a kernel of this file = loading its code object (milliseconds): rj_create pays that, not a build
__global__ void k_st_noop() {}
hipError_t warm_stitch_kernels(hipStream_t st) {
  hipLaunchKernelGGL(k_st_noop, dim3(1), dim3(1), 0, st);
  return hipGetLastError();
}

void stitch_output_bounds(uint64_t nc, uint64_t ne, uint32_t cap, uint64_t* max_pieces, uint64_t* max_runs) {
  // runs: ceil(path / cap) per path <= ne / cap + paths; pieces: one per chain + one per run boundary inside a chain
  *max_runs = ne / cap + nc + 1;
  *max_pieces = nc + *max_runs + 1;
}

hipError_t stitch_runs_device(hipStream_t st, const int64_t* pts, const uint32_t* eb, uint64_t nc64, uint64_t ne, uint32_t cap,
                              uint32_t* piece_begin, uint32_t* piece_len, uint32_t* run_first, uint64_t* nruns, uint64_t* npieces,
                              uint32_t* stats, char** scratch, size_t* scratch_bytes) {
  *nruns = *npieces = 0;
  if (stats) stats[0] = stats[1] = stats[2] = stats[3] = 0;
  if (nc64 == 0) return hipMemsetAsync(run_first, 0, 4, st);
  if (nc64 >= (1ull << 30) || cap == 0) return hipErrorInvalidValue;  // (path keys are 2 bits wider than chain ids)
  const uint32_t nc = (uint32_t) nc64, ni = 2 * nc;
  int rounds = 2;
  while ((1ull << (rounds - 2)) <= nc && rounds < kMaxRounds) rounds++;
  // ---- scratch: one allocation, carved (sizes first, then the pointers) --------------------------
  size_t sort_bytes = 0, scan64_bytes = 0, scan32_bytes = 0;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, sort_bytes, (const uint64_t*) nullptr, (uint64_t*) nullptr, (const uint32_t*) nullptr,
                                           (uint32_t*) nullptr, (size_t) ni, 0, kKeyBits, st);
  if (e != hipSuccess) return e;
  e = rocprim::exclusive_scan(nullptr, scan64_bytes, (const uint64_t*) nullptr, (uint64_t*) nullptr, (uint64_t) 0, (size_t) 2 * ni,
                              rocprim::plus<uint64_t>(), st);
  if (e != hipSuccess) return e;
  e = rocprim::exclusive_scan(nullptr, scan32_bytes, (const uint32_t*) nullptr, (uint32_t*) nullptr, 0u, (size_t) nc, rocprim::plus<uint32_t>(), st);
  if (e != hipSuccess) return e;
  size_t temp_bytes = sort_bytes > scan64_bytes ? sort_bytes : scan64_bytes;
  if (scan32_bytes > temp_bytes) temp_bytes = scan32_bytes;
  Arena A;
  uint64_t *kx, *ky, *ska, *skb, *head, *base;
  uint32_t *va, *vb, *partner, *ch_key, *ch_off, *ch_total, *ch_rank, *slot_chain, *slot_pieces, *pbase;
  uint8_t *ch_back, *in_loop;
  Dir* dir;
  Node *n0, *n1;
  Meta* meta;
  void* temp;
  auto carve = [&]() {
    A.used = 0;
    meta = A.take<Meta>(1);
    // (the four key arrays are dead once the junctions are paired: the paths' totals and their scan live there later)
    kx = A.take<uint64_t>(4 * (uint64_t) ni);
    ky = kx + ni; ska = ky + ni; skb = ska + ni;
    head = kx; base = kx + 2 * (uint64_t) ni;
    va = A.take<uint32_t>(ni); vb = A.take<uint32_t>(ni);
    dir = A.take<Dir>(ni);
    partner = A.take<uint32_t>(ni);
    n0 = A.take<Node>(ni); n1 = A.take<Node>(ni);
    ch_key = A.take<uint32_t>(nc); ch_off = A.take<uint32_t>(nc); ch_total = A.take<uint32_t>(nc); ch_rank = A.take<uint32_t>(nc);
    slot_chain = A.take<uint32_t>(nc); slot_pieces = A.take<uint32_t>(nc); pbase = A.take<uint32_t>(nc);
    ch_back = A.take<uint8_t>(nc); in_loop = A.take<uint8_t>(nc);
    temp = A.take<char>(temp_bytes);
  };
  carve();
  A.size = A.used;
  // (the caller's grow-only block: freeing a block of its own cost every first build of a map 0.25 ms -- hipFree waits
  //  for the device and unmaps -- a sixth of the headline map's)
  if (*scratch_bytes < A.size) {
    (void) hipFree(*scratch);
    *scratch = nullptr; *scratch_bytes = 0;
    if ((e = hipMalloc((void**) scratch, A.size)) != hipSuccess) return e;
    *scratch_bytes = A.size;
  }
  A.base = *scratch;
  carve();
  Link* links = nullptr;
  Meta hm;
  const int B = blocks_for(ni), Bc = blocks_for(nc);
  const int Br = B > 2048 ? 2048 : B;  // the ranking rounds: one counter update per block
  do {
    if ((e = hipMemsetAsync(meta, 0, sizeof(Meta), st)) != hipSuccess) break;
    if ((e = hipMemsetAsync(partner, 0xFF, 4 * (size_t) ni, st)) != hipSuccess) break;
    if ((e = hipMemsetAsync(in_loop, 0, nc, st)) != hipSuccess) break;
    // 1. end points -> keys; 2. sort by y, then stably by x; pair every junction
    hipLaunchKernelGGL(k_st_end_keys, dim3(B > 2048 ? 2048 : B), dim3(kThreads), 0, st, ni, pts, eb, kx, ky, dir, va, meta);
    // (a map of closed rings has nothing to pair -- a ring's ends never take part -- and the two sorts of its 2 nc chain
    //  ends were half of this stage on the lake-shaped maps: one more look at the count decides)
    if ((e = hipMemcpyAsync(&hm, meta, sizeof(Meta), hipMemcpyDeviceToHost, st)) != hipSuccess) break;
    if ((e = hipStreamSynchronize(st)) != hipSuccess) break;
    size_t tb = temp_bytes;
    if (hm.closed_chains != nc) {
      if ((e = rocprim::radix_sort_pairs(temp, tb, ky, ska, va, vb, (size_t) ni, 0, kKeyBits, st)) != hipSuccess) break;
      hipLaunchKernelGGL(k_st_gather_keys, dim3(B), dim3(kThreads), 0, st, ni, kx, vb, skb);
      tb = temp_bytes;
      if ((e = rocprim::radix_sort_pairs(temp, tb, skb, ska, vb, va, (size_t) ni, 0, kKeyBits, st)) != hipSuccess) break;
      hipLaunchKernelGGL(k_st_pair, dim3(B), dim3(kThreads), 0, st, ni, ska, va, ky, dir, partner);
    }
    // 3. list ranking
    hipLaunchKernelGGL(k_st_rank_init, dim3(B), dim3(kThreads), 0, st, ni, partner, eb, n0, n1);
    for (int r = 0; r < rounds; r++)
      hipLaunchKernelGGL(k_st_rank_round, dim3(Br), dim3(kThreads), 0, st, ni, (r & 1) ? n1 : n0, (r & 1) ? n0 : n1, meta, r, 0);
    hipLaunchKernelGGL(k_st_rank_done, dim3(1), dim3(1), 0, st, meta, rounds, 0);
    if ((e = hipGetLastError()) != hipSuccess) break;
    if ((e = hipMemcpyAsync(&hm, meta, sizeof(Meta), hipMemcpyDeviceToHost, st)) != hipSuccess) break;
    if ((e = hipStreamSynchronize(st)) != hipSuccess) break;
    const uint32_t on_loops = hm.done_round ? hm.act[hm.done_round - 1] : 0;
    if (stats) { stats[0] = hm.done_round; stats[1] = on_loops; stats[3] = hm.closed_chains; }
    int second = 0;
    if (on_loops) {
      // 3b. closed loops of paired chains (rare): find each loop's smallest incidence, open the loop there, rank again
      if ((e = hipMalloc((void**) &links, 2 * (size_t) ni * sizeof(Link))) != hipSuccess) break;
      Link *l0 = links, *l1 = links + ni;
      hipLaunchKernelGGL(k_st_cyc_init, dim3(B), dim3(kThreads), 0, st, ni, n0, n1, meta, partner, l0, l1);
      int cr = 1;
      while ((1ull << cr) < nc) cr++;
      for (int r = 0; r < cr; r++) hipLaunchKernelGGL(k_st_cyc_round, dim3(B), dim3(kThreads), 0, st, ni, (r & 1) ? l1 : l0, (r & 1) ? l0 : l1);
      hipLaunchKernelGGL(k_st_cyc_break, dim3(B), dim3(kThreads), 0, st, ni, (cr & 1) ? l1 : l0, partner, in_loop);
      hipLaunchKernelGGL(k_st_rerank_init, dim3(B), dim3(kThreads), 0, st, ni, n0, n1, meta, partner, eb);
      for (int r = 0; r < rounds; r++)
        hipLaunchKernelGGL(k_st_rank_round, dim3(Br), dim3(kThreads), 0, st, ni, (r & 1) ? n1 : n0, (r & 1) ? n0 : n1, meta, r, 1);
      hipLaunchKernelGGL(k_st_rank_done, dim3(1), dim3(1), 0, st, meta, rounds, 1);
      second = 1;
    }
    // 4. chains in their paths: totals at the paths' keys, scan, slots, scan, pieces
    if ((e = hipMemsetAsync(head, 0, 16 * (size_t) ni, st)) != hipSuccess) break;
    hipLaunchKernelGGL(k_st_orient, dim3(Bc), dim3(kThreads), 0, st, nc, n0, n1, meta, second, eb, in_loop, cap, ch_key, ch_off, ch_total, ch_rank,
                       ch_back, head);
    tb = temp_bytes;
    if ((e = rocprim::exclusive_scan(temp, tb, head, base, (uint64_t) 0, (size_t) 2 * ni, rocprim::plus<uint64_t>(), st)) != hipSuccess) break;
    hipLaunchKernelGGL(k_st_place, dim3(Bc), dim3(kThreads), 0, st, nc, eb, cap, ch_key, ch_off, ch_total, ch_rank, ch_back, base, slot_chain, slot_pieces);
    tb = temp_bytes;
    if ((e = rocprim::exclusive_scan(temp, tb, slot_pieces, pbase, 0u, (size_t) nc, rocprim::plus<uint32_t>(), st)) != hipSuccess) break;
    hipLaunchKernelGGL(k_st_emit, dim3(Bc), dim3(kThreads), 0, st, nc, eb, cap, ch_key, ch_off, ch_total, base, slot_chain, slot_pieces, pbase,
                       piece_begin, piece_len, run_first, meta);
    if ((e = hipGetLastError()) != hipSuccess) break;
    if ((e = hipMemcpyAsync(&hm, meta, sizeof(Meta), hipMemcpyDeviceToHost, st)) != hipSuccess) break;
    if ((e = hipStreamSynchronize(st)) != hipSuccess) break;
    if (second) {
      if (stats) stats[2] = hm.done_round2;
      if (hm.done_round2 && hm.act2[hm.done_round2 - 1]) { e = hipErrorUnknown; break; }  // (cannot happen: every loop was opened)
    }
    *nruns = hm.nruns;
    *npieces = hm.npieces;
  } while (0);
  (void) hipFree(links);
  return e;
}

}  // namespace rj
