// rj_grid.hip -- `-mode=grid` on the device: the reference's uniform-grid index and
// its two query kernels, for the three-way comparison grid / lbvh / rt that the paper runs.
//   UniformGrid::AddMapsToGrid / AddMapToGrid   src/grid/uniform_grid.h:132-349
//   iterate_cell (edge -> covered cells)          src/grid/uniform_grid.h:45-86
//   calculate_cell                                src/grid/cell.h:16-22
//   LSIGrid: intersect_one_cell                   src/app/lsi_grid.h:19-78
//   PIPGrid::Query + cell acceptance              src/app/pip_grid.h:37-70, src/algo/pip.h:98-114
// Layout: one CSR per map -- begin[g*g + 1] (u32) and the eids of every cell in ASCENDING eid order
// (the PIP tie rule is visit-order dependent; ascending is what the oracle does).  Built as
// (cell, eid) 64-bit keys + one radix sort instead of the reference's count / scan / atomic fill,
// whose order inside a cell is whatever the atomics produced.
#include "rj_kernels.h"

#include <cstring>
#include <rocprim/device/device_scan.hpp>

namespace rj {

namespace {

constexpr int64_t kGridIMin = -((int64_t) 1 << 46);  // INTERNAL_MIN (scaling.h:45)

__device__ __forceinline__ int cell_of_int(int64_t v, double scale) {  // cell.h:16-22, integer argument
  return (int) ((double) (v - kGridIMin) * scale);
}
__device__ __forceinline__ int cell_of_double(double v, double scale) {
  return (int) ((v - (double) kGridIMin) * scale);
}
// rational argument: operator-(rational, integer) builds and simplifies num - t den over den
// (rational.h:390-398), operator double divides (rational.h:190-192)
__device__ __forceinline__ int cell_of_rat(const Rat& r, double scale) {
  const Rat d = rat_make((i128) ((u128) r.num - (u128) (i128) kGridIMin * (u128) r.den), r.den);
  return (int) (rat_to_double(d) * scale);
}

struct CellRange {
  int x1, x2, y1, y2;
};
__device__ __forceinline__ CellRange cells_of(const Seg& s, double scale) {  // uniform_grid.h:63-77
  int ax = cell_of_int(s.x1, scale), ay = cell_of_int(s.y1, scale);
  int bx = cell_of_int(s.x2, scale), by = cell_of_int(s.y2, scale);
  CellRange r;
  r.x1 = ax < bx ? ax : bx; r.x2 = ax < bx ? bx : ax;
  r.y1 = ay < by ? ay : by; r.y2 = ay < by ? by : ay;
  return r;
}

// pass 1: per-cell counts + total number of (cell, edge) incidences
__global__ __launch_bounds__(256) void k_grid_count(const Seg* __restrict__ seg, uint64_t ne, int g, double scale,
                                                    uint32_t* __restrict__ counts,
                                                    unsigned long long* __restrict__ total) {
  unsigned long long mine = 0;
  for (uint64_t e = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; e < ne; e += (uint64_t) gridDim.x * blockDim.x) {
    const CellRange r = cells_of(seg[e], scale);
    for (int j = r.y1; j <= r.y2; j++)
      for (int i = r.x1; i <= r.x2; i++) atomicAdd(&counts[(size_t) j * g + i], 1u);
    mine += (unsigned long long) (r.x2 - r.x1 + 1) * (r.y2 - r.y1 + 1);
  }
  for (int d = 32; d; d >>= 1) mine += __shfl_down(mine, d, 64);
  if (lane_id() == 0 && mine) atomicAdd(total, mine);
}

// pass 2: (cell << 32 | eid) keys, appended with one atomic per wave
__global__ __launch_bounds__(256) void k_grid_emit(const Seg* __restrict__ seg, uint64_t ne, int g, double scale,
                                                   uint64_t* __restrict__ keys, unsigned long long* __restrict__ cursor) {
  const int lane = lane_id();
  const uint64_t stride = (uint64_t) gridDim.x * blockDim.x;
  for (uint64_t base = blockIdx.x * (uint64_t) blockDim.x + (threadIdx.x & ~63u); base < ne; base += stride) {  // wave-uniform
    const uint64_t e = base + lane;
    CellRange r = {0, -1, 0, -1};
    if (e < ne) r = cells_of(seg[e], scale);
    const unsigned long long n = e < ne ? (unsigned long long) (r.x2 - r.x1 + 1) * (r.y2 - r.y1 + 1) : 0;
    unsigned long long incl = n;  // inclusive scan over the wave
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned long long t = __shfl_up(incl, d, 64);
      if (lane >= d) incl += t;
    }
    const unsigned long long wave_total = __shfl(incl, 63, 64);
    unsigned long long b = 0;
    if (lane == 0 && wave_total) b = atomicAdd(cursor, wave_total);
    b = __shfl(b, 0, 64);
    unsigned long long pos = b + incl - n;
    for (int j = r.y1; j <= r.y2; j++)
      for (int i = r.x1; i <= r.x2; i++) keys[pos++] = ((uint64_t) ((size_t) j * g + i) << 32) | (uint32_t) e;
  }
}

__global__ __launch_bounds__(256) void k_grid_unpack(const uint64_t* __restrict__ keys, uint64_t n, uint32_t* __restrict__ eids) {
  for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x)
    eids[i] = (uint32_t) keys[i];
}

// LSIGrid (lsi_grid.h:19-78, 112-121): in every cell all (map-0 edge, map-1 edge) pairs; a hit is
// reported only by the cell that contains the computed intersection point.  One wave per cell
// (64 cells are inspected per step, the non-empty ones are then worked through by all lanes).
__global__ __launch_bounds__(256) void k_lsi_grid(GridLsiArgs A) {
  const int lane = lane_id();
  const uint64_t ncells = (uint64_t) A.g * A.g;
  const uint64_t wave = (blockIdx.x * (uint64_t) blockDim.x + threadIdx.x) >> 6;
  const uint64_t nwaves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
  for (uint64_t c0 = wave * 64; c0 < ncells; c0 += nwaves * 64) {
    const uint64_t cme = c0 + lane;
    uint32_t b0 = 0, n0 = 0, b1 = 0, n1 = 0;
    if (cme < ncells) {
      b0 = A.begin0[cme]; n0 = A.begin0[cme + 1] - b0;
      b1 = A.begin1[cme]; n1 = A.begin1[cme + 1] - b1;
    }
    uint64_t active = __ballot(n0 && n1);
    while (active) {
      const int l = __builtin_ctzll(active);
      active &= active - 1;
      const uint32_t cb0 = bcast((int32_t) b0, l), cn0 = bcast((int32_t) n0, l);
      const uint32_t cb1 = bcast((int32_t) b1, l), cn1 = bcast((int32_t) n1, l);
      const uint64_t c = c0 + l;
      const int cx = (int) (c % (uint64_t) A.g), cy = (int) (c / (uint64_t) A.g);
      const uint64_t npairs = (uint64_t) cn0 * cn1;
      for (uint64_t t0 = 0; t0 < npairs; t0 += 64) {
        const uint64_t t = t0 + lane;
        bool hit = false;
        uint32_t e0 = 0, e1 = 0;
        if (t < npairs) {
          e0 = A.eids0[cb0 + (uint32_t) (t / cn1)];
          e1 = A.eids1[cb1 + (uint32_t) (t % cn1)];
          const Seg s0 = A.seg0[e0], s1 = A.seg1[e1];
          if (lsi_test(s0, s1)) {  // e1 = map-0 edge, e2 = map-1 edge (lsi_grid.h:103-104)
            Rat x, y;
            lsi_point(s0, make_eqn(s0), s1, make_eqn(s1), &x, &y);
            hit = cell_of_rat(x, A.scale) == cx && cell_of_rat(y, A.scale) == cy;  // lsi_grid.h:62-74
          }
        }
        const uint64_t hm = __ballot(hit);
        if (hm) {
          unsigned long long base = 0;
          if (lane == 0) base = atomicAdd(A.counter, (unsigned long long) __popcll(hm));
          base = ((unsigned long long) __builtin_amdgcn_readfirstlane((uint32_t) (base >> 32)) << 32) |
                 __builtin_amdgcn_readfirstlane((uint32_t) base);
          const unsigned long long pos = base + rank_below(hm);
          if (hit && pos < A.cap) reinterpret_cast<uint2*>(A.out)[pos] = make_uint2(e0, e1);
        }
      }
    }
  }
}

// PIPGrid::Query (pip_grid.h:37-70): walk up the point's column of cells; in each cell visit the
// base map's edges in ascending eid with the predicate of pip.h:31-96; accept the cell's best edge
// only when the hit lies in this cell (pip.h:98-114).
__global__ __launch_bounds__(256) void k_pip_grid(GridPipArgs A) {
  for (uint64_t ip = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; ip < A.n; ip += (uint64_t) gridDim.x * blockDim.x) {
    const int64_t px = A.pts[2 * ip], py = A.pts[2 * ip + 1];
    const int cx = cell_of_int(px, A.scale), cy = cell_of_int(py, A.scale);
    uint32_t closest = 0xFFFFFFFFu;
    int32_t face = 0;
    for (int ccy = cy; ccy < A.g && closest == 0xFFFFFFFFu; ccy++) {
      const size_t c = (size_t) ccy * A.g + cx;
      const uint32_t b = A.begin[c], e = A.begin[c + 1];
      double best_y = __builtin_inf();
      uint32_t best = 0xFFFFFFFFu;
      Seg best_seg = {0, 0, 0, 0};
      for (uint32_t k = b; k < e; k++) {
        const uint32_t eid = A.eids[k];
        const Seg s = A.base.seg[eid];
        double yy;
        if (!pip_eval_y(s, px, py, A.query_map_id, &yy)) continue;
        if (yy > best_y) continue;  // pip.h:73-75
        if (yy == best_y) {         // pip.h:77-93
          const bool flag = pip_slope(s) > pip_slope(best_seg);
          if ((A.query_map_id && !flag) || (flag && !A.query_map_id)) continue;
        }
        best_y = yy;
        best = eid;
        best_seg = s;
      }
      if (best == 0xFFFFFFFFu) continue;
      const int64_t y_max = best_seg.y1 > best_seg.y2 ? best_seg.y1 : best_seg.y2;  // pip.h:96
      if (cell_of_int(y_max, A.scale) == ccy || !(cell_of_double(best_y, A.scale) > ccy)) {
        closest = best;
        const uint32_t ch = A.base.edge_chain[best];
        face = (int32_t) (best_seg.x1 < best_seg.x2 ? A.base.right[ch] : A.base.left[ch]);  // map.h:79-87
      }
    }
    A.closest[ip] = closest;
    if (A.face) A.face[ip] = face;
  }
}

static inline int grid_blocks(uint64_t items, int per_block, int max_blocks) {
  uint64_t b = (items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  return (int) (b > (uint64_t) max_blocks ? max_blocks : b);
}

}  // namespace

hipError_t launch_grid_count(hipStream_t st, const Seg* seg, uint64_t ne, int g, double scale, uint32_t* counts,
                             unsigned long long* total) {
  if (ne) hipLaunchKernelGGL(k_grid_count, dim3(grid_blocks(ne, 256, 8192)), dim3(256), 0, st, seg, ne, g, scale, counts, total);
  return hipGetLastError();
}
hipError_t launch_grid_emit(hipStream_t st, const Seg* seg, uint64_t ne, int g, double scale, uint64_t* keys,
                            unsigned long long* cursor) {
  if (ne) hipLaunchKernelGGL(k_grid_emit, dim3(grid_blocks(ne, 256, 8192)), dim3(256), 0, st, seg, ne, g, scale, keys, cursor);
  return hipGetLastError();
}
hipError_t launch_grid_unpack(hipStream_t st, const uint64_t* keys, uint64_t n, uint32_t* eids) {
  if (n) hipLaunchKernelGGL(k_grid_unpack, dim3(grid_blocks(n, 256, 8192)), dim3(256), 0, st, keys, n, eids);
  return hipGetLastError();
}
__global__ void k_grid_noop() {}
hipError_t warm_grid_kernels(hipStream_t st) {  // (see warm_stitch_kernels)
  hipLaunchKernelGGL(k_grid_noop, dim3(1), dim3(1), 0, st);
  return hipGetLastError();
}
hipError_t scan_cell_counts(hipStream_t st, void* temp, size_t& temp_bytes, const uint32_t* counts, uint32_t* begin, uint64_t n) {
  return rocprim::exclusive_scan(temp, temp_bytes, counts, begin, 0u, (size_t) n, rocprim::plus<uint32_t>(), st);
}
hipError_t launch_lsi_grid(hipStream_t st, const GridLsiArgs& a) {
  const uint64_t ncells = (uint64_t) a.g * a.g;
  hipLaunchKernelGGL(k_lsi_grid, dim3(grid_blocks(ncells, 256, 16384)), dim3(256), 0, st, a);
  return hipGetLastError();
}
hipError_t launch_pip_grid(hipStream_t st, const GridPipArgs& a) {
  if (a.n) hipLaunchKernelGGL(k_pip_grid, dim3(grid_blocks(a.n, 256, 16384)), dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace rj
