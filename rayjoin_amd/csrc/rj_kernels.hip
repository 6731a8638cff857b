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
// edge_begin[c] = row_index[c] - c = first eid of chain c (strictly increasing), [nc+1]
__global__ __launch_bounds__(256) void k_build_segs(const int64_t* __restrict__ pts,
                                                    const uint32_t* __restrict__ edge_begin,
                                                    uint32_t nc, uint64_t ne, Seg* __restrict__ seg,
                                                    uint32_t* __restrict__ edge_chain) {
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

// 64-bit Morton key straight from the int64 midpoint (no float, 32 bits per axis, y is the MSB)
__global__ __launch_bounds__(256) void k_morton(const Seg* __restrict__ seg, uint64_t ne,
                                                uint64_t* __restrict__ keys,
                                                uint32_t* __restrict__ vals) {
  for (uint64_t e = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; e < ne;
       e += (uint64_t) gridDim.x * blockDim.x) {
    Seg s = seg[e];
    uint64_t mx = (uint64_t) (((s.x1 + s.x2) >> 1) + kCoordOffset);  // 47 bits
    uint64_t my = (uint64_t) (((s.y1 + s.y2) >> 1) + kCoordOffset);
    uint32_t ux = (uint32_t) (mx >> 15), uy = (uint32_t) (my >> 15);
    keys[e] = (spread32(uy) << 1) | spread32(ux);
    vals[e] = (uint32_t) e;
  }
}

__global__ __launch_bounds__(256) void k_gather_sorted(const Seg* __restrict__ seg,
                                                       const uint32_t* __restrict__ order,
                                                       uint64_t ne, uint64_t n0p,
                                                       Seg* __restrict__ sseg,
                                                       uint32_t* __restrict__ seid,
                                                       QBox* __restrict__ box0) {
  for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n0p;
       i += (uint64_t) gridDim.x * blockDim.x) {
    Seg s = {0, 0, 0, 0};
    QBox b = {kEmptyMin, kEmptyMin, kEmptyMax, kEmptyMax};
    uint32_t id = 0xFFFFFFFFu;
    if (i < ne) {
      id = order[i];
      s = seg[id];
      b.x0 = quant(s.x1 < s.x2 ? s.x1 : s.x2);
      b.x1 = quant(s.x1 < s.x2 ? s.x2 : s.x1);
      b.y0 = quant(s.y1 < s.y2 ? s.y1 : s.y2);
      b.y1 = quant(s.y1 < s.y2 ? s.y2 : s.y1);
    }
    sseg[i] = s;
    seid[i] = id;
    box0[i] = b;
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

// =============================================================================================
// LSI: wave-cooperative traversal, LDS stack, ballot-compacted candidate pairs, dense predicate
// =============================================================================================
struct LsiWaveLds {
  uint32_t stack[kStackEntries];
  uint2 pairs[kPairBuf];  // (query eid, sorted base slot)
  uint2 hits[kPairBuf];   // (eid map 0, eid map 1)
};


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
      hit = lsi_test(bs, make_eqn(bs), qs, make_eqn(qs));
      h.x = beid; h.y = pr.x;
    } else {
      hit = lsi_test(qs, make_eqn(qs), bs, make_eqn(bs));
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

template <bool STATS>
__global__ __launch_bounds__(256) void k_lsi(LsiArgs A) {
  __shared__ LsiWaveLds lds[4];
  const int lane = lane_id();
  const int wib = threadIdx.x >> 6;
  LsiWaveLds& L = lds[wib];
  const uint64_t nwaves = (uint64_t) gridDim.x * 4;
  const uint64_t nq = A.qend - A.qbeg;
  const uint64_t ngroups = (nq + 63) >> 6;
  const DeviceBvh& T = A.bvh;
  int np = 0, nh = 0;  // wave-uniform fill of L.pairs / L.hits
  unsigned long long st_leaf = 0, st_tests = 0, st_nodes = 0, st_box = 0;
  long long tk_node = 0, tk_leaf = 0, tk_head = 0;  // STATS: cycle stamps
  const long long tk_begin = STATS ? clock64() : 0;

  for (uint64_t g = (uint64_t) blockIdx.x * 4 + wib; g < ngroups; g += nwaves) {
    const long long tkg = STATS ? clock64() : 0;
    const uint64_t q = A.qbeg + g * 64 + lane;
    const bool valid = q < A.qend;
    int32_t qx0 = kEmptyMin, qy0 = kEmptyMin, qx1 = kEmptyMax, qy1 = kEmptyMax;
    if (valid) {
      Seg s = A.qseg[q];
      qx0 = quant(s.x1 < s.x2 ? s.x1 : s.x2);
      qx1 = quant(s.x1 < s.x2 ? s.x2 : s.x1);
      qy0 = quant(s.y1 < s.y2 ? s.y1 : s.y2);
      qy1 = quant(s.y1 < s.y2 ? s.y2 : s.y1);
    }
    const int32_t gx0 = wave_min(qx0), gy0 = wave_min(qy0);
    const int32_t gx1 = wave_max(qx1), gy1 = wave_max(qy1);

    // A child is pushed only if SOME lane's own query box overlaps it: the union box is just
    // a cheap first filter.  The wave therefore visits exactly the union of the nodes its 64
    // queries need, whatever the spatial coherence of the group.
    auto refine = [&](const QBox& b, uint64_t um) -> uint64_t {
      uint64_t keep = 0;
      while (um) {
        const int c = __builtin_ctzll(um);
        um &= um - 1;
        const int32_t cx0 = bcast(b.x0, c), cy0 = bcast(b.y0, c);
        const int32_t cx1 = bcast(b.x1, c), cy1 = bcast(b.y1, c);
        if (__ballot(qx0 <= cx1 && cx0 <= qx1 && qy0 <= cy1 && cy0 <= qy1)) keep |= 1ull << c;
      }
      return keep;
    };
    int sp = 0;
    {  // top level: <= 64 nodes, one per lane
      QBox b = T.lvl[T.top][lane];
      uint64_t m = refine(b, __ballot(overlap(b, gx0, gy0, gx1, gy1)));
      if ((m >> lane) & 1) L.stack[rank_below(m)] = ((uint32_t) T.top << 28) | (uint32_t) lane;
      sp = __popcll(m);
      wave_lds_fence();
    }
    if (STATS) tk_head += clock64() - tkg;
    while (sp > 0) {
      uint32_t e = __builtin_amdgcn_readfirstlane(L.stack[sp - 1]);
      --sp;
      const int lvl = (int) (e >> 28);
      const uint32_t idx = e & 0x0FFFFFFFu;
      const long long tk0 = STATS ? clock64() : 0;
      if (lvl > 1) {
        QBox b = T.lvl[lvl - 1][(uint64_t) idx * 64 + lane];
        uint64_t m = refine(b, __ballot(overlap(b, gx0, gy0, gx1, gy1)));
        if ((m >> lane) & 1) L.stack[sp + rank_below(m)] = ((uint32_t) (lvl - 1) << 28) | (idx * 64 + lane);
        sp += __popcll(m);
        if (STATS) st_nodes++;
        wave_lds_fence();
        if (STATS) tk_node += clock64() - tk0;
      } else {
        // leaf block: 64 base segments, one box per lane
        const uint32_t slot0 = idx * 64;
        QBox bb = T.box0[(uint64_t) slot0 + lane];
        uint64_t bm = __ballot(overlap(bb, gx0, gy0, gx1, gy1));
        if (STATS) st_leaf++;
        while (bm) {
          const int b = __builtin_ctzll(bm);
          bm &= bm - 1;
          const int32_t bx0 = bcast(bb.x0, b), by0 = bcast(bb.y0, b);
          const int32_t bx1 = bcast(bb.x1, b), by1 = bcast(bb.y1, b);
          // (invalid lanes hold an empty box and can never overlap)
          const bool c = qx0 <= bx1 && bx0 <= qx1 && qy0 <= by1 && by0 <= qy1;
          const uint64_t cm = __ballot(c);
          if (STATS) st_box++;
          if (cm) {
            if (c) L.pairs[np + rank_below(cm)] = make_uint2((uint32_t) q, slot0 + b);
            np += __popcll(cm);
            wave_lds_fence();
            if (np >= 64) lsi_drain<STATS>(L, np, nh, 64, A, lane, st_tests);
          }
        }
        if (STATS) tk_leaf += clock64() - tk0;
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
    atomicMax(&A.stats[9], (unsigned long long) tk_total);
  }
}

// =============================================================================================
// LSI intersection points (per hit only): rational point, clamp, narrowing store
// =============================================================================================
__global__ __launch_bounds__(256) void k_lsi_points(const Seg* __restrict__ seg0,
                                                    const Seg* __restrict__ seg1,
                                                    const uint32_t* __restrict__ pairs, uint64_t n,
                                                    XsectRec* __restrict__ out) {
  for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n;
       i += (uint64_t) gridDim.x * blockDim.x) {
    uint32_t e0 = pairs[2 * i], e1 = pairs[2 * i + 1];
    Seg s1 = seg0[e0], s2 = seg1[e1];
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

__global__ __launch_bounds__(256) void k_swap_halves(uint64_t* __restrict__ v, uint64_t n) {
  for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n;
       i += (uint64_t) gridDim.x * blockDim.x) {
    uint64_t x = v[i];
    v[i] = (x << 32) | (x >> 32);
  }
}

// =============================================================================================
// PIP: upward ray through the same tree; per-lane best with pruning; compacted (point, edge)
// candidates evaluated densely and merged back through an LDS mailbox
// =============================================================================================
struct PipWaveLds {
  uint32_t stack[kStackEntries];
  uint2 pairs[kPairBuf];  // (query lane, sorted base slot)
  uint32_t mailbox[64];
  double res_yy[64];
  double res_slope[64];
  uint32_t res_eid[64];
};


__device__ __forceinline__ int32_t quant_best(double yy) {
  // conservative quantised upper bound of a finite best y (+1 margin, see DESIGN.md "PIP pruning")
  double t = (yy + (double) kCoordOffset) * (1.0 / 65536.0);
  if (!(t < 2147483000.0)) return 0x7FFFFFFF;
  if (t < -1.0) return -1;
  return (int32_t) t + 1;
}

template <bool STATS>
__global__ __launch_bounds__(256) void k_pip(PipArgs A) {
  __shared__ PipWaveLds lds[4];
  const int lane = lane_id();
  const int wib = threadIdx.x >> 6;
  PipWaveLds& L = lds[wib];
  const uint64_t nwaves = (uint64_t) gridDim.x * 4;
  const uint64_t ngroups = (A.n + 63) >> 6;
  const DeviceBvh& T = A.bvh;
  const int qm = A.query_map_id;
  unsigned long long st_leaf = 0, st_tests = 0, st_nodes = 0, st_box = 0;
  long long tk_drain = 0, tk_leaf = 0, tk_node = 0, tk_total = 0, tk_rounds = 0;  // STATS: cycle stamps
  const long long tk_begin = STATS ? clock64() : 0;

  for (uint64_t g = (uint64_t) blockIdx.x * 4 + wib; g < ngroups; g += nwaves) {
    const uint64_t ip = g * 64 + lane;
    const bool valid = ip < A.n;
    int64_t px = 0, py = 0;
    if (valid) {
      longlong2 p = reinterpret_cast<const longlong2*>(A.pts)[ip];
      px = p.x; py = p.y;
    }
    const int32_t qx = quant(px), qy = quant(py);
    const int32_t gx0 = wave_min(valid ? qx : kEmptyMin);
    const int32_t gx1 = wave_max(valid ? qx : kEmptyMax);
    const int32_t gy0 = wave_min(valid ? qy : kEmptyMin);
    double best_yy = __builtin_inf(), best_slope = 0.0;
    uint32_t best_eid = 0xFFFFFFFFu;
    int32_t lane_qbest = valid ? 0x7FFFFFFF : -1;  // quantised bound on this lane's best y
    int32_t gbest = 0x7FFFFFFF;                    // wave max of lane_qbest
    int np = 0;
    L.mailbox[lane] = 0xFFFFFFFFu;

    auto drain = [&](int n) {
      // evaluate the top n (<= 64) candidates densely, then deliver results to their query lane
      const long long tk0 = STATS ? clock64() : 0;
      bool pending = false;
      int ql = 0;
      double yy = 0, slope = 0;
      uint32_t eid = 0;
      uint2 pr = make_uint2(0, 0);
      if (lane < n) pr = L.pairs[np - n + lane];
      ql = (int) pr.x;
      // the query lane's point (all lanes take part in the shuffle)
      const int64_t qpx = ((int64_t) __shfl((int) (px >> 32), ql, 64) << 32) |
                          (uint32_t) __shfl((int) (uint32_t) px, ql, 64);
      const int64_t qpy = ((int64_t) __shfl((int) (py >> 32), ql, 64) << 32) |
                          (uint32_t) __shfl((int) (uint32_t) py, ql, 64);
      if (lane < n) {
        Seg bs = T.sseg[pr.y];
        eid = T.seid[pr.y];
        pending = pip_eval(bs, qpx, qpy, qm, &yy, &slope);
      }
      np -= n;
      if (STATS) st_tests += n;
      while (__ballot(pending)) {
        if (STATS) tk_rounds++;
        if (pending) L.mailbox[ql] = (uint32_t) lane;  // one winner per query lane
        wave_lds_fence();
        const bool win = pending && L.mailbox[ql] == (uint32_t) lane;
        if (win) {
          L.res_yy[ql] = yy;
          L.res_slope[ql] = slope;
          L.res_eid[ql] = eid;
        }
        wave_lds_fence();
        if (L.mailbox[lane] != 0xFFFFFFFFu) {
          const double ryy = L.res_yy[lane], rsl = L.res_slope[lane];
          const uint32_t reid = L.res_eid[lane];
          if (pip_better(ryy, rsl, reid, best_yy, best_slope, best_eid, qm)) {
            best_yy = ryy; best_slope = rsl; best_eid = reid;
          }
          L.mailbox[lane] = 0xFFFFFFFFu;
        }
        pending = pending && !win;
        wave_lds_fence();
      }
      if (valid && best_eid != 0xFFFFFFFFu) lane_qbest = quant_best(best_yy);
      gbest = wave_max(lane_qbest);
      if (STATS) tk_drain += clock64() - tk0;
    };

    // push-time per-lane culling (see k_lsi): a child survives only if SOME lane's upward ray
    // can still hit it given that lane's current best
    auto refine = [&](const QBox& b, uint64_t um) -> uint64_t {
      uint64_t keep = 0;
      while (um) {
        const int c = __builtin_ctzll(um);
        um &= um - 1;
        const int32_t cx0 = bcast(b.x0, c), cy0 = bcast(b.y0, c);
        const int32_t cx1 = bcast(b.x1, c), cy1 = bcast(b.y1, c);
        if (__ballot(cx0 <= qx && qx <= cx1 && cy1 >= qy - 1 && cy0 <= lane_qbest)) keep |= 1ull << c;
      }
      return keep;
    };
    int sp = 0;
    {
      QBox b = T.lvl[T.top][lane];
      uint64_t m = refine(b, __ballot(b.x0 <= gx1 && gx0 <= b.x1 && b.y1 >= gy0 - 1));
      const int cnt = __popcll(m);
      // reversed so that lane 0's child (lowest Morton = lowest y half) pops first
      if ((m >> lane) & 1) L.stack[cnt - 1 - rank_below(m)] = ((uint32_t) T.top << 28) | (uint32_t) lane;
      sp = cnt;
      wave_lds_fence();
    }
    while (sp > 0) {
      uint32_t e = __builtin_amdgcn_readfirstlane(L.stack[sp - 1]);
      --sp;
      const int lvl = (int) (e >> 28);
      const uint32_t idx = e & 0x0FFFFFFFu;
      if (lvl > 1) {
        const long long tk0 = STATS ? clock64() : 0;
        QBox b = T.lvl[lvl - 1][(uint64_t) idx * 64 + lane];
        uint64_t m = refine(b, __ballot(b.x0 <= gx1 && gx0 <= b.x1 && b.y1 >= gy0 - 1 && b.y0 <= gbest));
        const int cnt = __popcll(m);
        if ((m >> lane) & 1) L.stack[sp + cnt - 1 - rank_below(m)] = ((uint32_t) (lvl - 1) << 28) | (idx * 64 + lane);
        sp += cnt;
        if (STATS) st_nodes++;
        wave_lds_fence();
        if (STATS) tk_node += clock64() - tk0;
      } else {
        const long long tk0 = STATS ? clock64() : 0;
        const long long tkd0 = tk_drain;
        const uint32_t slot0 = idx * 64;
        QBox bb = T.box0[(uint64_t) slot0 + lane];
        uint64_t bm = __ballot(bb.x0 <= gx1 && gx0 <= bb.x1 && bb.y1 >= gy0 - 1 && bb.y0 <= gbest);
        if (STATS) st_leaf++;
        while (bm) {
          const int b = __builtin_ctzll(bm);
          bm &= bm - 1;
          const int32_t bx0 = bcast(bb.x0, b), by0 = bcast(bb.y0, b);
          const int32_t bx1 = bcast(bb.x1, b), by1 = bcast(bb.y1, b);
          const bool c = bx0 <= qx && qx <= bx1 && by1 >= qy - 1 && by0 <= lane_qbest;
          const uint64_t cm = __ballot(c);
          if (STATS) st_box++;
          if (cm) {
            if (c) L.pairs[np + rank_below(cm)] = make_uint2((uint32_t) lane, slot0 + b);
            np += __popcll(cm);
            wave_lds_fence();
            if (np >= 64) drain(64);
          }
        }
        // drain eagerly: the sooner a lane knows its best, the more of the column above it is
        // pruned (a partial drain costs far less than one more leaf block)
        if (np > 0) drain(np);
        if (STATS) tk_leaf += (clock64() - tk0) - (tk_drain - tkd0);
      }
    }
    if (np > 0) drain(np);
    if (valid) {
      A.closest[ip] = best_eid;
      if (A.face) {
        int32_t f = 0;  // EXTERIOR_FACE_ID
        if (best_eid != 0xFFFFFFFFu) {
          Seg s = A.base.seg[best_eid];
          uint32_t c = A.base.edge_chain[best_eid];
          f = (int32_t) (s.x1 < s.x2 ? A.base.right[c] : A.base.left[c]);  // map.h:79-87
        }
        A.face[ip] = f;
      }
    }
  }
  if (STATS && lane == 0 && A.stats) {
    tk_total = clock64() - tk_begin;
    atomicAdd(&A.stats[0], st_leaf);
    atomicAdd(&A.stats[1], st_tests);
    atomicAdd(&A.stats[2], st_nodes);
    atomicAdd(&A.stats[3], st_box);
    atomicAdd(&A.stats[4], (unsigned long long) tk_total);
    atomicAdd(&A.stats[5], (unsigned long long) tk_node);
    atomicAdd(&A.stats[6], (unsigned long long) tk_leaf);
    atomicAdd(&A.stats[7], (unsigned long long) tk_drain);
    atomicAdd(&A.stats[8], (unsigned long long) tk_rounds);
    atomicMax(&A.stats[9], (unsigned long long) tk_total);
  }
}

// =============================================================================================
// launch wrappers
// =============================================================================================
static inline int grid_for(uint64_t work_items, int per_block, int max_blocks) {
  uint64_t b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > (uint64_t) max_blocks) b = max_blocks;
  return (int) b;
}

hipError_t launch_build_segs(hipStream_t st, const int64_t* pts, const uint32_t* edge_begin,
                             uint32_t nc, uint64_t ne, Seg* seg, uint32_t* edge_chain) {
  if (ne == 0) return hipSuccess;
  hipLaunchKernelGGL(k_build_segs, dim3(grid_for(ne, 256, 8192)), dim3(256), 0, st, pts, edge_begin,
                     nc, ne, seg, edge_chain);
  return hipGetLastError();
}

hipError_t launch_morton(hipStream_t st, const Seg* seg, uint64_t ne, uint64_t* keys, uint32_t* vals) {
  if (ne == 0) return hipSuccess;
  hipLaunchKernelGGL(k_morton, dim3(grid_for(ne, 256, 8192)), dim3(256), 0, st, seg, ne, keys, vals);
  return hipGetLastError();
}

hipError_t sort_pairs_u64_u32(hipStream_t st, void* temp, size_t& temp_bytes, const uint64_t* kin,
                              uint64_t* kout, const uint32_t* vin, uint32_t* vout, uint64_t n) {
  return rocprim::radix_sort_pairs(temp, temp_bytes, kin, kout, vin, vout, (size_t) n, 0, 64, st);
}

hipError_t sort_keys_u64(hipStream_t st, void* temp, size_t& temp_bytes, const uint64_t* kin,
                         uint64_t* kout, uint64_t n) {
  return rocprim::radix_sort_keys(temp, temp_bytes, kin, kout, (size_t) n, 0, 64, st);
}

hipError_t launch_swap_halves(hipStream_t st, uint64_t* v, uint64_t n) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_swap_halves, dim3(grid_for(n, 256, 8192)), dim3(256), 0, st, v, n);
  return hipGetLastError();
}

hipError_t launch_gather_sorted(hipStream_t st, const Seg* seg, const uint32_t* order, uint64_t ne,
                                uint64_t n0p, Seg* sseg, uint32_t* seid, QBox* box0) {
  hipLaunchKernelGGL(k_gather_sorted, dim3(grid_for(n0p, 256, 8192)), dim3(256), 0, st, seg, order,
                     ne, n0p, sseg, seid, box0);
  return hipGetLastError();
}

hipError_t launch_reduce_level(hipStream_t st, const QBox* child, uint64_t n_child_alloc, QBox* parent,
                               uint64_t n_parent_alloc) {
  hipLaunchKernelGGL(k_reduce_level, dim3(grid_for(n_parent_alloc, 4, 8192)), dim3(256), 0, st, child,
                     n_child_alloc, parent, n_parent_alloc);
  return hipGetLastError();
}

hipError_t launch_lsi(hipStream_t st, const LsiArgs& a, bool stats, int max_blocks) {
  uint64_t ngroups = (a.qend - a.qbeg + 63) / 64;
  int grid = grid_for(ngroups, 4, max_blocks);
  if (stats)
    hipLaunchKernelGGL(k_lsi<true>, dim3(grid), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(k_lsi<false>, dim3(grid), dim3(256), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_lsi_points(hipStream_t st, const Seg* seg0, const Seg* seg1, const uint32_t* pairs,
                             uint64_t n, XsectRec* out) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_lsi_points, dim3(grid_for(n, 256, 4096)), dim3(256), 0, st, seg0, seg1, pairs, n, out);
  return hipGetLastError();
}

hipError_t launch_pip(hipStream_t st, const PipArgs& a, bool stats, int max_blocks) {
  uint64_t ngroups = (a.n + 63) / 64;
  int grid = grid_for(ngroups, 4, max_blocks);
  if (stats)
    hipLaunchKernelGGL(k_pip<true>, dim3(grid), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(k_pip<false>, dim3(grid), dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace rj
