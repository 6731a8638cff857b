// rj_stitch.h -- polyline runs of a map, the data-parallel form (index build).
//
// A leaf of the index is a run of <= `cap` consecutive edges of ONE POLYLINE -- the analogue of the reference's RT
// grouping (src/rt/primitive.h:120-260: one AABB around consecutive edges of a chain, looped over in the IS program,
// src/algo/rt_lsi_custom.cu:31-44); the whole construction replaces the thrust passes of the reference's build
// (deps/lbvh/lbvh/bvh.cuh:277-481).  A CDB chain ends at every junction, so polylines are stitched through the
// junctions first: at every shared end point the incident chains are paired by straightest continuation (smallest
// cosine between their directions there, at most 120 degrees of turn; closed chains and hubs of more than 16 chains
// are left alone), the pairs are followed into paths and every path is cut into ceil(len / cap) near-equal runs.
// A run is a short list of pieces (eid ranges of the chains it crosses).
//
// Until round 4 this was a host pass (a hash join of the chain ends on <= 16 threads behind a read-back of the
// whole point array).  Here every step is one function per element -- incidence i = 2 c + end, chain c, or slot --
// that rj_stitch.hip runs as a grid-stride kernel and tests/hosttwin/stitch_twin.cc runs as a plain loop (a
// test-only twin, never a fallback), with a radix sort and two prefix sums in between:
//
//   end_keys      per incidence: the end point as a sort key, the unit direction INTO the chain
//   (sort by y, then stably by x: the incidences of a junction become neighbours, ascending)
//   pair_node     per junction (its first incidence): the greedy straightest-continuation pairing -> partner[]
//   rank_*        list ranking by pointer jumping over the 2 nc directed walks (every path appears once per
//                 direction): edges and chains from here to the end of the walk, and the free end it exits by
//   cyc_*         closed loops of paired chains (no free end: the ranking never ends for them): the smallest
//                 incidence of each loop by pointer doubling, the loop is opened there and ranked again
//   chain_orient  per chain: the walk that starts at the smaller free end is the path's direction; the chain's
//                 offset and rank in its path; the path's first chain leaves (chains, runs) at the path's key
//   (exclusive scan over the keys: open paths by their first incidence, then loops -- the order of the host pass)
//   chain_place   per chain: its slot in path order, how many pieces it is cut into
//   (exclusive scan over the slots)
//   chain_emit    per slot: the pieces, and run_first[] for every run that starts inside this chain
//
// The result is array-for-array what the host pass produced (tests/test_stitch_twin.py compares both).
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define RJ_SHD __host__ __device__ __forceinline__
#else
#define RJ_SHD inline
#endif

namespace rj {
namespace stitch {

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint64_t kNoKey = ~0ull;          // the sort key of an end that continues nothing (sorts behind every point)
constexpr int kMaxDeg = 16;                 // a junction of more chains than this is a hub: left alone
constexpr int64_t kKeyOffset = (int64_t) 1 << 46;  // scaled coordinates lie in [-2^46, 2^46)
constexpr unsigned kKeyBits = 48;           // (47 significant bits; kNoKey differs from every real key below bit 48 too)
constexpr int kMaxRounds = 34;

struct Dir {
  float x, y;
};
struct alignas(16) Node {  // a directed walk from incidence i: over chain i >> 1, on through partner[i ^ 1], ...
  uint32_t succ;           // the incidence the walk enters next (kNone: the walk ends behind what is summed here)
  uint32_t dist;           // edges of the chains from here to `succ` (exclusive)
  uint32_t cnt;            // ... and how many chains that is
  uint32_t tail;           // the last incidence entered so far
};
struct Link {              // pointer doubling over the closed loops: the smallest incidence seen so far
  uint32_t succ, mn;
};
// what the stages leave for each other and for the host (device memory, zeroed before the first stage)
struct Meta {
  uint32_t act[kMaxRounds];   // ranking: incidences still on their way after round r
  uint32_t done_round;        // the first round that found nothing left to do (0: not set yet) -- the final states are in buffer done_round & 1
  uint32_t act2[kMaxRounds];  // the same for the second ranking (after the loops were opened)
  uint32_t done_round2;
  uint32_t nruns, npieces;    // totals (chain_emit)
  uint32_t closed_chains;     // chains that start and end on the same point (rings): end_keys
};

RJ_SHD uint32_t chain_len(const uint32_t* eb, uint32_t c) { return eb[c + 1] - eb[c]; }

// ---- 1. the ends of the chains ------------------------------------------------------------------
// incidence i = 2 c + end (0: the chain's first point, 1: its last); point indices of chain c: eb[c] + c .. eb[c + 1] + c
// -> true for the first incidence of a closed chain (the caller counts the rings of the map)
RJ_SHD bool end_keys(uint32_t i, const int64_t* pts, const uint32_t* eb, uint64_t* kx, uint64_t* ky, Dir* dir) {
  const uint32_t c = i >> 1;
  const uint64_t p0 = (uint64_t) eb[c] + c, p1 = (uint64_t) eb[c + 1] + c;
  const uint64_t p = (i & 1) ? p1 : p0, q = (i & 1) ? p1 - 1 : p0 + 1;  // q: the vertex next to this end, inside the chain
  const int64_t x = pts[2 * p], y = pts[2 * p + 1];
  const double vx = (double) (pts[2 * q] - x), vy = (double) (pts[2 * q + 1] - y);
  const double n = sqrt(vx * vx + vy * vy);
  Dir d;
  d.x = n > 0 ? (float) (vx / n) : 0.0f;
  d.y = n > 0 ? (float) (vy / n) : 0.0f;
  dir[i] = d;
  // a closed chain (a polygon) starts and ends here, and a zero-length first edge has no direction: nothing to continue
  const bool closed = pts[2 * p0] == pts[2 * p1] && pts[2 * p0 + 1] == pts[2 * p1 + 1];
  const bool valid = !closed && !(d.x == 0.0f && d.y == 0.0f);
  kx[i] = valid ? (uint64_t) (x + kKeyOffset) : kNoKey;
  ky[i] = valid ? (uint64_t) (y + kKeyOffset) : kNoKey;
  return closed && !(i & 1);
}

// ---- 2. junctions --------------------------------------------------------------------------------
// Sorted position j (skx = the sorted x keys, sv = the incidence at each position, ky by incidence).  The first
// position of every junction pairs its incidences; partner[] is kNone everywhere before.
RJ_SHD void pair_node(uint64_t j, uint64_t n, const uint64_t* skx, const uint32_t* sv, const uint64_t* ky, const Dir* dir,
                      uint32_t* partner) {
  const uint64_t x = skx[j];
  if (x == kNoKey) return;
  const uint64_t y = ky[sv[j]];
  if (j > 0 && skx[j - 1] == x && ky[sv[j - 1]] == y) return;  // not the first of its junction
  uint32_t asc[kMaxDeg];
  int m = 0;
  for (uint64_t k = j; k < n && skx[k] == x && ky[sv[k]] == y; k++) {
    if (m == kMaxDeg) return;  // a hub
    asc[m++] = sv[k];
  }
  if (m < 2) return;
  // (the host pass met the incidences of a junction in descending order; ties between equally straight pairs go to
  //  the first pair in that order)
  uint32_t at[kMaxDeg];
  for (int u = 0; u < m; u++) at[u] = asc[m - 1 - u];
  bool used[kMaxDeg];
  for (int u = 0; u < m; u++) used[u] = false;
  for (;;) {
    float best = -0.5f;  // cos of the angle between the two directions AWAY from the junction: -1 = straight on
    int bi = -1, bj = -1;
    for (int u = 0; u < m; u++)
      for (int v = u + 1; v < m; v++) {
        if (used[u] || used[v] || (at[u] >> 1) == (at[v] >> 1)) continue;
        const float d = dir[at[u]].x * dir[at[v]].x + dir[at[u]].y * dir[at[v]].y;
        if (d < best) { best = d; bi = u; bj = v; }
      }
    if (bi < 0) break;
    used[bi] = used[bj] = true;
    partner[at[bi]] = at[bj];
    partner[at[bj]] = at[bi];
  }
}

// ---- 3. list ranking -----------------------------------------------------------------------------
RJ_SHD Node rank_start(uint32_t i, const uint32_t* partner, const uint32_t* eb) {
  Node a;
  a.succ = partner[i ^ 1];
  a.dist = chain_len(eb, i >> 1);
  a.cnt = 1;
  a.tail = i;
  return a;
}
RJ_SHD void rank_init(uint32_t i, const uint32_t* partner, const uint32_t* eb, Node* a, Node* b) { a[i] = b[i] = rank_start(i, partner, eb); }
// one round of pointer jumping, in -> out; true while the walk from i has not reached its end.  A walk that ended in
// the round before still has its old state in `out` (written two rounds ago): copied once, then both hold it.
RJ_SHD bool rank_round(uint32_t i, const Node* in, Node* out) {
  Node a = in[i];
  if (a.succ == kNone) {
    if (out[i].succ != kNone) out[i] = a;
    return false;
  }
  const Node b = in[a.succ];
  a.dist += b.dist;
  a.cnt += b.cnt;
  a.tail = b.tail;
  a.succ = b.succ;
  out[i] = a;
  return a.succ != kNone;
}
// does round r still have work?  (nothing left, or nothing ended in the round before: only closed loops are left)
RJ_SHD bool rank_round_needed(const uint32_t* act, int r) {
  if (r == 0) return true;
  if (act[r - 1] == 0) return false;
  return !(r >= 2 && act[r - 1] == act[r - 2]);
}

// ---- 3b. closed loops ----------------------------------------------------------------------------
// F = the ranking's final states: an incidence whose walk never ended lies on a loop
RJ_SHD void cyc_init(uint32_t i, const Node* F, const uint32_t* partner, Link* a, Link* b) {
  Link l;
  l.succ = F[i].succ != kNone ? partner[i ^ 1] : kNone;
  l.mn = i;
  a[i] = b[i] = l;
}
RJ_SHD void cyc_round(uint32_t i, const Link* in, Link* out) {
  const Link a = in[i];
  if (a.succ == kNone) return;
  const Link b = in[a.succ];
  Link o;
  o.succ = b.succ;
  o.mn = a.mn < b.mn ? a.mn : b.mn;
  out[i] = o;
}
// The two directed walks around a loop hold the two incidences of its smallest chain c as their minima; the host pass
// entered that chain at its first point (incidence 2 c) and went round from there: the loop is opened at that
// incidence's junction, which makes it an open path whose smaller free end is 2 c.
RJ_SHD void cyc_break(uint32_t i, const Link* S, uint32_t* partner, uint8_t* in_loop) {
  if (S[i].succ == kNone) return;
  in_loop[i >> 1] = 1;
  if (S[i].mn == i && !(i & 1)) {
    const uint32_t p = partner[i];
    partner[i] = kNone;
    partner[p] = kNone;
  }
}
// second ranking: the loops start over from the opened links, everything else keeps its final state (in both buffers)
RJ_SHD void rerank_init(uint32_t i, const Node* F, const uint32_t* partner, const uint32_t* eb, Node* a, Node* b) {
  const Node f = F[i];
  a[i] = b[i] = f.succ != kNone ? rank_start(i, partner, eb) : f;
}

// ---- 4. the chains in their paths ----------------------------------------------------------------
RJ_SHD uint32_t runs_of(uint32_t total, uint32_t cap) { return (total + cap - 1) / cap; }
// run r of a path of `total` edges cut into k runs covers path positions [total r / k, total (r + 1) / k)
RJ_SHD uint64_t run_begin(uint32_t total, uint32_t k, uint64_t r) { return (uint64_t) total * r / k; }
RJ_SHD uint32_t run_of(uint64_t pos, uint32_t total, uint32_t k) { return (uint32_t) (((pos + 1) * k - 1) / total); }

// Entering chain c at incidence i leads to the free end F[i].tail ^ 1.  The path starts at the smaller of its two free
// ends; the chain is walked away from it.  key = where the path's totals go: its first incidence (open paths), behind
// all of those the loops by theirs (ni = 2 nc).
RJ_SHD void chain_orient(uint32_t c, const Node* F, const uint32_t* eb, const uint8_t* in_loop, uint32_t cap, uint32_t ni,
                         uint32_t* ch_key, uint32_t* ch_off, uint32_t* ch_total, uint32_t* ch_rank, uint8_t* ch_back,
                         uint64_t* head) {
  const Node f0 = F[2 * c], f1 = F[2 * c + 1];
  const uint32_t e0 = f0.tail ^ 1, e1 = f1.tail ^ 1;
  const uint32_t s = e0 < e1 ? e0 : e1;
  const bool fwd = e1 == s;  // entering at the last point leads back to the start: the path enters at the first point
  const Node& on = fwd ? f0 : f1;    // the walk in the path's direction, from this chain on
  const Node& back = fwd ? f1 : f0;  // the walk against it: this chain and everything before it
  const uint32_t len = chain_len(eb, c);
  const uint32_t total = on.dist + back.dist - len;
  const uint32_t key = in_loop[c] ? ni + s : s;
  ch_key[c] = key;
  ch_off[c] = back.dist - len;
  ch_total[c] = total;
  ch_rank[c] = back.cnt - 1;
  ch_back[c] = fwd ? 0 : 1;
  if (back.cnt == 1) head[key] = (uint64_t) (on.cnt + back.cnt - 1) | ((uint64_t) runs_of(total, cap) << 32);
}

// base = the exclusive scan of head[]: low word = chains of the paths before this one, high word = their runs
RJ_SHD void chain_place(uint32_t c, const uint32_t* eb, uint32_t cap, const uint32_t* ch_key, const uint32_t* ch_off,
                        const uint32_t* ch_total, const uint32_t* ch_rank, const uint8_t* ch_back, const uint64_t* base,
                        uint32_t* slot_chain, uint32_t* slot_pieces) {
  const uint32_t slot = (uint32_t) base[ch_key[c]] + ch_rank[c];
  const uint32_t len = chain_len(eb, c), total = ch_total[c], k = runs_of(total, cap);
  const uint64_t o = ch_off[c];
  slot_chain[slot] = (c << 1) | ch_back[c];
  slot_pieces[slot] = run_of(o + len - 1, total, k) - run_of(o, total, k) + 1;
}

// pbase = the exclusive scan of slot_pieces[].  The chain's pieces in run order; a run that starts inside this chain (or
// exactly at its first edge) starts at the piece written for it here.
RJ_SHD void chain_emit(uint32_t slot, uint32_t nc, const uint32_t* eb, uint32_t cap, const uint32_t* ch_key, const uint32_t* ch_off,
                       const uint32_t* ch_total, const uint64_t* base, const uint32_t* slot_chain, const uint32_t* slot_pieces,
                       const uint32_t* pbase, uint32_t* piece_begin, uint32_t* piece_len, uint32_t* run_first, Meta* meta) {
  const uint32_t sc = slot_chain[slot], c = sc >> 1;
  const bool backwards = sc & 1;
  const uint32_t len = chain_len(eb, c), total = ch_total[c], k = runs_of(total, cap);
  const uint64_t o = ch_off[c];
  const uint32_t run_base = (uint32_t) (base[ch_key[c]] >> 32);
  const uint32_t r_lo = run_of(o, total, k), r_hi = run_of(o + len - 1, total, k);
  uint32_t p = pbase[slot];
  for (uint32_t r = r_lo; r <= r_hi; r++, p++) {
    const uint64_t b = run_begin(total, k, r), e = run_begin(total, k, (uint64_t) r + 1);
    const uint64_t lo = b > o ? b : o, hi = e < o + len ? e : o + len;
    const uint32_t used = (uint32_t) (lo - o), take = (uint32_t) (hi - lo);
    // entered at its first point: the next `take` eids from the front; at its last point: from the back
    piece_begin[p] = backwards ? eb[c + 1] - used - take : eb[c] + used;
    piece_len[p] = take;
    if (b >= o) run_first[run_base + r] = p;
  }
  if (slot == nc - 1) {  // the last chain of the last path: the totals
    const uint32_t nruns = run_base + k, npieces = pbase[slot] + slot_pieces[slot];
    run_first[nruns] = npieces;
    meta->nruns = nruns;
    meta->npieces = npieces;
  }
}

}  // namespace stitch
}  // namespace rj
