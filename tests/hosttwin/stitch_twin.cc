// Test-only: (1) the HOST stitching pass of rounds 1-3 (a hash join of the chain ends, then a sequential walk), kept
// as the reference the data-parallel form must reproduce array for array; (2) a host twin of that data-parallel form:
// the per-element stage functions of rayjoin_amd/csrc/rj_stitch.h -- the very source the HIP kernels run -- driven by
// plain loops, std::stable_sort and a serial prefix sum.  Never linked into the product; the product path is HIP only.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <thread>
#include <vector>

#include "rj_stitch.h"

namespace {

struct RunSet {
  std::vector<uint32_t> piece_begin, piece_len, run_first;  // run_first[nruns + 1]
};

RunSet stitch_runs(const int64_t* xy, const std::vector<uint32_t>& eb, uint64_t cap_edges) {
  RunSet R;
  const size_t nc = eb.empty() ? 0 : eb.size() - 1;
  constexpr uint32_t kNone = 0xFFFFFFFFu;
  // incidence i = 2 c + end (0: the chain's first point, 1: its last); the point indices of chain c: first = eb[c] + c
  auto first_pt = [&](size_t c) { return (uint64_t) eb[c] + c; };
  auto last_pt = [&](size_t c) { return (uint64_t) eb[c + 1] + c; };
  // The end points and the directions of the chains AT their ends, gathered once in chain order (the passes below
  // touch them at random)
  std::vector<int64_t> ex(2 * nc), ey(2 * nc);
  std::vector<float> dirx(2 * nc), diry(2 * nc);
  std::vector<uint64_t> hsh(2 * nc);
  for (size_t c = 0; c < nc; c++) {
    const uint64_t p0 = first_pt(c), p1 = last_pt(c);
    for (int end = 0; end < 2; end++) {
      const uint32_t i = (uint32_t) (2 * c + end);
      const uint64_t p = end ? p1 : p0, q = end ? p1 - 1 : p0 + 1;  // q: the vertex next to this end, inside the chain
      ex[i] = xy[2 * p]; ey[i] = xy[2 * p + 1];
      const double vx = (double) (xy[2 * q] - ex[i]), vy = (double) (xy[2 * q + 1] - ey[i]);
      const double n = std::sqrt(vx * vx + vy * vy);
      dirx[i] = n > 0 ? (float) (vx / n) : 0.0f;
      diry[i] = n > 0 ? (float) (vy / n) : 0.0f;
      uint64_t v = ((uint64_t) ex[i] * 0x9E3779B97F4A7C15ull) ^ (((uint64_t) ey[i] + 0x7F4A7C15ull) * 0xC2B2AE3D27D4EB4Full);
      hsh[i] = v ^ (v >> 29);
    }
  }
  // 1 + 2, in parallel over hash partitions (a node lives in exactly one): an open-addressing table of the end points
  // -> that node's incidence list, then the incidences of every node paired by straightest continuation
  std::vector<uint32_t> partner(2 * nc, kNone);
  unsigned nthreads = std::thread::hardware_concurrency();
  nthreads = nthreads < 1 ? 1 : (nthreads > 16 ? 16 : nthreads);
  if (nc < 50000) nthreads = 1;
  auto part = [&](unsigned t) {
    size_t cap = 16;
    while (cap < (4 * nc) / nthreads + 16) cap <<= 1;
    std::vector<uint32_t> slot_head(cap, kNone), inc_next(2 * nc, kNone);
    for (uint32_t i = 0; i < 2 * nc; i++) {
      if ((hsh[i] >> 40) % nthreads != t) continue;
      size_t sl = (size_t) hsh[i] & (cap - 1);
      for (;;) {
        const uint32_t head = slot_head[sl];
        if (head == kNone) { slot_head[sl] = i; break; }
        if (ex[head] == ex[i] && ey[head] == ey[i]) { inc_next[i] = head; slot_head[sl] = i; break; }
        sl = (sl + 1) & (cap - 1);
      }
    }
    uint32_t at[16];
    for (size_t sl = 0; sl < cap; sl++) {
      if (slot_head[sl] == kNone) continue;
      int n = 0;
      bool hub = false;
      for (uint32_t i = slot_head[sl]; i != kNone; i = inc_next[i]) {
        if (ex[i] == ex[i ^ 1] && ey[i] == ey[i ^ 1]) continue;  // a closed chain (a polygon): it starts and ends here, nothing to continue
        if (dirx[i] == 0.0f && diry[i] == 0.0f) continue;
        if (n == 16) { hub = true; break; }                      // a hub of more than 16 chains: leave them be
        at[n++] = i;
      }
      if (hub || n < 2) continue;
      bool used[16] = {false};
      for (;;) {
        float best = -0.5f;  // cos of the angle between the two directions AWAY from the node: -1 = straight on
        int bi = -1, bj = -1;
        for (int u = 0; u < n; u++)
          for (int v = u + 1; v < n; v++) {
            if (used[u] || used[v] || (at[u] >> 1) == (at[v] >> 1)) continue;
            const float d = dirx[at[u]] * dirx[at[v]] + diry[at[u]] * diry[at[v]];
            if (d < best) { best = d; bi = u; bj = v; }
          }
        if (bi < 0) break;
        used[bi] = used[bj] = true;
        partner[at[bi]] = at[bj];
        partner[at[bj]] = at[bi];
      }
    }
  };
  if (nthreads == 1) {
    part(0);
  } else {
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nthreads; t++) pool.emplace_back(part, t);
    for (auto& th : pool) th.join();
  }
  // 3. follow the pairs into paths, cut every path into near-equal runs of <= 64 edges
  std::vector<bool> visited(nc, false);
  std::vector<uint32_t> path;  // incidences through which the path ENTERS its chains
  R.run_first.push_back(0);
  auto emit = [&]() {
    uint64_t total = 0;
    for (uint32_t i : path) total += eb[(i >> 1) + 1] - eb[i >> 1];
    if (!total) return;
    const uint64_t k = (total + cap_edges - 1) / cap_edges;
    uint64_t run = 0, done = 0, run_end = total / k;  // run `run` covers path positions [total run / k, total (run + 1) / k)
    for (uint32_t i : path) {
      const size_t c = i >> 1;
      const uint32_t len = eb[c + 1] - eb[c];
      uint32_t used_c = 0;  // edges of this chain already handed out, counted from the end the path entered by
      while (used_c < len) {
        const uint32_t take = (uint32_t) std::min<uint64_t>(len - used_c, run_end - done);
        // entered at its first point: the next `take` eids from the front; at its last point: from the back
        R.piece_begin.push_back((i & 1) ? eb[c + 1] - used_c - take : eb[c] + used_c);
        R.piece_len.push_back(take);
        used_c += take;
        done += take;
        if (done == run_end && done < total) {
          R.run_first.push_back((uint32_t) R.piece_begin.size());
          run++;
          run_end = total * (run + 1) / k;
        }
      }
    }
    R.run_first.push_back((uint32_t) R.piece_begin.size());
  };
  auto walk = [&](uint32_t enter) {
    path.clear();
    for (uint32_t i = enter; i != kNone && !visited[i >> 1]; i = partner[i ^ 1]) {
      visited[i >> 1] = true;
      path.push_back(i);
    }
    emit();
  };
  for (uint32_t i = 0; i < 2 * nc; i++)  // paths start at an end that continues nothing
    if (partner[i] == kNone && !visited[i >> 1] && eb[(i >> 1) + 1] > eb[i >> 1]) walk(i);
  for (uint32_t i = 0; i < 2 * nc; i += 2)  // what is left are closed loops of paired chains
    if (!visited[i >> 1] && eb[(i >> 1) + 1] > eb[i >> 1]) walk(i);
  return R;
}


}  // namespace

using namespace rj::stitch;

extern "C" {

// the reference pass: returns the counts, fills the arrays (sized by the caller: pieces <= 2 nc + ne / cap + 1, runs + 1 entries)
int stitch_ref(const int64_t* xy, const uint32_t* eb, uint64_t nc, uint64_t cap, uint32_t* piece_begin, uint32_t* piece_len,
               uint32_t* run_first, uint64_t* nruns, uint64_t* npieces) {
  std::vector<uint32_t> ebv(eb, eb + (nc ? nc + 1 : 0));
  const RunSet R = stitch_runs(xy, ebv, cap);
  *nruns = R.run_first.size() - 1;
  *npieces = R.piece_begin.size();
  std::copy(R.piece_begin.begin(), R.piece_begin.end(), piece_begin);
  std::copy(R.piece_len.begin(), R.piece_len.end(), piece_len);
  std::copy(R.run_first.begin(), R.run_first.end(), run_first);
  return 0;
}

// the twin of rj_stitch.hip's stitch_runs_device(): same stages, same order; stats[0] = ranking rounds run,
// stats[1] = incidences on closed loops, stats[2] = rounds of the second ranking, stats[3] = closed chains (rings)
int stitch_twin(const int64_t* pts, const uint32_t* eb, uint64_t nc64, uint64_t cap64, uint32_t* piece_begin, uint32_t* piece_len,
                uint32_t* run_first, uint64_t* nruns, uint64_t* npieces, uint64_t* stats) {
  const uint32_t nc = (uint32_t) nc64, cap = (uint32_t) cap64, ni = 2 * nc;
  *nruns = *npieces = 0;
  run_first[0] = 0;
  if (stats) stats[0] = stats[1] = stats[2] = stats[3] = 0;
  if (!nc) return 0;
  std::vector<uint64_t> kx(ni), ky(ni);
  std::vector<Dir> dir(ni);
  uint64_t rings = 0;
  for (uint32_t i = 0; i < ni; i++) rings += end_keys(i, pts, eb, kx.data(), ky.data(), dir.data()) ? 1 : 0;
  if (stats) stats[3] = rings;
  // sort by y, then stably by x (two stable radix sorts on the device)
  std::vector<uint32_t> sv(ni);
  std::iota(sv.begin(), sv.end(), 0u);
  std::stable_sort(sv.begin(), sv.end(), [&](uint32_t a, uint32_t b) { return ky[a] < ky[b]; });
  std::stable_sort(sv.begin(), sv.end(), [&](uint32_t a, uint32_t b) { return kx[a] < kx[b]; });
  std::vector<uint64_t> skx(ni);
  for (uint32_t j = 0; j < ni; j++) skx[j] = kx[sv[j]];
  std::vector<uint32_t> partner(ni, kNone);
  for (uint32_t j = 0; j < ni; j++) pair_node(j, ni, skx.data(), sv.data(), ky.data(), dir.data(), partner.data());
  // list ranking
  int rounds = 2;
  while ((1ull << (rounds - 2)) <= nc && rounds < kMaxRounds) rounds++;
  std::vector<Node> buf[2] = {std::vector<Node>(ni), std::vector<Node>(ni)};
  Meta meta;
  memset(&meta, 0, sizeof(meta));
  auto rank = [&](uint32_t* act, uint32_t* done) {
    for (int r = 0; r < rounds; r++) {
      if (!rank_round_needed(act, r)) { if (!*done) *done = (uint32_t) r; continue; }
      uint32_t n = 0;
      for (uint32_t i = 0; i < ni; i++) n += rank_round(i, buf[r & 1].data(), buf[(r + 1) & 1].data()) ? 1 : 0;
      act[r] = n;
    }
    if (!*done) *done = (uint32_t) rounds;
  };
  for (uint32_t i = 0; i < ni; i++) rank_init(i, partner.data(), eb, buf[0].data(), buf[1].data());
  rank(meta.act, &meta.done_round);
  if (stats) stats[0] = meta.done_round;
  const Node* F = buf[meta.done_round & 1].data();
  const uint32_t on_loops = meta.act[meta.done_round - 1];
  std::vector<uint8_t> in_loop(nc, 0);
  if (on_loops) {
    if (stats) stats[1] = on_loops;
    std::vector<Link> lk[2] = {std::vector<Link>(ni), std::vector<Link>(ni)};
    for (uint32_t i = 0; i < ni; i++) cyc_init(i, F, partner.data(), lk[0].data(), lk[1].data());
    int cr = 1;
    while ((1ull << cr) < nc) cr++;
    for (int r = 0; r < cr; r++)
      for (uint32_t i = 0; i < ni; i++) cyc_round(i, lk[r & 1].data(), lk[(r + 1) & 1].data());
    for (uint32_t i = 0; i < ni; i++) cyc_break(i, lk[cr & 1].data(), partner.data(), in_loop.data());
    const int fb = meta.done_round & 1;
    for (uint32_t i = 0; i < ni; i++) rerank_init(i, buf[fb].data(), partner.data(), eb, buf[0].data(), buf[1].data());
    rank(meta.act2, &meta.done_round2);
    if (stats) stats[2] = meta.done_round2;
    F = buf[meta.done_round2 & 1].data();
    if (meta.act2[meta.done_round2 - 1]) return 1;  // (cannot happen: every loop was opened)
  }
  std::vector<uint32_t> ch_key(nc), ch_off(nc), ch_total(nc), ch_rank(nc);
  std::vector<uint8_t> ch_back(nc);
  std::vector<uint64_t> head(2 * (size_t) ni, 0), base(2 * (size_t) ni);
  for (uint32_t c = 0; c < nc; c++)
    chain_orient(c, F, eb, in_loop.data(), cap, ni, ch_key.data(), ch_off.data(), ch_total.data(), ch_rank.data(), ch_back.data(), head.data());
  uint64_t acc = 0;
  for (size_t k = 0; k < head.size(); k++) { base[k] = acc; acc += head[k]; }
  std::vector<uint32_t> slot_chain(nc), slot_pieces(nc), pbase(nc);
  for (uint32_t c = 0; c < nc; c++)
    chain_place(c, eb, cap, ch_key.data(), ch_off.data(), ch_total.data(), ch_rank.data(), ch_back.data(), base.data(), slot_chain.data(), slot_pieces.data());
  uint32_t pacc = 0;
  for (uint32_t s = 0; s < nc; s++) { pbase[s] = pacc; pacc += slot_pieces[s]; }
  for (uint32_t s = 0; s < nc; s++)
    chain_emit(s, nc, eb, cap, ch_key.data(), ch_off.data(), ch_total.data(), base.data(), slot_chain.data(), slot_pieces.data(), pbase.data(),
               piece_begin, piece_len, run_first, &meta);
  *nruns = meta.nruns;
  *npieces = meta.npieces;
  return 0;
}

}  // extern "C"
