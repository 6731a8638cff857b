// output_chain.h -- assembling and writing the overlay's output chains on the host, with the
// behaviour of the reference's WriteOutputChain (src/app/output_chain.h:42-205): walk every chain
// of both maps, cut it at each intersection, label the pieces with the face of the other map
// (vertex faces from the PIP pass, piece-between-two-intersections faces from the mid-point PIP),
// drop pieces that touch no face pair, number the (face, face) pairs in order of first use, number
// distinct points in order of first use, write CDB with 6 fixed decimals.
#pragma once
#include <cstdio>
#include <fstream>
#include <map>
#include <unordered_map>
#include <vector>

#include "context.h"

namespace rayjoin {

struct OutputChain {
  std::vector<Point2d> points;
  uint32_t first_point_idx = 0, last_point_idx = 0;
  int64_t left_polygon_id = 0, right_polygon_id = 0, other_map_polygon_id = 0;
};

struct PointKey {
  double x, y;
  bool operator==(const PointKey& o) const { return x == o.x && y == o.y; }
};
struct PointKeyHash {
  size_t operator()(const PointKey& p) const {
    uint64_t a, b;
    static_assert(sizeof(double) == 8, "");
    __builtin_memcpy(&a, &p.x, 8);
    __builtin_memcpy(&b, &p.y, 8);
    return std::hash<uint64_t>()(a * 0x9E3779B97F4A7C15ull ^ b);
  }
};

// xsects[im]: records ordered by (eid[im], position along the edge) with mid_point_polygon_id set
// (rj_overlay_edge_xsects); point_in_polygon[im][p]: face, in the other map, of vertex p of map im
inline void WriteOutputChain(Context& ctx, const std::vector<rj_xsect> xsects[2],
                             const std::vector<int32_t> point_in_polygon[2], const char* path,
                             size_t* n_chains_out = nullptr, size_t* n_faces_out = nullptr) {
  const Scaling& scaling = ctx.get_scaling();
  std::vector<OutputChain> out;
  auto flush = [&out](OutputChain& oc) {
    auto& pts = oc.points;
    if (pts.empty()) return;
    if (oc.left_polygon_id * oc.other_map_polygon_id != 0 || oc.right_polygon_id * oc.other_map_polygon_id != 0) {
      OutputChain keep = oc;
      keep.points.clear();
      for (auto& p : pts)  // consecutive duplicates collapse
        if (keep.points.empty() || !(keep.points.back().x == p.x && keep.points.back().y == p.y)) keep.points.push_back(p);
      out.push_back(std::move(keep));
    }
    pts.clear();
  };
  auto xsect_point = [&scaling](const rj_xsect& x) { return Point2d{scaling.UnscaleX(x.x_num), scaling.UnscaleY(x.y_num)}; };

  for (int im = 0; im < 2; im++) {
    const PlanarGraph& g = *ctx.get_planar_graph(im);
    const auto& xs = xsects[im];
    // records of one edge are contiguous: eid -> [begin, end)
    std::unordered_map<uint32_t, std::pair<size_t, size_t>> runs;
    for (size_t i = 0; i < xs.size();) {
      size_t j = i;
      while (j < xs.size() && xs[j].eid[im] == xs[i].eid[im]) j++;
      runs[xs[i].eid[im]] = {i, j};
      i = j;
    }
    for (size_t ic = 0; ic < g.chains.size(); ic++) {
      const uint32_t begin_pid = g.row_index[ic], end_pid = g.row_index[ic + 1];
      OutputChain oc;
      oc.left_polygon_id = g.chains[ic].left_polygon_id;
      oc.right_polygon_id = g.chains[ic].right_polygon_id;
      for (uint32_t pid = begin_pid; pid < end_pid; pid++) {
        oc.other_map_polygon_id = point_in_polygon[im][pid];
        oc.points.push_back(g.points[pid]);
        if (pid + 1 == end_pid) continue;
        auto it = runs.find((uint32_t) (pid - ic));
        if (it == runs.end()) continue;
        const size_t b = it->second.first, e = it->second.second;
        oc.points.push_back(xsect_point(xs[b]));
        for (size_t k = b; k + 1 < e; k++) {
          flush(oc);
          oc.other_map_polygon_id = xs[k].mid_point_polygon_id;
          oc.points.push_back(xsect_point(xs[k]));
          oc.points.push_back(xsect_point(xs[k + 1]));
        }
        flush(oc);
        oc.points.push_back(xsect_point(xs[e - 1]));
      }
      flush(oc);
    }
  }

  std::map<std::pair<int64_t, int64_t>, size_t> face_ids;
  auto create_polygon = [&face_ids](int64_t a, int64_t b) -> size_t {
    if (a == 0 || b == 0) return 0;
    auto k = std::make_pair(a, b);
    auto it = face_ids.find(k);
    if (it != face_ids.end()) return it->second;
    size_t id = face_ids.size() + 1;
    face_ids[k] = id;
    return id;
  };
  std::unordered_map<PointKey, uint32_t, PointKeyHash> point_ids;
  for (auto& ch : out) {
    const int64_t o = ch.other_map_polygon_id;
    ch.left_polygon_id = ch.left_polygon_id < o ? create_polygon(ch.left_polygon_id, o) : create_polygon(o, ch.left_polygon_id);
    ch.right_polygon_id = ch.right_polygon_id < o ? create_polygon(ch.right_polygon_id, o) : create_polygon(o, ch.right_polygon_id);
    for (auto& p : ch.points) point_ids.emplace(PointKey{p.x, p.y}, (uint32_t) point_ids.size());
    ch.first_point_idx = point_ids[PointKey{ch.points.front().x, ch.points.front().y}];
    ch.last_point_idx = point_ids[PointKey{ch.points.back().x, ch.points.back().y}];
  }
  std::cerr << "Total chains: " << out.size() << " Total faces: " << face_ids.size() << std::endl;
  FILE* fp = fopen(path, "w");
  if (!fp) throw std::runtime_error(std::string("Cannot open ") + path);
  for (size_t i = 0; i < out.size(); i++) {
    const auto& ch = out[i];
    fprintf(fp, "%zu %zu %u %u %ld %ld\n", i + 1, ch.points.size(), ch.first_point_idx, ch.last_point_idx,
            (long) ch.left_polygon_id, (long) ch.right_polygon_id);
    for (auto& p : ch.points) fprintf(fp, "%.6f %.6f\n", p.x, p.y);
  }
  fclose(fp);
  if (n_chains_out) *n_chains_out = out.size();
  if (n_faces_out) *n_faces_out = face_ids.size();
}

}  // namespace rayjoin
