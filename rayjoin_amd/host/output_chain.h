// output_chain.h -- the overlay's output map: every chain of both input maps cut at its intersections into PIECES,
// each piece labelled with the face of the other map it lies in, written as a CDB map whose faces are the (face of
// map 0, face of map 1) pairs.  The file is the contract: it must be byte-identical to what the reference's
// WriteOutputChain writes (src/app/output_chain.h:42-205; its test/test_overlay.sh diffs this file across modes) --
// piece order, which pieces are dropped, the order in which face pairs and points get their numbers, "%.6f".
//
// Organised around the pieces, in three passes over plain arrays:
//   1. cut     one merge of a map's chains with its intersection records (which arrive ordered by edge and by position
//              along the edge, i.e. already chain by chain: eid = point index - chain index) -> piece descriptors;
//   2. label   a piece that holds a vertex lies in that vertex's face (the vertices' PIP pass), a piece between two
//              cuts of ONE edge in the face of its mid-point (the mid-point PIP pass); pieces outside the other map or
//              along a chain with no face on either side are dropped;
//   3. number  face pairs and distinct points in order of first use, then print.
#pragma once
#include <cstdio>
#include <cstring>
#include <map>
#include <unordered_map>
#include <vector>

#include "context.h"

namespace rayjoin {

// A maximal stretch of one chain between two cuts.  `head` / `tail`: index (into the map's intersection records) of
// the cut it starts / ends at, -1 where it starts / ends with the chain itself; [v_begin, v_end): the chain's own
// vertices inside it (none when both cuts lie on one edge).
struct ChainPiece {
  uint32_t chain;
  int64_t head, tail;
  uint32_t v_begin, v_end;
  int32_t other_face;  // face of the OTHER map the piece lies in (0 = outside)
};

// pass 1 + 2 for one map.  xs: this map's records ordered by (eid[im], position along the edge) with
// mid_point_polygon_id set (rj_overlay_edge_xsects); vertex_face[p]: face, in the other map, of vertex p.
inline std::vector<ChainPiece> CutChainsIntoPieces(const PlanarGraph& g, int im, const std::vector<rj_xsect>& xs,
                                                   const std::vector<int32_t>& vertex_face) {
  std::vector<ChainPiece> pieces;
  pieces.reserve(g.chains.size() + xs.size());
  size_t x = 0;  // the records are consumed in order: an edge id grows with the chain and inside the chain
  for (uint32_t c = 0; c < g.chains.size(); c++) {
    const uint32_t p_first = g.row_index[c], p_end = g.row_index[c + 1];
    const uint32_t eid_end = p_end - 1 - c;  // one past the chain's last edge
    int64_t head = -1;
    uint32_t v = p_first;
    for (; x < xs.size() && xs[x].eid[im] < eid_end; x++) {
      const uint32_t after = xs[x].eid[im] + c + 1;  // the vertex behind the cut edge's start: the first one NOT in this piece
      if (head >= 0 && xs[head].eid[im] == xs[x].eid[im])
        pieces.push_back({c, head, (int64_t) x, v, v, xs[head].mid_point_polygon_id});
      else
        pieces.push_back({c, head, (int64_t) x, v, after, vertex_face[after - 1]});
      head = (int64_t) x;
      v = after;
    }
    pieces.push_back({c, head, -1, v, p_end, vertex_face[p_end - 1]});
  }
  return pieces;
}

class OutputMapWriter {
 public:
  explicit OutputMapWriter(const Scaling& scaling) : scaling_(scaling) {}

  // the pieces of one input map, in piece order; a piece is kept when it lies inside the other map and its chain
  // borders a face (left * other != 0 || right * other != 0, src/app/output_chain.h:58-60)
  void Add(const PlanarGraph& g, const std::vector<rj_xsect>& xs, const std::vector<ChainPiece>& pieces) {
    std::vector<Point2d> pts;
    for (const ChainPiece& pc : pieces) {
      const int64_t left = g.chains[pc.chain].left_polygon_id, right = g.chains[pc.chain].right_polygon_id;
      if (pc.other_face == 0 || (left == 0 && right == 0)) continue;
      pts.clear();
      if (pc.head >= 0) Append(pts, CutPoint(xs[pc.head]));
      for (uint32_t p = pc.v_begin; p < pc.v_end; p++) Append(pts, g.points[p]);
      if (pc.tail >= 0) Append(pts, CutPoint(xs[pc.tail]));
      Row r;
      r.left = FaceOf(left, pc.other_face);
      r.right = FaceOf(right, pc.other_face);
      r.first_point = PointId(pts.front());
      for (size_t i = 1; i + 1 < pts.size(); i++) PointId(pts[i]);
      r.last_point = PointId(pts.back());
      r.points_begin = points_.size();
      points_.insert(points_.end(), pts.begin(), pts.end());
      r.points_end = points_.size();
      rows_.push_back(r);
    }
  }
  size_t n_chains() const { return rows_.size(); }
  size_t n_faces() const { return face_pairs_.size(); }

  void Write(const char* path) const {
    FILE* fp = fopen(path, "w");
    if (!fp) throw std::runtime_error(std::string("Cannot open ") + path);
    for (size_t i = 0; i < rows_.size(); i++) {
      const Row& r = rows_[i];
      fprintf(fp, "%zu %zu %u %u %zu %zu\n", i + 1, r.points_end - r.points_begin, r.first_point, r.last_point, r.left, r.right);
      for (size_t k = r.points_begin; k < r.points_end; k++) fprintf(fp, "%.6f %.6f\n", points_[k].x, points_[k].y);
    }
    fclose(fp);
  }

 private:
  struct Row {
    size_t left, right, points_begin, points_end;
    uint32_t first_point, last_point;
  };
  struct Bits128 {
    uint64_t x, y;
    bool operator==(const Bits128& o) const { return x == o.x && y == o.y; }
  };
  struct Bits128Hash {
    size_t operator()(const Bits128& b) const { return (size_t) ((b.x * 0x9E3779B97F4A7C15ull) ^ (b.y + (b.x >> 29))); }
  };

  Point2d CutPoint(const rj_xsect& r) const { return Point2d{scaling_.UnscaleX(r.x_num), scaling_.UnscaleY(r.y_num)}; }
  // (a cut that falls on a vertex unscales to that vertex: written once)
  static void Append(std::vector<Point2d>& pts, const Point2d& p) {
    if (pts.empty() || pts.back().x != p.x || pts.back().y != p.y) pts.push_back(p);
  }
  // the output face of (a face of this map, a face of the other): pairs are unordered, numbered from 1 in order of
  // first use; no face on this side of the chain -> 0
  size_t FaceOf(int64_t mine, int64_t other) {
    if (mine == 0 || other == 0) return 0;
    const std::pair<int64_t, int64_t> key = mine < other ? std::make_pair(mine, other) : std::make_pair(other, mine);
    return face_pairs_.emplace(key, face_pairs_.size() + 1).first->second;
  }
  // distinct coordinates, numbered from 0 in order of first use (-0.0 and 0.0 are one coordinate, as operator== on doubles says)
  uint32_t PointId(const Point2d& p) {
    const double x = p.x + 0.0, y = p.y + 0.0;
    Bits128 k;
    std::memcpy(&k.x, &x, 8);
    std::memcpy(&k.y, &y, 8);
    return point_ids_.emplace(k, (uint32_t) point_ids_.size()).first->second;
  }

  const Scaling& scaling_;
  std::vector<Row> rows_;
  std::vector<Point2d> points_;
  std::map<std::pair<int64_t, int64_t>, size_t> face_pairs_;
  std::unordered_map<Bits128, uint32_t, Bits128Hash> point_ids_;
};

// xsects[im] / point_in_polygon[im]: see CutChainsIntoPieces
inline void WriteOutputMap(const Scaling& scaling, const PlanarGraph* const graphs[2], const std::vector<rj_xsect> xsects[2],
                           const std::vector<int32_t> point_in_polygon[2], const char* path,
                           size_t* n_chains_out = nullptr, size_t* n_faces_out = nullptr) {
  OutputMapWriter out(scaling);
  for (int im = 0; im < 2; im++)
    out.Add(*graphs[im], xsects[im], CutChainsIntoPieces(*graphs[im], im, xsects[im], point_in_polygon[im]));
  std::cerr << "Total chains: " << out.n_chains() << " Total faces: " << out.n_faces() << std::endl;
  out.Write(path);
  if (n_chains_out) *n_chains_out = out.n_chains();
  if (n_faces_out) *n_faces_out = out.n_faces();
}

// the reference's entry point (src/app/output_chain.h:42-48)
inline void WriteOutputChain(Context& ctx, const std::vector<rj_xsect> xsects[2],
                             const std::vector<int32_t> point_in_polygon[2], const char* path,
                             size_t* n_chains_out = nullptr, size_t* n_faces_out = nullptr) {
  const PlanarGraph* const graphs[2] = {ctx.get_planar_graph(0).get(), ctx.get_planar_graph(1).get()};
  WriteOutputMap(ctx.get_scaling(), graphs, xsects, point_in_polygon, path, n_chains_out, n_faces_out);
}

}  // namespace rayjoin
