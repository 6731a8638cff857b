// planar_graph.h -- CDB loader surface of the reference, host C++:
//   Chain / PlanarGraph            src/map/planar_graph.h:24-40
//   read_pgraph (text CDB)         :42-126   (blank/#/% lines skipped; fatal on np<2, repeated
//                                             point, unparsable line, trailing incomplete chain)
//   serialize/deserialize (.bin)   :129-220  (byte-compatible)
//   load_from                      :223-252  (<prefix>/<path with '/'->'-'>.bin cache)
// Errors throw std::runtime_error (the reference CHECK-aborts).
#pragma once
#include <dirent.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <fstream>
#include <limits>
#include <memory>
#include <numeric>
#include <map>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace rayjoin {

struct Chain {
  int64_t id, first_point_idx, last_point_idx, left_polygon_id, right_polygon_id;
};
struct Point2d {
  double x, y;
};
struct BoundingBox {
  double min_x = std::numeric_limits<double>::max(), min_y = std::numeric_limits<double>::max();
  double max_x = -std::numeric_limits<double>::max(), max_y = -std::numeric_limits<double>::max();
};
struct PlanarGraph {
  std::vector<Chain> chains;
  std::vector<uint32_t> row_index;
  std::vector<Point2d> points;
  BoundingBox bb;
  size_t n_edges() const { return points.size() - chains.size(); }
};

inline std::shared_ptr<PlanarGraph> read_pgraph(const char* path, int verbose = 0) {
  std::ifstream ifs(path);
  if (!ifs.is_open()) throw std::runtime_error(std::string("Cannot open file ") + path);
  auto pg = std::make_shared<PlanarGraph>();
  auto& g = *pg;
  std::string line;
  int64_t np = 0;
  bool have_last = false;
  Point2d last{0, 0};
  size_t lno = 0;
  double min_len = std::numeric_limits<double>::max(), max_len = 0, sum_len = 0;
  while (std::getline(ifs, line)) {
    lno++;
    if (line.empty() || line[0] == '#' || line[0] == '%') continue;
    std::istringstream iss(line);
    bool bad;
    if (np == 0) {
      Chain c{};
      bad = !(iss >> c.id >> np >> c.first_point_idx >> c.last_point_idx >> c.left_polygon_id >> c.right_polygon_id);
      bad |= np < 2;
      if (!bad) {
        g.chains.push_back(c);
        g.row_index.push_back((uint32_t) g.points.size());
      }
      have_last = false;
    } else {
      Point2d p{};
      bad = !(iss >> p.x >> p.y);
      if (have_last) {
        double len = std::sqrt((p.x - last.x) * (p.x - last.x) + (p.y - last.y) * (p.y - last.y));
        min_len = std::min(min_len, len); max_len = std::max(max_len, len); sum_len += len;
        bad |= p.x == last.x && p.y == last.y;
      }
      g.bb.min_x = std::min(g.bb.min_x, p.x); g.bb.max_x = std::max(g.bb.max_x, p.x);
      g.bb.min_y = std::min(g.bb.min_y, p.y); g.bb.max_y = std::max(g.bb.max_y, p.y);
      g.points.push_back(p);
      last = p; have_last = true;
      np--;
    }
    if (bad) {
      std::ostringstream m;
      m << "Bad line. Check your dataset! " << path << "[" << lno << "]: " << line;
      throw std::runtime_error(m.str());
    }
  }
  if (!g.points.empty()) g.row_index.push_back((uint32_t) g.points.size());
  if (np != 0) throw std::runtime_error(std::string(path) + ": trailing incomplete chain");
  if (verbose)
    std::cerr << "Map " << path << " is loaded, chains: " << g.chains.size() << " points: " << g.points.size()
              << " edges: " << g.n_edges() << ", min seg len: " << min_len << ", max seg len: " << max_len
              << ", avg seg len: " << (g.n_edges() ? sum_len / g.n_edges() : 0.0) << std::endl;
  return pg;
}

inline void serialize_pgraph(const PlanarGraph& g, const char* path) {
  std::ofstream ofs(path, std::ios::out | std::ios::binary);
  if (!ofs.good()) throw std::runtime_error(std::string("Cannot write ") + path);
  auto W = [&](const void* p, size_t n) { ofs.write(reinterpret_cast<const char*>(p), n); };
  uint64_t magic = 0xabcdabcd, nc = g.chains.size(), nri = g.row_index.size(), npt = g.points.size();
  W(&magic, 8); W(&nc, 8); W(&nri, 8); W(&npt, 8);
  W(g.chains.data(), nc * sizeof(Chain));
  W(g.row_index.data(), nri * 4);
  W(g.points.data(), npt * 16);
  W(&g.bb.min_x, 8); W(&g.bb.min_y, 8); W(&g.bb.max_x, 8); W(&g.bb.max_y, 8);
  W(&magic, 8);
}

inline std::shared_ptr<PlanarGraph> deserialize_pgraph(const char* path) {
  std::ifstream ifs(path, std::ios::in | std::ios::binary);
  if (!ifs.good()) throw std::runtime_error(std::string("Cannot open ") + path);
  auto pg = std::make_shared<PlanarGraph>();
  auto R = [&](void* p, size_t n) { ifs.read(reinterpret_cast<char*>(p), n); };
  uint64_t magic = 0, nc = 0, nri = 0, npt = 0;
  R(&magic, 8);
  if (magic != 0xabcdabcd) throw std::runtime_error(std::string(path) + ": bad checksum");
  R(&nc, 8); R(&nri, 8); R(&npt, 8);
  pg->chains.resize(nc); pg->row_index.resize(nri); pg->points.resize(npt);
  R(pg->chains.data(), nc * sizeof(Chain));
  R(pg->row_index.data(), nri * 4);
  R(pg->points.data(), npt * 16);
  R(&pg->bb.min_x, 8); R(&pg->bb.min_y, 8); R(&pg->bb.max_x, 8); R(&pg->bb.max_y, 8);
  R(&magic, 8);
  if (!ifs.good() || magic != 0xabcdabcd) throw std::runtime_error(std::string(path) + ": bad trailing checksum");
  return pg;
}

inline std::shared_ptr<PlanarGraph> load_from(const std::string& path, const std::string& serialize_prefix, int verbose = 0) {
  std::string escaped = path;
  std::replace(escaped.begin(), escaped.end(), '/', '-');
  if (!serialize_prefix.empty()) {
    DIR* dir = opendir(serialize_prefix.c_str());
    if (dir) closedir(dir);
    else if (mkdir(serialize_prefix.c_str(), 0755)) throw std::runtime_error("Cannot create dir " + serialize_prefix);
  }
  std::string ser = serialize_prefix + '/' + escaped + ".bin";
  if (access(ser.c_str(), R_OK) == 0) return deserialize_pgraph(ser.c_str());
  auto pg = read_pgraph(path.c_str(), verbose);
  if (!serialize_prefix.empty() && access(serialize_prefix.c_str(), W_OK) == 0) serialize_pgraph(*pg, ser.c_str());
  return pg;
}

// Samplers of the paper's scalability runs (planar_graph.h:255-399; flags -sample, -sample_map_id,
// -sample_rate, -seed).  Both use std::mt19937 + std::shuffle like the reference, so with the same
// libstdc++ and a non-zero seed the sampled map is the same one.
namespace detail {
inline void append_points(PlanarGraph& out, const PlanarGraph& in, const std::vector<size_t>& pids) {
  out.row_index.push_back((uint32_t) out.points.size());
  for (size_t pid : pids) {
    const Point2d& p = in.points[pid];
    out.points.push_back(p);
    out.bb.min_x = std::min(out.bb.min_x, p.x);
    out.bb.max_x = std::max(out.bb.max_x, p.x);
    out.bb.min_y = std::min(out.bb.min_y, p.y);
    out.bb.max_y = std::max(out.bb.max_y, p.y);
  }
}
}  // namespace detail

// "-sample map": every chain survives with its two end points; of its interior points a random
// sample_rate fraction (at least one) is kept, in the original order -- a down-scaled map with the
// original topology.
inline std::shared_ptr<PlanarGraph> sample_map_from(const PlanarGraph& g, float sample_rate, int seed = 0) {
  std::random_device rd;
  std::mt19937 gen(seed == 0 ? rd() : seed);
  auto out = std::make_shared<PlanarGraph>();
  out->chains = g.chains;
  std::vector<size_t> pids;
  for (size_t ic = 0; ic < g.chains.size(); ic++) {
    const size_t begin = g.row_index[ic], end = g.row_index[ic + 1];
    pids.assign(1, begin);
    if (end - begin > 2) {
      for (size_t pid = begin + 1; pid + 1 < end; pid++) pids.push_back(pid);
      std::shuffle(pids.begin() + 1, pids.end(), gen);
      pids.resize(std::max((size_t) 2, (size_t) (pids.size() * sample_rate)));
      std::sort(pids.begin() + 1, pids.end());
    }
    pids.push_back(end - 1);
    detail::append_points(*out, g, pids);
  }
  if (!out->points.empty()) out->row_index.push_back((uint32_t) out->points.size());
  return out;
}

// "-sample edges": a random sample_rate fraction of all edges; the surviving edges of a chain are
// re-packed into one chain (consecutive survivors stay connected, gaps are bridged by the chain's
// next surviving point), chains without survivors disappear and chain ids are renumbered from 0.
inline std::shared_ptr<PlanarGraph> sample_edges_from(const PlanarGraph& g, float sample_rate, int seed = 0) {
  std::vector<std::pair<size_t, size_t>> edges;  // (chain, first point of the edge)
  edges.reserve(g.points.size());
  for (size_t ic = 0; ic < g.chains.size(); ic++)
    for (size_t pid = g.row_index[ic]; pid + 1 < g.row_index[ic + 1]; pid++) edges.emplace_back(ic, pid);
  std::random_device rd;
  std::mt19937 gen(seed == 0 ? rd() : seed);
  std::shuffle(edges.begin(), edges.end(), gen);
  edges.resize((size_t) (edges.size() * sample_rate));
  std::map<size_t, std::vector<size_t>> by_chain;
  for (const auto& e : edges) by_chain[e.first].push_back(e.second);
  auto out = std::make_shared<PlanarGraph>();
  int64_t chain_id = 0;
  std::vector<size_t> pids;
  for (const auto& kv : by_chain) {
    Chain c = g.chains[kv.first];
    c.id = chain_id++;
    out->chains.push_back(c);
    pids.clear();
    for (size_t p1 : kv.second) {
      pids.push_back(p1);
      pids.push_back(p1 + 1);
    }
    std::sort(pids.begin(), pids.end());
    pids.erase(std::unique(pids.begin(), pids.end()), pids.end());
    detail::append_points(*out, g, pids);
  }
  if (!out->points.empty()) out->row_index.push_back((uint32_t) out->points.size());
  return out;
}

}  // namespace rayjoin
