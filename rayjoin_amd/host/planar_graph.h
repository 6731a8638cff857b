// planar_graph.h -- CDB loader surface of the reference, host C++:
//   Chain / PlanarGraph            src/map/planar_graph.h:24-40
//   read_pgraph (text CDB)         :42-126   (blank/#/% lines skipped; fatal on np<2, repeated
//                                             point, unparsable line, trailing incomplete chain)
//                                            serial form + a chunked parallel form with the same
//                                            results and the same errors (read_pgraph_parallel)
//   serialize/deserialize (.bin)   :129-220  (byte-compatible)
//   load_from                      :223-252  (<prefix>/<path with '/'->'-'>.bin cache)
// Errors throw std::runtime_error (the reference CHECK-aborts).
#pragma once
#include <dirent.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <memory>
#include <numeric>
#include <map>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
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

inline std::shared_ptr<PlanarGraph> read_pgraph_serial(const char* path, int verbose = 0) {
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

// ---- parallel text parse -----------------------------------------------------------------------
// The reference reads a 50 M-point CDB file in 25 s with the loop above ("Read map",
// expr/draw/scal_lsi_synthetic/gaussian_1000000.log:73).  Same grammar, same results, same errors,
// on every host core:
//   A  (parallel) the file is mmap'ed and cut into byte ranges; each thread lists the start of every
//      line getline would hand to the parser (non-empty, not starting with '#' or '%');
//   B  (serial, headers only) which line is a chain header depends on the np of the header before
//      it, so the headers are walked in order -- one short parse per CHAIN, not per point;
//   C  (parallel) chains are dealt to threads by point count and their point lines parsed straight
//      into their final slots (row_index is known from B).
// A token is parsed by std::from_chars when it is a plain decimal number and the line has exactly the
// expected shape; any other line goes through the very istringstream extraction of the serial loop,
// so odd inputs ("+1.5", "2.5abc", hex floats, overflow) behave identically.  The first bad line in
// FILE order is the one reported, as the serial loop would.
namespace detail {

inline bool cdb_space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r'; }

// [-]digits  -> true and value; anything else (sign '+', overflow, other characters) -> false
inline bool fast_i64(const char*& p, const char* e, int64_t* v) {
  while (p < e && cdb_space(*p)) p++;
  const char* t = p;
  while (t < e && !cdb_space(*t)) t++;
  if (p == t) return false;
  const char* q = p;
  if (*q == '-') q++;
  if (q == t) return false;
  for (const char* c = q; c < t; c++)
    if (*c < '0' || *c > '9') return false;
  auto r = std::from_chars(p, t, *v);
  if (r.ec != std::errc() || r.ptr != t) return false;
  p = t;
  return true;
}

// [-](digits[.digits] | .digits)[(e|E)[+-]digits]
inline bool fast_f64(const char*& p, const char* e, double* v) {
  while (p < e && cdb_space(*p)) p++;
  const char* t = p;
  while (t < e && !cdb_space(*t)) t++;
  if (p == t) return false;
  const char* q = p;
  if (*q == '-') q++;
  int digits = 0;
  while (q < t && *q >= '0' && *q <= '9') { q++; digits++; }
  if (q < t && *q == '.') {
    q++;
    while (q < t && *q >= '0' && *q <= '9') { q++; digits++; }
  }
  if (!digits) return false;
  if (q < t && (*q == 'e' || *q == 'E')) {
    q++;
    if (q < t && (*q == '+' || *q == '-')) q++;
    int ed = 0;
    while (q < t && *q >= '0' && *q <= '9') { q++; ed++; }
    if (!ed) return false;
  }
  if (q != t) return false;
  auto r = std::from_chars(p, t, *v);
  if (r.ec != std::errc() || r.ptr != t) return false;
  p = t;
  return true;
}

// the serial loop's extraction, for lines the fast path declines
inline bool slow_header(const char* b, const char* e, Chain* c, int64_t* np) {
  std::istringstream iss(std::string(b, e));
  return (bool) (iss >> c->id >> *np >> c->first_point_idx >> c->last_point_idx >> c->left_polygon_id >> c->right_polygon_id);
}
inline bool slow_point(const char* b, const char* e, Point2d* p) {
  std::istringstream iss(std::string(b, e));
  return (bool) (iss >> p->x >> p->y);
}
inline bool parse_header(const char* b, const char* e, Chain* c, int64_t* np) {
  const char* p = b;
  Chain t{};
  int64_t n = 0;
  if (fast_i64(p, e, &t.id) && fast_i64(p, e, &n) && fast_i64(p, e, &t.first_point_idx) && fast_i64(p, e, &t.last_point_idx) &&
      fast_i64(p, e, &t.left_polygon_id) && fast_i64(p, e, &t.right_polygon_id)) {
    *c = t; *np = n;
    return true;  // (whatever follows the sixth field is ignored by the serial loop too)
  }
  *c = Chain{}; *np = 0;
  return slow_header(b, e, c, np);
}
inline bool parse_point(const char* b, const char* e, Point2d* pt) {
  const char* p = b;
  Point2d t{};
  if (fast_f64(p, e, &t.x) && fast_f64(p, e, &t.y)) { *pt = t; return true; }
  *pt = Point2d{};
  return slow_point(b, e, pt);
}

inline int loader_threads() {
  if (const char* s = getenv("RAYJOIN_LOADER_THREADS")) { int v = atoi(s); if (v > 0) return v; }
  unsigned hc = std::thread::hardware_concurrency();
  return (int) std::max(1u, std::min(hc ? hc : 1u, 16u));  // a GPU box gives one GPU's share of the host: 16 cores
}

}  // namespace detail

inline std::shared_ptr<PlanarGraph> read_pgraph_parallel(const char* path, int verbose, int nthreads) {
  const int fd = open(path, O_RDONLY);
  if (fd < 0) throw std::runtime_error(std::string("Cannot open file ") + path);
  struct stat st{};
  if (fstat(fd, &st) != 0) { close(fd); throw std::runtime_error(std::string("Cannot open file ") + path); }
  const size_t size = (size_t) st.st_size;
  auto pg = std::make_shared<PlanarGraph>();
  if (size == 0) { close(fd); return pg; }
  void* map = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  if (map == MAP_FAILED) throw std::runtime_error(std::string("Cannot map file ") + path);
  const char* buf = static_cast<const char*>(map);
  struct Unmap { void* p; size_t n; ~Unmap() { munmap(p, n); } } unmap{map, size};
  const int T = std::max(1, nthreads);
  auto run = [&](auto&& fn) {  // fn(t) on T threads
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back(fn, t);
    fn(0);
    for (auto& x : th) x.join();
  };
  // ---- A: content lines (start offsets) per byte range
  std::vector<std::vector<uint64_t>> starts(T);
  run([&](int t) {
    size_t b = size * (size_t) t / T, e = size * (size_t) (t + 1) / T;
    if (t > 0) {  // first line that STARTS in [b, e)
      const void* nl = b ? memchr(buf + b - 1, '\n', size - (b - 1)) : nullptr;
      b = nl ? (size_t) (static_cast<const char*>(nl) - buf) + 1 : size;
    }
    auto& v = starts[t];
    v.reserve((e > b ? e - b : 0) / 24 + 16);
    size_t s = b;
    while (s < e && s < size) {
      const void* nl = memchr(buf + s, '\n', size - s);
      const size_t le = nl ? (size_t) (static_cast<const char*>(nl) - buf) : size;
      if (le > s && buf[s] != '#' && buf[s] != '%') v.push_back(s);
      s = le + 1;
    }
  });
  std::vector<uint64_t> first(T + 1, 0);  // global index of each range's first content line
  for (int t = 0; t < T; t++) first[t + 1] = first[t] + starts[t].size();
  const uint64_t nlines = first[T];
  auto line_at = [&](uint64_t i, const char** b, const char** e) {
    const int t = (int) (std::upper_bound(first.begin(), first.end(), i) - first.begin()) - 1;
    const uint64_t s = starts[t][i - first[t]];
    const void* nl = memchr(buf + s, '\n', size - s);
    *b = buf + s;
    *e = nl ? static_cast<const char*>(nl) : buf + size;
  };
  auto lno_of = [&](const char* b) {  // 1-based physical line number (error path only)
    return (size_t) std::count(buf, b, '\n') + 1;
  };
  auto bad_line = [&](const char* b, const char* e) {
    std::ostringstream m;
    m << "Bad line. Check your dataset! " << path << "[" << lno_of(b) << "]: " << std::string(b, e);
    return std::runtime_error(m.str());
  };
  // ---- B: headers in order
  auto& g = *pg;
  struct Span { uint64_t line; int64_t np; };  // first point line and declared point count of a chain
  std::vector<Span> spans;
  const char* bad_b = nullptr;  // earliest bad line found so far
  const char* bad_e = nullptr;
  uint64_t npts = 0;
  bool incomplete = false;
  for (uint64_t i = 0; i < nlines;) {
    const char *b, *e;
    line_at(i, &b, &e);
    Chain c{};
    int64_t np = 0;
    bool bad = !detail::parse_header(b, e, &c, &np);
    bad |= np < 2;
    if (bad) { bad_b = b; bad_e = e; break; }  // (point lines before it are still to be checked, in C)
    g.chains.push_back(c);
    g.row_index.push_back((uint32_t) npts);
    int64_t have = np;
    if (i + 1 + (uint64_t) np > nlines) { have = (int64_t) (nlines - i - 1); incomplete = true; }
    spans.push_back(Span{i + 1, have});
    npts += (uint64_t) have;
    i += 1 + (uint64_t) np;
  }
  g.points.resize(npts);
  // ---- C: points, chains dealt to threads by point count
  struct Part { const char *bad_b = nullptr, *bad_e = nullptr; BoundingBox bb; double mn = std::numeric_limits<double>::max(), mx = 0, sum = 0; };
  std::vector<Part> part(T);
  const size_t nch = spans.size();
  run([&](int t) {
    const uint64_t lo_pt = npts * (uint64_t) t / T, hi_pt = npts * (uint64_t) (t + 1) / T;
    // chains whose first point index lies in [lo_pt, hi_pt)
    size_t c0 = std::lower_bound(g.row_index.begin(), g.row_index.begin() + nch, (uint32_t) std::min<uint64_t>(lo_pt, 0xFFFFFFFFull)) - g.row_index.begin();
    size_t c1 = t + 1 == T ? nch : std::lower_bound(g.row_index.begin(), g.row_index.begin() + nch, (uint32_t) std::min<uint64_t>(hi_pt, 0xFFFFFFFFull)) - g.row_index.begin();
    Part& P = part[t];
    for (size_t c = c0; c < c1 && !P.bad_b; c++) {
      Point2d last{0, 0};
      for (int64_t k = 0; k < spans[c].np; k++) {
        const char *b, *e;
        line_at(spans[c].line + (uint64_t) k, &b, &e);
        Point2d p{};
        bool bad = !detail::parse_point(b, e, &p);
        if (k > 0) {
          const double len = std::sqrt((p.x - last.x) * (p.x - last.x) + (p.y - last.y) * (p.y - last.y));
          P.mn = std::min(P.mn, len); P.mx = std::max(P.mx, len); P.sum += len;
          bad |= p.x == last.x && p.y == last.y;
        }
        if (bad) { P.bad_b = b; P.bad_e = e; break; }
        P.bb.min_x = std::min(P.bb.min_x, p.x); P.bb.max_x = std::max(P.bb.max_x, p.x);
        P.bb.min_y = std::min(P.bb.min_y, p.y); P.bb.max_y = std::max(P.bb.max_y, p.y);
        g.points[(size_t) g.row_index[c] + (size_t) k] = p;
        last = p;
      }
    }
  });
  double min_len = std::numeric_limits<double>::max(), max_len = 0, sum_len = 0;
  for (auto& P : part) {
    if (P.bad_b && (!bad_b || P.bad_b < bad_b)) { bad_b = P.bad_b; bad_e = P.bad_e; }
    g.bb.min_x = std::min(g.bb.min_x, P.bb.min_x); g.bb.max_x = std::max(g.bb.max_x, P.bb.max_x);
    g.bb.min_y = std::min(g.bb.min_y, P.bb.min_y); g.bb.max_y = std::max(g.bb.max_y, P.bb.max_y);
    min_len = std::min(min_len, P.mn); max_len = std::max(max_len, P.mx); sum_len += P.sum;
  }
  if (bad_b) throw bad_line(bad_b, bad_e);
  if (!g.points.empty()) g.row_index.push_back((uint32_t) g.points.size());
  if (incomplete) throw std::runtime_error(std::string(path) + ": trailing incomplete chain");
  if (verbose)
    std::cerr << "Map " << path << " is loaded, chains: " << g.chains.size() << " points: " << g.points.size()
              << " edges: " << g.n_edges() << ", min seg len: " << min_len << ", max seg len: " << max_len
              << ", avg seg len: " << (g.n_edges() ? sum_len / g.n_edges() : 0.0) << std::endl;
  return pg;
}

// text CDB -> PlanarGraph: the parallel form on files worth it, the reference's loop otherwise
inline std::shared_ptr<PlanarGraph> read_pgraph(const char* path, int verbose = 0) {
  const int T = detail::loader_threads();
  struct stat st{};
  long min_bytes = 1 << 20;  // below this the threads cost more than they save
  if (const char* mb = getenv("RAYJOIN_LOADER_MIN_BYTES")) min_bytes = atol(mb);
  if (T > 1 && stat(path, &st) == 0 && st.st_size >= min_bytes) return read_pgraph_parallel(path, verbose, T);
  return read_pgraph_serial(path, verbose);
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
