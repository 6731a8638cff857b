// context.h -- Scaling + Context of the reference, host C++ over the C ABI.
//   Scaling<double,int64_t,17>    src/map/scaling.h:32-136
//   Context                       src/context.h:31-88 (joint bbox -> one Scaling; LoadToDevice)
// Scaled coordinates are produced HERE, once, with a separate multiply and add (this file must
// be compiled with -ffp-contract=off; SURVEY 7 hard part 7) and shipped to the GPU as int64.
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rayjoin_amd.h"
#include "planar_graph.h"

namespace rayjoin {

class Scaling {
 public:
  Scaling() = default;
  // fused: x * rx + delta as ONE std::fma -- what nvcc's default -fmad=true makes of the reference's
  // device lambda (src/map/map.h:171-180); the default is scaling.h's separate multiply and add.
  explicit Scaling(const BoundingBox& bb, bool fused = false) : fused_(fused) {
    double max_x = bb.max_x + 1, min_x = bb.min_x - 1, max_y = bb.max_y + 1, min_y = bb.min_y - 1;  // config.h:4
    rx_ = (double) internal_range_ / (max_x - min_x);
    ry_ = (double) internal_range_ / (max_y - min_y);
    rrx_ = 1 / rx_;
    rry_ = 1 / ry_;
    deltax_ = 0.5 * ((internal_max_ + internal_min_) - (max_x + min_x) * rx_);
    deltay_ = 0.5 * ((internal_max_ + internal_min_) - (max_y + min_y) * ry_);
    ddeltax_ = 0.5 * ((max_x + min_x) - (internal_max_ + internal_min_) * rrx_);
    ddeltay_ = 0.5 * ((max_y + min_y) - (internal_max_ + internal_min_) * rry_);
  }
  int64_t ScaleX(double x) const { if (fused_) return (int64_t) std::fma(x, rx_, deltax_); double t = x * rx_; return (int64_t) (t + deltax_); }
  int64_t ScaleY(double y) const { if (fused_) return (int64_t) std::fma(y, ry_, deltay_); double t = y * ry_; return (int64_t) (t + deltay_); }
  double UnscaleX(int64_t v) const { double t = (double) v * rrx_; return t + ddeltax_; }
  double UnscaleY(int64_t v) const { double t = (double) v * rry_; return t + ddeltay_; }
  int64_t get_internal_min() const { return internal_min_; }
  int64_t get_internal_max() const { return internal_max_; }
  int64_t get_internal_range() const { return internal_range_; }

 private:
  int64_t internal_max_ = INT64_MAX >> 17, internal_min_ = INT64_MIN >> 17;
  int64_t internal_range_ = (INT64_MAX >> 17) - (INT64_MIN >> 17);
  double rx_ = 0, ry_ = 0, rrx_ = 0, rry_ = 0, deltax_ = 0, deltay_ = 0, ddeltax_ = 0, ddeltay_ = 0;
  bool fused_ = false;
};

struct RjError : std::runtime_error {
  int code;
  RjError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

inline void rj_check(rj_handle h, int rc, const char* what) {
  if (rc != RJ_OK) throw RjError(rc, std::string(what) + ": " + rj_last_error_string(h));
}

// host-side image of one scaled map (what rj_upload_map takes)
struct HostMap {
  struct point_t { int64_t x, y; };  // a scaled point as the device holds it (Map::point_t, src/map/map.h:53: cuda_vec<int64_t>::type_2d)
  std::vector<int64_t> xy;          // 2*np
  std::vector<uint32_t> row_index;  // nc+1
  std::vector<int64_t> left, right; // nc
  size_t n_points() const { return xy.size() / 2; }
  size_t n_chains() const { return left.size(); }
  size_t n_edges() const { return n_points() - n_chains(); }
  // contiguous chain range of shard `rank` of `nranks`, balanced by edge count; a chain range is a
  // contiguous eid range [e0, e1) and point range [p0, p1) because eid = p_idx - ichain (map.h:200-203)
  void shard(int nranks, int rank, size_t* e0, size_t* e1, size_t* p0, size_t* p1) const {
    const size_t nc = n_chains(), ne = n_edges();
    auto cut = [&](int r) -> size_t {  // first chain c with (#edges before c) >= r * ne / nranks
      if (r <= 0) return 0;
      if (r >= nranks) return nc;
      const double target = (double) r * (double) ne / nranks;
      size_t lo = 0, hi = nc;
      while (lo < hi) {
        size_t mid = (lo + hi) / 2;
        if ((double) (row_index[mid] - mid) >= target) hi = mid; else lo = mid + 1;
      }
      return lo;
    };
    const size_t c0 = cut(rank), c1 = std::max(cut(rank + 1), c0);
    *e0 = row_index[c0] - c0; *e1 = row_index[c1] - c1;
    *p0 = row_index[c0]; *p1 = row_index[c1];
  }
};

// Stream (src/util/stream.h:13-27): the stream a Query is enqueued on.  The reference's owns a
// non-blocking cudaStream_t; here it names either the handle's own non-blocking stream (what
// Context::get_stream() returns) or a caller-owned hipStream_t, and Query(stream, ...) points the
// handle at it through rj_set_stream.
class Stream {
 public:
  explicit Stream(rj_handle h) : h_(h), own_(true) {}
  Stream(rj_handle h, void* hip_stream) : h_(h), native_(hip_stream), own_(false) {}
  void Sync() const { rj_check(h_, rj_sync(h_), "rj_sync"); }
  // make the handle work on this stream
  void Bind() const {
    rj_check(h_, own_ ? rj_set_option(h_, "own_stream", 1) : rj_set_stream(h_, native_), "rj_set_stream");
  }
  void* native() const { return native_; }
  void* cuda_stream() const { return native_; }  // (the reference's accessor, src/util/stream.h:55: nullptr = the handle's own stream)

 private:
  rj_handle h_;
  void* native_ = nullptr;
  bool own_;
};

class Context {
 public:
  // the type names the reference's operators take from their CONTEXT_T (src/context.h:19-24)
  using coord_t = double;
  using internal_coord_t = int64_t;
  using coefficient_t = __int128;
  using map_t = HostMap;
  explicit Context(const std::array<std::shared_ptr<PlanarGraph>, 2>& pgs, int device = 0, bool fused_scaling = false) : pgraphs_(pgs) {
    for (auto& g : pgs)
      if (g) {
        bb_.min_x = std::min(bb_.min_x, g->bb.min_x); bb_.max_x = std::max(bb_.max_x, g->bb.max_x);
        bb_.min_y = std::min(bb_.min_y, g->bb.min_y); bb_.max_y = std::max(bb_.max_y, g->bb.max_y);
      }
    scaling_ = Scaling(bb_, fused_scaling);
    std::cerr << "Bounding Box, Bottom-left: (" << bb_.min_x << ", " << bb_.min_y << "), Top-right: (" << bb_.max_x
              << ", " << bb_.max_y << ")" << std::endl;
    int rc = rj_create(device, &h_);
    if (rc != RJ_OK) throw RjError(rc, "rj_create failed: no usable HIP device (the HIP path is the only compute path)");
    stream_.reset(new Stream(h_));
  }
  ~Context() { if (h_) rj_destroy(h_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;

  void LoadToDevice() {  // context.h:76-88
    for (int im = 0; im < 2; im++)
      if (pgraphs_[im]) {
        auto& g = *pgraphs_[im];
        auto m = std::make_shared<HostMap>();
        m->xy.resize(2 * g.points.size());
        for (size_t i = 0; i < g.points.size(); i++) {
          m->xy[2 * i] = scaling_.ScaleX(g.points[i].x);
          m->xy[2 * i + 1] = scaling_.ScaleY(g.points[i].y);
        }
        m->row_index = g.row_index;
        for (auto& c : g.chains) { m->left.push_back(c.left_polygon_id); m->right.push_back(c.right_polygon_id); }
        set_map(im, m);
      }
  }
  void set_map(int im, std::shared_ptr<HostMap> m) {
    maps_[im] = m;
    rj_check(h_, rj_upload_map(h_, im, m->xy.data(), m->n_points(), m->row_index.data(), m->left.data(),
                               m->right.data(), m->n_chains()), "rj_upload_map");
  }
  std::shared_ptr<HostMap> get_map(int im) { return maps_[im]; }
  std::shared_ptr<PlanarGraph> get_planar_graph(int im) { return pgraphs_[im]; }
  const Scaling& get_scaling() const { return scaling_; }
  const BoundingBox& get_bounding_box() const { return bb_; }
  rj_handle handle() { return h_; }
  Stream& get_stream() { return *stream_; }  // context.h:119

 private:
  std::unique_ptr<Stream> stream_;
  std::array<std::shared_ptr<PlanarGraph>, 2> pgraphs_;
  std::array<std::shared_ptr<HostMap>, 2> maps_;
  BoundingBox bb_;
  Scaling scaling_;
  rj_handle h_ = nullptr;
};

}  // namespace rayjoin
