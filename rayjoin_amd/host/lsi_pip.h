// lsi_pip.h -- the reference's query operators for -mode=lbvh and -mode=grid over the C ABI.
//   LSI  {Init, Query, get_xsects, CopyTo}     src/app/lsi.h:8-43, src/app/lsi_lbvh.h:17-98
//   PIP  {Init, Query, get_closest_eids}       src/app/pip.h:9-38, src/app/pip_lbvh.h:14-142
//   LSIGrid / PIPGrid                          src/app/lsi_grid.h:80-131, src/app/pip_grid.h:14-70
// Same method names, signatures (Query(Stream&, int query_map_id[, points]), src/app/lsi.h:27,
// src/app/pip.h:24), argument meaning and lifetime rules (results stay valid until the next Query or
// destruction); errors are exceptions carrying the C-ABI status (the reference throws from
// CUDA_CHECK, src/util/exception.h:150-158); a full queue throws instead of being UB.
#pragma once
#include <vector>

#include "context.h"

namespace rayjoin {

class LSI {
 public:
  using xsect_t = rj_xsect;
  explicit LSI(Context& ctx) : ctx_(ctx) {}
  virtual ~LSI() { if (queue_) rj_dev_free(ctx_.handle(), queue_); if (xsects_) rj_dev_free(ctx_.handle(), xsects_); }
  virtual void Init(size_t max_n_xsects) {
    std::cerr << "Queue size: " << max_n_xsects * sizeof(xsect_t) / 1024 / 1024 << " MB" << std::endl;
    cap_ = max_n_xsects;
    rj_check(ctx_.handle(), rj_dev_alloc(ctx_.handle(), 8 * (cap_ ? cap_ : 1), (void**) &queue_), "rj_dev_alloc");
  }
  virtual void Query(Stream& stream, int query_map_id) = 0;
  size_t size() const { return n_; }
  size_t local_size() const { return n_local_; }
  // restrict Query to the eid range of this rank's shard of the query map (default: every edge)
  void set_query_range(size_t e0, size_t e1) { e0_ = e0; e1_ = e1; ranged_ = true; }
  // all-gather-v of every rank's intersection queue (rj_comm_init first); afterwards this object
  // holds ALL pairs, in rank order
  uint64_t AllGather(size_t capacity) {
    rj_handle h = ctx_.handle();
    uint32_t* all = nullptr;
    rj_check(h, rj_dev_alloc(h, 8 * (capacity ? capacity : 1), (void**) &all), "rj_dev_alloc");
    uint64_t total = 0;
    int rc = rj_allgather_pairs(h, queue_, n_, all, capacity, nullptr, &total);
    if (rc != RJ_OK) { rj_dev_free(h, all); rj_check(h, rc, "rj_allgather_pairs"); }
    n_local_ = n_;
    rj_dev_free(h, queue_);
    queue_ = all;
    n_ = total;
    cap_ = capacity;
    return total;
  }
  // 48-byte Intersection records on the host, sorted by (eid[0], eid[1])
  void CopyTo(std::vector<xsect_t>& out) {
    rj_handle h = ctx_.handle();
    rj_check(h, rj_sort_pairs(h, queue_, n_), "rj_sort_pairs");
    if (xsects_) { rj_dev_free(h, xsects_); xsects_ = nullptr; }
    rj_check(h, rj_dev_alloc(h, 48 * (n_ ? n_ : 1), (void**) &xsects_), "rj_dev_alloc");
    rj_check(h, rj_lsi_points(h, queue_, n_, xsects_), "rj_lsi_points");
    out.resize(n_);
    rj_check(h, rj_memcpy_d2h(h, out.data(), xsects_, 48 * n_), "rj_memcpy_d2h");
  }
  Context& get_context() { return ctx_; }

 protected:
  Context& ctx_;
  uint32_t* queue_ = nullptr;  // (eid map 0, eid map 1) pairs, device
  rj_xsect* xsects_ = nullptr;
  size_t cap_ = 0, n_ = 0, n_local_ = 0;
  size_t e0_ = 0, e1_ = 0;
  bool ranged_ = false;
};

class LSILBVH : public LSI {
 public:
  explicit LSILBVH(Context& ctx) : LSI(ctx) {}
  void Query(Stream& stream, int query_map_id) override {
    stream.Bind();
    uint64_t n = 0;
    const size_t qb = ranged_ ? e0_ : 0, qe = ranged_ ? e1_ : ctx_.get_map(query_map_id)->n_edges();
    int rc = rj_lsi_query(ctx_.handle(), 1 - query_map_id, query_map_id, qb, qe, cap_, queue_, &n);
    n_ = n < cap_ ? n : cap_;
    rj_check(ctx_.handle(), rc, "rj_lsi_query");
  }
};

// -mode=grid: the uniform grid holds both maps (UniformGrid::AddMapsToGrid); the result does not
// depend on query_map_id (lsi_grid.h:103-104) and cannot be restricted to an eid range.
class LSIGrid : public LSI {
 public:
  explicit LSIGrid(Context& ctx) : LSI(ctx) {}
  void Query(Stream& stream, int /*query_map_id*/) override {
    stream.Bind();
    if (ranged_ && !(e0_ == 0 && e1_ == ctx_.get_map(1)->n_edges()))
      throw RjError(RJ_E_INVALID, "LSIGrid: -mode=grid joins the two whole maps (no shards)");
    uint64_t n = 0;
    int rc = rj_lsi_query_grid(ctx_.handle(), cap_, queue_, &n);
    n_ = n < cap_ ? n : cap_;
    rj_check(ctx_.handle(), rc, "rj_lsi_query_grid");
  }
};

class PIP {
 public:
  explicit PIP(Context& ctx) : ctx_(ctx) {}
  virtual ~PIP() { if (closest_) rj_dev_free(ctx_.handle(), closest_); if (faces_) rj_dev_free(ctx_.handle(), faces_); }
  virtual void Init(size_t n_points) {
    cap_ = n_points;
    rj_check(ctx_.handle(), rj_dev_alloc(ctx_.handle(), 4 * (cap_ ? cap_ : 1), (void**) &closest_), "rj_dev_alloc");
    rj_check(ctx_.handle(), rj_dev_alloc(ctx_.handle(), 4 * (cap_ ? cap_ : 1), (void**) &faces_), "rj_dev_alloc");
  }
  // query_points_dev == nullptr: every vertex of the query map (RunPIPQuery, run_query.cu:346)
  virtual void Query(Stream& stream, int query_map_id, const int64_t* query_points_dev, size_t n) = 0;
  void get_closest_eids(std::vector<uint32_t>& out) {
    out.resize(n_);
    rj_check(ctx_.handle(), rj_memcpy_d2h(ctx_.handle(), out.data(), closest_, 4 * n_), "rj_memcpy_d2h");
  }
  void get_face_ids(std::vector<int32_t>& out) {
    out.resize(n_);
    rj_check(ctx_.handle(), rj_memcpy_d2h(ctx_.handle(), out.data(), faces_, 4 * n_), "rj_memcpy_d2h");
  }

 protected:
  Context& ctx_;
  uint32_t* closest_ = nullptr;
  int32_t* faces_ = nullptr;
  size_t cap_ = 0, n_ = 0;
};

class PIPLBVH : public PIP {
 public:
  explicit PIPLBVH(Context& ctx) : PIP(ctx) {}
  void Query(Stream& stream, int query_map_id, const int64_t* query_points_dev, size_t n) override {
    stream.Bind();
    if (n > cap_) throw RjError(RJ_E_INVALID, "PIP::Query: more points than Init() reserved");
    rj_check(ctx_.handle(), rj_pip_query(ctx_.handle(), 1 - query_map_id, query_map_id, query_points_dev, 0, n,
                                         closest_, faces_), "rj_pip_query");
    n_ = n;
  }
};

class PIPGrid : public PIP {
 public:
  explicit PIPGrid(Context& ctx) : PIP(ctx) {}
  void Query(Stream& stream, int query_map_id, const int64_t* query_points_dev, size_t n) override {
    stream.Bind();
    if (n > cap_) throw RjError(RJ_E_INVALID, "PIP::Query: more points than Init() reserved");
    rj_check(ctx_.handle(), rj_pip_query_grid(ctx_.handle(), 1 - query_map_id, query_map_id, query_points_dev, 0, n,
                                              closest_, faces_), "rj_pip_query_grid");
    n_ = n;
  }
};

}  // namespace rayjoin
