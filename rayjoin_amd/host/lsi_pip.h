// lsi_pip.h -- the reference's query operators for -mode=lbvh and -mode=grid over the C ABI.
//   LSI<CONTEXT_T>  {Init, Query, get_xsects, CopyTo}   src/app/lsi.h:8-43, src/app/lsi_lbvh.h:17-98
//   PIP<CONTEXT_T>  {Init, Query, get_closest_eids}     src/app/pip.h:9-38, src/app/pip_lbvh.h:14-142
//   LSIGrid / PIPGrid                                   src/app/lsi_grid.h:80-131, src/app/pip_grid.h:14-70
//   ArrayView<T>                                        src/util/array_view.h:9-30 (a non-owning view of DEVICE memory)
// Same class templates, method names, signatures (Query(Stream&, int query_map_id[, ArrayView<point_t>]),
// src/app/lsi.h:27, src/app/pip.h:23), argument meaning and lifetime rules: after Query the operator holds its
// results ON THE DEVICE -- LSI the 48-byte Intersection records (get_xsects(), src/app/lsi.h:33: valid until the
// next Query or destruction), PIP the closest eids -- and copies to the host only when asked.  Errors are
// exceptions carrying the C-ABI status (the reference throws from CUDA_CHECK, src/util/exception.h:150-158); a full
// queue throws instead of being UB.
#pragma once
#include <vector>

#include "context.h"

namespace rayjoin {

template <typename T>
class ArrayView {
 public:
  ArrayView() = default;
  ArrayView(T* data, size_t size) : data_(data), size_(size) {}
  T* data() { return data_; }
  const T* data() const { return data_; }
  size_t size() const { return size_; }
  bool empty() const { return size_ == 0; }

 private:
  T* data_ = nullptr;
  size_t size_ = 0;
};

template <typename CONTEXT_T>
class LSI {
 public:
  using xsect_t = rj_xsect;  // layout of dev::Intersection<int64_t> (src/algo/lsi.h:15-27)
  explicit LSI(CONTEXT_T& ctx) : ctx_(ctx) {}
  virtual ~LSI() {
    if (queue_) rj_dev_free(ctx_.handle(), queue_);
    if (xsects_) rj_dev_free(ctx_.handle(), xsects_);
  }
  virtual void Init(size_t max_n_xsects) {
    std::cerr << "Queue size: " << max_n_xsects * sizeof(xsect_t) / 1024 / 1024 << " MB" << std::endl;
    cap_ = max_n_xsects;
    rj_check(ctx_.handle(), rj_dev_alloc(ctx_.handle(), 8 * (cap_ ? cap_ : 1), (void**) &queue_), "rj_dev_alloc");
    rj_check(ctx_.handle(), rj_dev_alloc(ctx_.handle(), sizeof(xsect_t) * (cap_ ? cap_ : 1), (void**) &xsects_), "rj_dev_alloc");
  }
  virtual void Query(Stream& stream, int query_map_id) = 0;
  CONTEXT_T& get_context() { return ctx_; }
  const CONTEXT_T& get_context() const { return ctx_; }
  // the records the last Query left, on the device, in queue order (src/app/lsi.h:33)
  ArrayView<xsect_t> get_xsects() { return ArrayView<xsect_t>(xsects_, n_); }
  // ... and the (eid map 0, eid map 1) pairs they were made from, same order
  ArrayView<uint32_t> get_pairs() { return ArrayView<uint32_t>(queue_, 2 * n_); }
  size_t size() const { return n_; }
  size_t local_size() const { return n_local_; }
  // restrict Query to the eid range of this rank's shard of the query map (default: every edge)
  void set_query_range(size_t e0, size_t e1) { e0_ = e0; e1_ = e1; ranged_ = true; }
  // all-gather-v of every rank's intersection queue (rj_comm_init first); afterwards this object
  // holds ALL pairs and their records, in rank order
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
    if (capacity > cap_) {
      rj_dev_free(h, xsects_);
      xsects_ = nullptr;
      rj_check(h, rj_dev_alloc(h, sizeof(xsect_t) * capacity, (void**) &xsects_), "rj_dev_alloc");
    }
    cap_ = capacity;
    rj_check(h, rj_lsi_points(h, queue_, n_, xsects_), "rj_lsi_points");
    return total;
  }
  // the records on the host, sorted by (eid[0], eid[1]) (src/app/lsi.h:37; the checker's order, run_overlay.cu:38-52)
  void CopyTo(std::vector<xsect_t>& out) {
    rj_handle h = ctx_.handle();
    rj_check(h, rj_sort_pairs(h, queue_, n_), "rj_sort_pairs");
    rj_check(h, rj_lsi_points(h, queue_, n_, xsects_), "rj_lsi_points");
    out.resize(n_);
    rj_check(h, rj_memcpy_d2h(h, out.data(), xsects_, sizeof(xsect_t) * n_), "rj_memcpy_d2h");
  }

 protected:
  CONTEXT_T& ctx_;
  uint32_t* queue_ = nullptr;  // (eid map 0, eid map 1) pairs, device
  xsect_t* xsects_ = nullptr;  // their 48-byte records, device
  size_t cap_ = 0, n_ = 0, n_local_ = 0;
  size_t e0_ = 0, e1_ = 0;
  bool ranged_ = false;
};

template <typename CONTEXT_T>
class LSILBVH : public LSI<CONTEXT_T> {
 public:
  explicit LSILBVH(CONTEXT_T& ctx) : LSI<CONTEXT_T>(ctx) {}
  // queue clear, traversal + predicate, the record of every hit, the count: one host sync (lsi_lbvh.h:27-98)
  void Query(Stream& stream, int query_map_id) override {
    stream.Bind();
    rj_handle h = this->ctx_.handle();
    const size_t qb = this->ranged_ ? this->e0_ : 0, qe = this->ranged_ ? this->e1_ : this->ctx_.get_map(query_map_id)->n_edges();
    rj_check(h, rj_lsi_query_async(h, 1 - query_map_id, query_map_id, qb, qe, this->cap_, this->queue_), "rj_lsi_query_async");
    rj_check(h, rj_lsi_points_async(h, this->queue_, this->cap_, this->xsects_), "rj_lsi_points_async");
    uint64_t n = 0;
    int rc = rj_lsi_query_finish(h, this->cap_, &n);
    this->n_ = n < this->cap_ ? n : this->cap_;
    rj_check(h, rc, "rj_lsi_query");
  }
};

// -mode=grid: the uniform grid holds both maps (UniformGrid::AddMapsToGrid); the result does not
// depend on query_map_id (lsi_grid.h:103-104) and cannot be restricted to an eid range.
template <typename CONTEXT_T>
class LSIGrid : public LSI<CONTEXT_T> {
 public:
  explicit LSIGrid(CONTEXT_T& ctx) : LSI<CONTEXT_T>(ctx) {}
  void Query(Stream& stream, int /*query_map_id*/) override {
    stream.Bind();
    rj_handle h = this->ctx_.handle();
    if (this->ranged_ && !(this->e0_ == 0 && this->e1_ == this->ctx_.get_map(1)->n_edges()))
      throw RjError(RJ_E_INVALID, "LSIGrid: -mode=grid joins the two whole maps (no shards)");
    uint64_t n = 0;
    int rc = rj_lsi_query_grid(h, this->cap_, this->queue_, &n);
    this->n_ = n < this->cap_ ? n : this->cap_;
    rj_check(h, rc, "rj_lsi_query_grid");
    rj_check(h, rj_lsi_points(h, this->queue_, this->n_, this->xsects_), "rj_lsi_points");
  }
};

template <typename CONTEXT_T>
class PIP {
 public:
  using point_t = typename CONTEXT_T::map_t::point_t;
  explicit PIP(CONTEXT_T& ctx) : ctx_(ctx) {}
  virtual ~PIP() {
    if (closest_) rj_dev_free(ctx_.handle(), closest_);
    if (faces_) rj_dev_free(ctx_.handle(), faces_);
  }
  virtual void Init(size_t n_points) {
    cap_ = n_points;
    rj_check(ctx_.handle(), rj_dev_alloc(ctx_.handle(), 4 * (cap_ ? cap_ : 1), (void**) &closest_), "rj_dev_alloc");
    rj_check(ctx_.handle(), rj_dev_alloc(ctx_.handle(), 4 * (cap_ ? cap_ : 1), (void**) &faces_), "rj_dev_alloc");
  }
  // query_points: a caller-owned DEVICE array (src/app/pip.h:23); a view with data() == nullptr and size() == n means
  // the first n vertices of the query map (RunPIPQuery queries every vertex, run_query.cu:346)
  virtual void Query(Stream& stream, int query_map_id, ArrayView<point_t> query_points) = 0;
  CONTEXT_T& get_context() { return ctx_; }
  // on the device (src/app/pip.h:30 returns the device vector), valid until the next Query
  ArrayView<uint32_t> get_closest_eids() { return ArrayView<uint32_t>(closest_, n_); }
  ArrayView<int32_t> get_face_ids() { return ArrayView<int32_t>(faces_, n_); }
  void get_closest_eids(std::vector<uint32_t>& out) {
    out.resize(n_);
    rj_check(ctx_.handle(), rj_memcpy_d2h(ctx_.handle(), out.data(), closest_, 4 * n_), "rj_memcpy_d2h");
  }
  void get_face_ids(std::vector<int32_t>& out) {
    out.resize(n_);
    rj_check(ctx_.handle(), rj_memcpy_d2h(ctx_.handle(), out.data(), faces_, 4 * n_), "rj_memcpy_d2h");
  }

 protected:
  CONTEXT_T& ctx_;
  uint32_t* closest_ = nullptr;
  int32_t* faces_ = nullptr;
  size_t cap_ = 0, n_ = 0;
};

template <typename CONTEXT_T>
class PIPLBVH : public PIP<CONTEXT_T> {
 public:
  using point_t = typename PIP<CONTEXT_T>::point_t;
  explicit PIPLBVH(CONTEXT_T& ctx) : PIP<CONTEXT_T>(ctx) {}
  void Query(Stream& stream, int query_map_id, ArrayView<point_t> query_points) override {
    stream.Bind();
    const size_t n = query_points.size();
    if (n > this->cap_) throw RjError(RJ_E_INVALID, "PIP::Query: more points than Init() reserved");
    rj_check(this->ctx_.handle(), rj_pip_query(this->ctx_.handle(), 1 - query_map_id, query_map_id, (const int64_t*) query_points.data(), 0, n,
                                               this->closest_, this->faces_), "rj_pip_query");
    this->n_ = n;
  }
};

template <typename CONTEXT_T>
class PIPGrid : public PIP<CONTEXT_T> {
 public:
  using point_t = typename PIP<CONTEXT_T>::point_t;
  explicit PIPGrid(CONTEXT_T& ctx) : PIP<CONTEXT_T>(ctx) {}
  void Query(Stream& stream, int query_map_id, ArrayView<point_t> query_points) override {
    stream.Bind();
    const size_t n = query_points.size();
    if (n > this->cap_) throw RjError(RJ_E_INVALID, "PIP::Query: more points than Init() reserved");
    rj_check(this->ctx_.handle(), rj_pip_query_grid(this->ctx_.handle(), 1 - query_map_id, query_map_id, (const int64_t*) query_points.data(), 0, n,
                                                    this->closest_, this->faces_), "rj_pip_query_grid");
    this->n_ = n;
  }
};

}  // namespace rayjoin
