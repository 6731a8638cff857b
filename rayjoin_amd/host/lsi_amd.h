// lsi_amd.h -- the adapter a RayJoin maintainer adds next to LSILBVH / PIPLBVH: two
// subclasses of the reference's operator interfaces that hand the query to librayjoin_amd.so.
//   LSIAMD<CONTEXT_T> : LSI<CONTEXT_T>    replaces LSILBVH<CONTEXT_T>   src/app/lsi_lbvh.h:17-98
//   PIPAMD<CONTEXT_T> : PIP<CONTEXT_T>    replaces PIPLBVH<CONTEXT_T>   src/app/pip_lbvh.h:14-142
// Written against nothing but the base-class surface both trees share -- LSI<CONTEXT_T>(ctx), virtual Init(size_t),
// virtual Query(Stream&, int[, ArrayView<point_t>]), Stream::cuda_stream(), ArrayView<T>(ptr, n) -- and the C ABI:
// it owns its device buffers through rj_dev_alloc and never touches the base classes' own queue, so it does not
// care whether that is a thrust-backed Queue (the reference) or this repository's.  Include it AFTER the tree's
// app/lsi.h + app/pip.h (here: host/lsi_pip.h).  Compiled and run by tests/test_gpu_cli.py through
// `query_exec -mode=amd` (same results and output files as -mode=lbvh).
#pragma once
#include <stdexcept>
#include <string>

#include "rayjoin_amd.h"

namespace rayjoin {

struct AmdError : std::runtime_error {
  int code;
  AmdError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
#define RJ_OK_OR_THROW(h, expr)                                                                     \
  do {                                                                                              \
    int _rc = (expr);                                                                               \
    if (_rc != RJ_OK) throw ::rayjoin::AmdError(_rc, std::string(#expr ": ") + rj_last_error_string(h)); \
  } while (0)

// the handle works on the caller's stream (a hipStream_t on ROCm); a Stream without one = the handle's own
inline void bind(rj_handle h, const Stream& stream) {
  if (stream.cuda_stream()) RJ_OK_OR_THROW(h, rj_set_stream(h, (void*) stream.cuda_stream()));
  else RJ_OK_OR_THROW(h, rj_set_option(h, "own_stream", 1));
}

template <typename CONTEXT_T>
class LSIAMD : public LSI<CONTEXT_T> {
 public:
  using xsect_t = typename LSI<CONTEXT_T>::xsect_t;
  static_assert(sizeof(xsect_t) == sizeof(rj_xsect), "rj_xsect is laid out like dev::Intersection<int64_t> (48 bytes)");

  LSIAMD(CONTEXT_T& ctx, rj_handle h) : LSI<CONTEXT_T>(ctx), h_(h) {}
  ~LSIAMD() override {
    if (pairs_) rj_dev_free(h_, pairs_);
    if (xsects_) rj_dev_free(h_, xsects_);
  }
  void Init(size_t max_n_xsects) override {  // Queue::Init (src/app/lsi.h:21-25)
    cap_ = max_n_xsects;
    RJ_OK_OR_THROW(h_, rj_dev_alloc(h_, 8 * (cap_ ? cap_ : 1), (void**) &pairs_));
    RJ_OK_OR_THROW(h_, rj_dev_alloc(h_, sizeof(rj_xsect) * (cap_ ? cap_ : 1), (void**) &xsects_));
  }
  void Query(Stream& stream, int query_map_id) override {
    bind(h_, stream);
    uint64_t ne = 0, n = 0;
    RJ_OK_OR_THROW(h_, rj_map_num_edges(h_, query_map_id, &ne));
    RJ_OK_OR_THROW(h_, rj_lsi_query_async(h_, 1 - query_map_id, query_map_id, 0, ne, cap_, pairs_));
    RJ_OK_OR_THROW(h_, rj_lsi_points_async(h_, pairs_, cap_, xsects_));  // the 48-byte records, same stream
    const int rc = rj_lsi_query_finish(h_, cap_, &n);                     // the one host sync (Queue::size)
    n_ = n < cap_ ? n : cap_;
    RJ_OK_OR_THROW(h_, rc);  // RJ_E_OVERFLOW: the queue was too small (UB in the reference, queue.h:37), n is the true count
  }
  // shadows LSI::get_xsects (src/app/lsi.h:33, not virtual there): callers that hold the operator by its base type go
  // through xsects_view(); run_query.cu:304 and the overlay's ComputeOutputPolygons hold the concrete type
  ArrayView<xsect_t> get_xsects() { return ArrayView<xsect_t>(reinterpret_cast<xsect_t*>(xsects_), n_); }
  ArrayView<uint32_t> get_pairs() { return ArrayView<uint32_t>(pairs_, 2 * n_); }
  size_t size() const { return n_; }

 private:
  rj_handle h_;
  uint32_t* pairs_ = nullptr;
  rj_xsect* xsects_ = nullptr;
  size_t cap_ = 0, n_ = 0;
};

template <typename CONTEXT_T>
class PIPAMD : public PIP<CONTEXT_T> {
 public:
  using point_t = typename CONTEXT_T::map_t::point_t;
  static_assert(sizeof(point_t) == 16, "a query point is two int64 (the scaled coordinates)");

  PIPAMD(CONTEXT_T& ctx, rj_handle h) : PIP<CONTEXT_T>(ctx), h_(h) {}
  ~PIPAMD() override { if (closest_) rj_dev_free(h_, closest_); }
  void Init(size_t n_points) override {
    cap_ = n_points;
    RJ_OK_OR_THROW(h_, rj_dev_alloc(h_, 4 * (cap_ ? cap_ : 1), (void**) &closest_));
  }
  void Query(Stream& stream, int query_map_id, ArrayView<point_t> query_points) override {  // src/app/pip.h:23
    bind(h_, stream);
    if (query_points.size() > cap_) throw AmdError(RJ_E_INVALID, "PIPAMD::Query: more points than Init() reserved");
    RJ_OK_OR_THROW(h_, rj_pip_query(h_, 1 - query_map_id, query_map_id, reinterpret_cast<const int64_t*>(query_points.data()), 0,
                                    query_points.size(), closest_, nullptr));
    n_ = query_points.size();
  }
  // index_t is uint32_t in both trees and the miss value 0xFFFFFFFF is the same (src/config.h: DONTKNOW)
  ArrayView<uint32_t> get_closest_eids() { return ArrayView<uint32_t>(closest_, n_); }

 private:
  rj_handle h_;
  uint32_t* closest_ = nullptr;
  size_t cap_ = 0, n_ = 0;
};

}  // namespace rayjoin
