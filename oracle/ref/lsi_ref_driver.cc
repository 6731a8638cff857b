// lsi_ref_driver.cc -- thin C entry points around the REFERENCE's own headers, compiled in
// place from /root/reference/src (never copied): algo/lsi.h (intersect_test, both overloads),
// util/rational.h (tcb::rational), grid/cell.h (calculate_cell), map/scaling.h +
// map/bounding_box.h + util/type_traits.h (Scaling<double, int64_t, 17>), config.h.
//
// TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile target `ref` into oracle/_ref/ (git-ignored).
// Used to (1) validate oracle/rj_oracle.c and (2) generate tests/golden/lsi_ref_vectors.json.
//
// How the headers compile on the host: the reference's util/util.h pulls cuda_runtime.h and the
// thrust CUDA backend, which this image lacks.  No stand-in header is written; instead its
// include guard (RAYJOIN_UTIL_H) is pre-defined on the command line, and the seven macros its own
// non-CUDA branch defines (DEV_HOST, DEV_HOST_INLINE, DEV_INLINE, MIN, MAX, MIN4, MAX4;
// src/util/util.h:13-22) are given with -D.  util/type_traits.h includes <cuda_runtime.h> for the
// vector types (long2, double2 ...): that header IS in this image -- NVIDIA's own, shipped inside
// the triton package (oracle/Makefile finds it) -- so Scaling compiles against the real thing.
// Everything the predicate and the scaling compute comes from the reference sources.  The edge
// equation (map.h needs thrust) is formed here from src/map/map.h:216-226's formula and the point
// type is ours (lsi.h is templated on POINT_T).  Compiled with -ffp-contract=off: Scaling is the
// HOST build of the reference's class (separate multiply and add).
#include <cstdint>
#include <cstring>

#include <cassert>
#include <cstdio>
#include <limits>

#include "config.h"
#include "util/rational.h"
#include "algo/lsi.h"
#include "grid/cell.h"
#include "map/bounding_box.h"
#include "map/scaling.h"

namespace {
struct Pt {
  long x, y;
  bool operator==(const Pt& o) const { return x == o.x && y == o.y; }
};
struct EdgeEq {
  __int128 a, b, c;
  EdgeEq(const Pt& p1, const Pt& p2) {
    a = p1.y - p2.y;
    b = p2.x - p1.x;
    c = -(__int128) p1.x * a - (__int128) p1.y * b;
    if (b < 0) {
      a = -a;
      b = -b;
      c = -c;
    }
  }
};
using RefScaling = rayjoin::Scaling<double>;  // <double, int64_t, 17>; its getters feed calculate_cell
RefScaling make_scaling(const double* bb) {
  rayjoin::BoundingBox<double> b;
  b.min_x = bb[0]; b.min_y = bb[1]; b.max_x = bb[2]; b.max_y = bb[3];
  return RefScaling(b);
}
void put128(int64_t* out, __int128 v) {
  out[0] = (int64_t) (unsigned __int128) v;
  out[1] = (int64_t) ((unsigned __int128) v >> 64);
}
}  // namespace

extern "C" {
int ref_intersect_test_segs(const int64_t* s1, const int64_t* s2) {
  Pt a1{s1[0], s1[1]}, a2{s1[2], s1[3]}, b1{s2[0], s2[1]}, b2{s2[2], s2[3]};
  EdgeEq e1(a1, a2), e2(b1, b2);
  return rayjoin::dev::intersect_test<EdgeEq, EdgeEq, Pt, __int128>(e1, a1, a2, e2, b1, b2);
}

// same output layout as rjo_intersect_point_segs (oracle/rj_oracle.c)
int ref_intersect_point_segs(const int64_t* s1, const int64_t* s2, int gsize, int64_t* out) {
  Pt a1{s1[0], s1[1]}, a2{s1[2], s1[3]}, b1{s2[0], s2[1]}, b2{s2[2], s2[3]};
  EdgeEq e1(a1, a2), e2(b1, b2);
  tcb::rational<__int128> x, y;
  if (!rayjoin::dev::intersect_test<EdgeEq, EdgeEq, Pt, __int128>(e1, a1, a2, e2, b1, b2, x, y))
    return 0;
  put128(out + 0, x.num());
  put128(out + 2, x.denom());
  put128(out + 4, y.num());
  put128(out + 6, y.denom());
  rayjoin::dev::Intersection<int64_t> xs;  // the 48-byte record the queue stores
  xs.x = x;
  xs.y = y;
  out[8] = xs.x.num();
  out[9] = xs.y.num();
  const RefScaling sc;  // default constructor: the internal range only (scaling.h:43-54)
  out[10] = rayjoin::dev::calculate_cell(gsize, sc, x);
  out[11] = rayjoin::dev::calculate_cell(gsize, sc, y);
  out[12] = xs.x.denom();
  out[13] = xs.y.denom();
  out[14] = (int64_t) sizeof(xs);
  return 1;
}

int ref_cell_of_int(int gsize, int64_t v) {
  const RefScaling sc;  // default constructor: the internal range only (scaling.h:43-54)
  return rayjoin::dev::calculate_cell(gsize, sc, v);
}
int ref_cell_of_double(int gsize, double v) {
  const RefScaling sc;  // default constructor: the internal range only (scaling.h:43-54)
  return rayjoin::dev::calculate_cell(gsize, sc, v);
}

// Scaling(bb).ScaleX/ScaleY over n points (src/map/scaling.h:56-93)
void ref_scale_points(const double* bb, const double* xy, uint64_t n, int64_t* out) {
  const RefScaling s = make_scaling(bb);
  for (uint64_t i = 0; i < n; i++) {
    out[2 * i] = s.ScaleX(xy[2 * i]);
    out[2 * i + 1] = s.ScaleY(xy[2 * i + 1]);
  }
}
// Scaling(bb).UnscaleX/UnscaleY (src/map/scaling.h:95-106)
void ref_unscale_points(const double* bb, const int64_t* xy, uint64_t n, double* out) {
  const RefScaling s = make_scaling(bb);
  for (uint64_t i = 0; i < n; i++) {
    out[2 * i] = s.UnscaleX(xy[2 * i]);
    out[2 * i + 1] = s.UnscaleY(xy[2 * i + 1]);
  }
}
// {internal_min, internal_max, internal_range, sizeof(Scaling)}
void ref_scaling_consts(int64_t* out) {
  const RefScaling s;
  out[0] = s.get_internal_min();
  out[1] = s.get_internal_max();
  out[2] = s.get_internal_range();
  out[3] = (int64_t) sizeof(RefScaling);
}
}
