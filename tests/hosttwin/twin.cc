// Test-only host build of the DEVICE predicate source (rayjoin_amd/csrc/rj_predicates.h): lets the
// CPU test suite check the exact functions the HIP kernels inline against the reference's golden
// vectors and the oracle.  Never linked into the product; the product path is HIP only.
#include <cstdint>

#include "rj_predicates.h"

using namespace rj;

extern "C" {

int twin_lsi_test(const int64_t* s1, const int64_t* s2) {
  const Seg a = {s1[0], s1[1], s1[2], s1[3]}, b = {s2[0], s2[1], s2[2], s2[3]};
  return lsi_test(a, b) ? 1 : 0;
}

// out[0..1] = the narrowing store of the intersection point (lsi.h:107-143), valid when the test is true
void twin_lsi_stored(const int64_t* s1, const int64_t* s2, int64_t* out) {
  const Seg a = {s1[0], s1[1], s1[2], s1[3]}, b = {s2[0], s2[1], s2[2], s2[3]};
  Rat x, y;
  lsi_point(a, make_eqn(a), b, make_eqn(b), &x, &y);
  out[0] = (int64_t) rat_to_double(x);
  out[1] = (int64_t) rat_to_double(y);
}

// the gcd-free store: returns 1 and fills out[0..1] when it decides, 0 when the caller must take twin_lsi_stored
int twin_lsi_stored_fast(const int64_t* s1, const int64_t* s2, int64_t* out) {
  const Seg a = {s1[0], s1[1], s1[2], s1[3]}, b = {s2[0], s2[1], s2[2], s2[3]};
  return lsi_stored_fast(a, b, &out[0], &out[1]) ? 1 : 0;
}

// one coordinate: (num, den) as 128-bit halves, clamp range -> 1 and *fast when decided; *slow = what
// rat_make + the clamp of lsi.h:121-141 + the narrowing store give
int twin_lsi_coord(int64_t nh, uint64_t nl, int64_t dh, uint64_t dl, int64_t t_min, int64_t t_max, int64_t* fast, int64_t* slow) {
  const i128 num = (i128) (((u128) (uint64_t) nh << 64) | nl), den = (i128) (((u128) (uint64_t) dh << 64) | dl);
  Rat x = rat_make(num, den);
  if (x.num < (i128) ((u128) (i128) t_min * (u128) x.den)) { x.num = t_min; x.den = 1; }
  if ((i128) ((u128) (i128) t_max * (u128) x.den) < x.num) { x.num = t_max; x.den = 1; }
  *slow = (int64_t) rat_to_double(x);
  return lsi_coord_fast(num, den, t_min, t_max, fast) ? 1 : 0;
}

// bulk form for the fuzz test: n pairs of segments (8 int64 each); counts[0] = predicate-true pairs,
// counts[1] = decided by the fast path, counts[2] = decided differently from the slow path (must stay 0)
void twin_lsi_stored_fuzz(const int64_t* segs, uint64_t n, uint64_t* counts) {
  counts[0] = counts[1] = counts[2] = 0;
  for (uint64_t k = 0; k < n; k++) {
    const int64_t* s = segs + 8 * k;
    const Seg a = {s[0], s[1], s[2], s[3]}, b = {s[4], s[5], s[6], s[7]};
    if (!lsi_test(a, b)) continue;
    counts[0]++;
    int64_t fx, fy;
    if (!lsi_stored_fast(a, b, &fx, &fy)) continue;
    counts[1]++;
    Rat x, y;
    lsi_point(a, make_eqn(a), b, make_eqn(b), &x, &y);
    if (fx != (int64_t) rat_to_double(x) || fy != (int64_t) rat_to_double(y)) counts[2]++;
  }
}

// returns 1 when the edge is a candidate; *yy = xsect_y, *slope = (double) a / (double) b
int twin_pip_eval(const int64_t* s, int64_t px, int64_t py, int query_map_id, double* yy, double* slope) {
  const Seg e = {s[0], s[1], s[2], s[3]};
  *slope = pip_slope(e);
  return pip_eval_y(e, px, py, query_map_id, yy) ? 1 : 0;
}

// the custom 128-bit -> double conversion next to the compiler's
void twin_i128_to_double(int64_t hi, uint64_t lo, double* custom, double* compiler) {
  const i128 v = (i128) (((u128) (uint64_t) hi << 64) | lo);
  *custom = i128_to_double(v);
  *compiler = (double) v;
}

// rational<__int128>(num, den).simplify() (rational.h:87-90,198-203): out = {num hi, num lo, den hi, den lo}
void twin_rat_make(int64_t nh, uint64_t nl, int64_t dh, uint64_t dl, uint64_t* out) {
  const Rat r = rat_make((i128) (((u128) (uint64_t) nh << 64) | nl), (i128) (((u128) (uint64_t) dh << 64) | dl));
  out[0] = (uint64_t) ((u128) r.num >> 64);
  out[1] = (uint64_t) r.num;
  out[2] = (uint64_t) ((u128) r.den >> 64);
  out[3] = (uint64_t) r.den;
}

int twin_pip_better(double yy, double slope, uint32_t eid, double byy, double bslope, uint32_t beid, int q) {
  return pip_better(yy, slope, eid, byy, bslope, beid, q) ? 1 : 0;
}
}
