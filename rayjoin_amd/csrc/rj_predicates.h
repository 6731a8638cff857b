// rj_predicates.h -- exact-arithmetic LSI / PIP predicates, shared by every HIP kernel.
//
// Numerically identical to the reference (file:line relative to /root/reference):
//   lsi_test()      dev::intersect_test, boolean form            src/algo/lsi.h:29-103
//   lsi_point()     intersection point + narrowing store         src/algo/lsi.h:107-143,
//                                                                 src/util/rational.h:87-90,190-203,335-343
//   pip_eval_y()/pip_slope()/pip_better()  "lowest edge above point"  src/algo/pip.h:31-96
//                                                                 == src/app/pip_lbvh.h:57-123
// The 80-byte dev::Edge (src/map/map.h:20-46) is never stored: what the predicates need of a, b, c
// is rebuilt from the two endpoints (map.h:216-226) -- for the two query predicates without ever
// forming c (see edge_side and pip_eval_y).
//
// RJ_HD lets tests compile these functions for the host (tests/hosttwin/twin.cc) -- a test-only twin,
// never a fallback: the product path is HIP only.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define RJ_HD __host__ __device__ __forceinline__
#else
#define RJ_HD inline
#endif

namespace rj {

typedef __int128 i128;
typedef unsigned __int128 u128;

struct Seg {  // scaled endpoints, 32 bytes
  int64_t x1, y1, x2, y2;
};

struct Eqn {  // a*x + b*y + c = 0 with b >= 0 (map.h:216-226)
  i128 a, b, c;
};

RJ_HD Eqn make_eqn(const Seg& s) {
  Eqn e;
  int64_t a = s.y1 - s.y2;
  int64_t b = s.x2 - s.x1;
  e.c = -(i128) s.x1 * a - (i128) s.y1 * b;
  if (b < 0) {
    a = -a;
    b = -b;
    e.c = -e.c;
  }
  e.a = a;
  e.b = b;
  return e;
}

// Sign of p.x*e.a + p.y*e.b + e.c (lsi.h:32-33) for the edge equation of segment e, WITHOUT
// building the equation: c = -x1 a - y1 b, so the value is a (px - x1) + b (py - y1) exactly
// (|a|, |b|, the differences < 2^47: each product < 2^94, no overflow), and normalising the
// equation to b >= 0 (map.h:216-226) only flips its sign when b < 0.  Two 128-bit products and a
// handful of registers per evaluation instead of three products and a live 128-bit a, b, c.
RJ_HD int edge_side(const Seg& e, int64_t px, int64_t py) {
  const int64_t a = e.y1 - e.y2, b = e.x2 - e.x1;
  const i128 v = (i128) a * (px - e.x1) + (i128) b * (py - e.y1);
  const int sg = (v > 0) - (v < 0);
  return b < 0 ? -sg : sg;
}
RJ_HD int sign64(int64_t v) { return (v > 0) - (v < 0); }

// intersect_test(e1 = map-0 edge, e2 = map-1 edge); the operand order is part of the semantics
// (simulation of simplicity, lsi.h:41-87).  Only signs are ever used (lsi.h:44-87).
RJ_HD bool lsi_test(const Seg& s1, const Seg& s2) {
  const int64_t a2 = s2.y1 - s2.y2, b2 = s2.x2 - s2.x1;  // normalised (a, b) of e2: negate if b < 0
  const int sa2 = b2 < 0 ? -sign64(a2) : sign64(a2), sb2 = b2 != 0;
  int u1 = edge_side(s2, s1.x1, s1.y1);  // e1_p1 against e2
  int u2 = edge_side(s2, s1.x2, s1.y2);
  if (u1 == 0) u1 = -sa2;
  if (u1 == 0) u1 = -sb2;
  if (u1 == 0) return false;
  if (u2 == 0) u2 = -sa2;
  if (u2 == 0) u2 = -sb2;
  if (u2 == 0) return false;
  if (u1 == u2) return false;
  const int64_t a1 = s1.y1 - s1.y2, b1 = s1.x2 - s1.x1;
  const int sa1 = b1 < 0 ? -sign64(a1) : sign64(a1), sb1 = b1 != 0;
  int v1 = edge_side(s1, s2.x1, s2.y1);  // e2_p1 against e1
  int v2 = edge_side(s1, s2.x2, s2.y2);
  if (v1 == 0) v1 = sa1;
  if (v1 == 0) v1 = sb1;
  if (v1 == 0) return false;
  if (v2 == 0) v2 = sa1;
  if (v2 == 0) v2 = sb1;
  if (v2 == 0) return false;
  if (v1 == v2) return false;
  if ((s1.x1 == s2.x1 && s1.y1 == s2.y1 && s1.x2 == s2.x2 && s1.y2 == s2.y2) ||
      (s1.x1 == s2.x2 && s1.y1 == s2.y2 && s1.x2 == s2.x1 && s1.y2 == s2.y1))
    return false;
  return true;
}

// ---- tcb::rational<__int128> pieces ------------------------------------------------------
struct Rat {
  i128 num, den;
};

RJ_HD u128 uabs128(i128 v) { return v < 0 ? (u128) 0 - (u128) v : (u128) v; }

RJ_HD int ctz128(u128 v) {
  uint64_t lo = (uint64_t) v;
  if (lo) return __builtin_ctzll(lo);
  return 64 + __builtin_ctzll((uint64_t) (v >> 64));
}

RJ_HD int clz128(u128 v) {  // v != 0
  const uint64_t hi = (uint64_t) (v >> 64);
  return hi ? __builtin_clzll(hi) : 64 + __builtin_clzll((uint64_t) v);
}

// a mod b (b != 0): one Euclid step.  When the quotient is short (< 2^46: always, for the
// numerators and denominators of map-like segments) it is estimated in double -- the estimate is
// within 1 of the true quotient, and the remainder is then fixed up exactly in integers; otherwise
// the generic 128-bit remainder.
RJ_HD u128 mod128(u128 a, u128 b) {
  if (a < b) return a;
  const int la = 128 - clz128(a), lb = 128 - clz128(b);
  if (la <= 126 && la - lb <= 45) {
    const double da = (double) (uint64_t) (a >> 64) * 18446744073709551616.0 + (double) (uint64_t) a;
    const double db = (double) (uint64_t) (b >> 64) * 18446744073709551616.0 + (double) (uint64_t) b;
    const uint64_t q = (uint64_t) (da / db);  // |q - floor(a / b)| <= 1: relative error < 2^-50 on a quotient < 2^46
    u128 p = (u128) q * b;                    // <= a + b < 2^127
    if (p > a) p -= b;
    u128 r = a - p;
    if (r >= b) r -= b;
    return r;
  }
  return a % b;
}

RJ_HD uint64_t gcd64(uint64_t a, uint64_t b) {  // binary GCD, both non-zero
  const int sh = __builtin_ctzll(a | b);
  a >>= __builtin_ctzll(a);
  do {
    b >>= __builtin_ctzll(b);
    if (a > b) {
      const uint64_t t = a;
      a = b;
      b = t;
    }
    b -= a;
  } while (b != 0);
  return a << sh;
}

// |gcd(a, b)| on magnitudes -- the value the reference's Euclid loop (rational.h:36-43) followed by
// abs (rational.h:200) yields.  Euclid steps (mod128) until both operands fit 64 bits -- one to
// three for a ~2^110 numerator against a ~2^65 denominator -- then a 64-bit binary GCD: about a
// quarter of the instructions of a 128-bit binary GCD from the start, and no 128-bit division.
RJ_HD u128 gcd_mag(u128 a, u128 b) {
  if (a < b) {
    const u128 t = a;
    a = b;
    b = t;
  }
  while (b != 0 && (uint64_t) (a >> 64) != 0) {  // invariant a >= b
    const u128 r = mod128(a, b);
    a = b;
    b = r;
  }
  if (b == 0) return a;
  return (u128) gcd64((uint64_t) a, (uint64_t) b);
}

// n / g for g | n, without a division: shift out g's power of two, multiply by the inverse of its
// odd part modulo 2^128 (exact for exact quotients)
RJ_HD u128 inv_odd128(u128 o) {
  u128 x = o;  // o * o = 1 (mod 8): 3 correct bits, doubled by every Newton step
  for (int i = 0; i < 6; i++) x *= (u128) 2 - o * x;
  return x;
}

// rational(num, denom) -> simplify()  (rational.h:87-90,198-203):
//   g = |gcd|; num = sign(den)*num / g; den = |den| / g
RJ_HD Rat rat_make(i128 num, i128 den) {
  Rat r;
  const u128 g = gcd_mag(uabs128(num), uabs128(den));
  if (g == 0) {
    r.num = num;
    r.den = den;
    return r;
  }
  const i128 sn = den < 0 ? (i128) ((u128) 0 - (u128) num) : num;
  u128 qn = uabs128(sn), qd = uabs128(den);
  if (g != 1) {
    const int sh = ctz128(g);
    const u128 inv = inv_odd128(g >> sh);
    qn = (qn >> sh) * inv;
    qd = (qd >> sh) * inv;
  }
  r.num = sn < 0 ? (i128) ((u128) 0 - qn) : (i128) qn;
  r.den = (i128) qd;
  return r;
}

RJ_HD double rat_to_double(const Rat& r) { return (double) r.num / (double) r.den; }

template <typename T>
RJ_HD T min4(T a, T b, T c, T d) {
  T m = a < b ? a : b, n = c < d ? c : d;
  return m < n ? m : n;
}
template <typename T>
RJ_HD T max4(T a, T b, T c, T d) {
  T m = a > b ? a : b, n = c > d ? c : d;
  return m > n ? m : n;
}

// lsi.h:117-141: exact point of a predicate-true pair, clamped to the 4 endpoints' range.
// int128 products wrap exactly like the reference's (two's complement).
RJ_HD void lsi_point(const Seg& s1, const Eqn& e1, const Seg& s2, const Eqn& e2, Rat* ox, Rat* oy) {
  u128 den = (u128) e1.a * (u128) e2.b - (u128) e2.a * (u128) e1.b;
  u128 nx = (u128) e2.c * (u128) e1.b - (u128) e1.c * (u128) e2.b;
  u128 ny = (u128) e2.a * (u128) e1.c - (u128) e1.a * (u128) e2.c;
  Rat x = rat_make((i128) nx, (i128) den), y = rat_make((i128) ny, (i128) den);
  int64_t t;
  t = min4(s1.x1, s1.x2, s2.x1, s2.x2);
  if (x.num < (i128) ((u128) (i128) t * (u128) x.den)) { x.num = t; x.den = 1; }
  t = max4(s1.x1, s1.x2, s2.x1, s2.x2);
  if ((i128) ((u128) (i128) t * (u128) x.den) < x.num) { x.num = t; x.den = 1; }
  t = min4(s1.y1, s1.y2, s2.y1, s2.y2);
  if (y.num < (i128) ((u128) (i128) t * (u128) y.den)) { y.num = t; y.den = 1; }
  t = max4(s1.y1, s1.y2, s2.y1, s2.y2);
  if ((i128) ((u128) (i128) t * (u128) y.den) < y.num) { y.num = t; y.den = 1; }
  *ox = x;
  *oy = y;
}

// ---- the narrowing store without the gcd ------------------------------------------------
// What is stored of an intersection is (int64_t) ((double) num' / (double) den') of the SIMPLIFIED, clamped rational
// (lsi.h:117-143, rational.h:335-343).  The gcd only matters through the rounding of num' and den': with
// v = num / den, q = floor(v), r = num - q den (den > 0 after the sign normalisation of rational.h:198-203):
//   * v < t_min or v > t_max (integers; v < t <=> q < t, v > t <=> q > t or q == t and r > 0): the clamp value, exactly;
//   * r == 0: the gcd is den, the simplified rational is q / 1, the store is q;
//   * otherwise the double quotient Q of the simplified rational is within |v| 2^-51 of v whatever the gcd was (three
//     roundings of 2^-53 each), so when v keeps that distance from q and from q + 1 -- r and den - r both above
//     |num| 2^-51 -- Q lies strictly between them and truncates toward zero to q (v > 0) or q + 1 (v < 0).
// Only a coordinate that comes closer to an integer than that without being one (1-2 % of them at 2^44) needs
// num' and den' themselves: the caller takes lsi_point for those.  No 128-bit gcd, no modular inverse, one
// double-estimated quotient per coordinate.  false = not decided here (also: den == 0, magnitudes the short
// quotient or the clamp's exact comparison cannot take).
RJ_HD bool lsi_coord_fast(i128 num, i128 den, int64_t t_min, int64_t t_max, int64_t* out) {
  if (den == 0) return false;
  const bool flip = den < 0;
  const u128 D = uabs128(den);
  const i128 sn = flip ? (i128) ((u128) 0 - (u128) num) : num;
  const u128 a = uabs128(sn);
  if ((uint64_t) (D >> 80) != 0 || (uint64_t) (a >> 126) != 0) return false;  // (t * den' cannot wrap below 2^80)
  uint64_t qa = 0;
  u128 ra = a;
  if (a >= D) {
    const int la = 128 - clz128(a), lb = 128 - clz128(D);
    if (la - lb > 47) return false;  // quotient < 2^48: the estimate below (relative error < 2^-50) is within 1 of it
    const double da = (double) (uint64_t) (a >> 64) * 18446744073709551616.0 + (double) (uint64_t) a;
    const double db = (double) (uint64_t) (D >> 64) * 18446744073709551616.0 + (double) (uint64_t) D;
    qa = (uint64_t) (da / db);  // within 1 of floor(a / D), as in mod128
    u128 p = (u128) qa * D;
    if (p > a) {
      p -= D;
      qa--;
    }
    ra = a - p;
    if (ra >= D) {
      ra -= D;
      qa++;
    }
  }
  // floor and remainder of the signed value
  int64_t q = (int64_t) qa;
  u128 r = ra;
  if (sn < 0) {
    q = ra == 0 ? -q : -q - 1;
    r = ra == 0 ? (u128) 0 : D - ra;
  }
  if (q < t_min) {
    *out = t_min;
    return true;
  }
  if (q > t_max || (q == t_max && r != 0)) {
    *out = t_max;
    return true;
  }
  if (r == 0) {
    *out = q;
    return true;
  }
  const u128 m = (a >> 51) + 1;
  if (r < m || D - r < m) return false;
  *out = sn < 0 ? q + 1 : q;
  return true;
}

// both stored coordinates of a predicate-true pair, or false (then: lsi_point + rat_to_double)
RJ_HD bool lsi_stored_fast(const Seg& s1, const Seg& s2, int64_t* ox, int64_t* oy) {
  // den, nx, ny of lsi.h:117-119 with the 64-bit a, b of map.h:216-226 kept 64-bit (the same values modulo 2^128,
  // a quarter of the multiplications of the all-128-bit form)
  int64_t a1 = s1.y1 - s1.y2, b1 = s1.x2 - s1.x1, a2 = s2.y1 - s2.y2, b2 = s2.x2 - s2.x1;
  u128 c1 = (u128) 0 - (u128) ((i128) s1.x1 * a1) - (u128) ((i128) s1.y1 * b1);
  u128 c2 = (u128) 0 - (u128) ((i128) s2.x1 * a2) - (u128) ((i128) s2.y1 * b2);
  if (b1 < 0) { a1 = -a1; b1 = -b1; c1 = (u128) 0 - c1; }
  if (b2 < 0) { a2 = -a2; b2 = -b2; c2 = (u128) 0 - c2; }
  const i128 den = (i128) a1 * b2 - (i128) a2 * b1;
  const i128 nx = (i128) (c2 * (u128) (i128) b1 - c1 * (u128) (i128) b2);
  const i128 ny = (i128) ((u128) (i128) a2 * c1 - (u128) (i128) a1 * c2);
  return lsi_coord_fast(nx, den, min4(s1.x1, s1.x2, s2.x1, s2.x2), max4(s1.x1, s1.x2, s2.x1, s2.x2), ox) &&
         lsi_coord_fast(ny, den, min4(s1.y1, s1.y2, s2.y1, s2.y2), max4(s1.y1, s1.y2, s2.y1, s2.y2), oy);
}

// ---- PIP ------------------------------------------------------------------------------
// (double) of a 128-bit integer, round-to-nearest-even like the compiler's conversion (what the
// reference's `(double) int128` is), but through the top 64 bits + a sticky bit instead of the
// generic 128-bit expansion: the u64 -> double conversion then does the one rounding (bit 0 only
// decides exact ties, it is far below the 53 kept bits) and the scaling by a power of two is exact.
// About 40 VALU instructions less per PIP evaluation on gfx950; checked bit for bit against the
// compiler's conversion by tests/test_device_predicates_on_host.py.
RJ_HD double i128_to_double(i128 v) {
  const bool neg = v < 0;
  const u128 m = neg ? (u128) 0 - (u128) v : (u128) v;
  const uint64_t hi = (uint64_t) (m >> 64), lo = (uint64_t) m;
  double d;
  if (hi == 0) {
    d = (double) lo;
  } else {
    const int sh = 64 - __builtin_clzll(hi);  // bits of hi in use: 1..64
    uint64_t top = sh == 64 ? hi : (hi << (64 - sh)) | (lo >> sh);
    const uint64_t rest = sh == 64 ? lo : lo << (64 - sh);
    top |= rest != 0;
    d = __builtin_ldexp((double) top, sh);
  }
  return neg ? -d : d;
}

// One (point, base edge) evaluation, pip.h:36-71.  Returns false when the edge is rejected
// outright (x range / point above edge); otherwise *yy = xsect_y.
// The reference divides double(-a px - c) by double(b) with (a, b, c) normalised to b >= 0.
// -a px - c = a (x1 - px) + b y1 exactly (c = -x1 a - y1 b; every term < 2^95), so two 128-bit
// products instead of three; and normalisation negates numerator and denominator together, which
// IEEE division cannot see -- so the raw (a, b) are used for the quotient, the normalised ones
// only for the simulation-of-simplicity substitutes.  b == 0 (vertical edge) never gets here: its
// x range is one point, which the range rule rejects.
RJ_HD bool pip_eval_y(const Seg& s, int64_t px, int64_t py, int query_map_id, double* yy) {
  const int64_t x_min = s.x1 < s.x2 ? s.x1 : s.x2;
  const int64_t x_max = s.x1 < s.x2 ? s.x2 : s.x1;
  if (px < x_min || px > x_max || px == (query_map_id == 0 ? x_min : x_max)) return false;
  const int64_t a = s.y1 - s.y2, b = s.x2 - s.x1;
  const i128 num = (i128) a * (s.x1 - px) + (i128) b * s.y1;
  const double xsect_y = i128_to_double(num) / (double) b;
  double diff_y = (double) py - xsect_y;
  if (diff_y == 0) {
    const int64_t an = b < 0 ? -a : a, bn = b < 0 ? -b : b;
    diff_y = (double) (query_map_id == 0 ? -an : an);
    if (diff_y == 0) diff_y = (double) (query_map_id == 0 ? -bn : bn);
  }
  if (diff_y > 0) return false;
  *yy = xsect_y;
  return true;
}

// (double) a / (double) b of the normalised edge equation: only needed to break a tie in xsect_y
RJ_HD double pip_slope(const Seg& s) {
  const int64_t a = s.y1 - s.y2, b = s.x2 - s.x1;
  return (double) (b < 0 ? -a : a) / (double) (b < 0 ? -b : b);
}

// Does candidate (yy, slope, eid) replace best (byy, bslope, beid)?  pip.h:73-95 made a total
// order: the reference's full ties (equal yy and slope) keep the first visited edge for
// query map 1 and the last visited for query map 0; with edges visited in ascending eid (what
// the oracle does) that is "smaller eid" for q==1 and "larger eid" for q==0.
RJ_HD bool pip_better(double yy, double slope, uint32_t eid, double byy, double bslope,
                      uint32_t beid, int query_map_id) {
  if (yy > byy) return false;
  if (yy < byy) return true;
  if (beid == 0xFFFFFFFFu) return true;  // no best yet (byy == +inf == yy cannot happen, but be safe)
  if (query_map_id) {
    if (slope > bslope) return true;
    if (slope < bslope) return false;
    return eid < beid;
  }
  if (slope > bslope) return false;
  if (slope < bslope) return true;
  return eid > beid;
}

}  // namespace rj
