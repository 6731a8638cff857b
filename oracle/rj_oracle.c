/*
 * rj_oracle.c -- CPU restatement of RayJoin's LSI / PIP query path (-mode=grid semantics).
 *
 * THIS FILE IS TEST INFRASTRUCTURE.  It is the parity oracle and the timed CPU baseline
 * (bench.py `cpu_baseline`, kind "port").  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The product path (rayjoin_amd/csrc) never
 * links, includes or calls anything in this directory.
 *
 * Parity pin: the predicate-level functions (intersect_test, intersection point,
 * rational arithmetic, calculate_cell) and Scaling are checked against vectors produced by the
 * reference's own headers compiled on the host (oracle/ref/, tests/golden/lsi_ref_vectors.json,
 * tests/golden/scaling_ref_vectors.json) and against the known answers recorded in SURVEY.md 8c.  The PIP predicate and all
 * dataset-level behaviour have no reference-owned fixture (the test/dataset files are missing from
 * the snapshot): for those rows parity is UNPINNED by the reference and rests on this
 * restatement cross-checked by brute force (see DESIGN.md "Oracle").
 *
 * Every function cites the reference file:line (relative to /root/reference) it follows.
 * Build: gcc -O3 -fwrapv -ffp-contract=off -fopenmp (see oracle/Makefile).
 *   -fwrapv            : the reference's __int128 products silently wrap (SURVEY 7 hard part 1)
 *   -ffp-contract=off  : no FMA fusion in scaling / PIP double arithmetic (SURVEY 7 hard part 7)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef __int128 i128;
typedef unsigned __int128 u128;

#define RJO_MISS 0xFFFFFFFFu /* static_cast<index_t>(DONTKNOW), src/app/pip_lbvh.h:44 */

/* ------------------------------------------------------------------------------------------
 * Scaling -- src/map/scaling.h:32-136  (Scaling<double,int64_t,17>)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int64_t imax, imin, irange;
  double rx, ry, rrx, rry, dx, dy, ddx, ddy;
} rjo_scaling;

/* scaling.h:43-74; bbox margin = SCALING_BOUNDING_BOX_MARGIN (1), src/config.h:4 */
void rjo_scaling_init(rjo_scaling* s, double min_x, double min_y, double max_x, double max_y) {
  s->imax = INT64_MAX >> 17; /* scaling.h:44 */
  s->imin = INT64_MIN >> 17; /* scaling.h:45 */
  s->irange = s->imax - s->imin;
  double mxx = max_x + 1, mnx = min_x - 1, mxy = max_y + 1, mny = min_y - 1; /* :57-60 */
  s->rx = (double) s->irange / (mxx - mnx);                                  /* :62 */
  s->ry = (double) s->irange / (mxy - mny);
  s->rrx = 1 / s->rx;
  s->rry = 1 / s->ry;
  /* :67-70 -- (internal_max_ + internal_min_) is an int64 sum (= -1) */
  s->dx = 0.5 * ((double) (s->imax + s->imin) - (mxx + mnx) * s->rx);
  s->dy = 0.5 * ((double) (s->imax + s->imin) - (mxy + mny) * s->ry);
  s->ddx = 0.5 * ((mxx + mnx) - (double) (s->imax + s->imin) * s->rrx);
  s->ddy = 0.5 * ((mxy + mny) - (double) (s->imax + s->imin) * s->rry);
}

/* scaling.h:79-93: internal = (int64)(x * r + delta), mul then add, truncation toward 0 */
void rjo_scale_points(const rjo_scaling* s, const double* xy, size_t n, int64_t* out) {
  for (size_t i = 0; i < n; i++) {
    double vx = xy[2 * i] * s->rx;
    double vy = xy[2 * i + 1] * s->ry;
    out[2 * i] = (int64_t) (vx + s->dx);
    out[2 * i + 1] = (int64_t) (vy + s->dy);
  }
}

/* scaling.h:100-106 */
void rjo_unscale_points(const rjo_scaling* s, const int64_t* xy, size_t n, double* out) {
  for (size_t i = 0; i < n; i++) {
    double vx = (double) xy[2 * i] * s->rrx;
    double vy = (double) xy[2 * i + 1] * s->rry;
    out[2 * i] = vx + s->ddx;
    out[2 * i + 1] = vy + s->ddy;
  }
}

/* ------------------------------------------------------------------------------------------
 * Map model -- src/map/map.h:20-46 (EdgeEquation / Edge), :187-230 (chain -> edge expansion)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int64_t x, y;
} rjo_pt;

typedef struct {
  i128 a, b, c; /* a*x + b*y + c = 0, b >= 0 */
  uint32_t eid, p1, p2, left, right;
} rjo_edge;

typedef struct {
  size_t np, nc, ne;
  rjo_pt* pts;
  rjo_edge* edges;
} rjo_map;

/* map.h:216-226 */
static void edge_eqn(rjo_edge* e, rjo_pt p1, rjo_pt p2) {
  e->a = (i128) (p1.y - p2.y);
  e->b = (i128) (p2.x - p1.x);
  e->c = -(i128) p1.x * e->a - (i128) p1.y * e->b;
  if (e->b < 0) {
    e->a = -e->a;
    e->b = -e->b;
    e->c = -e->c;
  }
}

/* map.h:187-230: chain i owns points [row_index[i], row_index[i+1]); eid = p_idx - ichain */
rjo_map* rjo_map_create(const int64_t* xy, size_t np, const uint32_t* row_index,
                        const int64_t* left, const int64_t* right, size_t nc) {
  rjo_map* m = (rjo_map*) calloc(1, sizeof(rjo_map));
  m->np = np;
  m->nc = nc;
  m->ne = np - nc; /* map.h:165 */
  m->pts = (rjo_pt*) malloc(sizeof(rjo_pt) * (np ? np : 1));
  m->edges = (rjo_edge*) aligned_alloc(16, sizeof(rjo_edge) * (m->ne ? m->ne : 1));
  for (size_t i = 0; i < np; i++) {
    m->pts[i].x = xy[2 * i];
    m->pts[i].y = xy[2 * i + 1];
  }
  for (size_t ic = 0; ic < nc; ic++) {
    for (uint32_t p = row_index[ic]; p + 1 < row_index[ic + 1]; p++) {
      rjo_edge* e = &m->edges[p - ic];
      e->eid = (uint32_t) (p - ic);
      e->p1 = p;
      e->p2 = p + 1;
      e->left = (uint32_t) left[ic]; /* int64 -> index_t truncation, map.h:45,206-207 */
      e->right = (uint32_t) right[ic];
      edge_eqn(e, m->pts[p], m->pts[p + 1]);
    }
  }
  return m;
}

/* free-standing segments (run_query.cu:102-144 GenerateLSIQueries: edge i = points 2i,2i+1) */
rjo_map* rjo_map_create_segments(const int64_t* xy, size_t ne) {
  rjo_map* m = (rjo_map*) calloc(1, sizeof(rjo_map));
  m->np = 2 * ne;
  m->nc = ne;
  m->ne = ne;
  m->pts = (rjo_pt*) malloc(sizeof(rjo_pt) * (m->np ? m->np : 1));
  m->edges = (rjo_edge*) aligned_alloc(16, sizeof(rjo_edge) * (ne ? ne : 1));
  for (size_t i = 0; i < m->np; i++) {
    m->pts[i].x = xy[2 * i];
    m->pts[i].y = xy[2 * i + 1];
  }
  for (size_t i = 0; i < ne; i++) {
    rjo_edge* e = &m->edges[i];
    e->eid = (uint32_t) i;
    e->p1 = (uint32_t) (2 * i);
    e->p2 = (uint32_t) (2 * i + 1);
    e->left = e->right = 0;
    edge_eqn(e, m->pts[e->p1], m->pts[e->p2]);
  }
  return m;
}

void rjo_map_free(rjo_map* m) {
  if (!m)
    return;
  free(m->pts);
  free(m->edges);
  free(m);
}

size_t rjo_map_num_edges(const rjo_map* m) { return m->ne; }
size_t rjo_map_num_points(const rjo_map* m) { return m->np; }

/* out[0..5] = a,b,c as (lo,hi) int64 words; out[6..10] = eid,p1,p2,left,right */
void rjo_map_get_edge(const rjo_map* m, size_t eid, int64_t* out) {
  const rjo_edge* e = &m->edges[eid];
  out[0] = (int64_t) (u128) e->a;
  out[1] = (int64_t) ((u128) e->a >> 64);
  out[2] = (int64_t) (u128) e->b;
  out[3] = (int64_t) ((u128) e->b >> 64);
  out[4] = (int64_t) (u128) e->c;
  out[5] = (int64_t) ((u128) e->c >> 64);
  out[6] = e->eid;
  out[7] = e->p1;
  out[8] = e->p2;
  out[9] = e->left;
  out[10] = e->right;
}

/* map.h:79-87 get_face_id; miss -> EXTERIOR_FACE_ID (map_overlay_lbvh.h:96-104, config.h:8) */
static int32_t face_id_of(const rjo_map* m, uint32_t eid) {
  if (eid == RJO_MISS)
    return 0;
  const rjo_edge* e = &m->edges[eid];
  return (int32_t) (m->pts[e->p1].x < m->pts[e->p2].x ? e->right : e->left);
}

void rjo_face_ids(const rjo_map* base, const uint32_t* eids, size_t n, int32_t* out) {
  for (size_t i = 0; i < n; i++)
    out[i] = face_id_of(base, eids[i]);
}

/* ------------------------------------------------------------------------------------------
 * tcb::rational<__int128> -- src/util/rational.h
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  i128 num, den;
} rat128;

/* rational.h:36-43 Euclid with C '%' (sign follows the dividend) */
static i128 gcd128(i128 a, i128 b) {
  while (b != 0) {
    i128 t = b;
    b = a % b;
    a = t;
  }
  return a;
}

/* rational.h:198-203 simplify(); rational(num,denom) ctor :87-90 */
static rat128 rat_make(i128 num, i128 den) {
  rat128 r;
  i128 g = gcd128(num, den);
  if (g < 0)
    g = -g;
  if (g == 0) { /* 0/0: unreachable for predicate-true pairs (the reference would trap) */
    r.num = num;
    r.den = den;
    return r;
  }
  i128 sgn = den < 0 ? -1 : 1;
  r.num = sgn * num / g;
  r.den = (den < 0 ? -den : den) / g;
  return r;
}

/* rational.h:190-192 operator double() */
static double rat_to_double(rat128 r) { return (double) r.num / (double) r.den; }

/* rational.h:335-343: rational < integer  <=>  num*1 < t*den (wrapping int128) */
static int rat_lt_int(rat128 r, int64_t t) { return r.num * (i128) 1 < (i128) t * r.den; }
/* integer < rational  <=>  t*den < num*1 */
static int int_lt_rat(int64_t t, rat128 r) { return (i128) t * r.den < r.num * (i128) 1; }

/* rational.h:390-398 operator-(rational, integer) -> rational{num*1 - t*den, den*1} (simplified) */
static rat128 rat_sub_int(rat128 r, int64_t t) {
  return rat_make(r.num * (i128) 1 - (i128) t * r.den, r.den * (i128) 1);
}

/* ------------------------------------------------------------------------------------------
 * LSI predicate -- src/algo/lsi.h:29-103 (boolean), :107-143 (intersection point)
 * ---------------------------------------------------------------------------------------- */
static inline i128 subedge(rjo_pt p, const rjo_edge* e) { /* lsi.h:32-33 */
  return (i128) p.x * e->a + (i128) p.y * e->b + e->c;
}

static int intersect_test(const rjo_edge* e1, rjo_pt e1_p1, rjo_pt e1_p2, const rjo_edge* e2,
                          rjo_pt e2_p1, rjo_pt e2_p2) {
  i128 e2_p1_agst_e1 = subedge(e2_p1, e1);
  i128 e2_p2_agst_e1 = subedge(e2_p2, e1);
  i128 e1_p1_agst_e2 = subedge(e1_p1, e2);
  i128 e1_p2_agst_e2 = subedge(e1_p2, e2);

  /* lsi.h:41-50: e1's endpoints on e2 -> perturb by -e2.a, then -e2.b */
  if (e1_p1_agst_e2 == 0)
    e1_p1_agst_e2 = -e2->a;
  if (e1_p1_agst_e2 == 0)
    e1_p1_agst_e2 = -e2->b;
  if (e1_p1_agst_e2 == 0)
    return 0;
  /* lsi.h:52-60 */
  if (e1_p2_agst_e2 == 0)
    e1_p2_agst_e2 = -e2->a;
  if (e1_p2_agst_e2 == 0)
    e1_p2_agst_e2 = -e2->b;
  if (e1_p2_agst_e2 == 0)
    return 0;
  /* lsi.h:64-67 */
  if ((e1_p1_agst_e2 > 0 && e1_p2_agst_e2 > 0) || (e1_p1_agst_e2 < 0 && e1_p2_agst_e2 < 0))
    return 0;
  /* lsi.h:70-87: e2's endpoints on e1 -> perturb by +e1.a, then +e1.b */
  if (e2_p1_agst_e1 == 0)
    e2_p1_agst_e1 = e1->a;
  if (e2_p1_agst_e1 == 0)
    e2_p1_agst_e1 = e1->b;
  if (e2_p1_agst_e1 == 0)
    return 0;
  if (e2_p2_agst_e1 == 0)
    e2_p2_agst_e1 = e1->a;
  if (e2_p2_agst_e1 == 0)
    e2_p2_agst_e1 = e1->b;
  if (e2_p2_agst_e1 == 0)
    return 0;
  /* lsi.h:88-91 */
  if ((e2_p1_agst_e1 > 0 && e2_p2_agst_e1 > 0) || (e2_p1_agst_e1 < 0 && e2_p2_agst_e1 < 0))
    return 0;
  /* lsi.h:97-100 identical (or reversed-identical) edges never intersect */
  if ((e1_p1.x == e2_p1.x && e1_p1.y == e2_p1.y && e1_p2.x == e2_p2.x && e1_p2.y == e2_p2.y) ||
      (e1_p1.x == e2_p2.x && e1_p1.y == e2_p2.y && e1_p2.x == e2_p1.x && e1_p2.y == e2_p1.y))
    return 0;
  return 1;
}

#define MIN2(a, b) ((a) < (b) ? (a) : (b))
#define MAX2(a, b) ((a) > (b) ? (a) : (b))
#define MIN4(a, b, c, d) (MIN2(MIN2(a, b), MIN2(c, d)))
#define MAX4(a, b, c, d) (MAX2(MAX2(a, b), MAX2(c, d)))

/* lsi.h:107-143 */
static int intersect_point(const rjo_edge* e1, rjo_pt e1_p1, rjo_pt e1_p2, const rjo_edge* e2,
                           rjo_pt e2_p1, rjo_pt e2_p2, rat128* xx, rat128* xy) {
  if (!intersect_test(e1, e1_p1, e1_p2, e2, e2_p1, e2_p2))
    return 0;
  i128 denom = e1->a * e2->b - e2->a * e1->b; /* :117 */
  i128 numx = e2->c * e1->b - e1->c * e2->b;  /* :118 (may wrap) */
  i128 numy = e2->a * e1->c - e1->a * e2->c;  /* :119 (may wrap) */
  rat128 x = rat_make(numx, denom), y = rat_make(numy, denom);
  int64_t t;
  t = MIN4(e1_p1.x, e1_p2.x, e2_p1.x, e2_p2.x); /* :124-127 */
  if (rat_lt_int(x, t)) {
    x.num = t;
    x.den = 1;
  }
  t = MAX4(e1_p1.x, e1_p2.x, e2_p1.x, e2_p2.x); /* :129-132 */
  if (int_lt_rat(t, x)) {
    x.num = t;
    x.den = 1;
  }
  t = MIN4(e1_p1.y, e1_p2.y, e2_p1.y, e2_p2.y); /* :134-137 */
  if (rat_lt_int(y, t)) {
    y.num = t;
    y.den = 1;
  }
  t = MAX4(e1_p1.y, e1_p2.y, e2_p1.y, e2_p2.y); /* :138-141 */
  if (int_lt_rat(t, y)) {
    y.num = t;
    y.den = 1;
  }
  *xx = x;
  *xy = y;
  return 1;
}

/* dev::Intersection<int64_t> -- lsi.h:10-26; 48 bytes.
 * Storing rational<int128> into rational<int64> goes through operator double() and the
 * converting constructor: num = (int64)((double)num/(double)den), denom = 1 (SURVEY fact 7;
 * rational.h:84-85,190-192). */
typedef struct {
  int64_t x_num, x_den, y_num, y_den;
  uint32_t eid[2];
  int32_t mid_point_polygon_id;
  int32_t _pad;
} rjo_xsect;

static void store_xsect(rjo_xsect* o, rat128 x, rat128 y, uint32_t eid0, uint32_t eid1) {
  o->x_num = (int64_t) rat_to_double(x);
  o->x_den = 1;
  o->y_num = (int64_t) rat_to_double(y);
  o->y_den = 1;
  o->eid[0] = eid0;
  o->eid[1] = eid1;
  o->mid_point_polygon_id = -1; /* DONTKNOW, lsi.h:25 */
  o->_pad = 0;
}

/* ------------------------------------------------------------------------------------------
 * calculate_cell -- src/grid/cell.h:16-22
 * ---------------------------------------------------------------------------------------- */
#define RJO_IMIN (INT64_MIN >> 17)
#define RJO_IRANGE ((INT64_MAX >> 17) - (INT64_MIN >> 17))

static inline double cell_scale(int gsize) { return (double) gsize / (double) RJO_IRANGE * 0.999; }
static inline int cell_of_int(int gsize, int64_t v) {
  return (int) ((double) (v - RJO_IMIN) * cell_scale(gsize));
}
static inline int cell_of_rat(int gsize, rat128 v) {
  return (int) (rat_to_double(rat_sub_int(v, RJO_IMIN)) * cell_scale(gsize));
}
static inline int cell_of_double(int gsize, double v) {
  return (int) ((v - (double) RJO_IMIN) * cell_scale(gsize));
}

/* ------------------------------------------------------------------------------------------
 * Single-pair entry points (golden-vector checks)
 *   seg = {x1,y1,x2,y2} scaled int64.  e1 = "map 0 edge", e2 = "map 1 edge".
 * ---------------------------------------------------------------------------------------- */
static void mk_edge(rjo_edge* e, const int64_t* s, rjo_pt* p1, rjo_pt* p2) {
  p1->x = s[0];
  p1->y = s[1];
  p2->x = s[2];
  p2->y = s[3];
  memset(e, 0, sizeof(*e));
  edge_eqn(e, *p1, *p2);
}

int rjo_intersect_test_segs(const int64_t* s1, const int64_t* s2) {
  rjo_edge e1, e2;
  rjo_pt a1, a2, b1, b2;
  mk_edge(&e1, s1, &a1, &a2);
  mk_edge(&e2, s2, &b1, &b2);
  return intersect_test(&e1, a1, a2, &e2, b1, b2);
}

/* out[0..7] = xnum(lo,hi) xden(lo,hi) ynum(lo,hi) yden(lo,hi); out[8],out[9] = stored int64 x,y;
 * out[10], out[11] = calculate_cell(gsize, x), calculate_cell(gsize, y) */
int rjo_intersect_point_segs(const int64_t* s1, const int64_t* s2, int gsize, int64_t* out) {
  rjo_edge e1, e2;
  rjo_pt a1, a2, b1, b2;
  rat128 x, y;
  mk_edge(&e1, s1, &a1, &a2);
  mk_edge(&e2, s2, &b1, &b2);
  if (!intersect_point(&e1, a1, a2, &e2, b1, b2, &x, &y))
    return 0;
  i128 v[4] = {x.num, x.den, y.num, y.den};
  for (int i = 0; i < 4; i++) {
    out[2 * i] = (int64_t) (u128) v[i];
    out[2 * i + 1] = (int64_t) ((u128) v[i] >> 64);
  }
  out[8] = (int64_t) rat_to_double(x);
  out[9] = (int64_t) rat_to_double(y);
  out[10] = cell_of_rat(gsize, x);
  out[11] = cell_of_rat(gsize, y);
  return 1;
}

int rjo_cell_of_int(int gsize, int64_t v) { return cell_of_int(gsize, v); }
int rjo_cell_of_double(int gsize, double v) { return cell_of_double(gsize, v); }

/* ------------------------------------------------------------------------------------------
 * PIP predicate -- src/algo/pip.h:31-96 == src/app/pip_lbvh.h:57-123
 *
 * State: best edge, best_y.  Returns 1 when `e` replaces the current best.
 * Full ties (equal xsect_y AND equal slope) are visit-order dependent in the reference
 * (q==1 keeps the first visited, q==0 the last visited); every caller here visits edges in
 * ascending eid order, which the HIP path reproduces as a total order
 * (q==1: smaller eid wins, q==0: larger eid wins).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  const rjo_edge* e;
  double y;
} pip_best;

static inline int pip_visit(const rjo_map* base, const rjo_edge* e, rjo_pt p, int query_map_id,
                            pip_best* best) {
  rjo_pt p1 = base->pts[e->p1], p2 = base->pts[e->p2];
  int64_t x_min = MIN2(p1.x, p2.x), x_max = MAX2(p1.x, p2.x);
  /* pip.h:44-47 */
  if (p.x < x_min || p.x > x_max || p.x == (query_map_id == 0 ? x_min : x_max))
    return 0;
  /* pip.h:53: (double)(int128) / int128 -> the divisor is converted to double */
  double xsect_y = (double) (-e->a * (i128) p.x - e->c) / (double) e->b;
  double diff_y = (double) p.y - xsect_y; /* :54 */
  if (diff_y == 0)
    diff_y = (double) (query_map_id == 0 ? -e->a : e->a); /* :56-58 */
  if (diff_y == 0)
    diff_y = (double) (query_map_id == 0 ? -e->b : e->b); /* :59-61 */
  if (diff_y > 0)
    return 0; /* :69-71 point above edge */
  if (xsect_y > best->y)
    return 0; /* :73-75 */
  if (xsect_y == best->y) { /* :77-93 */
    double cur = (double) e->a / (double) e->b;
    double bst = (double) best->e->a / (double) best->e->b;
    int flag = cur > bst;
    if ((query_map_id && !flag) || (flag && !query_map_id))
      return 0;
  }
  best->y = xsect_y;
  best->e = e;
  return 1;
}

/* one (point, edge) evaluation for golden/unit tests: returns 0 = rejected, 1 = accepted;
 * *yy receives xsect_y when the x-range test passed (NaN otherwise) */
int rjo_pip_single(const int64_t* seg, const int64_t* pt, int query_map_id, double* yy) {
  rjo_map m;
  rjo_pt pts[2] = {{seg[0], seg[1]}, {seg[2], seg[3]}};
  rjo_edge e;
  memset(&e, 0, sizeof(e));
  e.p1 = 0;
  e.p2 = 1;
  edge_eqn(&e, pts[0], pts[1]);
  m.pts = pts;
  m.edges = &e;
  pip_best b = {NULL, INFINITY};
  rjo_pt p = {pt[0], pt[1]};
  int r = pip_visit(&m, &e, p, query_map_id, &b);
  *yy = NAN;
  int64_t x_min = MIN2(pts[0].x, pts[1].x), x_max = MAX2(pts[0].x, pts[1].x);
  if (!(p.x < x_min || p.x > x_max || p.x == (query_map_id == 0 ? x_min : x_max)))
    *yy = (double) (-e.a * (i128) p.x - e.c) / (double) e.b;
  return r;
}

/* ------------------------------------------------------------------------------------------
 * Brute force (O(N*M)) -- definitionally "all predicate-true (map0 edge, map1 edge) pairs"
 * and "global best edge over all base edges visited in ascending eid order"
 * ---------------------------------------------------------------------------------------- */
/* returns total count (may exceed cap); pairs[2*i] = eid map0, pairs[2*i+1] = eid map1 */
uint64_t rjo_lsi_brute(const rjo_map* m0, const rjo_map* m1, uint32_t* pairs, uint64_t cap) {
  uint64_t n = 0;
#pragma omp parallel for schedule(dynamic, 64)
  for (size_t i = 0; i < m0->ne; i++) {
    const rjo_edge* e1 = &m0->edges[i];
    rjo_pt a1 = m0->pts[e1->p1], a2 = m0->pts[e1->p2];
    for (size_t j = 0; j < m1->ne; j++) {
      const rjo_edge* e2 = &m1->edges[j];
      if (intersect_test(e1, a1, a2, e2, m1->pts[e2->p1], m1->pts[e2->p2])) {
        uint64_t k;
#pragma omp atomic capture
        k = n++;
        if (k < cap) {
          pairs[2 * k] = (uint32_t) i;
          pairs[2 * k + 1] = (uint32_t) j;
        }
      }
    }
  }
  return n;
}

void rjo_pip_brute(const rjo_map* base, int query_map_id, const int64_t* pts, size_t n,
                   uint32_t* out) {
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; i++) {
    rjo_pt p = {pts[2 * i], pts[2 * i + 1]};
    pip_best b = {NULL, INFINITY};
    for (size_t j = 0; j < base->ne; j++)
      pip_visit(base, &base->edges[j], p, query_map_id, &b);
    out[i] = b.e ? b.e->eid : RJO_MISS;
  }
}

/* intersection records for an explicit pair list (row a4): pairs = (eid map0, eid map1) */
void rjo_lsi_points(const rjo_map* m0, const rjo_map* m1, const uint32_t* pairs, uint64_t n,
                    rjo_xsect* out) {
#pragma omp parallel for schedule(static)
  for (uint64_t k = 0; k < n; k++) {
    const rjo_edge* e1 = &m0->edges[pairs[2 * k]];
    const rjo_edge* e2 = &m1->edges[pairs[2 * k + 1]];
    rat128 x = {0, 1}, y = {0, 1};
    int hit = intersect_point(e1, m0->pts[e1->p1], m0->pts[e1->p2], e2, m1->pts[e2->p1],
                              m1->pts[e2->p2], &x, &y);
    store_xsect(&out[k], x, y, e1->eid, e2->eid);
    if (!hit)
      out[k].mid_point_polygon_id = -2; /* marks "pair is not an intersection" for tests */
  }
}

/* ------------------------------------------------------------------------------------------
 * Uniform grid -- src/grid/uniform_grid.h:45-86 (iterate_cell), :132-245 (AddMapsToGrid)
 * Deterministic restatement: cells are filled in ascending eid order (the reference's
 * atomicAdd fill order is nondeterministic).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int gsize;
  uint32_t* ne0;  /* per cell: #edges of map 0 */
  uint32_t* ne1;  /* per cell: #edges of map 1 */
  uint64_t* begin; /* per cell: offset into eids; map-0 eids first, then map-1 (lsi_grid.h:35,41) */
  uint32_t* eids;
  uint64_t total;
} rjo_grid;

static inline void edge_cell_range(const rjo_map* m, const rjo_edge* e, int g, int* ix1, int* ix2,
                                   int* iy1, int* iy2) {
  rjo_pt p1 = m->pts[e->p1], p2 = m->pts[e->p2];
  int ax = cell_of_int(g, p1.x), ay = cell_of_int(g, p1.y); /* uniform_grid.h:63-66 */
  int bx = cell_of_int(g, p2.x), by = cell_of_int(g, p2.y);
  if (ax > bx) { int t = ax; ax = bx; bx = t; } /* :74-77 */
  if (ay > by) { int t = ay; ay = by; by = t; }
  *ix1 = ax; *ix2 = bx; *iy1 = ay; *iy2 = by;
}

/* m1 may be NULL (AddMapToGrid with a single map, uniform_grid.h:247-349) */
rjo_grid* rjo_grid_build(const rjo_map* m0, const rjo_map* m1, int gsize) {
  rjo_grid* gr = (rjo_grid*) calloc(1, sizeof(rjo_grid));
  size_t ncell = (size_t) gsize * gsize;
  gr->gsize = gsize;
  gr->ne0 = (uint32_t*) calloc(ncell, 4);
  gr->ne1 = (uint32_t*) calloc(ncell, 4);
  gr->begin = (uint64_t*) calloc(ncell + 1, 8);
  const rjo_map* maps[2] = {m0, m1};
  for (int im = 0; im < 2; im++) { /* count, :164-177 */
    const rjo_map* m = maps[im];
    if (!m) continue;
    uint32_t* ne = im ? gr->ne1 : gr->ne0;
    for (size_t k = 0; k < m->ne; k++) {
      int x1, x2, y1, y2;
      edge_cell_range(m, &m->edges[k], gsize, &x1, &x2, &y1, &y2);
      for (int i = x1; i <= x2; i++)
        for (int j = y1; j <= y2; j++)
          ne[(size_t) j * gsize + i]++;
    }
  }
  for (size_t c = 0; c < ncell; c++) /* scan, :184-196 */
    gr->begin[c + 1] = gr->begin[c] + gr->ne0[c] + gr->ne1[c];
  gr->total = gr->begin[ncell];
  gr->eids = (uint32_t*) malloc(4 * (gr->total ? gr->total : 1));
  uint32_t* fill = (uint32_t*) calloc(ncell, 4);
  for (int im = 0; im < 2; im++) { /* fill, :215-227 */
    const rjo_map* m = maps[im];
    if (!m) continue;
    for (size_t k = 0; k < m->ne; k++) {
      int x1, x2, y1, y2;
      edge_cell_range(m, &m->edges[k], gsize, &x1, &x2, &y1, &y2);
      for (int i = x1; i <= x2; i++)
        for (int j = y1; j <= y2; j++) {
          size_t c = (size_t) j * gsize + i;
          gr->eids[gr->begin[c] + fill[c]++] = m->edges[k].eid;
        }
    }
  }
  free(fill);
  return gr;
}

void rjo_grid_free(rjo_grid* g) {
  if (!g) return;
  free(g->ne0); free(g->ne1); free(g->begin); free(g->eids); free(g);
}

uint64_t rjo_grid_total(const rjo_grid* g) { return g->total; }

/* LSI over the grid -- src/app/lsi_grid.h:19-78 (intersect_one_cell), :112-121.
 * e1 = map 0 edge, e2 = map 1 edge always (lsi_grid.h:103-104).  A hit is emitted only by the
 * cell containing the computed intersection point (:62-74).  Returns total count. */
uint64_t rjo_lsi_grid(const rjo_map* m0, const rjo_map* m1, const rjo_grid* gr, rjo_xsect* out,
                      uint64_t cap) {
  uint64_t n = 0;
  int g = gr->gsize;
#pragma omp parallel for schedule(dynamic, 1)
  for (int cy = 0; cy < g; cy++) {
    for (int cx = 0; cx < g; cx++) {
      size_t c = (size_t) cy * g + cx;
      uint32_t n0 = gr->ne0[c], n1 = gr->ne1[c];
      if (!n0 || !n1) continue;
      const uint32_t* ids = gr->eids + gr->begin[c];
      for (uint32_t i = 0; i < n0; i++) {
        const rjo_edge* e1 = &m0->edges[ids[i]];
        rjo_pt a1 = m0->pts[e1->p1], a2 = m0->pts[e1->p2];
        for (uint32_t j = 0; j < n1; j++) {
          const rjo_edge* e2 = &m1->edges[ids[n0 + j]];
          rat128 x, y;
          if (intersect_point(e1, a1, a2, e2, m1->pts[e2->p1], m1->pts[e2->p2], &x, &y)) {
            if (cell_of_rat(g, x) == cx && cell_of_rat(g, y) == cy) {
              uint64_t k;
#pragma omp atomic capture
              k = n++;
              if (k < cap) store_xsect(&out[k], x, y, e1->eid, e2->eid);
            }
          }
        }
      }
    }
  }
  return n;
}

/* PIP over the grid -- src/app/pip_grid.h:37-70 + src/algo/pip.h:14-115.
 * The grid holds the base map only (run_query.cu:381, AddMapToGrid(ctx, 0)). */
void rjo_pip_grid(const rjo_map* base, int base_map_id, const rjo_grid* gr, const int64_t* pts,
                  size_t n, uint32_t* out) {
  int g = gr->gsize;
  int query_map_id = !base_map_id; /* pip.h:19 */
  const uint32_t* cnt = base_map_id ? gr->ne1 : gr->ne0;
#pragma omp parallel for schedule(dynamic, 1024)
  for (size_t ip = 0; ip < n; ip++) {
    rjo_pt p = {pts[2 * ip], pts[2 * ip + 1]};
    int cx = cell_of_int(g, p.x), cy = cell_of_int(g, p.y); /* pip_grid.h:43-44 */
    uint32_t closest = RJO_MISS;
    for (int ccy = cy; ccy < g; ccy++) { /* pip_grid.h:51 */
      size_t c = (size_t) ccy * g + cx;
      uint32_t ne = cnt[c];
      /* dst map = 0: begin + ie; dst map = 1: begin + ne0 + ie (pip.h:29-32) */
      const uint32_t* ids = gr->eids + gr->begin[c] + (base_map_id ? gr->ne0[c] : 0);
      pip_best b = {NULL, INFINITY};
      int64_t best_y_max = 0;
      for (uint32_t ie = 0; ie < ne; ie++) {
        const rjo_edge* e = &base->edges[ids[ie]];
        if (pip_visit(base, e, p, query_map_id, &b))
          best_y_max = MAX2(base->pts[e->p1].y, base->pts[e->p2].y); /* pip.h:96 */
      }
      if (!b.e) continue;
      /* pip.h:98-114: accept only when the hit lies in this cell */
      if (cell_of_int(g, best_y_max) == ccy || !(cell_of_double(g, b.y) > ccy)) {
        closest = b.e->eid;
        break;
      }
    }
    out[ip] = closest;
  }
}


/* ------------------------------------------------------------------------------------------
 * Overlay: per-map ordering of intersections + mid-point faces
 * -- src/app/map_overlay_lbvh.h:109-265 (ComputeOutputPolygons), same steps in
 *    src/app/map_overlay_grid.h.  Deterministic restatement: the reference's thrust::sort calls
 *    are unstable; ties (same edge, same squared distance) are ordered by the other map's eid.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  rjo_xsect x;
  u128 d2;
  uint32_t mine, other;
} ovl_rec;

static int ovl_cmp(const void* a, const void* b) {
  const ovl_rec *p = (const ovl_rec*) a, *q = (const ovl_rec*) b;
  if (p->mine != q->mine) return p->mine < q->mine ? -1 : 1;
  if (p->d2 != q->d2) return p->d2 < q->d2 ? -1 : 1; /* :204-213 */
  if (p->other != q->other) return p->other < q->other ? -1 : 1;
  return 0;
}

/* pairs = n (eid map 0, eid map 1); out = n records ordered for map `im`; gsize = grid for the
 * mid-point PIP (results do not depend on it) */
void rjo_overlay_edge_xsects(const rjo_map* m0, const rjo_map* m1, int im, const uint32_t* pairs,
                             uint64_t n, int gsize, rjo_xsect* out) {
  if (n == 0) return;
  const rjo_map* mine = im ? m1 : m0;
  const rjo_map* base = im ? m0 : m1;
  ovl_rec* r = (ovl_rec*) malloc(sizeof(ovl_rec) * n);
  rjo_lsi_points(m0, m1, pairs, n, out);
  for (uint64_t i = 0; i < n; i++) {
    r[i].x = out[i];
    r[i].x.mid_point_polygon_id = -1;
    r[i].mine = out[i].eid[im];
    r[i].other = out[i].eid[1 - im];
    rjo_pt p1 = mine->pts[mine->edges[r[i].mine].p1];
    i128 dx = (i128) out[i].x_num - p1.x, dy = (i128) out[i].y_num - p1.y;
    r[i].d2 = (u128) (dx * dx) + (u128) (dy * dy);
  }
  qsort(r, n, sizeof(ovl_rec), ovl_cmp);
  /* mid-points of consecutive intersections on one edge (:215-227):
   * x1 + (x2 - x1)/2 as a rational, stored through operator double -> trunc((x1 + x2)/2) */
  int64_t* mid = (int64_t*) malloc(16 * n);
  for (uint64_t i = 0; i < n; i++) {
    int64_t mx = r[i].x.x_num, my = r[i].x.y_num;
    if (i + 1 < n && r[i + 1].mine == r[i].mine) {
      rat128 x = rat_make((i128) r[i].x.x_num + r[i + 1].x.x_num, 2);
      rat128 y = rat_make((i128) r[i].x.y_num + r[i + 1].x.y_num, 2);
      mx = (int64_t) rat_to_double(x);
      my = (int64_t) rat_to_double(y);
    }
    mid[2 * i] = mx;
    mid[2 * i + 1] = my;
  }
  uint32_t* eids = (uint32_t*) malloc(4 * n);
  rjo_grid* g = rjo_grid_build(im ? base : NULL, im ? NULL : base, gsize); /* base map id = 1 - im */
  rjo_pip_grid(base, 1 - im, g, mid, n, eids);                             /* query map id = im (:232-236) */
  rjo_grid_free(g);
  for (uint64_t i = 0; i < n; i++) {
    if (i + 1 < n && r[i + 1].mine == r[i].mine) r[i].x.mid_point_polygon_id = face_id_of(base, eids[i]);
    out[i] = r[i].x;
  }
  free(eids);
  free(mid);
  free(r);
}

int rjo_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void rjo_set_num_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n);
#else
  (void) n;
#endif
}
