// query_exec -- RayJoin's LSI / PIP benchmark driver (src/query.cc, src/run_query.cu) with the
// MI355X-native -mode=lbvh path behind it.  Same flags, same phases, same stderr timing format.
//   -mode=lbvh : software LBVH in hand-written HIP (this repository)
//   -mode=grid : the reference's uniform grid, also in HIP (rj_grid.hip), -grid_size as in src/flags.cc
//   -mode=rt   : rejected -- gfx950 has no ray-tracing units (the reference's OptiX path)
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <random>

#include "context.h"
#include "flags.h"
#include "lsi_pip.h"
#include "lsi_amd.h"
#include "timer.h"

using namespace rayjoin;

namespace {

void Usage(const char* argv0) {
  std::cerr << "Usage: " << argv0 << " -poly1 <base.cdb> [-poly2 <query.cdb>] -query lsi|pip -mode lbvh|grid [-grid_size 2048]\n"
            << "  [-serialize <dir>] [-xsect_factor 0.2] [-warmup 5] [-repeat 5] [-seed N] [-gen_n 10000]\n"
            << "  [-gen_t 0.1] [-output <pairs.txt>] [-device 0] [-v 1]\n"
            << "  [-sample map|edges -sample_map_id 0|1 -sample_rate 0.5 [-sample_output <map.bin>]]\n";
}

// GenerateLSIQueries (run_query.cu:102-144)
std::shared_ptr<HostMap> GenerateLSIQueries(const Flags& f, Context& ctx) {
  auto bb = ctx.get_bounding_box();
  auto& sc = ctx.get_scaling();
  std::random_device rd;
  std::mt19937 gen(f.seed == 0 ? rd() : f.seed);
  std::uniform_real_distribution<> dx(bb.min_x, bb.max_x), dy(bb.min_y, bb.max_y), dt(0, f.gen_t);
  auto m = std::make_shared<HostMap>();
  size_t ne = f.gen_n;
  for (size_t i = 0; i < ne; i++) {
    double x1 = dx(gen), y1 = dy(gen), x2 = dx(gen), y2 = dy(gen);
    double len = std::sqrt((x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1));
    double ux = (x2 - x1) / len, uy = (y2 - y1) / len, t = dt(gen);
    m->xy.push_back(sc.ScaleX(x1)); m->xy.push_back(sc.ScaleY(y1));
    m->xy.push_back(sc.ScaleX(x1 + t * ux)); m->xy.push_back(sc.ScaleY(y1 + t * uy));
    m->row_index.push_back((uint32_t) (2 * i));
    m->left.push_back(0); m->right.push_back(0);
  }
  m->row_index.push_back((uint32_t) (2 * ne));
  return m;
}

// GeneratePIPQueries (run_query.cu:147-167)
std::vector<int64_t> GeneratePIPQueries(const Flags& f, Context& ctx) {
  auto bb = ctx.get_bounding_box();
  auto& sc = ctx.get_scaling();
  std::random_device rd;
  std::mt19937 gen(f.seed == 0 ? rd() : f.seed);
  std::uniform_real_distribution<> dx(bb.min_x, bb.max_x), dy(bb.min_y, bb.max_y);
  std::vector<int64_t> pts;
  for (int i = 0; i < f.gen_n; i++) {
    double x = dx(gen), y = dy(gen);
    pts.push_back(sc.ScaleX(x)); pts.push_back(sc.ScaleY(y));
  }
  return pts;
}

// one process per GPU: rank 0 creates the RCCL id and publishes it through a file
void InitComm(const Flags& f, Context& ctx) {
  if (f.nranks <= 1 && f.comm_file.empty()) return;
  if (f.comm_file.empty()) throw std::invalid_argument("-nranks > 1 needs -comm_file <path shared by all ranks>");
  uint8_t id[RJ_COMM_ID_BYTES];
  if (f.rank == 0) {
    if (rj_comm_unique_id(id) != RJ_OK) throw std::runtime_error("rj_comm_unique_id failed");
    std::string tmp = f.comm_file + ".tmp";
    std::ofstream(tmp, std::ios::binary).write(reinterpret_cast<char*>(id), sizeof(id));
    if (rename(tmp.c_str(), f.comm_file.c_str())) throw std::runtime_error("cannot publish " + f.comm_file);
  } else {
    for (int tries = 0;; tries++) {
      std::ifstream in(f.comm_file, std::ios::binary);
      if (in.read(reinterpret_cast<char*>(id), sizeof(id))) break;
      if (tries > 600) throw std::runtime_error("timed out waiting for " + f.comm_file);
      usleep(100000);
    }
  }
  rj_check(ctx.handle(), rj_comm_init(ctx.handle(), f.nranks, f.rank, id), "rj_comm_init");
}

// -sample map|edges -sample_map_id 0|1 -sample_rate r -seed s (src/flags.cc:20-27): thin out one of
// the two input maps before anything is scaled or uploaded (the paper's scalability runs).
void ApplySampling(const Flags& f, std::shared_ptr<PlanarGraph>& base, std::shared_ptr<PlanarGraph>* query) {
  if (f.sample.empty()) return;
  if (f.sample != "map" && f.sample != "edges") throw std::invalid_argument("Invalid sample option: " + f.sample + " (map|edges)");
  if (f.sample_map_id != 0 && f.sample_map_id != 1) throw std::invalid_argument("-sample needs -sample_map_id 0|1");
  if (!(f.sample_rate > 0 && f.sample_rate <= 1)) throw std::invalid_argument("-sample_rate must be in (0, 1]");
  std::shared_ptr<PlanarGraph>* target = f.sample_map_id == 0 ? &base : query;
  if (!target || !*target) throw std::invalid_argument("-sample_map_id 1 needs -poly2");
  *target = f.sample == "map" ? sample_map_from(**target, (float) f.sample_rate, f.seed)
                              : sample_edges_from(**target, (float) f.sample_rate, f.seed);
  if (f.v >= 1)
    std::cerr << (f.sample == "map" ? "Map is sampled" : "Edges are sampled") << ", chains: " << (*target)->chains.size()
              << " points: " << (*target)->points.size() << " edges: " << (*target)->n_edges() << "\n";
  if (!f.sample_output.empty()) serialize_pgraph(**target, f.sample_output.c_str());
}

// -profile: per-stage index build times, the analogue of the reference's "LBVH Profiling result"
// (deps/lbvh/lbvh/bvh.cuh:464-474) / grid profile (src/grid/uniform_grid.h:238-244), from HIP events
void PrintBuildProfile(Context& ctx, bool grid) {
  rj_handle h = ctx.handle();
  float total = 0;
  rj_last_ms(h, RJ_T_BUILD, &total);
  if (grid) {
    printf("Grid Profiling result:\nTotal: %.3lf\n", (double) total);
    return;
  }
  float keys = 0, sort = 0, leaves = 0, levels = 0;
  rj_last_ms(h, RJ_T_BUILD_KEYS, &keys);
  rj_last_ms(h, RJ_T_BUILD_SORT, &sort);
  rj_last_ms(h, RJ_T_BUILD_LEAVES, &leaves);
  rj_last_ms(h, RJ_T_BUILD_LEVELS, &levels);
  printf("LBVH Profiling result:\nSort keys (Hilbert): %.3lf\nRadix sort: %.3lf\nLeaf blocks (gather, boxes, "
         "occupancy bitmap): %.3lf\nUpper levels + sibling order: %.3lf\nTotal: %.3lf\n",
         (double) keys, (double) sort, (double) leaves, (double) levels, (double) total);
}

// CheckPIPResult (run_query.cu:22-99): run the grid PIP on the same points and compare.  A different
// eid is not yet a wrong answer -- two edges may have the same coordinates -- so differing eids are
// compared by the scaled endpoints of their edges, as the reference does.
bool CheckPIPResult(Context& ctx, const Flags& f, const int64_t* d_pts, size_t n_points, const std::vector<uint32_t>& res) {
  rj_handle h = ctx.handle();
  std::cerr << "Checking point in polygon" << std::endl;
  rj_check(h, rj_build_grid(h, 0, f.grid_size), "rj_build_grid");
  PIPGrid<Context> pip_grid(ctx);
  pip_grid.Init(n_points);
  pip_grid.Query(ctx.get_stream(), 1, ArrayView<HostMap::point_t>((HostMap::point_t*) d_pts, n_points));
  std::vector<uint32_t> ans;
  pip_grid.get_closest_eids(ans);
  auto base = ctx.get_map(0);
  auto endpoints = [&](uint32_t eid, int64_t out[4]) {  // eid = p_idx - ichain (map.h:198-207)
    size_t lo = 0, hi = base->n_chains();          // largest chain c with row_index[c] - c <= eid
    while (hi - lo > 1) {
      size_t mid = (lo + hi) / 2;
      if ((size_t) base->row_index[mid] - mid <= eid) lo = mid; else hi = mid;
    }
    const size_t p = (size_t) eid + lo;
    out[0] = base->xy[2 * p]; out[1] = base->xy[2 * p + 1]; out[2] = base->xy[2 * p + 2]; out[3] = base->xy[2 * p + 3];
  };
  size_t n_diff = 0;
  for (size_t i = 0; i < n_points; i++) {
    if (ans[i] == res[i]) continue;
    bool diff = (ans[i] == RJ_MISS_EID) != (res[i] == RJ_MISS_EID);
    if (!diff) {
      int64_t a[4], b[4];
      endpoints(ans[i], a);
      endpoints(res[i], b);
      diff = a[0] != b[0] || a[1] != b[1] || a[2] != b[2] || a[3] != b[3];
    }
    if (diff && n_diff < 10) printf("point %zu ans %u, res %u\n", i, ans[i], res[i]);
    n_diff += diff;
  }
  if (n_diff)
    std::cerr << "Map: 0 Total points: " << n_points << " n diff: " << n_diff << " Error rate: "
              << (double) n_diff / n_points * 100 << " %" << std::endl;
  else
    std::cerr << "Map: 0 passed check" << std::endl;
  return n_diff == 0;
}

void CheckMode(const Flags& f) {
  // "amd": the same LBVH path through the adapter classes a RayJoin maintainer would add (host/lsi_amd.h)
  if (f.mode == "lbvh" || f.mode == "grid" || f.mode == "amd") return;
  if (f.mode == "rt") throw std::runtime_error("-mode=rt needs RT cores/OptiX; MI355X (gfx950) has none: use -mode=lbvh");
  throw std::runtime_error("Invalid index type: " + f.mode);
}

void RunLSIQuery(const Flags& f) {  // run_query.cu:169-314
  PhaseTimer tm;
  tm.start();
  tm.next("Read map 0");
  auto base = load_from(f.poly1, f.serialize, f.v);
  std::unique_ptr<Context> ctx;
  std::shared_ptr<HostMap> gen_queries;
  if (f.poly2.empty()) {
    ApplySampling(f, base, nullptr);
    tm.next("Generate Workloads");
    ctx.reset(new Context({base, nullptr}, f.device, f.scale_fma));
    gen_queries = GenerateLSIQueries(f, *ctx);
  } else {
    tm.next("Read map 1");
    auto query = load_from(f.poly2, f.serialize, f.v);
    ApplySampling(f, base, &query);
    ctx.reset(new Context({base, query}, f.device, f.scale_fma));
  }
  tm.next("Create App");
  const bool grid = f.mode == "grid";
  if (grid && f.nranks > 1) throw std::invalid_argument("-mode=grid joins the two whole maps: no -nranks");
  const bool amd = f.mode == "amd";
  if (amd && f.nranks > 1) throw std::invalid_argument("-mode=amd is the single-GPU adapter: use -mode=lbvh with -nranks");
  LSIAMD<Context>* lsi_amd = amd ? new LSIAMD<Context>(*ctx, ctx->handle()) : nullptr;
  std::unique_ptr<LSI<Context>> lsi_p(amd ? (LSI<Context>*) lsi_amd : grid ? (LSI<Context>*) new LSIGrid<Context>(*ctx) : (LSI<Context>*) new LSILBVH<Context>(*ctx));
  LSI<Context>& lsi = *lsi_p;
  // (get_xsects / size are not virtual in the reference's LSI either: the adapter's results are read through its own type)
  auto n_found = [&]() { return lsi_amd ? lsi_amd->size() : lsi.size(); };
  auto d_pairs = [&]() { return lsi_amd ? lsi_amd->get_pairs() : lsi.get_pairs(); };
  auto d_xsects = [&]() { return lsi_amd ? lsi_amd->get_xsects() : lsi.get_xsects(); };
  tm.next("Load Data");
  ctx->LoadToDevice();
  if (gen_queries) ctx->set_map(1, gen_queries);
  tm.next("Init");
  size_t queue_cap = (size_t) ((ctx->get_map(0)->n_edges() + ctx->get_map(1)->n_edges()) * f.xsect_factor);
  std::cerr << "Queue capacity: " << queue_cap << std::endl;
  lsi.Init(queue_cap);
  InitComm(f, *ctx);
  size_t e0 = 0, e1 = ctx->get_map(1)->n_edges(), p0 = 0, p1 = 0;
  if (f.nranks > 1) {
    ctx->get_map(1)->shard(f.nranks, f.rank, &e0, &e1, &p0, &p1);
    std::cerr << "Rank " << f.rank << "/" << f.nranks << ": query eids [" << e0 << ", " << e1 << ")" << std::endl;
  }
  if (!amd) lsi.set_query_range(e0, e1);
  tm.next("Build Index");
  if (grid) {  // run_query.cu:247-249: AddMapsToGrid
    rj_check(ctx->handle(), rj_build_grid(ctx->handle(), 0, f.grid_size), "rj_build_grid");
    rj_check(ctx->handle(), rj_build_grid(ctx->handle(), 1, f.grid_size), "rj_build_grid");
  } else {
    rj_check(ctx->handle(), rj_build_lbvh(ctx->handle(), 0), "rj_build_lbvh");
  }
  if (f.profile) PrintBuildProfile(*ctx, grid);
  tm.next("Warmup");
  Stream& stream = ctx->get_stream();
  for (int i = 0; i < f.warmup; i++) lsi.Query(stream, 1);
  tm.next("Query", f.repeat);
  float kernel_ms = 0;
  for (int i = 0; i < f.repeat; i++) {
    if (f.v) std::cerr << "Iter: " << i << std::endl;
    lsi.Query(stream, 1);
    float ms = 0;
    rj_last_ms(ctx->handle(), RJ_T_LSI_KERNEL, &ms);
    kernel_ms += ms;
  }
  tm.next("Cleanup");
  if (!f.comm_file.empty()) {  // all-gather-v of the intersection queues over RCCL
    uint64_t total = lsi.AllGather(queue_cap);
    std::cerr << "Rank " << f.rank << ": local intersections " << lsi.local_size() << ", all ranks " << total << std::endl;
  }
  std::cerr << "Intersections: " << n_found() << " Queue Load Factor: " << (double) n_found() / (queue_cap ? queue_cap : 1)
            << std::endl;
  if (f.repeat > 0) std::cerr << "LSI kernel (HIP events): " << kernel_ms / f.repeat << " ms" << std::endl;
  if (!f.output.empty()) {
    // d_xsects = lsi->get_xsects() (run_query.cu:304): the records Query left on the device, in queue order; the file
    // lists them by (eid[0], eid[1]) -- sorted on the host, where the reference's checker sorts (run_overlay.cu:38-52)
    std::vector<rj_xsect> xs(d_xsects().size());
    rj_check(ctx->handle(), rj_memcpy_d2h(ctx->handle(), xs.data(), d_xsects().data(), sizeof(rj_xsect) * xs.size()), "rj_memcpy_d2h");
    std::sort(xs.begin(), xs.end(), [](const rj_xsect& a, const rj_xsect& b) { return a.eid[0] != b.eid[0] ? a.eid[0] < b.eid[0] : a.eid[1] < b.eid[1]; });
    (void) d_pairs;
    FILE* fp = fopen(f.output.c_str(), "w");
    if (!fp) throw std::runtime_error("Cannot write " + f.output);
    for (auto& x : xs) fprintf(fp, "%u %u %ld %ld\n", x.eid[0], x.eid[1], (long) x.x_num, (long) x.y_num);
    fclose(fp);
  }
  tm.end();
}

void RunPIPQuery(const Flags& f) {  // run_query.cu:316-462
  PhaseTimer tm;
  tm.start();
  tm.next("Read map 0");
  auto base = load_from(f.poly1, f.serialize, f.v);
  std::unique_ptr<Context> ctx;
  std::vector<int64_t> gen_pts;
  int64_t* d_pts = nullptr;
  size_t n_points = 0;
  if (f.poly2.empty()) {
    ApplySampling(f, base, nullptr);
    tm.next("Generate Workloads");
    ctx.reset(new Context({base, nullptr}, f.device, f.scale_fma));
    gen_pts = GeneratePIPQueries(f, *ctx);
    tm.next("Load Data");
    ctx->LoadToDevice();
    n_points = gen_pts.size() / 2;
    rj_check(ctx->handle(), rj_dev_alloc(ctx->handle(), 16 * (n_points ? n_points : 1), (void**) &d_pts), "rj_dev_alloc");
    rj_check(ctx->handle(), rj_memcpy_h2d(ctx->handle(), d_pts, gen_pts.data(), 16 * n_points), "rj_memcpy_h2d");
  } else {
    tm.next("Read map 1");
    auto query = load_from(f.poly2, f.serialize, f.v);
    ApplySampling(f, base, &query);
    ctx.reset(new Context({base, query}, f.device, f.scale_fma));
    tm.next("Load Data");
    ctx->LoadToDevice();
    n_points = ctx->get_map(1)->n_points();
  }
  tm.next("Create App");
  const bool grid = f.mode == "grid";
  const bool amd = f.mode == "amd";
  PIPAMD<Context>* pip_amd = amd ? new PIPAMD<Context>(*ctx, ctx->handle()) : nullptr;
  std::unique_ptr<PIP<Context>> pip_p(amd ? (PIP<Context>*) pip_amd : grid ? (PIP<Context>*) new PIPGrid<Context>(*ctx) : (PIP<Context>*) new PIPLBVH<Context>(*ctx));
  PIP<Context>& pip = *pip_p;
  const ArrayView<HostMap::point_t> query_points((HostMap::point_t*) d_pts, n_points);  // (no array: every vertex of the query map)
  tm.next("Init");
  pip.Init(n_points);
  tm.next("Build Index");
  if (grid)  // run_query.cu:381: AddMapToGrid(ctx, 0)
    rj_check(ctx->handle(), rj_build_grid(ctx->handle(), 0, f.grid_size), "rj_build_grid");
  else
    rj_check(ctx->handle(), rj_build_lbvh(ctx->handle(), 0), "rj_build_lbvh");
  if (f.profile) PrintBuildProfile(*ctx, grid);
  tm.next("Warmup");
  Stream& stream = ctx->get_stream();
  for (int i = 0; i < f.warmup; i++) pip.Query(stream, 1, query_points);
  tm.next("Query", f.repeat);
  float kernel_ms = 0;
  for (int i = 0; i < f.repeat; i++) {
    pip.Query(stream, 1, query_points);
    float ms = 0;
    rj_last_ms(ctx->handle(), RJ_T_PIP_KERNEL, &ms);
    kernel_ms += ms;
  }
  std::vector<uint32_t> eids;
  if (pip_amd) {
    eids.resize(pip_amd->get_closest_eids().size());
    rj_check(ctx->handle(), rj_memcpy_d2h(ctx->handle(), eids.data(), pip_amd->get_closest_eids().data(), 4 * eids.size()), "rj_memcpy_d2h");
  } else {
    pip.get_closest_eids(eids);
  }
  bool check_ok = true;
  if (f.check && !grid) {  // run_query.cu:449-455
    tm.next("Check");
    check_ok = CheckPIPResult(*ctx, f, d_pts, n_points, eids);
  }
  tm.next("Cleanup");
  size_t hits = 0;
  for (auto e : eids) hits += e != RJ_MISS_EID;
  std::cerr << "Points: " << n_points << " Hits: " << hits << std::endl;
  if (f.repeat > 0) std::cerr << "PIP kernel (HIP events): " << kernel_ms / f.repeat << " ms" << std::endl;
  if (!f.output.empty()) {
    std::vector<int32_t> faces;
    if (pip_amd) {  // (the adapter mirrors the reference's PIP, which has no face ids: a second query through the operator that has)
      PIPLBVH<Context> with_faces(*ctx);
      with_faces.Init(n_points);
      with_faces.Query(stream, 1, query_points);
      with_faces.get_face_ids(faces);
    } else {
      pip.get_face_ids(faces);
    }
    FILE* fp = fopen(f.output.c_str(), "w");
    if (!fp) throw std::runtime_error("Cannot write " + f.output);
    for (size_t i = 0; i < eids.size(); i++) fprintf(fp, "%u %d\n", eids[i], faces[i]);
    fclose(fp);
  }
  if (d_pts) rj_dev_free(ctx->handle(), d_pts);
  tm.end();
  if (!check_ok) throw std::runtime_error("-check: the LBVH result differs from -mode=grid");
}

}  // namespace

int main(int argc, char* argv[]) {
  if (argc == 1) {
    Usage(argv[0]);
    return 1;
  }
  Flags f;
  try {
    f.Parse(argc, argv);
    if (f.poly1.empty()) throw std::invalid_argument("-poly1 is required");
    CheckMode(f);
    if (f.query == "lsi") RunLSIQuery(f);
    else if (f.query == "pip") RunPIPQuery(f);
    else throw std::invalid_argument("Invalid query: " + f.query);
  } catch (const std::invalid_argument& e) {
    std::cerr << "ERROR: " << e.what() << std::endl;
    Usage(argv[0]);
    return 2;
  } catch (const std::exception& e) {
    std::cerr << "FATAL: " << e.what() << std::endl;
    return 3;
  }
  return 0;
}
