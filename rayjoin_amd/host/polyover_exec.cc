// polyover_exec -- RayJoin's polygon-overlay driver (src/overlay.cc, src/run_overlay.cu:143-228)
// on the MI355X-native LSI / PIP path.  Same flags, phases and stderr timing format.
//   MapOverlayLBVH::{Init, BuildIndex, IntersectEdge, LocateVerticesInOtherMap,
//                    ComputeOutputPolygons, WriteResult}     src/app/map_overlay_lbvh.h:25-270
//   -mode=grid (MapOverlayGrid, src/app/map_overlay_grid.h) runs the same stages on the device-side
//   uniform grid; -check compares the LBVH results with the grid's, as run_overlay.cu:18-141 does.
#include <iostream>

#include "context.h"
#include "flags.h"
#include "output_chain.h"
#include "timer.h"

using namespace rayjoin;

namespace {

class MapOverlayLBVH {
 public:
  MapOverlayLBVH(Context& ctx, double xsect_factor, bool grid = false, int grid_size = 2048)
      : ctx_(ctx), xsect_factor_(xsect_factor), grid_(grid), grid_size_(grid_size) {}
  ~MapOverlayLBVH() {
    rj_handle h = ctx_.handle();
    if (pairs_) rj_dev_free(h, pairs_);
    for (int im = 0; im < 2; im++) {
      if (closest_[im]) rj_dev_free(h, closest_[im]);
      if (faces_[im]) rj_dev_free(h, faces_[im]);
    }
  }
  void Init() {  // map_overlay_lbvh.h:25-40
    rj_handle h = ctx_.handle();
    size_t n_edges = ctx_.get_map(0)->n_edges() + ctx_.get_map(1)->n_edges();
    cap_ = (size_t) (xsect_factor_ * n_edges);
    rj_check(h, rj_dev_alloc(h, 8 * (cap_ ? cap_ : 1), (void**) &pairs_), "rj_dev_alloc");
    for (int im = 0; im < 2; im++) {
      size_t np = ctx_.get_map(im)->n_points();
      rj_check(h, rj_dev_alloc(h, 4 * (np ? np : 1), (void**) &closest_[im]), "rj_dev_alloc");
      rj_check(h, rj_dev_alloc(h, 4 * (np ? np : 1), (void**) &faces_[im]), "rj_dev_alloc");
    }
  }
  void BuildIndex() {  // :42-58: an LBVH over each map (grid mode: AddMapsToGrid)
    for (int im = 0; im < 2; im++) {
      if (grid_) rj_check(ctx_.handle(), rj_build_grid(ctx_.handle(), im, grid_size_), "rj_build_grid");
      else rj_check(ctx_.handle(), rj_build_lbvh(ctx_.handle(), im), "rj_build_lbvh");
    }
  }
  void IntersectEdge(int query_map_id) {  // :60-71
    uint64_t n = 0;
    int rc = grid_ ? rj_lsi_query_grid(ctx_.handle(), cap_, pairs_, &n)
                   : rj_lsi_query(ctx_.handle(), 1 - query_map_id, query_map_id, 0,
                                  ctx_.get_map(query_map_id)->n_edges(), cap_, pairs_, &n);
    rj_check(ctx_.handle(), rc, "rj_lsi_query");
    n_xsects_ = n;
    std::cerr << "Intersections: " << n << std::endl;
  }
  void LocateVerticesInOtherMap(int query_map_id) {  // :73-107
    const size_t np = ctx_.get_map(query_map_id)->n_points();
    rj_check(ctx_.handle(),
             grid_ ? rj_pip_query_grid(ctx_.handle(), 1 - query_map_id, query_map_id, nullptr, 0, np,
                                       closest_[query_map_id], faces_[query_map_id])
                   : rj_pip_query(ctx_.handle(), 1 - query_map_id, query_map_id, nullptr, 0, np,
                                  closest_[query_map_id], faces_[query_map_id]),
             "rj_pip_query");
  }
  void ComputeOutputPolygons() {  // :109-265
    rj_handle h = ctx_.handle();
    for (int im = 0; im < 2; im++) {
      rj_xsect* d = nullptr;
      rj_check(h, rj_dev_alloc(h, 48 * (n_xsects_ ? n_xsects_ : 1), (void**) &d), "rj_dev_alloc");
      int rc = rj_overlay_edge_xsects(h, im, pairs_, n_xsects_, d);
      xsects_[im].resize(n_xsects_);
      if (rc == RJ_OK) rc = rj_memcpy_d2h(h, xsects_[im].data(), d, 48 * n_xsects_);
      rj_dev_free(h, d);
      rj_check(h, rc, "rj_overlay_edge_xsects");
    }
  }
  void WriteResult(const char* path) {  // :267-270
    std::vector<int32_t> pip[2];
    for (int im = 0; im < 2; im++) {
      pip[im].resize(ctx_.get_map(im)->n_points());
      rj_check(ctx_.handle(), rj_memcpy_d2h(ctx_.handle(), pip[im].data(), faces_[im], 4 * pip[im].size()), "rj_memcpy_d2h");
    }
    WriteOutputChain(ctx_, xsects_, pip, path);
  }
  // CheckResult (run_overlay.cu:18-141): the same stages through -mode=grid must give the same
  // intersections and the same located edges.  Runs the device-side grid next to the LBVH results.
  bool CheckAgainstGrid(int grid_size) {
    rj_handle h = ctx_.handle();
    bool ok = true;
    for (int im = 0; im < 2; im++) rj_check(h, rj_build_grid(h, im, grid_size), "rj_build_grid");
    uint32_t* p2 = nullptr;
    rj_check(h, rj_dev_alloc(h, 8 * (cap_ ? cap_ : 1), (void**) &p2), "rj_dev_alloc");
    uint64_t n2 = 0;
    int rc = rj_lsi_query_grid(h, cap_, p2, &n2);
    ok = rc == RJ_OK && n2 == n_xsects_;
    if (ok) {
      std::vector<uint32_t> a(2 * n2), b(2 * n2);
      rj_sort_pairs(h, p2, n2);
      rj_sort_pairs(h, pairs_, n_xsects_);
      rj_memcpy_d2h(h, a.data(), p2, 8 * n2);
      rj_memcpy_d2h(h, b.data(), pairs_, 8 * n2);
      ok = a == b;
    }
    rj_dev_free(h, p2);
    std::cerr << (ok ? "LSI passed check" : "LSI check FAILED") << std::endl;
    for (int im = 0; im < 2 && ok; im++) {
      const size_t np = ctx_.get_map(im)->n_points();
      uint32_t* c2 = nullptr;
      rj_check(h, rj_dev_alloc(h, 4 * (np ? np : 1), (void**) &c2), "rj_dev_alloc");
      rc = rj_pip_query_grid(h, 1 - im, im, nullptr, 0, np, c2, nullptr);
      std::vector<uint32_t> a(np), b(np);
      rj_memcpy_d2h(h, a.data(), c2, 4 * np);
      rj_memcpy_d2h(h, b.data(), closest_[im], 4 * np);
      rj_dev_free(h, c2);
      ok = rc == RJ_OK && a == b;
      std::cerr << "Map " << im << (ok ? ": PIP passed check" : ": PIP check FAILED") << std::endl;
    }
    return ok;
  }

 private:
  Context& ctx_;
  double xsect_factor_;
  bool grid_;
  int grid_size_;
  size_t cap_ = 0, n_xsects_ = 0;
  uint32_t* pairs_ = nullptr;
  uint32_t* closest_[2] = {nullptr, nullptr};
  int32_t* faces_[2] = {nullptr, nullptr};
  std::vector<rj_xsect> xsects_[2];
};

void RunOverlay(const Flags& f) {  // run_overlay.cu:143-228
  PhaseTimer tm;
  tm.start();
  tm.next("Read map 0");
  auto g1 = load_from(f.poly1, f.serialize, f.v);
  tm.next("Read map 1");
  auto g2 = load_from(f.poly2, f.serialize, f.v);
  tm.next("Create App");
  Context ctx({g1, g2}, f.device, f.scale_fma);
  MapOverlayLBVH overlay(ctx, f.xsect_factor, f.mode == "grid", f.grid_size);
  tm.next("Load Data");
  ctx.LoadToDevice();
  tm.next("Init");
  overlay.Init();
  tm.next("Build Index");
  overlay.BuildIndex();
  tm.next("Intersection edges");
  overlay.IntersectEdge(0);
  for (int im = 0; im < 2; im++) {
    tm.next("Map " + std::to_string(im) + ": Locate vertices in other map");
    overlay.LocateVerticesInOtherMap(im);
  }
  tm.next("Computer output polygons");
  overlay.ComputeOutputPolygons();
  if (f.check && f.mode != "grid") {  // run_overlay.cu:199-204: compare with -mode=grid
    tm.next("Check result");
    if (!overlay.CheckAgainstGrid(f.grid_size)) throw std::runtime_error("result differs from -mode=grid");
  }
  if (!f.output.empty()) {
    tm.next("Write to file");
    overlay.WriteResult(f.output.c_str());
  }
  tm.end();
}

}  // namespace

int main(int argc, char* argv[]) {
  if (argc == 1) {
    std::cerr << "Usage: " << argv[0] << " -poly1 <map0.cdb> -poly2 <map1.cdb> -mode lbvh|grid [-grid_size 2048] [-output <result.cdb>]\n"
              << "  [-serialize <dir>] [-xsect_factor 0.2] [-check] [-device 0] [-v 1]\n";
    return 1;
  }
  Flags f;
  try {
    f.Parse(argc, argv);
    if (f.poly1.empty() || f.poly2.empty()) throw std::invalid_argument("-poly1 and -poly2 are required");
    if (f.mode == "rt") throw std::runtime_error("-mode=rt needs RT cores/OptiX; MI355X (gfx950) has none: use -mode=lbvh");
    if (f.mode != "lbvh" && f.mode != "grid") throw std::runtime_error("Illegal mode: " + f.mode);
    RunOverlay(f);
  } catch (const std::invalid_argument& e) {
    std::cerr << "ERROR: " << e.what() << std::endl;
    return 2;
  } catch (const std::exception& e) {
    std::cerr << "FATAL: " << e.what() << std::endl;
    return 3;
  }
  return 0;
}
