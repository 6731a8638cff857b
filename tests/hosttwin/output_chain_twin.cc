// Test-only driver of the product's output-map writer (rayjoin_amd/host/output_chain.h) WITHOUT a GPU: the planar
// graphs come from the CDB files, the intersection records and vertex faces from files the test wrote (the oracle
// pipeline's), the scaling from the joint bounding box exactly as Context does it.  tests/test_overlay.py diffs the
// file it writes with the committed overlay answer.
//   output_chain_twin map0.cdb map1.cdb xs0.bin xs1.bin pip0.bin pip1.bin out.cdb
#include <cstdio>
#include <vector>

#include "output_chain.h"

using namespace rayjoin;

template <typename T>
static std::vector<T> slurp(const char* path) {
  FILE* fp = fopen(path, "rb");
  if (!fp) throw std::runtime_error(std::string("cannot read ") + path);
  fseek(fp, 0, SEEK_END);
  const long bytes = ftell(fp);
  fseek(fp, 0, SEEK_SET);
  std::vector<T> v((size_t) bytes / sizeof(T));
  if (bytes && fread(v.data(), sizeof(T), v.size(), fp) != v.size()) throw std::runtime_error("short read");
  fclose(fp);
  return v;
}

int main(int argc, char** argv) {
  if (argc != 8) return 2;
  try {
    std::shared_ptr<PlanarGraph> g[2] = {load_from(argv[1], "", 0), load_from(argv[2], "", 0)};
    BoundingBox bb;
    for (auto& m : g) {
      bb.min_x = std::min(bb.min_x, m->bb.min_x); bb.max_x = std::max(bb.max_x, m->bb.max_x);
      bb.min_y = std::min(bb.min_y, m->bb.min_y); bb.max_y = std::max(bb.max_y, m->bb.max_y);
    }
    const Scaling scaling(bb);
    const std::vector<rj_xsect> xs[2] = {slurp<rj_xsect>(argv[3]), slurp<rj_xsect>(argv[4])};
    const std::vector<int32_t> pip[2] = {slurp<int32_t>(argv[5]), slurp<int32_t>(argv[6])};
    const PlanarGraph* const graphs[2] = {g[0].get(), g[1].get()};
    size_t nch = 0, nfc = 0;
    WriteOutputMap(scaling, graphs, xs, pip, argv[7], &nch, &nfc);
    printf("%zu %zu\n", nch, nfc);
  } catch (const std::exception& e) {
    fprintf(stderr, "FATAL: %s\n", e.what());
    return 3;
  }
  return 0;
}
