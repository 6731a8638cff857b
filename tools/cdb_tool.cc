// cdb_tool -- loader diagnostics (host only, no GPU):
//   cdb_tool parse <in.cdb> <threads> [out.bin]   time read_pgraph (threads = 1: the reference's serial
//                                                  loop; > 1: the chunked parallel form), optionally
//                                                  write the .bin cache image of what was read
//   cdb_tool bin2txt <in.bin> <out.cdb>            write a .bin cache file as CDB text ("%.9f", like misc/shp2cdb.py:35)
// Build: g++ -O2 -std=c++17 -pthread -I rayjoin_amd/host tools/cdb_tool.cc -o tools/cdb_tool
#include <chrono>
#include <cstdio>
#include <cstring>
#include <iostream>

#include "planar_graph.h"

using namespace rayjoin;

int main(int argc, char** argv) {
  try {
    if (argc >= 4 && !strcmp(argv[1], "parse")) {
      const int threads = atoi(argv[3]);
      const auto t0 = std::chrono::steady_clock::now();
      auto g = threads <= 1 ? read_pgraph_serial(argv[2], 0) : read_pgraph_parallel(argv[2], 0, threads);
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      printf("{\"threads\": %d, \"read_map_ms\": %.1f, \"chains\": %zu, \"points\": %zu, \"edges\": %zu}\n", threads, ms,
             g->chains.size(), g->points.size(), g->n_edges());
      if (argc >= 5) serialize_pgraph(*g, argv[4]);
      return 0;
    }
    if (argc == 4 && !strcmp(argv[1], "bin2txt")) {
      auto g = deserialize_pgraph(argv[2]);
      FILE* f = fopen(argv[3], "w");
      if (!f) throw std::runtime_error(std::string("Cannot write ") + argv[3]);
      std::vector<char> big(1 << 22);
      setvbuf(f, big.data(), _IOFBF, big.size());
      for (size_t c = 0; c < g->chains.size(); c++) {
        const auto& ch = g->chains[c];
        const size_t b = g->row_index[c], e = g->row_index[c + 1];
        fprintf(f, "%ld %zu %ld %ld %ld %ld\n", (long) ch.id, e - b, (long) ch.first_point_idx, (long) ch.last_point_idx,
                (long) ch.left_polygon_id, (long) ch.right_polygon_id);
        for (size_t k = b; k < e; k++) fprintf(f, "%.9f %.9f\n", g->points[k].x, g->points[k].y);
      }
      fclose(f);
      return 0;
    }
  } catch (const std::exception& e) {
    std::cerr << "FATAL: " << e.what() << std::endl;
    return 3;
  }
  std::cerr << "usage: cdb_tool parse <in.cdb> <threads> [out.bin] | cdb_tool bin2txt <in.bin> <out.cdb>\n";
  return 1;
}
