// flags.h -- the reference's command-line surface (src/flags.cc:3-34, src/flags.h) without
// gflags (not installed here): -flag=value, -flag value, --flag=value, -boolflag, -noboolflag.
#pragma once
#include <cstdlib>
#include <map>
#include <stdexcept>
#include <string>

namespace rayjoin {

struct Flags {
  // defaults = src/flags.cc
  std::string poly1, poly2, output, mode, serialize, sample, query;
  int grid_size = 2048;
  double xsect_factor = 0.2;
  bool box = false, check = true, fau = false, histo = false, profile = false;
  int warmup = 5, repeat = 5;
  int ag = 1, ag_iter = 5, win = 32;
  double enlarge = 5;
  int sample_map_id = -1;
  double sample_rate = 1;
  int seed = 0;
  double gen_t = 0.1;
  int gen_n = 10000;
  int v = 0;       // glog verbosity (-v=1 in expr/run_query.sh)
  int device = 0;  // ours: GPU ordinal
  int nranks = 1, rank = 0;  // ours: one process per GPU; the query map is sharded by chain range
  std::string comm_file;     // ours: rank 0 writes the RCCL unique id here, the others read it
  std::string sample_output; // ours: write the sampled map (-sample) as a .bin cache file, for inspection/tests
  bool scale_fma = false;    // ours: scale with one fma per coordinate (nvcc's contraction of map.h:171-180) instead of scaling.h's multiply + add

  static bool parse_bool(const std::string& s) {
    if (s == "" || s == "1" || s == "true" || s == "t" || s == "yes" || s == "y") return true;
    if (s == "0" || s == "false" || s == "f" || s == "no" || s == "n") return false;
    throw std::invalid_argument("bad boolean value '" + s + "'");
  }

  void set(const std::string& k, const std::string& val, bool has_val) {
#define RJ_S(name) if (k == #name) { name = val; return; }
#define RJ_I(name) if (k == #name) { name = std::stoi(val); return; }
#define RJ_D(name) if (k == #name) { name = std::stod(val); return; }
#define RJ_B(name) if (k == #name) { name = has_val ? parse_bool(val) : true; return; } \
                   if (k == "no" #name) { name = false; return; }
    RJ_S(poly1) RJ_S(poly2) RJ_S(output) RJ_S(mode) RJ_S(serialize) RJ_S(sample) RJ_S(query)
    RJ_I(grid_size) RJ_D(xsect_factor) RJ_B(box) RJ_B(check) RJ_B(fau) RJ_I(warmup) RJ_I(repeat)
    RJ_I(ag) RJ_I(ag_iter) RJ_I(win) RJ_D(enlarge) RJ_I(sample_map_id) RJ_D(sample_rate)
    RJ_I(seed) RJ_D(gen_t) RJ_I(gen_n) RJ_B(histo) RJ_B(profile) RJ_B(scale_fma) RJ_I(v) RJ_I(device) RJ_I(nranks) RJ_I(rank) RJ_S(comm_file) RJ_S(sample_output)
#undef RJ_S
#undef RJ_I
#undef RJ_D
#undef RJ_B
    if (k == "lb" || k == "logtostderr" || k == "alsologtostderr" || k == "stderrthreshold") return;  // seen in the logs
    throw std::invalid_argument("unknown command line flag '" + k + "'");
  }

  static bool is_bool(const std::string& k) {
    static const char* b[] = {"box", "check", "fau", "histo", "profile", "scale_fma"};
    for (auto n : b) if (k == n || k == std::string("no") + n) return true;
    return false;
  }

  void Parse(int argc, char** argv) {
    for (int i = 1; i < argc; i++) {
      std::string a = argv[i];
      if (a.size() < 2 || a[0] != '-') throw std::invalid_argument("unexpected argument '" + a + "'");
      a = a.substr(a[1] == '-' ? 2 : 1);
      auto eq = a.find('=');
      if (eq != std::string::npos) {
        set(a.substr(0, eq), a.substr(eq + 1), true);
      } else if (is_bool(a)) {
        set(a, "", false);
      } else {
        if (i + 1 >= argc) throw std::invalid_argument("flag '" + a + "' is missing its argument");
        set(a, argv[++i], true);
      }
    }
  }
};

}  // namespace rayjoin
