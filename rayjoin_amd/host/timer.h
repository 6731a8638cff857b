// timer.h -- phase timer with the reference's exact stderr format, which its plotting scripts
// regex ("Timing results:" / " - <phase>: <ms> ms"; src/util/timer.h:57-80,
// expr/draw/print_query.py:28-39).
#pragma once
#include <sys/time.h>

#include <iostream>
#include <string>
#include <tuple>
#include <vector>

namespace rayjoin {
class PhaseTimer {
 public:
  static double now() {
    struct timeval tv;
    gettimeofday(&tv, nullptr);
    return tv.tv_sec + tv.tv_usec / 1000000.0;
  }
  void start() { t_.clear(); }
  void next(const std::string& name, int repeat = 1) { t_.emplace_back(name, now(), repeat); }
  void end() {
    next("end");
    std::cerr << "Timing results:" << std::endl;
    for (size_t i = 0; i + 1 < t_.size(); i++) {
      double dt = std::get<1>(t_[i + 1]) - std::get<1>(t_[i]);
      std::cerr << " - " << std::get<0>(t_[i]) << ": " << dt * 1000 / std::get<2>(t_[i]) << " ms" << std::endl;
      std::cerr << std::endl;
    }
    t_.clear();
  }

 private:
  std::vector<std::tuple<std::string, double, int>> t_;
};
}  // namespace rayjoin
