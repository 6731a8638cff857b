// Test-only: the C++ host's Scaling (rayjoin_amd/host/context.h) behind a C entry point, so the CPU
// test suite can hold it against the reference's golden vectors.  Built with the host's own flags
// (-ffp-contract=off, rayjoin_amd/host/Makefile).
#include <cstdint>

#include "context.h"

extern "C" {
void host_scale_points(const double* bb, const double* xy, uint64_t n, int64_t* out) {
  rayjoin::BoundingBox b;
  b.min_x = bb[0]; b.min_y = bb[1]; b.max_x = bb[2]; b.max_y = bb[3];
  const rayjoin::Scaling s(b);
  for (uint64_t i = 0; i < n; i++) {
    out[2 * i] = s.ScaleX(xy[2 * i]);
    out[2 * i + 1] = s.ScaleY(xy[2 * i + 1]);
  }
}
void host_unscale_points(const double* bb, const int64_t* xy, uint64_t n, double* out) {
  rayjoin::BoundingBox b;
  b.min_x = bb[0]; b.min_y = bb[1]; b.max_x = bb[2]; b.max_y = bb[3];
  const rayjoin::Scaling s(b);
  for (uint64_t i = 0; i < n; i++) {
    out[2 * i] = s.UnscaleX(xy[2 * i]);
    out[2 * i + 1] = s.UnscaleY(xy[2 * i + 1]);
  }
}
void host_scaling_consts(int64_t* out) {
  const rayjoin::Scaling s;
  out[0] = s.get_internal_min(); out[1] = s.get_internal_max(); out[2] = s.get_internal_range();
}
}
