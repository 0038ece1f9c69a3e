// LDS-tiled lower-envelope line passes (Y and X) -- placeholder until the tiled kernels land;
// reporting handled = false makes the launchers fall back to the pruned-search kernels.
#include "vgt_internal.hpp"

namespace vgt
{
hipError_t LaunchPassYHull(const int16_t*, int32_t*, const SdfParams&, hipStream_t, bool* handled)
{
  *handled = false;
  return hipSuccess;
}
hipError_t LaunchPassXHullFinalize(const int32_t*, float*, uint32_t*, const SdfParams&, hipStream_t,
                                   bool* handled)
{
  *handled = false;
  return hipSuccess;
}
}  // namespace vgt
