// Blocking host calls of vgt_hip_sdf_from_occupancy_f32 on a sequence of small cubes (edges from argv), 200 calls each:
// ms per call, per edge, in the order given.  g++ -O2 -o small_seq small_seq.cc -L../../voxelized_geometry_tools_amd -lvgt_hip
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../include/vgt_hip.h"

int main(int argc, char** argv)
{
  vgt_hip_ctx* ctx = nullptr;
  if (vgt_hip_create(0, -1, &ctx) != 0) { std::printf("no device: %s\n", vgt_hip_last_error()); return 1; }
  for (int a = 1; a < argc; a++)
  {
    const int64_t n = std::atoll(argv[a]);
    std::vector<float> occ(static_cast<size_t>(n * n * n), 0.0f), sdf(occ.size());
    for (int64_t i = 0; i < n * n; i++) occ[static_cast<size_t>(i * 7 % (n * n * n))] = 1.0f;
    float lo = 0, hi = 0;
    for (int r = 0; r < 5; r++) vgt_hip_sdf_from_occupancy_f32(ctx, occ.data(), n, n, n, 0.01, 1, 0, sdf.data(), &lo, &hi);
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < 200; r++) vgt_hip_sdf_from_occupancy_f32(ctx, occ.data(), n, n, n, 0.01, 1, 0, sdf.data(), &lo, &hi);
    const double ms = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / 200 * 1e3;
    std::printf("%lld^3: %.4f ms per call (max %g)\n", static_cast<long long>(n), ms, hi);
  }
  vgt_hip_destroy(ctx);
  return 0;
}
