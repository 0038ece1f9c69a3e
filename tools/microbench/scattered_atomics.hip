// Stand-alone micro-benchmark (not part of the library): how many 4-byte global atomic increments per second the chip
// sustains on a tracking grid of the raycaster's size, i.e. the bound the raycast kernel lives under ("bound
// by L2 / Infinity-Cache atomic throughput, not by streaming HBM").  Patterns:
//   scattered    every lane of every wave instruction increments a uniformly random cell of the 128 MiB grid (int32
//                [2 x 256^3], the seen-free counter of a random cell): what one-atomic-per-visit raycasting of an
//                incoherent cloud would do
//   ray_like     the 64 lanes of a wave walk 64 neighbouring rays: consecutive steps move to a neighbouring cell, lanes
//                stay within a few cells of each other (cells of one or two cache lines per wave instruction)
//   one_line     all 64 lanes of a wave instruction hit distinct counters of ONE 256-byte stretch (the best case)
// Output: one JSON line {"pattern": G atomics / s, ...}.   make -C tools/microbench scattered_atomics && ./scattered_atomics
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                                     \
  do                                                                                 \
  {                                                                                  \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess)                                                            \
    {                                                                                \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                   \
      return 1;                                                                      \
    }                                                                                \
  } while (0)

__device__ __forceinline__ uint32_t Mix(uint32_t x)
{
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}

constexpr int kSteps = 256;

__global__ __launch_bounds__(256) void Scattered(int32_t* grid, uint32_t cells, uint32_t seed)
{
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t h = Mix(tid * 2654435761u + seed);
  for (int s = 0; s < kSteps; s++)
  {
    h = Mix(h + 0x9e3779b9u);
    atomicAdd(&grid[2ull * (h % cells)], 1);
  }
}

__global__ __launch_bounds__(256) void RayLike(int32_t* grid, int n, uint32_t seed)
{
  // a wave = 64 rays leaving one point in nearly the same direction: lane l is offset by (l % 8, l / 8) cells sideways
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) / 64;
  const int lane = threadIdx.x & 63;
  const uint32_t h = Mix(wave * 2654435761u + seed);
  int x = 8 + static_cast<int>(h % (n - 16)), y = 8 + static_cast<int>((h >> 8) % (n - 16)) + lane / 8;
  int z = static_cast<int>((h >> 16) % (n - kSteps - 8)) + lane % 8;
  for (int s = 0; s < kSteps; s++)
  {
    atomicAdd(&grid[2ull * ((static_cast<uint64_t>(x) * n + y) * n + z)], 1);
    z += 1;  // the walk's dominant axis is the contiguous one here
    if ((s & 7) == 7) y = y + 1 < n ? y + 1 : y;
  }
}

__global__ __launch_bounds__(256) void OneLine(int32_t* grid, uint32_t cells, uint32_t seed)
{
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) / 64;
  const int lane = threadIdx.x & 63;
  uint32_t h = Mix(wave * 2654435761u + seed);
  for (int s = 0; s < kSteps; s++)
  {
    h = Mix(h + 0x9e3779b9u);
    const uint64_t base = (h % (cells / 32 - 1)) * 64ull;  // 64 int32 = 256 bytes
    atomicAdd(&grid[base + lane], 1);
  }
}

int main()
{
  const int n = 256;
  const uint32_t cells = static_cast<uint32_t>(n) * n * n;
  int32_t* grid = nullptr;
  CHECK(hipMalloc(&grid, 2ull * cells * sizeof(int32_t)));
  CHECK(hipMemset(grid, 0, 2ull * cells * sizeof(int32_t)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int blocks = 256 * 32, threads = 256;
  const double atomics = static_cast<double>(blocks) * threads * kSteps;
  std::printf("{\"grid\": \"int32[2 x 256^3] = 128 MiB\", \"atomics_per_launch\": %.0f", atomics);
  for (int pattern = 0; pattern < 3; pattern++)
  {
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++)
    {
      CHECK(hipEventRecord(e0));
      if (pattern == 0)
        hipLaunchKernelGGL(Scattered, dim3(blocks), dim3(threads), 0, 0, grid, cells, 17u + rep);
      else if (pattern == 1)
        hipLaunchKernelGGL(RayLike, dim3(blocks), dim3(threads), 0, 0, grid, n, 17u + rep);
      else
        hipLaunchKernelGGL(OneLine, dim3(blocks), dim3(threads), 0, 0, grid, cells, 17u + rep);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (rep && ms < best) best = ms;
    }
    const char* names[3] = {"scattered", "ray_like", "one_line"};
    std::printf(", \"%s_G_atomics_per_s\": %.2f", names[pattern], atomics / (best * 1e-3) / 1e9);
  }
  std::printf("}\n");
  return 0;
}
