// Micro-benchmark: what the memory system gives a line pass for a given tile shape, with no
// computation.  Geometry of the X pass (rows ny*nz elements apart) and of the Y pass (rows nz
// apart) at 1024^3; a tile = all n rows x W adjacent z positions (W*4-byte row segments).
//   copy<W>      : global -> registers -> global, 16-byte chunks, no LDS
//   staged<W>    : global -> LDS tile -> (thread = line x band of 32 rows) -> global, the skeleton
//                  of the round-1 hull pass
//   staged16<W>  : as staged, but the store phase writes 16-byte chunks (thread = row x 4 lines)
// Build: hipcc --offload-arch=gfx950 -O3 -o tile_copy tile_copy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Geom { int n; int ztiles; long long row_stride; long long outer_stride; int xcd_swizzle; };

__device__ __forceinline__ int TileId(const Geom& g)
{
  int tile_id = blockIdx.x;
  if (g.xcd_swizzle)
  {
    const int group = 8 * g.ztiles;
    if (tile_id < (int)gridDim.x / group * group)
    {
      const int local = tile_id % group;
      tile_id = tile_id - local + (local % 8) * g.ztiles + local / 8;
    }
  }
  return tile_id;
}

template <typename InT, int W, int K>
__global__ void copy_kernel(const InT* __restrict__ in, float* __restrict__ out, Geom g)
{
  constexpr int kVecIn = 16 / sizeof(InT);          // elements per 16-byte input chunk
  constexpr int kChunksPerRow = W / kVecIn;
  const int tile_id = TileId(g);
  const long long base = (long long)(tile_id / g.ztiles) * g.outer_stride + (long long)(tile_id % g.ztiles) * W;
  const int total = g.n * kChunksPerRow;
  using Chunk = __attribute__((__vector_size__(16))) int;
  for (int c0 = threadIdx.x; c0 < total; c0 += blockDim.x * K)
  {
    Chunk buf[K];
#pragma unroll
    for (int k = 0; k < K; k++)
    {
      const int c = c0 + k * blockDim.x;
      if (c < total) buf[k] = *reinterpret_cast<const Chunk*>(in + base + (long long)(c / kChunksPerRow) * g.row_stride + (c % kChunksPerRow) * kVecIn);
    }
#pragma unroll
    for (int k = 0; k < K; k++)
    {
      const int c = c0 + k * blockDim.x;
      if (c < total)
      {
        const InT* v = reinterpret_cast<const InT*>(&buf[k]);
        float* dst = out + base + (long long)(c / kChunksPerRow) * g.row_stride + (c % kChunksPerRow) * kVecIn;
#pragma unroll
        for (int u = 0; u < kVecIn; u += 4)
        {
          float4 q = make_float4((float)v[u], (float)v[u + 1], (float)v[u + 2], (float)v[u + 3]);
          *reinterpret_cast<float4*>(dst + u) = q;
        }
      }
    }
  }
}

template <typename InT, int W, int MODE>  // MODE 0: band store (4 B per lane, W*4-byte segments), 1: 16-byte chunk store
__global__ void staged_kernel(const InT* __restrict__ in, float* __restrict__ out, Geom g)
{
  extern __shared__ __align__(16) unsigned char smem[];
  int32_t* F = reinterpret_cast<int32_t*>(smem);
  constexpr int kVecIn = 16 / sizeof(InT);
  constexpr int kChunksPerRow = W / kVecIn;
  constexpr int K = 4;
  const int tile_id = TileId(g);
  const long long base = (long long)(tile_id / g.ztiles) * g.outer_stride + (long long)(tile_id % g.ztiles) * W;
  const int total = g.n * kChunksPerRow;
  using Chunk = __attribute__((__vector_size__(16))) int;
  for (int c0 = threadIdx.x; c0 < total; c0 += blockDim.x * K)
  {
    Chunk buf[K];
#pragma unroll
    for (int k = 0; k < K; k++)
    {
      const int c = c0 + k * blockDim.x;
      if (c < total) buf[k] = *reinterpret_cast<const Chunk*>(in + base + (long long)(c / kChunksPerRow) * g.row_stride + (c % kChunksPerRow) * kVecIn);
    }
#pragma unroll
    for (int k = 0; k < K; k++)
    {
      const int c = c0 + k * blockDim.x;
      if (c < total)
      {
        const InT* v = reinterpret_cast<const InT*>(&buf[k]);
        int32_t* dst = F + (c / kChunksPerRow) * W + (c % kChunksPerRow) * kVecIn;
#pragma unroll
        for (int u = 0; u < kVecIn; u += 4) *reinterpret_cast<int4*>(dst + u) = make_int4(v[u], v[u + 1], v[u + 2], v[u + 3]);
      }
    }
  }
  __syncthreads();
  if (MODE == 0)
  {
    const int w = threadIdx.x % W, band = threadIdx.x / W;
    const int r0 = band * 32;
    if (r0 < g.n)
    {
      float* dst = out + base + (long long)r0 * g.row_stride + w;
      const int32_t* col = F + r0 * W + w;
#pragma unroll
      for (int k = 0; k < 32; k++) dst[(long long)k * g.row_stride] = (float)col[k * W];
    }
  }
  else
  {
    constexpr int kOutChunksPerRow = W / 4;
    const int total_out = g.n * kOutChunksPerRow;
    for (int c = threadIdx.x; c < total_out; c += blockDim.x)
    {
      const int row = c / kOutChunksPerRow, part = c % kOutChunksPerRow;
      const int4 v = *reinterpret_cast<const int4*>(F + row * W + part * 4);
      *reinterpret_cast<float4*>(out + base + (long long)row * g.row_stride + part * 4) = make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w);
    }
  }
}

// plain streaming copy for reference (int32 -> float, 16 bytes per lane)
__global__ void stream_kernel(const int4* __restrict__ in, float4* __restrict__ out, long long n4)
{
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < n4; i += stride)
  {
    const int4 v = in[i];
    out[i] = make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w);
  }
}

template <typename F>
float TimeIt(F launch)
{
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(a, 0);
  for (int i = 0; i < 3; i++) launch();
  hipEventRecord(b, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  hipEventDestroy(a); hipEventDestroy(b);
  return ms / 3;
}

int main()
{
  const long long N = 1024, NV = N * N * N;
  void *in, *out;
  CHECK(hipMalloc(&in, NV * 4));
  CHECK(hipMalloc(&out, NV * 4));
  CHECK(hipMemset(in, 1, NV * 4));
  CHECK(hipMemset(out, 0, NV * 4));
  {
    float ms = TimeIt([&] { hipLaunchKernelGGL(stream_kernel, dim3(256 * 16), dim3(256), 0, 0, (const int4*)in, (float4*)out, NV / 4); });
    printf("stream copy int32->float 8 GiB: %.3f ms (%.2f TB/s)\n", ms, 8.0 * NV / ms * 1e-9);
  }
  for (int pass = 0; pass < 2; pass++)  // 0 = X geometry (int32 in), 1 = Y geometry (int16 in)
  {
    for (int swz = 0; swz < 2; swz++)
    {
      printf("== %s pass geometry, xcd swizzle %d (ms per 2^30 voxels)\n", pass == 0 ? "X" : "Y", swz);
#define RUN_COPY(T, W, THREADS)                                                                         \
  {                                                                                                     \
    Geom g{(int)N, (int)(N / W), pass == 0 ? N * N : N, pass == 0 ? N : N * N, swz};                    \
    const int blocks = (int)(N * (N / W));                                                              \
    float ms = TimeIt([&] { hipLaunchKernelGGL((copy_kernel<T, W, 8>), dim3(blocks), dim3(THREADS), 0, 0, (const T*)in, (float*)out, g); }); \
    printf("  copy   W=%3d threads=%4d: %.3f ms\n", W, THREADS, ms);                                    \
  }
#define RUN_STAGED(T, W, THREADS, MODE)                                                                 \
  {                                                                                                     \
    Geom g{(int)N, (int)(N / W), pass == 0 ? N * N : N, pass == 0 ? N : N * N, swz};                    \
    const int blocks = (int)(N * (N / W));                                                              \
    const size_t lds = N * W * 4;                                                                       \
    auto kern = staged_kernel<T, W, MODE>;                                                              \
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);       \
    float ms = TimeIt([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(THREADS), lds, 0, (const T*)in, (float*)out, g); }); \
    printf("  staged W=%3d threads=%4d mode=%d: %.3f ms\n", W, THREADS, MODE, ms);                      \
  }
      if (pass == 0)
      {
        RUN_COPY(int32_t, 8, 256) RUN_COPY(int32_t, 16, 256) RUN_COPY(int32_t, 32, 256) RUN_COPY(int32_t, 64, 256)
        RUN_COPY(int32_t, 64, 1024) RUN_COPY(int32_t, 256, 1024)
        RUN_STAGED(int32_t, 8, 256, 0) RUN_STAGED(int32_t, 8, 256, 1) RUN_STAGED(int32_t, 16, 512, 0) RUN_STAGED(int32_t, 16, 512, 1)
        RUN_STAGED(int32_t, 32, 1024, 0) RUN_STAGED(int32_t, 32, 1024, 1)
      }
      else
      {
        RUN_COPY(int16_t, 8, 256) RUN_COPY(int16_t, 16, 256) RUN_COPY(int16_t, 32, 256) RUN_COPY(int16_t, 64, 256)
        RUN_COPY(int16_t, 64, 1024) RUN_COPY(int16_t, 256, 1024)
        RUN_STAGED(int16_t, 8, 256, 0) RUN_STAGED(int16_t, 8, 256, 1) RUN_STAGED(int16_t, 16, 512, 0) RUN_STAGED(int16_t, 16, 512, 1)
        RUN_STAGED(int16_t, 32, 1024, 0) RUN_STAGED(int16_t, 32, 1024, 1)
      }
    }
  }
  return 0;
}
