// Pieces shared by the LDS-tiled line-pass kernels (edt_hull_kernels.hip).
#pragma once

#include "edt_device.hpp"

namespace vgt
{
constexpr int kBandRows = 32;

__device__ __forceinline__ uint32_t LowMask(int bits)  // bits in [0, 32]
{
  return (bits >= 32) ? ~0u : ((1u << bits) - 1u);
}

// Geometry of one line pass: a tile = all n rows of the pass axis x W adjacent Z positions.
struct TileGeom
{
  int n;                 // rows along the pass axis
  int nz;                // extent of the contiguous axis
  int ztiles;            // tiles along Z
  int64_t row_stride;    // elements between consecutive rows
  int64_t outer_stride;  // elements between consecutive outer indices
  int nx, ny;            // full grid (finalize)
  int pass_axis;         // 0 = X pass (outer = y), 1 = Y pass (outer = x)
  double resolution;
  int add_virtual_border;
  int vector_io;         // rows and base pointers are 16-byte aligned
  int debug_skip;        // timing experiments only (VGT_HULL_SKIP)
  int z_offset;          // Z slab (multi-GPU): global z of local z = 0
  int nz_global;         // Z extent of the whole grid (virtual border)
  int outer_begin;       // a launch over part of the outer axis: index of its first outer position (virtual border)
};

// Cooperative load of a tile into LDS as signed squared int32, F[row * W + line].  Rows are
// contiguous 4*W-byte (2*W for int16) segments in HBM; 16-byte chunks per lane when the tile is
// full and aligned, several loads in flight per lane either way.  Lines beyond nz get +kInf32.
template <typename InT, int W>
__device__ __forceinline__ void LoadTile(const InT* __restrict__ in, int32_t* __restrict__ F,
                                         int n, int64_t base, int z0, const TileGeom& g)
{
  struct Wrap
  {
    int32_t* F;
  } t{F};
  {
    // elements per chunk: 16 bytes, or a whole row when the row is shorter than that
    constexpr int kVec16 = 16 / static_cast<int>(sizeof(InT));
    constexpr int kVec = (W < kVec16) ? W : kVec16;
    constexpr int kChunksPerRow = W / kVec;
    constexpr int kBatch = 4;
    if (g.vector_io && (z0 + W <= g.nz))
    {
      using Chunk = __attribute__((__vector_size__(kVec * sizeof(InT)))) int;
      const int total = n * kChunksPerRow;
      for (int c0 = threadIdx.x; c0 < total; c0 += blockDim.x * kBatch)
      {
        Chunk buf[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; k++)
        {
          const int c = c0 + k * blockDim.x;
          if (c < total)
          {
            const int row = c / kChunksPerRow;
            const int part = c % kChunksPerRow;
            buf[k] = *reinterpret_cast<const Chunk*>(
                in + base + static_cast<int64_t>(row) * g.row_stride + part * kVec);
          }
        }
#pragma unroll
        for (int k = 0; k < kBatch; k++)
        {
          const int c = c0 + k * blockDim.x;
          if (c < total)
          {
            const int row = c / kChunksPerRow;
            const int part = c % kChunksPerRow;
            const InT* vals = reinterpret_cast<const InT*>(&buf[k]);
            int32_t* dst = t.F + row * W + part * kVec;
#pragma unroll
            for (int u = 0; u < kVec; u += 4)
            {
              int4 q;
              q.x = ToSignedSquare(vals[u + 0]);
              q.y = ToSignedSquare(vals[u + 1]);
              q.z = ToSignedSquare(vals[u + 2]);
              q.w = ToSignedSquare(vals[u + 3]);
              *reinterpret_cast<int4*>(dst + u) = q;
            }
          }
        }
      }
    }
    else
    {
      constexpr int kScalarBatch = 8;
      const int total = n * W;
      for (int e0 = threadIdx.x; e0 < total; e0 += blockDim.x * kScalarBatch)
      {
        InT buf[kScalarBatch];
#pragma unroll
        for (int k = 0; k < kScalarBatch; k++)
        {
          const int e = e0 + k * blockDim.x;
          const int row = e / W;
          const int ww = e % W;
          buf[k] = InT(0);
          if (e < total && z0 + ww < g.nz)
            buf[k] = in[base + static_cast<int64_t>(row) * g.row_stride + ww];
        }
#pragma unroll
        for (int k = 0; k < kScalarBatch; k++)
        {
          const int e = e0 + k * blockDim.x;
          if (e < total) t.F[e] = (z0 + (e % W) < g.nz) ? ToSignedSquare(buf[k]) : kInf32;
        }
      }
    }
  }
}
}  // namespace vgt
