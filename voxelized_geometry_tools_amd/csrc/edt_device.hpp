// Device-side helpers shared by the EDT translation units.
#pragma once

#include "vgt_internal.hpp"

namespace vgt
{
constexpr int kWaveSize = 64;
constexpr int kNumXcd = 8;  // MI355X: 8 XCDs, workgroup b of a launch goes to XCD b % 8

// Square of a row / voxel offset (|v| <= 16384): 24-bit multiply, full rate on CDNA (the 32-bit
// v_mul_lo is quarter rate).
__device__ __forceinline__ int32_t Sq(int v) { return __mul24(v, v); }
// The same as single instructions, for hot loops where the compiler keeps the library wrapper's
// (redundant) 24-bit sign-extension shifts around the multiply: v * v and v * v + add.
__device__ __forceinline__ int32_t SqAsm(int v)
{
  int32_t r;
  asm("v_mul_i32_i24 %0, %1, %1" : "=v"(r) : "v"(v));
  return r;
}
__device__ __forceinline__ int32_t SqPlusAsm(int v, int32_t add)
{
  int32_t r;
  asm("v_mad_i32_i24 %0, %1, %1, %2" : "=v"(r) : "v"(v), "v"(add));
  return r;
}

// Decoding of the intermediate encodings into (class, squared distance so far).
__device__ __forceinline__ void Decode(int16_t v, bool& negative, int32_t& f)
{
  negative = v < 0;
  const int32_t a = negative ? -static_cast<int32_t>(v) : static_cast<int32_t>(v);
  f = (a == kInf16) ? kInf32 : a * a;
}
__device__ __forceinline__ void Decode(int32_t v, bool& negative, int32_t& f)
{
  negative = v < 0;
  f = negative ? -v : v;
}
// int16 pass-1 value -> signed squared int32 encoding used by the later passes.
__device__ __forceinline__ int32_t ToSignedSquare(int16_t v)
{
  bool negative;
  int32_t f;
  Decode(v, negative, f);
  return negative ? -f : f;
}
__device__ __forceinline__ int32_t ToSignedSquare(int32_t v) { return v; }

// Order-preserving float <-> uint32 map so min / max can use integer atomics.
__device__ __forceinline__ uint32_t EncodeOrdered(float v)
{
  const uint32_t b = __float_as_uint(v);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float DecodeOrdered(uint32_t e)
{
  const uint32_t b = (e & 0x80000000u) ? (e & 0x7fffffffu) : ~e;
  return __uint_as_float(b);
}

// float(sqrt(double(d2)) * resolution), the reference's final conversion
// (signed_distance_field_generation.hpp:98-105), bit for bit.
__device__ __forceinline__ float SqrtTimesResolutionExact(int32_t d2, double resolution)
{
  return static_cast<float>(sqrt(static_cast<double>(d2)) * resolution);
}

// Same value, cheaper on the common path.  One Newton step in double from a float rsqrt seed gives
// sqrt(d2) to a relative error below 2^-43 (seed error e0 < 2^-22: the step leaves
// e0^2 / 2 + e0 * e_h), so the product with the resolution is within ~2^10 double ulps of the
// reference's doubly rounded product.  Rounding either one to float gives the same result unless
// the product lies that close to the midpoint of two floats, i.e. its 29 bits below the float
// mantissa are within 2^10 of 2^28; a window of 2^13 is tested (`unsure`: 2^-15 of all values) and the
// caller takes the exact path for those, as for everything when the float result could be subnormal or
// overflow (resolution outside (1e-30, 1e30)) and for d2 <= 0.
// tests/test_gpu_sdf.py::test_fast_finalize_matches_exact_for_every_d2 compares the two over all d2 in [0, 2^31).
__device__ __forceinline__ float FastSqrtTimesResolution(uint32_t d2, double resolution, bool& unsure)
{
  const float xf = static_cast<float>(d2);
  const float y0 = __frsqrt_rn(xf);
  const double gd = static_cast<double>(xf * y0);
  const double hd = static_cast<double>(0.5f * y0);
  const double rem = fma(-gd, gd, static_cast<double>(d2));  // exact: gd has 24 significant bits
  const double p = fma(rem, hd, gd) * resolution;
  // the 29 bits below the float mantissa within 2^13 of one half: (low29 - (2^28 - 8191)) mod 2^29 < 16383
  const uint32_t low = static_cast<uint32_t>(__double_as_longlong(p));
  unsure = ((low << 3) - ((0x10000000u - 8191u) << 3)) < (16383u << 3);
  return static_cast<float>(p);
}
__device__ __forceinline__ float SqrtTimesResolution(int32_t d2, double resolution)
{
  const bool range_ok = (resolution > 1.0e-30) && (resolution < 1.0e30);
  if (d2 <= 0 || !range_ok) return SqrtTimesResolutionExact(d2, resolution);
  bool unsure;
  const float fast = FastSqrtTimesResolution(static_cast<uint32_t>(d2), resolution, unsure);
  return unsure ? SqrtTimesResolutionExact(d2, resolution) : fast;
}

// Virtual border (signed_distance_field_generation.hpp:134-284) through its closed form
// min(d2, b^2), b = distance to the nearest virtual border cell.
__device__ __forceinline__ int32_t ClampToVirtualBorder(int32_t d2, int x, int y, int z, int nx, int ny, int nz)
{
  int32_t b = kInf32;
  if (nx > 1) b = min(b, min(x + 1, nx - x));
  if (ny > 1) b = min(b, min(y + 1, ny - y));
  if (nz > 1) b = min(b, min(z + 1, nz - z));
  return (b != kInf32) ? min(d2, Sq(b)) : d2;
}

// Squared voxel distance (after the border clamp) -> float SDF value.
__device__ __forceinline__ float DistanceToSdf(int32_t d2, bool negative, double resolution)
{
  const float dist = (d2 == kInf32) ? __uint_as_float(0x7f800000u) : SqrtTimesResolution(d2, resolution);
  return negative ? -dist : dist;
}

// Final conversion shared by every X-pass implementation: squared voxel distance -> float SDF.
__device__ __forceinline__ float FinalizeSdf(int32_t d2, bool negative, int x, int y, int z,
                                             int nx, int ny, int nz, double resolution,
                                             int add_virtual_border)
{
  if (add_virtual_border) d2 = ClampToVirtualBorder(d2, x, y, z, nx, ny, nz);
  return DistanceToSdf(d2, negative, resolution);
}

// Wave + block reduction of the ordered encodings, one atomic pair per block.
// Every thread of the block must call it (contains a barrier).
__device__ __forceinline__ void BlockMinMax(uint32_t lo, uint32_t hi, uint32_t* minmax_enc)
{
  __shared__ uint32_t s_lo[16], s_hi[16];
  for (int off = kWaveSize / 2; off > 0; off >>= 1)
  {
    lo = min(lo, static_cast<uint32_t>(__shfl_xor(static_cast<int>(lo), off)));
    hi = max(hi, static_cast<uint32_t>(__shfl_xor(static_cast<int>(hi), off)));
  }
  const int lane = threadIdx.x & (kWaveSize - 1);
  const int wave = threadIdx.x / kWaveSize;
  if (lane == 0)
  {
    s_lo[wave] = lo;
    s_hi[wave] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    const int nwaves = (blockDim.x + kWaveSize - 1) / kWaveSize;
    for (int w = 1; w < nwaves; w++)
    {
      lo = min(lo, s_lo[w]);
      hi = max(hi, s_hi[w]);
    }
    atomicMin(&minmax_enc[0], lo);
    atomicMax(&minmax_enc[1], hi);
  }
}
}  // namespace vgt
