// Host stand-in for <hip/hip_runtime.h>, for tests/cpp/sweep_emulation.cc ONLY: lets the lane-per-line sweep kernels of
// csrc/edt_sweep_kernels.hip be compiled by g++ and run one lane at a time on the CPU (the lanes of these kernels
// do not exchange data: wave votes only steer wave-uniform loops, and every lane has its own ring column), so that the
// stack / ring / spill bookkeeping can be checked against a brute-force transform without a GPU.  Test infrastructure,
// never part of the product.
#pragma once
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include <algorithm>

#define VGT_HOST_EMULATION 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __restrict__
#define __shared__ static
#define __launch_bounds__(...)
#define __align__(x)

struct dim3
{
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct EmulatedIndex
{
  unsigned x, y, z;
};
inline thread_local EmulatedIndex threadIdx{0, 0, 0}, blockIdx{0, 0, 0}, blockDim{64, 1, 1}, gridDim{1, 1, 1};

using hipStream_t = void*;
using hipError_t = int;
using hipEvent_t = void*;
constexpr hipError_t hipSuccess = 0;
constexpr hipError_t hipErrorInvalidValue = 1;
inline hipError_t hipGetLastError() { return hipSuccess; }

struct uint2
{
  uint32_t x, y;
};
struct uint4
{
  uint32_t x, y, z, w;
};
struct int4
{
  int32_t x, y, z, w;
};
inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }

using std::max;
using std::min;

inline int __mul24(int a, int b) { return static_cast<int>(static_cast<uint32_t>(a) * static_cast<uint32_t>(b)); }
#define __builtin_assume(cond) assert(cond)  // (the emulation checks every assumption the kernels hand to the optimiser)
inline uint32_t __umul24(uint32_t a, uint32_t b) { return (a & 0xffffffu) * (b & 0xffffffu); }
inline int __clz(int v) { return v == 0 ? 32 : __builtin_clz(static_cast<unsigned>(v)); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __clzll(long long v) { return v == 0 ? 64 : __builtin_clzll(static_cast<unsigned long long>(v)); }
inline float __frsqrt_rn(float x) { return 1.0f / std::sqrt(x); }
inline float __frcp_rn(float x) { return 1.0f / x; }
inline float __fsqrt_rn(float x) { return std::sqrt(x); }
inline long long __double_as_longlong(double v)
{
  long long r;
  std::memcpy(&r, &v, 8);
  return r;
}
inline float __uint_as_float(uint32_t v)
{
  float r;
  std::memcpy(&r, &v, 4);
  return r;
}
inline uint32_t __float_as_uint(float v)
{
  uint32_t r;
  std::memcpy(&r, &v, 4);
  return r;
}
// one lane per "wave": votes and shuffles see this lane only
// One lane runs at a time, so a vote holds the lane's own bit -- plus, now and then, the bit of an imaginary other lane:
// code guarded by a vote must be correct for a lane that did not ask for it.
inline unsigned long long vgt_emulated_vote(bool pred)
{
  static unsigned long long state = 0x9E3779B97F4A7C15ull;
  state = state * 6364136223846793005ull + 1442695040888963407ull;
  const bool other = ((state >> 59) == 0);  // 1 in 32
  return (pred ? 1ull : 0ull) | (other ? 2ull : 0ull);
}
#define __builtin_amdgcn_ballot_w64(pred) vgt_emulated_vote(pred)
#define __builtin_amdgcn_sbfe(value, offset, width) \
  (static_cast<int32_t>(static_cast<uint32_t>(value) << (32 - (offset) - (width))) >> (32 - (width)))
#define __builtin_amdgcn_alignbit(hi, lo, shift) \
  (static_cast<uint32_t>(((static_cast<uint64_t>(hi) << 32) | static_cast<uint64_t>(lo)) >> (shift)))
#define __builtin_amdgcn_readfirstlane(value) (value)
inline uint32_t vgt_emulated_bitreverse32(uint32_t v)
{
  uint32_t r = 0;
  for (int i = 0; i < 32; i++) r |= ((v >> i) & 1u) << (31 - i);
  return r;
}
#ifndef __clang__
#define __builtin_bitreverse32(value) vgt_emulated_bitreverse32(value)
#endif
inline int __any(int pred) { return pred != 0; }
inline int __shfl_xor(int v, int) { return v; }
inline void __syncthreads() {}
inline uint32_t atomicMin(uint32_t* p, uint32_t v)
{
  const uint32_t old = *p;
  *p = std::min(old, v);
  return old;
}
inline uint32_t atomicMax(uint32_t* p, uint32_t v)
{
  const uint32_t old = *p;
  *p = std::max(old, v);
  return old;
}

// every (block, thread) of the launch, one after the other
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...)            \
  do                                                                         \
  {                                                                          \
    const dim3 emu_grid = (grid), emu_block = (block);                       \
    (void)(lds);                                                             \
    (void)(stream);                                                          \
    gridDim = EmulatedIndex{emu_grid.x, 1, 1};                               \
    blockDim = EmulatedIndex{emu_block.x, 1, 1};                             \
    for (unsigned emu_b = 0; emu_b < emu_grid.x; emu_b++)                    \
      for (unsigned emu_t = 0; emu_t < emu_block.x; emu_t++)                 \
      {                                                                      \
        blockIdx = EmulatedIndex{emu_b, 0, 0};                               \
        threadIdx = EmulatedIndex{emu_t, 0, 0};                              \
        kernel(__VA_ARGS__);                                                 \
      }                                                                      \
  } while (0)
