// Pointcloud raycast voxelization for gfx950: per-ray 3-D DDA with atomic seen-free /
// seen-filled accumulation, and the per-voxel combine-and-filter.
//
// Behaviour follows the reference's device kernels (src/voxelized_geometry_tools/
// cuda_voxelization_helpers.cu:73-356 "RaycastPoint", :358-426 "FilterGrids") including their
// quirks (final voxel marked first and never as free-by-walk, `t2 > tmax` slab update, the
// 1e-10 nudge that vanishes in float).  This TU is compiled with -ffp-contract=off so the
// float arithmetic is the plain left-to-right evaluation the oracle restates; sqrt and
// division are correctly rounded (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt).
// Differences from the reference kernels, none of which change results on grids the
// reference can index: 64-bit cell indices (the reference's int32 index overflows at 2^30
// cells), tracking counters addressed as (cell*2 + {0 free, 1 filled}).
//
// The walk.  A ray's DDA state is kept as counters (cell index moved by a per-axis delta, steps left per axis, steps
// left in all) so that a step is a few compares and selects: no three-way branch, no 64-bit index arithmetic.
//
// Accumulation.  All rays of a call leave from one point, so rays of similar direction visit the same voxels for a long
// stretch: large clouds are first ordered by direction (a counting sort over cube-map bins, Morton order inside a face),
// so that a workgroup's rays form a narrow cone, and the workgroup counts its seen-free visits in an LDS table (cell ->
// count) that it flushes with ONE global atomic per distinct cell every few steps of the walk.  Integer additions
// commute, so the tracking counts are bit-identical to the one-atomic-per-visit formulation; a visit that finds its
// slot taken by another cell simply goes to global memory directly.  What the kernel costs, in the order it was
// found (profiles/r5/experiments.md, "Raycaster"): the walk's instructions; the flush's scattered atomics, which the L2
// retires by the 128-byte line (hence a table whose slot order is address order); lanes of a wave on ONE LDS address,
// which the LDS serves one after the other (hence a wave's lanes spread over the workgroup's cone).
#include "vgt_internal.hpp"

#include <cmath>
#include <mutex>
#include <type_traits>
#include <vector>

namespace vgt
{
namespace
{
template <typename Real>
struct RaycastTraits;
// ToIndex: what `static_cast<integer>(std::floor(x))` of the reference gives where IT runs.  The float walk restates
// the device kernels (cuda_voxelization_helpers.cu:140-144, :229-240): on the device the cast saturates and turns NaN
// into 0, which is what v_cvt_i32_f32 does as well.  The double walk restates the CPU voxelizer
// (cpu_pointcloud_voxelization.cpp:107, :181, :294-297), whose cast on x86-64 answers "indefinite" (the most negative
// integer) to NaN: never a voxel of the grid.  It matters for a ray of length zero seen from outside the grid (its
// direction is 0 / 0 and its entry point NaN): the device kernels start it in voxel (0, 0, 0), the CPU voxelizer drops it.
template <>
struct RaycastTraits<float>
{
  using Grid = RaycastGridF32;
  static constexpr float kFlat = 1e-10f;
  static constexpr float kNudge = 1e-10f;
  // (spelled out: a bare float -> int cast of NaN or of an out-of-range value is undefined in C++, whatever v_cvt_i32_f32
  // does with it; the compiler folds this back into the one conversion instruction -- the same contract as the oracle's
  // device_index_f32)
  static __device__ __forceinline__ int32_t ToIndex(float floored)
  {
    if (isnan(floored)) return 0;
    if (floored >= 2147483648.0f) return INT32_MAX;
    if (floored <= -2147483648.0f) return INT32_MIN;
    return static_cast<int32_t>(floored);
  }
};
template <>
struct RaycastTraits<double>
{
  using Grid = RaycastGridF64;
  static constexpr double kFlat = 1e-10;
  static constexpr double kNudge = 1e-10;
  static __device__ __forceinline__ int32_t ToIndex(double floored)
  {
    // The CPU voxelizer's index is 64 bits wide (the oracle's host_index_f64): NaN and everything outside int64 is x86's
    // "indefinite", the most negative integer; a value inside int64 but outside int32 keeps its sign -- all that matters
    // about an index that far outside the grid is which way the walk steps away from it.
    if (!(floored > -9223372036854775808.0 && floored < 9223372036854775808.0)) return INT32_MIN;
    if (floored >= 2147483648.0) return INT32_MAX;
    if (floored <= -2147483648.0) return INT32_MIN;
    return static_cast<int32_t>(floored);
  }
};

template <typename Real>
__device__ __forceinline__ Real AxisT(Real point, Real ray, Real lo, Real hi)
{
  // GetAxisTValue, cuda_voxelization_helpers.cu:52-71
  if (ray > Real(0)) return fabs((hi - point) / ray);
  if (ray < -Real(0)) return fabs((point - lo) / ray);
  return static_cast<Real>(INFINITY);
}

__device__ __forceinline__ bool InGrid(const int32_t idx[3], const int32_t counts[3])
{
  return idx[0] >= 0 && idx[0] < counts[0] && idx[1] >= 0 && idx[1] < counts[1] && idx[2] >= 0 &&
         idx[2] < counts[2];
}

__device__ __forceinline__ int64_t CellIndex(const int32_t idx[3], const int32_t counts[3])
{
  return (static_cast<int64_t>(idx[0]) * counts[1] + idx[1]) * counts[2] + idx[2];
}

// Rays of one cloud all leave from the same voxel, so during the first steps of the walk the
// lanes of a wave hit a handful of addresses (1M rays -> 1M increments of ONE counter; a single
// address retires roughly one atomic per 11-13 ns, MI355X_MICROARCH.md "fanin").  For those
// steps the wave combines equal addresses first: the leader of each group adds the group's
// population count.  Integer adds commute, so the counts stay bit-exact.  After kMaxRounds
// distinct addresses the remaining lanes fall back to one atomic each.
constexpr int kAggregatedSteps = 12;
constexpr int kMaxRounds = 8;

__device__ __forceinline__ void AggregatedIncrement(int32_t* __restrict__ tracking, int64_t index)
{
  uint64_t pending = __ballot(1);
  const int lane = static_cast<int>(__lane_id());
  bool mine = true;
  for (int round = 0; round < kMaxRounds && pending; round++)
  {
    const int leader = __ffsll(static_cast<long long>(pending)) - 1;
    const int lo = __builtin_amdgcn_readlane(static_cast<int>(index), leader);
    const int hi = __builtin_amdgcn_readlane(static_cast<int>(index >> 32), leader);
    const int64_t group = (static_cast<int64_t>(hi) << 32) | static_cast<uint32_t>(lo);
    const bool member = mine && (index == group);
    const uint64_t same = __ballot(member);
    if (lane == leader) atomicAdd(&tracking[group], static_cast<int32_t>(__popcll(same)));
    if (member) mine = false;
    pending &= ~same;
  }
  if (mine) atomicAdd(&tracking[index], 1);
}

// ---- per-workgroup accumulation table (see the header) ----
// Tuned on config 3 (profiles/r5/experiments.md, "Raycaster"): 512 rays per workgroup, 8 slots per ray (4096 slots = a
// 16 x 16 x 16 window, 32 KiB of LDS), a flush every 16 steps.
constexpr int kTableThreads = 512;        // workgroup size of the table kernel unless the caller fixes one
constexpr int kPlainThreads = 256;        // ... of the kernel without the table
constexpr int kTableSlotsPerThread = 8;
constexpr int kWalkSegment = 16;          // steps between flushes
constexpr uint32_t kEmptyKey = 0xffffffffu;

// (LDS pointers carry their address space, so that the two places a visit can go -- the table or global memory -- stay two
// instructions, ds_add and global_atomic_add, instead of one flat atomic on a selected address)
using LdsWord = __attribute__((address_space(3))) uint32_t;

__device__ __forceinline__ void LdsCount(LdsWord* word, uint32_t by)
{
  __hip_atomic_fetch_add(word, by, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ uint32_t LdsClaim(LdsWord* word, uint32_t expected, uint32_t desired)
{
  __hip_atomic_compare_exchange_strong(word, &expected, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
  return expected;  // the word's value before the exchange
}

// The table is DIRECT-MAPPED BY POSITION: the slot of voxel (x, y, z) is (x mod 2^bx, y mod 2^by, z mod 2^bz), z
// fastest -- a window of 16 x 16 x 16 voxels for 4096 slots that the workgroup's cone moves through, a voxel that maps
// onto a slot held by another voxel going to global memory directly.  What this buys is the ORDER of the flush: 64
// consecutive slots are four z-rows of 16 voxels, so one wave instruction of the flush touches a handful of 128-byte
// lines of the tracking grid (a hashed table's flush touches 64; tools/sim/raycast_table_sim.c counts 5.3 M line
// transactions per 1 M-point cloud against 18.7 M, and the L2 retires scattered atomics by the line).
struct VisitTable
{
  LdsWord* keys;
  LdsWord* counts;
  int slots;            // power of two
  uint32_t mask[3];     // the slot index's field of each axis
  __device__ __forceinline__ VisitTable(uint32_t* words, int table_slots)
      : keys((LdsWord*)words), counts((LdsWord*)words + table_slots), slots(table_slots)
  {
    const int bits = 31 - __clz(table_slots);
    const int bx = bits / 3, by = (bits + 1) / 3, bz = (bits + 2) / 3;
    mask[0] = ((1u << bx) - 1u) << (by + bz);
    mask[1] = ((1u << by) - 1u) << bz;
    mask[2] = (1u << bz) - 1u;
  }
  __device__ __forceinline__ uint32_t Slot(const int32_t idx[3]) const
  {
    const uint32_t unit0 = mask[0] & (0u - mask[0]), unit1 = mask[1] & (0u - mask[1]);
    return ((static_cast<uint32_t>(idx[0]) * unit0) & mask[0]) | ((static_cast<uint32_t>(idx[1]) * unit1) & mask[1]) |
           (static_cast<uint32_t>(idx[2]) & mask[2]);
  }
  // what a step of `direction` (+1 / -1) along `axis` adds to the slot index, inside the axis' field: one unit, or the
  // whole field (= -1 modulo the field's size)
  __device__ __forceinline__ uint32_t SlotStep(int axis, int32_t direction) const
  {
    return direction > 0 ? (mask[axis] & (0u - mask[axis])) : mask[axis];
  }
  __device__ __forceinline__ void Clear()
  {
    for (int s = threadIdx.x; s < slots; s += blockDim.x)
    {
      keys[s] = kEmptyKey;
      counts[s] = 0u;
    }
  }
  // A seen-free visit of `cell` (cell < 2^32 - 1) is recorded in two halves with a step of the walk between them, so
  // that the wave never waits for the LDS: Lookup() asks for the key of the cell's slot; Record(), a DDA step later,
  // counts the visit if the slot holds the cell, or -- a first visit: the slot is empty -- sends off a claim at once,
  // whose answer is what the lane OWES looking at in its next Record(), or -- the slot is another cell's -- sends the
  // visit to global memory.  A claim is a compare-and-swap (empty -> cell) whose answer settles it: the slot now holds
  // the cell (count the visit there) or another lane's cell won the slot (global).  The sooner a claim lands, the fewer of
  // the rays that enter the voxel in the next steps still see the slot empty and claim it again.
  struct Owed
  {
    uint32_t cell;
    uint32_t slot;     // kNoSlot: nothing owed
    uint32_t claimed;  // the claim's answer (in flight until it is looked at)
  };
  static constexpr uint32_t kNoSlot = 0xffffffffu;
  __device__ __forceinline__ uint32_t Lookup(uint32_t slot) const { return keys[slot]; }
  __device__ __forceinline__ void Settle(Owed& owed, int32_t* __restrict__ tracking)
  {
    if (owed.slot != kNoSlot)
    {
      const bool mine = owed.claimed == kEmptyKey || owed.claimed == owed.cell;
      if (mine) LdsCount(counts + owed.slot, 1u);
      if (!mine) atomicAdd(&tracking[static_cast<int64_t>(owed.cell) * 2], 1);
      owed.slot = kNoSlot;
    }
  }
  __device__ __forceinline__ void Record(uint32_t cell, uint32_t slot, uint32_t key, Owed& owed,
                                         int32_t* __restrict__ tracking)
  {
    Settle(owed, tracking);
    if (key == cell)
      LdsCount(counts + slot, 1u);
    else if (key == kEmptyKey)
    {
      owed.cell = cell;
      owed.slot = slot;
      owed.claimed = LdsClaim(keys + slot, kEmptyKey, cell);
    }
    else
      atomicAdd(&tracking[static_cast<int64_t>(cell) * 2], 1);
  }
  __device__ __forceinline__ void Flush(int32_t* __restrict__ tracking)
  {
    for (int s = threadIdx.x; s < slots; s += blockDim.x)
    {
      const uint32_t key = keys[s];
      if (key != kEmptyKey)
      {
        atomicAdd(&tracking[static_cast<int64_t>(key) * 2], static_cast<int32_t>(counts[s]));
        keys[s] = kEmptyKey;
        counts[s] = 0u;
      }
    }
  }
};

// ---- rays of a workgroup dealt to its waves by the length of their walk ----
// A wave runs until its longest ray is done, so a wave of mixed rays idles a third of its lanes (config 3: ranges
// uniform in [0.5, 4] m, 155 steps on average, 250 for the longest of 64).  Before the walk the workgroup therefore
// sorts its rays by the expected number of steps (a counting sort in LDS over kLengthBuckets buckets) and hands them
// out in that order: wave 0 the shortest 64, the last wave the longest.  The rays' walk state changes lanes through
// the LDS the table uses afterwards: 16 words per ray, which the table's 2 x 8 words per thread provide.  A wave's lanes
// are then rays of like length from all over the workgroup's cone, which also keeps them off each other's LDS
// addresses (see the header).
constexpr int kLengthBuckets = 256;
typedef uint32_t Word4 __attribute__((ext_vector_type(4)));
using LdsWord4 = __attribute__((address_space(3))) Word4;

// -> this thread's rank among the workgroup's threads by `bucket` (ties in arrival order).  `histogram` and `start`:
// kLengthBuckets words each.
__device__ __forceinline__ uint32_t RankInWorkgroup(uint32_t bucket, uint32_t* histogram, uint32_t* start)
{
  for (int b = threadIdx.x; b < kLengthBuckets; b += blockDim.x) histogram[b] = 0u;
  __syncthreads();
  const uint32_t arrival = atomicAdd(&histogram[bucket], 1u);
  __syncthreads();
  if (threadIdx.x < 64u)
  {
    // wave 0: four buckets per lane, an exclusive scan across the lanes
    constexpr int kPerLane = kLengthBuckets / 64;
    uint32_t count[kPerLane], sum = 0u;
    for (int k = 0; k < kPerLane; k++)
    {
      count[k] = histogram[threadIdx.x * kPerLane + k];
      sum += count[k];
    }
    uint32_t inclusive = sum;
    for (int d = 1; d < 64; d <<= 1)
    {
      const uint32_t below = __shfl_up(inclusive, d, 64);
      if (threadIdx.x >= static_cast<unsigned>(d)) inclusive += below;
    }
    uint32_t running = inclusive - sum;
    for (int k = 0; k < kPerLane; k++)
    {
      start[threadIdx.x * kPerLane + k] = running;
      running += count[k];
    }
  }
  __syncthreads();
  return start[bucket] + arrival;
}

// ---- ordering of a cloud by ray direction (counting sort, see the header) ----
// Bins: 6 cube-map faces x 32 x 32 cells, Morton order inside a face (a workgroup's consecutive rays then span a compact
// patch of neighbouring cells).  Two kernels over chunks of kSortChunk rays: the first bins the rays, counts its chunk per
// bin in LDS and adds the counts to the bins' totals; the second scans the totals (every workgroup for itself, in LDS),
// takes its chunk's share of every bin with one atomic on the bin's cursor, and places its rays with LDS atomics.  The
// order of rays inside a bin is whatever the atomics make it: it decides which rays share a workgroup, never a count.
constexpr int kFaceBits = 5;
constexpr int kSortBins = 6 << (2 * kFaceBits);               // 6144
constexpr int kSortChunk = 4096;                              // rays per workgroup of the two kernels
constexpr int kSortThreads = 256;
constexpr int64_t kSortMinPoints = 16384;                     // smaller clouds: the extra launches cost more than they save
static_assert(kSortBins % kSortThreads == 0 && kSortBins < 65536 && kSortChunk < 65536, "16-bit bins and counts");
__device__ __forceinline__ uint32_t SpreadBits(uint32_t x)    // 0b..cba -> 0b..0c0b0a
{
  x &= 0xffffu;
  x = (x | (x << 8)) & 0x00ff00ffu;
  x = (x | (x << 4)) & 0x0f0f0f0fu;
  x = (x | (x << 2)) & 0x33333333u;
  x = (x | (x << 1)) & 0x55555555u;
  return x;
}

template <typename Real>
__device__ __forceinline__ uint32_t DirectionBin(const Real* __restrict__ points, int64_t point_stride, int64_t i,
                                                 const Real* __restrict__ T)
{
  const float px = static_cast<float>(points[point_stride * i + 0]);
  const float py = static_cast<float>(points[point_stride * i + 1]);
  const float pz = static_cast<float>(points[point_stride * i + 2]);
  if (!(isfinite(px) && isfinite(py) && isfinite(pz))) return kSortBins - 1;
  const float d[3] = {static_cast<float>(T[0]) * px + static_cast<float>(T[4]) * py + static_cast<float>(T[8]) * pz,
                      static_cast<float>(T[1]) * px + static_cast<float>(T[5]) * py + static_cast<float>(T[9]) * pz,
                      static_cast<float>(T[2]) * px + static_cast<float>(T[6]) * py + static_cast<float>(T[10]) * pz};
  const float ax = fabsf(d[0]), ay = fabsf(d[1]), az = fabsf(d[2]);
  int axis = 0;
  float major = ax;
  if (ay > major)
  {
    axis = 1;
    major = ay;
  }
  if (az > major)
  {
    axis = 2;
    major = az;
  }
  if (!(major > 0.0f && isfinite(major))) return 0;
  const float u = d[(axis + 1) % 3] / major, v = d[(axis + 2) % 3] / major;  // in [-1, 1]
  const int cells = 1 << kFaceBits;
  const int iu = min(cells - 1, max(0, static_cast<int>((u + 1.0f) * (0.5f * cells))));
  const int iv = min(cells - 1, max(0, static_cast<int>((v + 1.0f) * (0.5f * cells))));
  const uint32_t face = static_cast<uint32_t>(axis * 2 + (d[axis] < 0.0f ? 1 : 0));
  return (face << (2 * kFaceBits)) | SpreadBits(static_cast<uint32_t>(iu)) | (SpreadBits(static_cast<uint32_t>(iv)) << 1);
}

// (two 16-bit counters per word: a chunk holds kSortChunk rays, so a counter cannot overflow)
__device__ __forceinline__ void CountBin(uint32_t* histogram, uint32_t bin)
{
  atomicAdd(&histogram[bin >> 1], 1u << ((bin & 1u) * 16u));
}
__device__ __forceinline__ uint32_t BinCount(const uint32_t* histogram, int bin)
{
  return (histogram[bin >> 1] >> ((bin & 1) * 16)) & 0xffffu;
}

// bin_total must be zero on entry.
template <typename Real>
__global__ __launch_bounds__(kSortThreads) void DirectionBinKernel(const Real* __restrict__ points, int64_t num_points,
                                                                  int64_t point_stride,
                                                                  const typename RaycastTraits<Real>::Grid g,
                                                                  uint16_t* __restrict__ bins,
                                                                  uint32_t* __restrict__ bin_total)
{
  __shared__ uint32_t histogram[kSortBins / 2];
  for (int b = threadIdx.x; b < kSortBins / 2; b += kSortThreads) histogram[b] = 0u;
  __syncthreads();
  const int64_t first = static_cast<int64_t>(blockIdx.x) * kSortChunk;
  const int64_t last = min(first + kSortChunk, num_points);
  // (constant trip counts, so that the loads of several rays are in flight together)
#pragma unroll 4
  for (int k = 0; k < kSortChunk / kSortThreads; k++)
  {
    const int64_t i = first + k * kSortThreads + threadIdx.x;
    if (i < last)
    {
      const uint32_t bin = DirectionBin<Real>(points, point_stride, i, g.xform);
      bins[i] = static_cast<uint16_t>(bin);
      CountBin(histogram, bin);
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kSortBins / kSortThreads; k++)
  {
    const int b = k * kSortThreads + threadIdx.x;
    const uint32_t count = BinCount(histogram, b);
    if (count) atomicAdd(&bin_total[b], count);
  }
}

// cursor must be zero on entry; bin_total complete.
__global__ __launch_bounds__(kSortThreads) void ScatterOrderKernel(const uint16_t* __restrict__ bins, int64_t num_points,
                                                                  const uint32_t* __restrict__ bin_total,
                                                                  uint32_t* __restrict__ cursor,
                                                                  uint32_t* __restrict__ order)
{
  __shared__ uint32_t next[kSortBins];           // the bins' totals, then their first positions, then this chunk's
  __shared__ uint32_t histogram[kSortBins / 2];  // this chunk's count per bin
  __shared__ uint32_t partial[kSortThreads];
  constexpr int kPerThread = kSortBins / kSortThreads;  // 24
  for (int b = threadIdx.x; b < kSortBins; b += kSortThreads) next[b] = bin_total[b];
  for (int b = threadIdx.x; b < kSortBins / 2; b += kSortThreads) histogram[b] = 0u;
  __syncthreads();
  const int64_t first = static_cast<int64_t>(blockIdx.x) * kSortChunk;
  const int64_t last = min(first + kSortChunk, num_points);
  constexpr int kRaysPerThread = kSortChunk / kSortThreads;  // 16
  uint32_t my_bin[kRaysPerThread];
#pragma unroll
  for (int k = 0; k < kRaysPerThread; k++)
  {
    const int64_t i = first + k * kSortThreads + threadIdx.x;
    my_bin[k] = i < last ? bins[i] : 0xffffffffu;
  }
#pragma unroll
  for (int k = 0; k < kRaysPerThread; k++)
    if (my_bin[k] != 0xffffffffu) CountBin(histogram, my_bin[k]);
  // exclusive scan of the totals: a thread's run of bins, then the runs' sums across the workgroup
  const int first_bin = threadIdx.x * kPerThread;
  uint32_t sum = 0;
  for (int k = 0; k < kPerThread; k++) sum += next[first_bin + k];
  partial[threadIdx.x] = sum;
  __syncthreads();
  for (int d = 1; d < kSortThreads; d <<= 1)
  {
    const uint32_t add = (threadIdx.x >= static_cast<unsigned>(d)) ? partial[threadIdx.x - d] : 0u;
    __syncthreads();
    partial[threadIdx.x] += add;
    __syncthreads();
  }
  uint32_t running = partial[threadIdx.x] - sum;
  for (int k = 0; k < kPerThread; k++)
  {
    const uint32_t total = next[first_bin + k];
    next[first_bin + k] = running;
    running += total;
  }
  __syncthreads();
  // this chunk's place inside every bin it has rays in
  // (all of a thread's atomics are sent before the first answer is used)
  uint32_t before[kPerThread];
#pragma unroll
  for (int k = 0; k < kPerThread; k++)
  {
    const int b = k * kSortThreads + threadIdx.x;
    const uint32_t count = BinCount(histogram, b);
    before[k] = count ? atomicAdd(&cursor[b], count) : 0u;
  }
#pragma unroll
  for (int k = 0; k < kPerThread; k++) next[k * kSortThreads + threadIdx.x] += before[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kRaysPerThread; k++)
    if (my_bin[k] != 0xffffffffu)
      order[atomicAdd(&next[my_bin[k]], 1u)] = static_cast<uint32_t>(first + k * kSortThreads + threadIdx.x);
}

// kTable: seen-free visits go through the workgroup's LDS table (needs num_cells < 2^32 - 1); `order`
// (optional) = the direction-sorted permutation of the points.
template <typename Real, bool kTable>
__global__ void RaycastKernel(const Real* __restrict__ points, int64_t num_points,
                              int64_t point_stride,  // elements between consecutive points (3 = packed xyz)
                              const uint32_t* __restrict__ order,
                              const typename RaycastTraits<Real>::Grid g,
                              int32_t* __restrict__ tracking, int table_slots)
{
  extern __shared__ uint32_t table_words[];  // kTable: 2 * table_slots words
  VisitTable table(table_words, table_slots);
  // (float rays only: a double ray's state does not fit the 16 words)
  constexpr bool kLengthSort = kTable && std::is_same<Real, float>::value;

  // Walk state.  The reference walks (cur, end, step) and tests cur against end and the grid every step
  // (cuda_voxelization_helpers.cu:283-352); the same walk is kept here as counters, so that a step is a handful of
  // selects instead of three-way branches and a 64-bit index computation:
  //   cell     linear index of cur, moved by delta[a] = step[a] * stride[a]
  //   lim[a]   steps axis a may still take: min(|end[a] - cur[a]|, steps to the grid's face).  Choosing an axis whose
  //            lim is 0 ends the walk: either cur[a] == end[a] (the reference's break) or the step leaves the grid
  //            (the reference steps, fails InGrid at the top of the next iteration and visits nothing more)
  //   total    sum over axes of |end[a] - cur[a]|: 0 <=> cur == end (the reference's test before every visit)
  // The first voxel is tested against the grid once; after that only the stepped axis can leave it.
  using Index = std::conditional_t<kTable, uint32_t, uint64_t>;  // kTable: num_cells < 2^32 - 1
  // With the table, wave w of the workgroup takes rays w, w + waves, w + 2 waves ... of the workgroup's share of the
  // direction-sorted cloud: the same cone per workgroup, but a wave's lanes are spread over all of it, so that fewer of
  // them stand in the same voxel at the same step (the LDS serves lanes on one address one after the other).
  const unsigned waves = blockDim.x / 64u;
  const int64_t ray_slot = static_cast<int64_t>(blockIdx.x) * blockDim.x +
                           (kTable && !kLengthSort ? (threadIdx.x % 64u) * waves + threadIdx.x / 64u : threadIdx.x);
  bool walking = ray_slot < num_points;
  Index cell = 0, total = 0, delta0 = 0, delta1 = 0, delta2 = 0;
  uint32_t lim0 = 0, lim1 = 0, lim2 = 0;
  uint32_t slot = 0, slot_step0 = 0, slot_step1 = 0, slot_step2 = 0;  // kTable: the cell's place in the table
  uint32_t expected_steps = 0;                                         // kLengthSort: the key of the re-deal
  Real t0 = Real(0), t1 = Real(0), t2 = Real(0), dt0 = Real(0), dt1 = Real(0), dt2 = Real(0);
  if (walking)
  {
    const int64_t i = order ? static_cast<int64_t>(order[ray_slot]) : ray_slot;
    const Real px = points[point_stride * i + 0];
    const Real py = points[point_stride * i + 1];
    const Real pz = points[point_stride * i + 2];
    walking = isfinite(px) && isfinite(py) && isfinite(pz);
    if (walking)
    {
      const Real* T = g.xform;
      // point and cloud origin in the grid frame
      const Real gp[3] = {T[0] * px + T[4] * py + T[8] * pz + T[12],
                          T[1] * px + T[5] * py + T[9] * pz + T[13],
                          T[2] * px + T[6] * py + T[10] * pz + T[14]};
      const Real origin[3] = {T[12], T[13], T[14]};

      // clip the ray to max_range
      const Real ray[3] = {gp[0] - origin[0], gp[1] - origin[1], gp[2] - origin[2]};
      const Real length = sqrt(ray[0] * ray[0] + ray[1] * ray[1] + ray[2] * ray[2]);
      const bool clipped = length > g.max_range;
      Real last[3] = {gp[0], gp[1], gp[2]};
      if (clipped)
      {
        for (int a = 0; a < 3; a++) last[a] = origin[a] + (ray[a] * (g.max_range / length));
      }

      // entry point: the origin itself, or where the ray enters the grid's box
      int32_t origin_idx[3];
      for (int a = 0; a < 3; a++)
        origin_idx[a] = RaycastTraits<Real>::ToIndex(floor(origin[a] * g.inverse_voxel_size));
      Real first[3] = {origin[0], origin[1], origin[2]};
      if (!InGrid(origin_idx, g.counts))
      {
        Real tmin = Real(0);
        Real tmax = g.max_range;
        Real dir[3];
        for (int a = 0; a < 3; a++) dir[a] = ray[a] / length;
        for (int a = 0; a < 3; a++)
        {
          if (!walking) break;
          if (fabs(dir[a]) < RaycastTraits<Real>::kFlat)
          {
            if (!(origin[a] >= Real(0) && origin[a] < g.grid_size[a])) walking = false;
          }
          else
          {
            const Real ood = Real(1) / dir[a];
            const Real tlow = (Real(0) - origin[a]) * ood;
            const Real thigh = (g.grid_size[a] - origin[a]) * ood;
            const Real t1 = (tlow <= thigh) ? tlow : thigh;
            const Real t2 = (tlow <= thigh) ? thigh : tlow;
            if (t1 > tmin) tmin = t1;
            if (t2 > tmax) tmax = t2;  // as the reference (cuda_voxelization_helpers.cu:206-209)
            if (tmin > tmax) walking = false;
          }
        }
        for (int a = 0; a < 3; a++)
          first[a] = origin[a] + (dir[a] * (tmin + RaycastTraits<Real>::kNudge));
      }

      if (walking)
      {
        const Real half = g.voxel_size * Real(0.5);
        int32_t cur[3], end[3];
        Real t[3], dt[3];
        uint32_t lim[3];
        int64_t move[3];
        uint64_t remaining = 0;
        const int64_t stride[3] = {static_cast<int64_t>(g.counts[1]) * g.counts[2], g.counts[2], 1};
        for (int a = 0; a < 3; a++)
        {
          cur[a] = RaycastTraits<Real>::ToIndex(floor(first[a] * g.inverse_voxel_size));
          end[a] = RaycastTraits<Real>::ToIndex(floor(last[a] * g.inverse_voxel_size));
          const int64_t diff = static_cast<int64_t>(end[a]) - cur[a];
          const int32_t step = (diff > 0) - (diff < 0);
          const Real centre = (static_cast<Real>(cur[a]) + Real(0.5)) * g.voxel_size;
          t[a] = AxisT<Real>(first[a], ray[a], centre - half, centre + half);
          dt[a] = fabs(g.voxel_size / ray[a]);
          const uint64_t apart = static_cast<uint64_t>(diff < 0 ? -diff : diff);  // < 2^32
          remaining += apart;
          // (meaningful only when cur is inside the grid, which is tested below)
          const uint32_t room = static_cast<uint32_t>(step > 0 ? g.counts[a] - 1 - cur[a] : cur[a]);
          lim[a] = static_cast<uint32_t>(apart < room ? apart : room);
          move[a] = step * stride[a];
        }
        // the end voxel is recorded first: seen-free if the ray was clipped, seen-filled otherwise
        if (InGrid(end, g.counts))
          atomicAdd(&tracking[CellIndex(end, g.counts) * 2 + (clipped ? 0 : 1)], 1);
        walking = remaining != 0 && InGrid(cur, g.counts);
        if (walking)
        {
          // (a walk inside the grid takes fewer steps than the grid has cells, so clamping `total` changes nothing)
          const uint64_t most = static_cast<uint64_t>(~Index(0));
          total = static_cast<Index>(remaining < most ? remaining : most);
          cell = static_cast<Index>(CellIndex(cur, g.counts));
          delta0 = static_cast<Index>(move[0]);  // two's complement: cell + delta wraps to the right index
          delta1 = static_cast<Index>(move[1]);
          delta2 = static_cast<Index>(move[2]);
          lim0 = lim[0], lim1 = lim[1], lim2 = lim[2];
          if constexpr (kTable)
          {
            slot = table.Slot(cur);
            slot_step0 = table.SlotStep(0, static_cast<int32_t>(move[0] > 0) - static_cast<int32_t>(move[0] < 0));
            slot_step1 = table.SlotStep(1, static_cast<int32_t>(move[1] > 0) - static_cast<int32_t>(move[1] < 0));
            slot_step2 = table.SlotStep(2, static_cast<int32_t>(move[2] > 0) - static_cast<int32_t>(move[2] < 0));
          }
          t0 = t[0], t1 = t[1], t2 = t[2];
          dt0 = dt[0], dt1 = dt[1], dt2 = dt[2];
          if constexpr (kLengthSort)
          {
            // steps until the ray's end or the grid's face, whichever comes first: the parameter at which the walk
            // crosses its last face inside the grid, then the crossings of every axis up to it
            float leaves = INFINITY;
            for (int a = 0; a < 3; a++)
              if (move[a] != 0)
                leaves = fminf(leaves, static_cast<float>(t[a]) +
                                           static_cast<float>(move[a] > 0 ? g.counts[a] - 1 - cur[a] : cur[a]) *
                                               static_cast<float>(dt[a]));
            float steps = 0.0f;
            for (int a = 0; a < 3; a++)
              if (move[a] != 0)
              {
                const float crossings = (leaves - static_cast<float>(t[a])) / static_cast<float>(dt[a]) + 1.0f;
                const float apart = static_cast<float>(end[a] > cur[a] ? end[a] - cur[a] : cur[a] - end[a]);
                steps += fminf(fmaxf(crossings, 0.0f), apart);
              }
            expected_steps = static_cast<uint32_t>(fminf(steps, 1.0e6f)) + 1u;
          }
        }
      }
    }
  }

  if constexpr (kLengthSort)
  {
    __shared__ uint32_t length_histogram[kLengthBuckets], length_start[kLengthBuckets];
    // (rays that do not walk at all: bucket 0)
    const uint32_t bucket = walking ? min(static_cast<uint32_t>(kLengthBuckets - 1), 1u + (expected_steps >> 1)) : 0u;
    const uint32_t rank = RankInWorkgroup(bucket, length_histogram, length_start);
    // the walk state goes to the thread of that rank: four 128-bit words per ray, word w of all rays together
    LdsWord4* exchange = (LdsWord4*)table_words;
    const uint32_t directions = (static_cast<int32_t>(delta0) > 0 ? 1u : 0u) | (static_cast<int32_t>(delta1) > 0 ? 2u : 0u) |
                                (static_cast<int32_t>(delta2) > 0 ? 4u : 0u) | (slot << 3);
    exchange[0 * blockDim.x + rank] = Word4{static_cast<uint32_t>(cell), static_cast<uint32_t>(delta0),
                                            static_cast<uint32_t>(delta1), static_cast<uint32_t>(delta2)};
    exchange[1 * blockDim.x + rank] = Word4{lim0, lim1, lim2, walking ? static_cast<uint32_t>(total) : 0u};
    exchange[2 * blockDim.x + rank] = Word4{__float_as_uint(t0), __float_as_uint(t1), __float_as_uint(t2), directions};
    exchange[3 * blockDim.x + rank] = Word4{__float_as_uint(dt0), __float_as_uint(dt1), __float_as_uint(dt2), 0u};
    __syncthreads();
    const Word4 a = exchange[0 * blockDim.x + threadIdx.x], b = exchange[1 * blockDim.x + threadIdx.x];
    const Word4 c = exchange[2 * blockDim.x + threadIdx.x], d = exchange[3 * blockDim.x + threadIdx.x];
    __syncthreads();
    cell = a.x, delta0 = a.y, delta1 = a.z, delta2 = a.w;
    lim0 = b.x, lim1 = b.y, lim2 = b.z, total = b.w;
    t0 = __uint_as_float(c.x), t1 = __uint_as_float(c.y), t2 = __uint_as_float(c.z);
    dt0 = __uint_as_float(d.x), dt1 = __uint_as_float(d.y), dt2 = __uint_as_float(d.z);
    slot = c.w >> 3;
    slot_step0 = table.SlotStep(0, (c.w & 1u) ? 1 : -1);
    slot_step1 = table.SlotStep(1, (c.w & 2u) ? 1 : -1);
    slot_step2 = table.SlotStep(2, (c.w & 4u) ? 1 : -1);
    walking = total != 0;
  }
  if constexpr (kTable) table.Clear();

  // The walk, in segments of kWalkSegment steps; with the table, the workgroup flushes it between segments.
  int walked = 0;
  VisitTable::Owed owed{0u, VisitTable::kNoSlot, 0u};
  for (;;)
  {
    if constexpr (kTable)
    {
      if (!__syncthreads_or(walking ? 1 : 0)) break;  // also orders Clear() / Flush() before the next inserts
    }
    else if (!walking)
      break;
    if (walking)
    {
      for (int s = 0; s < kWalkSegment; s++, walked++)
      {
        const Index here = cell;
        const uint32_t here_slot = slot;
        uint32_t lookup = 0u;
        if constexpr (kTable)
        {
          lookup = table.Lookup(here_slot);
          __builtin_amdgcn_sched_barrier(0);  // the step below is what the wave does while the LDS answers
        }
        else if (walked < kAggregatedSteps)  // `walked` is the same for every lane still in the loop
          AggregatedIncrement(tracking, static_cast<int64_t>(cell) * 2);
        else
          atomicAdd(&tracking[static_cast<int64_t>(cell) * 2], 1);
        // the axis whose boundary comes first (cuda_voxelization_helpers.cu:300-352: X if t.x is the least or tied
        // least, else Y if t.y is, else Z)
        const bool ax = (t0 <= t1) & (t0 <= t2);
        const bool ay = !ax & (t1 <= t0) & (t1 <= t2);
        const bool az = !(ax | ay);
        const uint32_t lim = ax ? lim0 : (ay ? lim1 : lim2);
        lim0 -= ax ? 1u : 0u;
        lim1 -= ay ? 1u : 0u;
        lim2 -= az ? 1u : 0u;
        t0 = ax ? t0 + dt0 : t0;
        t1 = ay ? t1 + dt1 : t1;
        t2 = az ? t2 + dt2 : t2;
        cell += ax ? delta0 : (ay ? delta1 : delta2);
        total -= 1;
        bool stop = lim == 0u || total == 0;
        if constexpr (kTable)
        {
          const uint32_t field = ax ? table.mask[0] : (ay ? table.mask[1] : table.mask[2]);
          const uint32_t moved = slot + (ax ? slot_step0 : (ay ? slot_step1 : slot_step2));
          slot = (moved & field) | (slot & ~field);
          // (the step's results are pinned here, or the compiler sinks the step below the LDS answers it should cover)
          uint32_t lim_here = lim;
          asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(lim0), "+v"(lim1), "+v"(lim2), "+v"(cell), "+v"(total),
                       "+v"(lim_here), "+v"(slot)::"memory");
          stop = lim_here == 0u || total == 0;
          __builtin_amdgcn_sched_barrier(0);
          table.Record(static_cast<uint32_t>(here), here_slot, lookup, owed, tracking);
        }
        if (stop)
        {
          walking = false;
          break;
        }
      }
    }
    if constexpr (kTable) table.Settle(owed, tracking);  // (the last step's claim)
    if constexpr (kTable)
    {
      __syncthreads();
      table.Flush(tracking);
    }
  }
}

// One thread per voxel; cells whose static occupancy is > 0.5 are left alone.
template <typename Ratio>
__global__ void FilterKernel(const int32_t* __restrict__ tracking, int64_t num_cells,
                             int32_t num_grids, Ratio percent_seen_free,
                             int32_t outlier_points_threshold, int32_t num_cameras_seen_free,
                             float* __restrict__ occupancy)
{
  for (int64_t cell = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
       cell < num_cells; cell += static_cast<int64_t>(gridDim.x) * blockDim.x)
  {
    if (!(occupancy[cell] <= 0.5f)) continue;
    int32_t seen_filled = 0, seen_free = 0;
    for (int32_t grid = 0; grid < num_grids; grid++)
    {
      const int2 counts = *reinterpret_cast<const int2*>(
          tracking + (static_cast<int64_t>(grid) * num_cells + cell) * 2);
      const int32_t free_count = counts.x;
      const int32_t filled_count = (counts.y >= outlier_points_threshold) ? counts.y : 0;
      if (free_count > 0 && filled_count > 0)
      {
        const Ratio ratio =
            static_cast<Ratio>(free_count) / static_cast<Ratio>(free_count + filled_count);
        if (ratio >= percent_seen_free)
          seen_free++;
        else
          seen_filled++;
      }
      else if (free_count > 0)
        seen_free++;
      else if (filled_count > 0)
        seen_filled++;
    }
    occupancy[cell] =
        (seen_filled > 0) ? 1.0f : ((seen_free >= num_cameras_seen_free) ? 0.0f : 0.5f);
  }
}
}  // namespace

// Device scratch for one raycast call of `num_points` points: the bins' totals and cursors, the direction-sorted
// permutation, the per-point bins.
size_t RaycastScratchBytes(int64_t num_points)
{
  if (num_points < kSortMinPoints) return 0;
  return (2 * static_cast<size_t>(kSortBins) + static_cast<size_t>(num_points) + 64) * sizeof(uint32_t) +
         static_cast<size_t>(num_points) * sizeof(uint16_t) + 1024;
}

namespace
{
// Whether the table kernel can run with `threads` threads and `table_lds` bytes of dynamic LDS on the current device:
// the attribute must be granted AND a workgroup must fit a CU with the kernel's static LDS on top (a target with less LDS
// than gfx950 grants the attribute and rejects the launch).  Decided once per device, workgroup size and table size --
// before anything of the sorted path has been launched -- and remembered; a refusal sends the call down the plain kernel.
template <typename Real>
bool TableKernelFits(int threads, size_t table_lds)
{
  struct Known
  {
    int device, threads;
    size_t lds;
    bool fits;
  };
  static std::mutex guard;
  static std::vector<Known> known;
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess)
  {
    (void)hipGetLastError();
    return false;
  }
  std::lock_guard<std::mutex> lock(guard);
  for (const Known& k : known)
    if (k.device == device && k.threads == threads && k.lds == table_lds) return k.fits;
  const void* const kernel = reinterpret_cast<const void*>(RaycastKernel<Real, true>);
  bool fits = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(table_lds)) == hipSuccess;
  if (fits)
  {
    int resident = 0;
    fits = hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, RaycastKernel<Real, true>, threads, table_lds) == hipSuccess &&
           resident > 0;
  }
  if (!fits) (void)hipGetLastError();
  known.push_back(Known{device, threads, table_lds, fits});
  return fits;
}

template <typename Real>
hipError_t LaunchRaycast(const Real* points_dev, int64_t num_points, int64_t point_stride,
                         const typename RaycastTraits<Real>::Grid& g, int32_t* tracking_dev,
                         int threads_per_block, void* scratch_dev, size_t scratch_bytes, hipStream_t stream)
{
  if (num_points <= 0) return hipSuccess;
  // threads_per_block <= 0: the caller leaves the workgroup size to the kernels (HIP_THREADS_PER_BLOCK not given)
  const int table_threads = threads_per_block > 0 ? threads_per_block : kTableThreads;
  const int plain_threads = threads_per_block > 0 ? threads_per_block : kPlainThreads;
  const int64_t num_cells = static_cast<int64_t>(g.counts[0]) * g.counts[1] * g.counts[2];
  const size_t need = RaycastScratchBytes(num_points);
  // Large clouds: order by direction and count visits per workgroup in LDS.  Needs the scratch, 32-bit
  // point and cell indices.
  // (the LDS the table kernel asks for must be granted -- checked BEFORE anything is launched, so that a refusal, e.g. a
  // large HIP_THREADS_PER_BLOCK, falls back to the plain kernel below instead of failing the call)
  int table_slots = 1024;
  while (table_slots < kTableSlotsPerThread * table_threads) table_slots <<= 1;
  const size_t table_lds = static_cast<size_t>(2 * table_slots) * sizeof(uint32_t);
  bool sorted_path = need > 0 && scratch_dev && scratch_bytes >= need && num_points < 0x7fffffffLL &&
                     num_cells < 0xffffffffLL;
  if (sorted_path) sorted_path = TableKernelFits<Real>(table_threads, table_lds);
  if (sorted_path)
  {
    const int num_chunks = static_cast<int>((num_points + kSortChunk - 1) / kSortChunk);
    uint32_t* bin_total = static_cast<uint32_t*>(scratch_dev);
    uint32_t* cursor = bin_total + kSortBins;
    uint32_t* order = cursor + kSortBins;
    uint16_t* bins = reinterpret_cast<uint16_t*>(order + ((num_points + 63) / 64 * 64));
    const hipError_t zeroed = hipMemsetAsync(bin_total, 0, 2 * static_cast<size_t>(kSortBins) * sizeof(uint32_t), stream);
    if (zeroed != hipSuccess) return zeroed;
    hipLaunchKernelGGL(DirectionBinKernel<Real>, dim3(num_chunks), dim3(kSortThreads), 0, stream, points_dev, num_points,
                       point_stride, g, bins, bin_total);
    hipLaunchKernelGGL(ScatterOrderKernel, dim3(num_chunks), dim3(kSortThreads), 0, stream, bins, num_points, bin_total,
                       cursor, order);
    const int64_t blocks = (num_points + table_threads - 1) / table_threads;
    hipLaunchKernelGGL((RaycastKernel<Real, true>), dim3(static_cast<unsigned>(blocks)), dim3(table_threads),
                       table_lds, stream, points_dev, num_points, point_stride, order, g, tracking_dev, table_slots);
    return hipGetLastError();
  }
  const int64_t blocks = (num_points + plain_threads - 1) / plain_threads;
  hipLaunchKernelGGL((RaycastKernel<Real, false>), dim3(static_cast<unsigned>(blocks)), dim3(plain_threads), 0,
                     stream, points_dev, num_points, point_stride, static_cast<const uint32_t*>(nullptr), g,
                     tracking_dev, 0);
  return hipGetLastError();
}
}  // namespace

hipError_t LaunchRaycastF32(const float* points_dev, int64_t num_points, int64_t point_stride,
                            const RaycastGridF32& g, int32_t* tracking_dev, int threads_per_block,
                            void* scratch_dev, size_t scratch_bytes, hipStream_t stream)
{
  return LaunchRaycast<float>(points_dev, num_points, point_stride, g, tracking_dev, threads_per_block, scratch_dev,
                              scratch_bytes, stream);
}

hipError_t LaunchRaycastF64(const double* points_dev, int64_t num_points, const RaycastGridF64& g,
                            int32_t* tracking_dev, int threads_per_block, void* scratch_dev, size_t scratch_bytes,
                            hipStream_t stream)
{
  return LaunchRaycast<double>(points_dev, num_points, int64_t{3}, g, tracking_dev, threads_per_block, scratch_dev,
                               scratch_bytes, stream);
}

// dst[i] += src[i]: the tracking counts of one share of a point cloud added to another's (the counts are
// integers, so the sum over shares equals the counts of the whole cloud whatever the split).
__global__ __launch_bounds__(256) void AccumulateCountsKernel(int32_t* __restrict__ dst,
                                                              const int32_t* __restrict__ src, int64_t count)
{
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const int64_t quads = count / 4;
  int4* const dst4 = reinterpret_cast<int4*>(dst);
  const int4* const src4 = reinterpret_cast<const int4*>(src);
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < quads; i += stride)
  {
    const int4 a = dst4[i];
    const int4 b = src4[i];
    dst4[i] = make_int4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
  }
  for (int64_t i = quads * 4 + static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < count; i += stride)
    dst[i] += src[i];
}

hipError_t LaunchAccumulateCounts(int32_t* dst_dev, const int32_t* src_dev, int64_t count, hipStream_t stream)
{
  if (count <= 0) return hipSuccess;
  int64_t blocks = (count / 4 + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(AccumulateCountsKernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, dst_dev,
                     src_dev, count);
  return hipGetLastError();
}

hipError_t LaunchFilter(const int32_t* tracking_dev, int64_t num_cells, int32_t num_grids,
                        double percent_seen_free, int32_t outlier_points_threshold,
                        int32_t num_cameras_seen_free, bool ratio_in_double, float* occupancy_dev,
                        int threads_per_block, hipStream_t stream)
{
  int64_t blocks = (num_cells + threads_per_block - 1) / threads_per_block;
  if (blocks > 256 * 64) blocks = 256 * 64;
  if (blocks < 1) blocks = 1;
  if (ratio_in_double)
    hipLaunchKernelGGL(FilterKernel<double>, dim3(static_cast<unsigned>(blocks)),
                       dim3(threads_per_block), 0, stream, tracking_dev, num_cells, num_grids,
                       percent_seen_free, outlier_points_threshold, num_cameras_seen_free,
                       occupancy_dev);
  else
    hipLaunchKernelGGL(FilterKernel<float>, dim3(static_cast<unsigned>(blocks)),
                       dim3(threads_per_block), 0, stream, tracking_dev, num_cells, num_grids,
                       static_cast<float>(percent_seen_free), outlier_points_threshold,
                       num_cameras_seen_free, occupancy_dev);
  return hipGetLastError();
}
}  // namespace vgt
