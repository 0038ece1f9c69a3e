// Exact signed Euclidean distance transform for gfx950 (MI355X): kernels and launchers.
//
// What is computed (reference: OccupancyMap::ExtractSignedDistanceField<float>,
// include/voxelized_geometry_tools/occupancy_map.hpp:174-210 ->
// signed_distance_field_generation.hpp:39-113 -> signed_distance_field_generation.cpp:258-391):
// for every voxel the exact squared Euclidean distance (integer, voxel units) to the nearest
// voxel of the OTHER class (filled vs free), turned into
//     sdf = float( sqrt(double(d2)) * resolution ), negated on filled voxels,
// +-inf when the other class is absent.  The reference runs two separate double-precision
// Felzenszwalb-Huttenlocher transforms (one per class) in X,Y,Z order; the result is
// order-independent and integral, so here both classes travel through three passes as ONE
// signed integer field (a voxel only ever needs the distance to the other class, and a voxel of
// the other class is a zero-valued site):
//     pass 1  Z (contiguous axis): nearest-site scan on the binarised input, wave ballots
//     pass 2  Y: lower envelope of parabolas over the squared pass-1 distances
//     pass 3  X: same, fused with sqrt / resolution / sign / virtual border / min-max.
//
// The default pipeline is pass 1 = class records (edt_record_kernels.hip), passes 2 and 3 = lane-per-line sweeps
// (edt_sweep_kernels.hip).  This file holds what the pipelines share (extrema, slab carries, the dispatch by variant)
// and, in VGT_HIP_TESTING builds only (libvgt_hip_testing.so, used by the parity tests to cross-check the default
// against an independent implementation), the int16 Z scan and the pruned-search line passes (EdtVariant::kBruteForce).
#include "edt_device.hpp"

#include <atomic>

namespace vgt
{
namespace
{
#ifdef VGT_HIP_TESTING
constexpr int kWave = kWaveSize;
constexpr int kScanBlock = 256;
constexpr int kScanWaves = kScanBlock / kWave;
constexpr int kMaxChunks = static_cast<int>(kMaxExtent / kWave);

// is_filled predicate of OccupancyMap (occupancy_map.hpp:181-205).
__device__ __forceinline__ bool IsFilled(float occupancy, int unknown_is_filled)
{
  return (occupancy > 0.5f) || (unknown_is_filled && (occupancy == 0.5f));
}
__device__ __forceinline__ bool IsFilled(uint8_t mask, int) { return mask != 0; }

// Slab summary halves (vgt_internal.hpp): `boundary` = slab-local z of the slab's first (last) voxel, at_filled /
// at_free = slab-local z of the first (last) voxel of each class, -1 when absent.
__device__ __forceinline__ uint16_t SummaryHalf(int at_filled, int at_free, int boundary, int z_offset)
{
  const bool filled = at_filled == boundary;
  const int other = filled ? at_free : at_filled;
  return static_cast<uint16_t>((filled ? kSlabFilledBit : 0u) |
                               (other < 0 ? kSlabNone : static_cast<uint16_t>(other + z_offset)));
}

// ---------------------------------------------------------------------------------------------
// Pass 1: one wave per Z line.  Each 64-voxel chunk becomes one ballot mask; a voxel's distance
// to the nearest voxel of the other class is a clz/ffs on that mask, falling back to the nearest
// such voxel in the chunks before / after (carried as scalars).  Input is read exactly once.
// ---------------------------------------------------------------------------------------------
template <typename InT>
__global__ __launch_bounds__(kScanBlock) void ScanZKernel(const InT* __restrict__ in,
                                                         int16_t* __restrict__ out,
                                                         int64_t num_lines, int nz,
                                                         int unknown_is_filled,
                                                         SlabLineSummary* __restrict__ summary,
                                                         int z_offset)
{
  // [wave][chunk]: ballot of "filled", then first position >= chunk end holding a filled /
  // free voxel (or -1).  Written and read by the same wave only.
  __shared__ uint64_t s_filled[kScanWaves][kMaxChunks];
  __shared__ int32_t s_next_filled[kScanWaves][kMaxChunks];
  __shared__ int32_t s_next_free[kScanWaves][kMaxChunks];

  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const int nchunks = (nz + kWave - 1) / kWave;
  volatile uint64_t* filled = s_filled[wave];
  volatile int32_t* next_filled = s_next_filled[wave];
  volatile int32_t* next_free = s_next_free[wave];

  for (int64_t line = static_cast<int64_t>(blockIdx.x) * kScanWaves + wave; line < num_lines;
       line += static_cast<int64_t>(gridDim.x) * kScanWaves)
  {
    const InT* src = in + line * nz;
    int16_t* dst = out + line * nz;

    for (int c = 0; c < nchunks; c++)
    {
      const int z = c * kWave + lane;
      const bool f = (z < nz) && IsFilled(src[z], unknown_is_filled);
      const uint64_t m = __ballot(f);
      if (lane == 0) filled[c] = m;
    }
    __builtin_amdgcn_wave_barrier();

    // Backward sweep (uniform per wave): nearest filled / free voxel after each chunk.
    if (lane == 0)
    {
      int32_t nf = -1, ne = -1;
      for (int c = nchunks - 1; c >= 0; c--)
      {
        next_filled[c] = nf;
        next_free[c] = ne;
        const int rem = nz - c * kWave;
        const uint64_t valid = (rem >= kWave) ? ~0ull : ((1ull << rem) - 1ull);
        const uint64_t F = filled[c];
        const uint64_t E = ~F & valid;
        if (F) nf = c * kWave + (__ffsll(static_cast<long long>(F)) - 1);
        if (E) ne = c * kWave + (__ffsll(static_cast<long long>(E)) - 1);
      }
      if (summary) summary[line].first = SummaryHalf(nf, ne, 0, z_offset);
    }
    __builtin_amdgcn_wave_barrier();

    // Forward sweep: per-lane distances.
    int32_t prev_filled = -1, prev_free = -1;  // last filled / free position before this chunk
    for (int c = 0; c < nchunks; c++)
    {
      const int rem = nz - c * kWave;
      const uint64_t valid = (rem >= kWave) ? ~0ull : ((1ull << rem) - 1ull);
      const uint64_t F = filled[c];
      const uint64_t E = ~F & valid;
      const int z = c * kWave + lane;
      if (z < nz)
      {
        const bool is_filled = (F >> lane) & 1ull;
        const uint64_t other = is_filled ? E : F;
        const int32_t prev_other = is_filled ? prev_free : prev_filled;
        const int32_t next_other = is_filled ? next_free[c] : next_filled[c];
        const uint64_t below = other & ((1ull << lane) - 1ull);
        const uint64_t above = (lane == kWave - 1) ? 0ull : (other >> (lane + 1));
        int32_t d_below = kInf16, d_above = kInf16;
        if (below)
          d_below = lane - (63 - __clzll(static_cast<long long>(below)));
        else if (prev_other >= 0)
          d_below = z - prev_other;
        if (above)
          d_above = __ffsll(static_cast<long long>(above));
        else if (next_other >= 0)
          d_above = next_other - z;
        const int32_t d = min(d_below, d_above);
        dst[z] = static_cast<int16_t>(is_filled ? -d : d);
      }
      if (F) prev_filled = c * kWave + (63 - __clzll(static_cast<long long>(F)));
      if (E) prev_free = c * kWave + (63 - __clzll(static_cast<long long>(E)));
    }
    if (summary && lane == 0) summary[line].last = SummaryHalf(prev_filled, prev_free, nz - 1, z_offset);
    __builtin_amdgcn_wave_barrier();
  }
}

// Multi-GPU: a voxel's distance along Z to the other class is the minimum of the slab-local
// distance and the distances to the nearest such voxel in the slabs below / above.
__global__ __launch_bounds__(256) void SlabFixupKernel(int16_t* __restrict__ io,
                                                      const SlabLineCarry* __restrict__ carries,
                                                      int64_t total, int nz, int z_offset)
{
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
  {
    const int64_t line = i / nz;
    const int z = static_cast<int>(i - line * nz) + z_offset;
    const SlabLineCarry c = carries[line];
    const int16_t v = io[i];
    const bool filled = v < 0;
    int32_t d = filled ? -static_cast<int32_t>(v) : static_cast<int32_t>(v);
    const int prev_other = filled ? c.prev_free : c.prev_filled;
    const int next_other = filled ? c.next_free : c.next_filled;
    if (prev_other >= 0) d = min(d, z - prev_other);
    if (next_other >= 0) d = min(d, next_other - z);
    io[i] = static_cast<int16_t>(filled ? -d : d);
  }
}

#endif  // VGT_HIP_TESTING
// Multi-GPU: per-line carries of slab `rank` from the gathered summaries of all slabs
// (summaries[slab][line], 4 bytes each, see vgt_internal.hpp): nearest filled / free voxel below = the last such
// voxel of the nearest lower slab that has one, above = the first such voxel of the nearest upper slab (-1 when
// absent).  A slab's first (last) voxel of one class is its first (last) voxel; where that is follows from SlabRange.
__global__ __launch_bounds__(256) void SlabCarriesKernel(const SlabLineSummary* __restrict__ summaries,
                                                        int world, int rank, int64_t lines, int nz_global,
                                                        SlabLineCarry* __restrict__ carries)
{
  const int share = nz_global / world, extra = nz_global % world;
  for (int64_t line = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; line < lines;
       line += static_cast<int64_t>(gridDim.x) * blockDim.x)
  {
    int prev_filled = -1, prev_free = -1, next_filled = -1, next_free = -1;
    for (int r = 0; r < rank; r++)
    {
      const uint16_t rec = summaries[static_cast<int64_t>(r) * lines + line].last;
      const int end = (r + 1) * share + min(r + 1, extra) - 1;  // the slab's last voxel
      const int other = (rec & kSlabNone) == kSlabNone ? -1 : static_cast<int>(rec & kSlabNone);
      const bool filled = (rec & kSlabFilledBit) != 0;
      prev_filled = max(prev_filled, filled ? end : other);
      prev_free = max(prev_free, filled ? other : end);
    }
    for (int r = world - 1; r > rank; r--)
    {
      const uint16_t rec = summaries[static_cast<int64_t>(r) * lines + line].first;
      const int begin = r * share + min(r, extra);  // the slab's first voxel
      const int other = (rec & kSlabNone) == kSlabNone ? -1 : static_cast<int>(rec & kSlabNone);
      const bool filled = (rec & kSlabFilledBit) != 0;
      const int first_filled = filled ? begin : other, first_free = filled ? other : begin;
      if (first_filled >= 0) next_filled = first_filled;
      if (first_free >= 0) next_free = first_free;
    }
    SlabLineCarry c;
    c.prev_filled = static_cast<int16_t>(prev_filled);
    c.next_filled = static_cast<int16_t>(next_filled);
    c.prev_free = static_cast<int16_t>(prev_free);
    c.next_free = static_cast<int16_t>(next_free);
    carries[line] = c;
  }
}

#ifdef VGT_HIP_TESTING
// Exact 1-D lower-envelope value at position q by outward search with pruning: a site at
// offset k can only improve the answer while k*k < best, and the first voxel of the other
// class (a zero-valued site) ends the search on both sides.  O(sqrt(answer)) per voxel.
template <typename InT>
__device__ __forceinline__ int32_t LineSearch(const InT* __restrict__ centre, int64_t stride,
                                              int q, int n, bool negative, int32_t own)
{
  int32_t best = own;
  for (int k = 1; k < n; k++)
  {
    const int32_t kk = k * k;
    if (kk >= best) break;
    const bool has_lo = (q - k) >= 0;
    const bool has_hi = (q + k) < n;
    if (!has_lo && !has_hi) break;
    if (has_lo)
    {
      bool neg;
      int32_t f;
      Decode(centre[-static_cast<int64_t>(k) * stride], neg, f);
      const int32_t cand = (neg != negative) ? kk : ((f == kInf32) ? kInf32 : kk + f);
      best = min(best, cand);
    }
    if (has_hi)
    {
      bool neg;
      int32_t f;
      Decode(centre[static_cast<int64_t>(k) * stride], neg, f);
      const int32_t cand = (neg != negative) ? kk : ((f == kInf32) ? kInf32 : kk + f);
      best = min(best, cand);
    }
  }
  return best;
}

__global__ __launch_bounds__(256) void PassYBruteKernel(const int16_t* __restrict__ in,
                                                       int32_t* __restrict__ out, int64_t total,
                                                       int ny, int nz)
{
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
  {
    const int y = static_cast<int>((i / nz) % ny);
    bool negative;
    int32_t own;
    Decode(in[i], negative, own);
    const int32_t best = LineSearch(in + i, static_cast<int64_t>(nz), y, ny, negative, own);
    out[i] = negative ? -best : best;
  }
}

__global__ __launch_bounds__(256) void PassXBruteFinalizeKernel(
    const int32_t* __restrict__ in, float* __restrict__ sdf, uint32_t* __restrict__ minmax_enc,
    int64_t total, int nx, int ny, int nz, double resolution, int add_virtual_border, int z_offset,
    int nz_global)
{
  uint32_t lo = 0xffffffffu, hi = 0u;
  const int64_t plane = static_cast<int64_t>(ny) * nz;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
  {
    const int x = static_cast<int>(i / plane);
    const int64_t r = i - static_cast<int64_t>(x) * plane;
    const int y = static_cast<int>(r / nz);
    const int z = static_cast<int>(r - static_cast<int64_t>(y) * nz);
    bool negative;
    int32_t own;
    Decode(in[i], negative, own);
    const int32_t best = LineSearch(in + i, plane, x, nx, negative, own);
    const float v = FinalizeSdf(best, negative, x, y, z + z_offset, nx, ny, nz_global, resolution,
                                add_virtual_border);
    sdf[i] = v;
    const uint32_t e = EncodeOrdered(v);
    lo = min(lo, e);
    hi = max(hi, e);
  }
  BlockMinMax(lo, hi, minmax_enc);
}

#endif  // VGT_HIP_TESTING
__global__ void InitMinMaxKernel(uint32_t* minmax_enc, int count)
{
  for (int i = static_cast<int>(threadIdx.x); i < count; i += static_cast<int>(blockDim.x))
  {
    minmax_enc[2 * i] = 0xffffffffu;
    minmax_enc[2 * i + 1] = 0u;
  }
}
__global__ void DecodeMinMaxKernel(const uint32_t* minmax_enc, float* out, int count)
{
  for (int i = static_cast<int>(threadIdx.x); i < 2 * count; i += static_cast<int>(blockDim.x))
    out[i] = DecodeOrdered(minmax_enc[i]);
}

int GridFor(int64_t work_items, int block)
{
  // Memory-bound grid-stride launches: enough blocks to fill 256 CUs several times over.
  const int64_t blocks = (work_items + block - 1) / block;
  const int64_t cap = 256 * 32;
  return static_cast<int>(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
}
}  // namespace

// Defined in edt_sweep_kernels.hip.
hipError_t LaunchPassXSweepFinalizeRange(const int32_t* in32, float* sdf, uint32_t* minmax_enc, SweepScratch scratch,
                                         const SdfParams& p, int64_t outer_begin, int64_t outer_count_or_all,
                                         hipStream_t stream);
#ifdef VGT_HIP_TESTING
namespace
{
template <typename InT>
hipError_t LaunchScanZ(const InT* in, int16_t* out16, const SdfParams& p, int unknown_is_filled,
                       SlabLineSummary* summary, hipStream_t stream)
{
  const int z_offset = static_cast<int>(p.z_offset);
  const int64_t lines = p.nx * p.ny;
  const int nz = static_cast<int>(p.nz);
  const int grid = GridFor(lines, kScanWaves);
  hipLaunchKernelGGL(ScanZKernel<InT>, dim3(grid), dim3(kScanBlock), 0, stream, in, out16, lines, nz, unknown_is_filled,
                     summary, z_offset);
  return hipGetLastError();
}
}  // namespace

hipError_t LaunchScanZFromOccupancy(const float* occupancy, int16_t* out16, const SdfParams& p,
                                    SlabLineSummary* summary, hipStream_t stream)
{
  return LaunchScanZ<float>(occupancy, out16, p, p.unknown_is_filled, summary, stream);
}

hipError_t LaunchScanZFromMask(const uint8_t* mask, int16_t* out16, const SdfParams& p,
                               SlabLineSummary* summary, hipStream_t stream)
{
  return LaunchScanZ<uint8_t>(mask, out16, p, 0, summary, stream);
}

namespace
{
// Diagnostic: compares the fast final conversion with the exact one over a range of squared
// distances; result[0] = number of differing values, result[1] = first differing d2 (or ~0).
__global__ __launch_bounds__(256) void FinalizeCheckKernel(int64_t first, int64_t count, double resolution,
                                                          unsigned long long* __restrict__ result)
{
  unsigned long long bad = 0, first_bad = ~0ull;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < count;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
  {
    const int32_t d2 = static_cast<int32_t>(first + i);
    const float fast = SqrtTimesResolution(d2, resolution);
    const float exact = SqrtTimesResolutionExact(d2, resolution);
    if (__float_as_uint(fast) != __float_as_uint(exact))
    {
      bad++;
      first_bad = min(first_bad, static_cast<unsigned long long>(d2));
    }
  }
  if (bad)
  {
    atomicAdd(&result[0], bad);
    atomicMin(&result[1], first_bad);
  }
}
}  // namespace

hipError_t LaunchFinalizeCheck(int64_t first, int64_t count, double resolution,
                               unsigned long long* result_dev, hipStream_t stream)
{
  hipLaunchKernelGGL(FinalizeCheckKernel, dim3(4096), dim3(256), 0, stream, first, count, resolution,
                     result_dev);
  return hipGetLastError();
}

#endif  // VGT_HIP_TESTING
hipError_t LaunchSlabCarries(const SlabLineSummary* summaries, int world, int rank, int64_t lines, int64_t nz_global,
                             SlabLineCarry* carries, hipStream_t stream)
{
  hipLaunchKernelGGL(SlabCarriesKernel, dim3(GridFor(lines, 256)), dim3(256), 0, stream, summaries, world, rank,
                     lines, static_cast<int>(nz_global), carries);
  return hipGetLastError();
}

#ifdef VGT_HIP_TESTING
hipError_t LaunchSlabFixup(int16_t* io16, const SlabLineCarry* carries, const SdfParams& p,
                           hipStream_t stream)
{
  const int64_t total = p.nx * p.ny * p.nz;
  hipLaunchKernelGGL(SlabFixupKernel, dim3(GridFor(total, 256)), dim3(256), 0, stream, io16,
                     carries, total, static_cast<int>(p.nz), static_cast<int>(p.z_offset));
  return hipGetLastError();
}

hipError_t LaunchPassY(const int16_t* in16, int32_t* out32, SweepScratch scratch, const SdfParams& p,
                       EdtVariant variant, hipStream_t stream)
{
  (void)scratch;
  (void)variant;  // (kBruteForce: the one cross-check pipeline)
  const int64_t total = p.nx * p.ny * p.nz;
  hipLaunchKernelGGL(PassYBruteKernel, dim3(GridFor(total, 256)), dim3(256), 0, stream, in16,
                     out32, total, static_cast<int>(p.ny), static_cast<int>(p.nz));
  return hipGetLastError();
}

#endif  // VGT_HIP_TESTING
hipError_t LaunchPassXFinalize(const int32_t* in32, float* sdf, uint32_t* minmax_enc,
                               SweepScratch scratch, const SdfParams& p, EdtVariant variant,
                               hipStream_t stream)
{
  // (The X pass keeps the sweeps beyond 64 rows even when a launch has few items: measured equal at 80 - 128 rows,
  // profiles/r5/short_vs_sweep.txt -- its rows pay for the final conversion either way.  The Y pass gains a third there.)
  if (variant == EdtVariant::kDefault && p.nx <= ShortLineRows())
    return LaunchPassXShortFinalizeRange(in32, sdf, minmax_enc, p, 0, -1, stream);
  if (variant == EdtVariant::kDefault)
    return LaunchPassXSweepFinalizeRange(in32, sdf, minmax_enc, scratch, p, 0, -1, stream);
#ifdef VGT_HIP_TESTING
  const int64_t total = p.nx * p.ny * p.nz;
  hipLaunchKernelGGL(PassXBruteFinalizeKernel, dim3(GridFor(total, 256)), dim3(256), 0, stream,
                     in32, sdf, minmax_enc, total, static_cast<int>(p.nx), static_cast<int>(p.ny),
                     static_cast<int>(p.nz), p.resolution, p.add_virtual_border,
                     static_cast<int>(p.z_offset),
                     static_cast<int>(p.nz_global > 0 ? p.nz_global : p.nz));
  return hipGetLastError();
#else
  return hipErrorInvalidValue;  // (the cross-check variants are not part of this build)
#endif
}

bool LinePassesTakeRanges(const SdfParams& p, EdtVariant variant)
{
  (void)p;
  return variant == EdtVariant::kDefault;
}

hipError_t LaunchPassXFinalizeRange(const int32_t* in32, float* sdf, uint32_t* minmax_enc, SweepScratch scratch,
                                    const SdfParams& p, EdtVariant variant, int64_t outer_begin, int64_t outer_count,
                                    hipStream_t stream)
{
  if (variant == EdtVariant::kDefault && p.nx <= ShortLineRows())
    return LaunchPassXShortFinalizeRange(in32, sdf, minmax_enc, p, outer_begin, outer_count, stream);
  if (variant == EdtVariant::kDefault)
    return LaunchPassXSweepFinalizeRange(in32, sdf, minmax_enc, scratch, p, outer_begin, outer_count, stream);
  return hipErrorInvalidValue;  // (the cross-check pipeline takes whole grids)
}

#ifdef VGT_HIP_TESTING
std::atomic<int> g_short_line_override{-1};
int ShortLineOverride() { return g_short_line_override.load(); }
void SetShortLineRows(int rows)
{
  g_short_line_override.store(rows < 0 ? -1 : (rows > kShortLineRowsFewItems ? kShortLineRowsFewItems : rows));
}
#else
int ShortLineOverride() { return -1; }
#endif

hipError_t LaunchInitMinMax(uint32_t* minmax_enc, hipStream_t stream, int64_t count)
{
  hipLaunchKernelGGL(InitMinMaxKernel, dim3(1), dim3(count > 1 ? 64 : 1), 0, stream, minmax_enc, static_cast<int>(count));
  return hipGetLastError();
}

hipError_t LaunchDecodeMinMax(const uint32_t* minmax_enc, float* minmax_out, hipStream_t stream, int64_t count)
{
  hipLaunchKernelGGL(DecodeMinMaxKernel, dim3(1), dim3(count > 1 ? 64 : 1), 0, stream, minmax_enc, minmax_out,
                     static_cast<int>(count));
  return hipGetLastError();
}
}  // namespace vgt
