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
// against independent implementations), the int16 Z scan and the pruned-search line passes (EdtVariant::kBruteForce);
// the LDS-tiled lower-envelope passes (variant 2) live in edt_hull_kernels.hip, also a testing-only file.
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

// Fast path for nz <= 64 * NCHUNK: all loads of a line are issued before the first ballot (NCHUNK
// independent 256-byte wave loads in flight), the chunk masks and carries live in scalar registers.
template <typename InT, int NCHUNK>
__global__ __launch_bounds__(kScanBlock) void ScanZUnrolledKernel(
    const InT* __restrict__ in, int16_t* __restrict__ out, int64_t num_lines, int nz,
    int unknown_is_filled, SlabLineSummary* __restrict__ summary, int z_offset)
{
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  for (int64_t line = static_cast<int64_t>(blockIdx.x) * kScanWaves + wave; line < num_lines;
       line += static_cast<int64_t>(gridDim.x) * kScanWaves)
  {
    const InT* src = in + line * nz;
    int16_t* dst = out + line * nz;
    InT v[NCHUNK];
#pragma unroll
    for (int c = 0; c < NCHUNK; c++)
    {
      const int z = c * kWave + lane;
      v[c] = (z < nz) ? src[z] : InT(0);
    }
    uint64_t F[NCHUNK], E[NCHUNK];
#pragma unroll
    for (int c = 0; c < NCHUNK; c++)
    {
      const int z = c * kWave + lane;
      F[c] = __ballot((z < nz) && IsFilled(v[c], unknown_is_filled));
      const int rem = nz - c * kWave;
      const uint64_t valid = (rem >= kWave) ? ~0ull : ((rem <= 0) ? 0ull : ((1ull << rem) - 1ull));
      E[c] = ~F[c] & valid;
    }
    int32_t next_filled[NCHUNK], next_free[NCHUNK];
    int32_t nf = -1, ne = -1;
#pragma unroll
    for (int c = NCHUNK - 1; c >= 0; c--)
    {
      next_filled[c] = nf;
      next_free[c] = ne;
      if (F[c]) nf = c * kWave + (__ffsll(static_cast<long long>(F[c])) - 1);
      if (E[c]) ne = c * kWave + (__ffsll(static_cast<long long>(E[c])) - 1);
    }
    int32_t prev_filled = -1, prev_free = -1;
#pragma unroll
    for (int c = 0; c < NCHUNK; c++)
    {
      const int z = c * kWave + lane;
      if (z < nz)
      {
        const bool is_filled = (F[c] >> lane) & 1ull;
        const uint64_t other = is_filled ? E[c] : F[c];
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
      if (F[c]) prev_filled = c * kWave + (63 - __clzll(static_cast<long long>(F[c])));
      if (E[c]) prev_free = c * kWave + (63 - __clzll(static_cast<long long>(E[c])));
    }
    if (summary && lane == 0)
    {
      SlabLineSummary out_summary;
      out_summary.first = SummaryHalf(nf, ne, 0, z_offset);
      out_summary.last = SummaryHalf(prev_filled, prev_free, nz - 1, z_offset);
      summary[line] = out_summary;
    }
  }
}

// Main path for nz % 4 == 0, nz <= 256 * NCH: every lane owns FOUR consecutive voxels (one 16-byte
// load, one 8-byte store), so a wave covers 256 voxels per chunk.  All loads of a line are in flight before
// the first ballot.
template <typename InT>
struct QuadOf;
template <>
struct QuadOf<float>
{
  using type = float4;
};
template <>
struct QuadOf<uint8_t>
{
  using type = uchar4;
};

// The scan works on class TRANSITIONS: the nearest voxel of the other class below voxel z is the voxel at the
// last index i < z with class(i) != class(i + 1), the nearest above is i + 1 for the first such i >= z -- one
// transition set serves both classes, so the kernel has no per-class masks, selects or carries.  The class of a
// voxel and the transitions are per-lane predicates (lane masks in scalar registers, combined by the scalar
// unit); every lane publishes the first and the last transition of its own four voxels, the nearest transition
// outside the quad is fetched from the nearest lane that has one (one ballot, two ds_bpermute per quad), and the
// four voxels take running selects over their quad's three inner transitions.
template <typename InT, int NCH>
__global__ __launch_bounds__(kScanBlock) void ScanZTransitionKernel(
    const InT* __restrict__ in, int16_t* __restrict__ out, int64_t num_lines, int nz,
    int unknown_is_filled, SlabLineSummary* __restrict__ summary, int z_offset)
{
  using Vec = typename QuadOf<InT>::type;
  constexpr int kNoneBelow = -40000, kNoneAbove = 80000;  // distances from these exceed kInf16
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  const uint64_t gt_mask = (lane == kWave - 1) ? 0ull : (~0ull << (lane + 1));
  for (int64_t line = static_cast<int64_t>(blockIdx.x) * kScanWaves + wave; line < num_lines;
       line += static_cast<int64_t>(gridDim.x) * kScanWaves)
  {
    const Vec* src = reinterpret_cast<const Vec*>(in + line * nz);
    short4* dst = reinterpret_cast<short4*>(out + line * nz);
    Vec v[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++)
    {
      const int grp = c * kWave + lane;
      if (grp * 4 < nz)
      {
        // (the field is read once: a streaming load, which also leaves the caches to the passes that follow)
        using Raw = InT __attribute__((ext_vector_type(4)));
        const Raw raw = __builtin_nontemporal_load(reinterpret_cast<const Raw*>(src) + grp);
        v[c].x = raw.x;
        v[c].y = raw.y;
        v[c].z = raw.z;
        v[c].w = raw.w;
      }
    }
    bool f[NCH][4];     // class of my four voxels (false beyond the line)
    int first0[NCH];    // class of the first voxel of my quad as an integer, for the lane below me
#pragma unroll
    for (int c = 0; c < NCH; c++)
    {
      const bool valid = (c * kWave + lane) * 4 < nz;
      f[c][0] = valid && IsFilled(v[c].x, unknown_is_filled);
      f[c][1] = valid && IsFilled(v[c].y, unknown_is_filled);
      f[c][2] = valid && IsFilled(v[c].z, unknown_is_filled);
      f[c][3] = valid && IsFilled(v[c].w, unknown_is_filled);
      first0[c] = f[c][0] ? 1 : 0;
    }
    bool t[NCH][4];     // t[c][k]: voxels base + k and base + k + 1 exist and differ in class
    uint64_t HT[NCH];   // lanes whose quad holds a transition
    int own_first[NCH], own_last[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++)
    {
      const int base = (c * kWave + lane) * 4;
      // class of the voxel after my quad: the first voxel of the next lane (of the next chunk for lane 63)
      int next0 = __shfl_down(first0[c], 1);
      if (c + 1 < NCH)
      {
        const int wrap = __builtin_amdgcn_readfirstlane(first0[c + 1]);
        next0 = (lane == kWave - 1) ? wrap : next0;
      }
      t[c][0] = (base + 1 < nz) && (f[c][0] != f[c][1]);
      t[c][1] = (base + 2 < nz) && (f[c][1] != f[c][2]);
      t[c][2] = (base + 3 < nz) && (f[c][2] != f[c][3]);
      t[c][3] = (base + 4 < nz) && (f[c][3] != (next0 != 0));
      HT[c] = __ballot(t[c][0] || t[c][1] || t[c][2] || t[c][3]);
      own_first[c] = t[c][0] ? base : (t[c][1] ? base + 1 : (t[c][2] ? base + 2 : base + 3));
      own_last[c] = t[c][3] ? base + 3 : (t[c][2] ? base + 2 : (t[c][1] ? base + 1 : base));
    }
    // scalar carries across chunks: first transition after the chunk, last transition before it
    int32_t next_t[NCH], prev_t[NCH];
    int32_t nt = kNoneAbove;
#pragma unroll
    for (int c = NCH - 1; c >= 0; c--)
    {
      next_t[c] = nt;
      if (HT[c]) nt = __builtin_amdgcn_readlane(own_first[c], __ffsll(static_cast<long long>(HT[c])) - 1);
    }
    int32_t pt = kNoneBelow;
#pragma unroll
    for (int c = 0; c < NCH; c++)
    {
      prev_t[c] = pt;
      if (HT[c]) pt = __builtin_amdgcn_readlane(own_last[c], 63 - __clzll(static_cast<long long>(HT[c])));
    }
#pragma unroll
    for (int c = 0; c < NCH; c++)
    {
      const int grp = c * kWave + lane;
      const int base = grp * 4;
      // nearest transition outside my quad, below and above
      const uint64_t tbelow = HT[c] & lt_mask, tabove = HT[c] & gt_mask;
      const int g_b = tbelow ? 63 - __clzll(static_cast<long long>(tbelow)) : lane;
      const int g_a = tabove ? __ffsll(static_cast<long long>(tabove)) - 1 : lane;
      const int32_t from_b = __shfl(own_last[c], g_b);
      const int32_t from_a = __shfl(own_first[c], g_a);
      const int32_t Pb = tbelow ? from_b : prev_t[c];
      const int32_t Pa = tabove ? from_a : next_t[c];
      if (base < nz)
      {
        // last transition below voxel k / first transition at or above it
        const int32_t b0 = Pb;
        const int32_t b1 = t[c][0] ? base : b0;
        const int32_t b2 = t[c][1] ? base + 1 : b1;
        const int32_t b3 = t[c][2] ? base + 2 : b2;
        const int32_t a3 = t[c][3] ? base + 3 : Pa;
        const int32_t a2 = t[c][2] ? base + 2 : a3;
        const int32_t a1 = t[c][1] ? base + 1 : a2;
        const int32_t a0 = t[c][0] ? base : a1;
        const int32_t below[4] = {b0, b1, b2, b3};
        const int32_t above[4] = {a0, a1, a2, a3};
        int16_t r[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
          const int32_t z = base + k;
          // the other class sits AT the transition below and one voxel past the transition above
          const int32_t d = min(min(z - below[k], above[k] + 1 - z), static_cast<int32_t>(kInf16));
          r[k] = static_cast<int16_t>(f[c][k] ? -d : d);
        }
        using RawOut = int16_t __attribute__((ext_vector_type(4)));
        RawOut packed;
        packed.x = r[0];
        packed.y = r[1];
        packed.z = r[2];
        packed.w = r[3];
        __builtin_nontemporal_store(packed, reinterpret_cast<RawOut*>(dst) + grp);
      }
    }
    if (summary)
    {
      // slab summaries (multi-GPU): first / last voxel of either class, from the class masks
      int32_t ff = -1, lf = -1, fe = -1, le = -1;
#pragma unroll
      for (int c = NCH - 1; c >= 0; c--)
      {
#pragma unroll
        for (int k = 3; k >= 0; k--)
        {
          const bool valid = (c * kWave + lane) * 4 + k < nz;
          const uint64_t F = __ballot(f[c][k]);
          const uint64_t E = __ballot(valid && !f[c][k]);
          // descending (c, k, lane-within-mask is handled by taking the lowest lane): keep the smallest position
          if (F)
          {
            const int pos = (c * kWave + __ffsll(static_cast<long long>(F)) - 1) * 4 + k;
            ff = (ff < 0 || pos < ff) ? pos : ff;
            const int last = (c * kWave + 63 - __clzll(static_cast<long long>(F))) * 4 + k;
            lf = (last > lf) ? last : lf;
          }
          if (E)
          {
            const int pos = (c * kWave + __ffsll(static_cast<long long>(E)) - 1) * 4 + k;
            fe = (fe < 0 || pos < fe) ? pos : fe;
            const int last = (c * kWave + 63 - __clzll(static_cast<long long>(E))) * 4 + k;
            le = (last > le) ? last : le;
          }
        }
      }
      if (lane == 0)
      {
        SlabLineSummary out_summary;
        out_summary.first = SummaryHalf(ff, fe, 0, z_offset);
        out_summary.last = SummaryHalf(lf, le, nz - 1, z_offset);
        summary[line] = out_summary;
      }
    }
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
// Same, eight consecutive voxels of one line per thread (nz % 8 == 0): one 16-byte load, one
// carry record, and a store only when a distance actually shrank.
__global__ __launch_bounds__(256) void SlabFixupVecKernel(int16_t* __restrict__ io,
                                                         const SlabLineCarry* __restrict__ carries,
                                                         int64_t total_groups, int groups_per_line,
                                                         int z_offset)
{
  for (int64_t gidx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; gidx < total_groups;
       gidx += static_cast<int64_t>(gridDim.x) * blockDim.x)
  {
    const int64_t line = gidx / groups_per_line;
    const int z0 = static_cast<int>(gidx - line * groups_per_line) * 8 + z_offset;
    const SlabLineCarry c = carries[line];
    uint4* ptr = reinterpret_cast<uint4*>(io) + gidx;
    using Raw = uint32_t __attribute__((ext_vector_type(4)));
    const Raw raw = __builtin_nontemporal_load(reinterpret_cast<const Raw*>(ptr));  // read once: streaming
    uint32_t words[4] = {raw.x, raw.y, raw.z, raw.w};
    bool changed = false;
#pragma unroll
    for (int k = 0; k < 8; k++)
    {
      const int16_t v = static_cast<int16_t>((words[k >> 1] >> ((k & 1) * 16)) & 0xffffu);
      const bool filled = v < 0;
      int32_t d = filled ? -static_cast<int32_t>(v) : static_cast<int32_t>(v);
      const int prev_other = filled ? c.prev_free : c.prev_filled;
      const int next_other = filled ? c.next_free : c.next_filled;
      const int z = z0 + k;
      int32_t nd = d;
      if (prev_other >= 0) nd = min(nd, z - prev_other);
      if (next_other >= 0) nd = min(nd, next_other - z);
      changed |= (nd != d);
      const uint32_t enc = static_cast<uint32_t>(static_cast<uint16_t>(static_cast<int16_t>(filled ? -nd : nd)));
      words[k >> 1] = (words[k >> 1] & ~(0xffffu << ((k & 1) * 16))) | (enc << ((k & 1) * 16));
    }
    if (changed) *ptr = make_uint4(words[0], words[1], words[2], words[3]);
  }
}

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

// Defined in edt_sweep_kernels.hip / edt_hull_kernels.hip.
#ifdef VGT_HIP_TESTING
hipError_t LaunchPassYSweep(const int16_t* in16, int32_t* out32, SweepScratch scratch, const SdfParams& p,
                            hipStream_t stream);
#endif  // VGT_HIP_TESTING
hipError_t LaunchPassXSweepFinalizeRange(const int32_t* in32, float* sdf, uint32_t* minmax_enc, SweepScratch scratch,
                                         const SdfParams& p, int64_t outer_begin, int64_t outer_count_or_all,
                                         hipStream_t stream);
#ifdef VGT_HIP_TESTING
hipError_t LaunchPassYHull(const int16_t* in16, int32_t* out32, const SdfParams& p,
                           hipStream_t stream, bool* handled);
hipError_t LaunchPassXHullFinalize(const int32_t* in32, float* sdf, uint32_t* minmax_enc,
                                   const SdfParams& p, hipStream_t stream, bool* handled);
bool HullPassesAreTiled(const SdfParams& p);
hipError_t LaunchPassXHullFinalizeRange(const int32_t* in32, float* sdf, uint32_t* minmax_enc,
                                        const SdfParams& p, int64_t outer_begin, int64_t outer_count_or_all,
                                        hipStream_t stream, bool* handled);

#endif  // VGT_HIP_TESTING
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
#define VGT_SCAN_CASE(N)                                                                        \
  hipLaunchKernelGGL((ScanZUnrolledKernel<InT, N>), dim3(grid), dim3(kScanBlock), 0, stream, in, \
                     out16, lines, nz, unknown_is_filled, summary, z_offset)
  // four voxels per lane when the lines allow 16-byte (float) / 4-byte (mask) vector accesses
  const bool quad_ok = (nz % 4 == 0) && nz <= 2048 &&
                       (reinterpret_cast<uintptr_t>(in) % (4 * sizeof(InT)) == 0) &&
                       (reinterpret_cast<uintptr_t>(out16) % 8 == 0);
#define VGT_QUAD_CASE(N)                                                                    \
  hipLaunchKernelGGL((ScanZTransitionKernel<InT, N>), dim3(grid), dim3(kScanBlock), 0, stream, in, \
                     out16, lines, nz, unknown_is_filled, summary, z_offset)
  if (quad_ok && nz <= 256)
    VGT_QUAD_CASE(1);
  else if (quad_ok && nz <= 512)
    VGT_QUAD_CASE(2);
  else if (quad_ok && nz <= 1024)
    VGT_QUAD_CASE(4);
  else if (quad_ok)
    VGT_QUAD_CASE(8);
  else if (nz <= 64)
    VGT_SCAN_CASE(1);
  else if (nz <= 128)
    VGT_SCAN_CASE(2);
  else if (nz <= 256)
    VGT_SCAN_CASE(4);
  else if (nz <= 512)
    VGT_SCAN_CASE(8);
  else if (nz <= 1024)
    VGT_SCAN_CASE(16);
  else
    hipLaunchKernelGGL(ScanZKernel<InT>, dim3(grid), dim3(kScanBlock), 0, stream, in, out16, lines,
                       nz, unknown_is_filled, summary, z_offset);
#undef VGT_SCAN_CASE
#undef VGT_QUAD_CASE
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
  if (p.nz % 8 == 0 && reinterpret_cast<uintptr_t>(io16) % 16 == 0)
  {
    const int64_t groups = total / 8;
    hipLaunchKernelGGL(SlabFixupVecKernel, dim3(GridFor(groups, 256)), dim3(256), 0, stream, io16, carries,
                       groups, static_cast<int>(p.nz / 8), static_cast<int>(p.z_offset));
    return hipGetLastError();
  }
  hipLaunchKernelGGL(SlabFixupKernel, dim3(GridFor(total, 256)), dim3(256), 0, stream, io16,
                     carries, total, static_cast<int>(p.nz), static_cast<int>(p.z_offset));
  return hipGetLastError();
}

hipError_t LaunchPassY(const int16_t* in16, int32_t* out32, SweepScratch scratch, const SdfParams& p,
                       EdtVariant variant, hipStream_t stream)
{
  if (IsSweepVariant(variant)) return LaunchPassYSweep(in16, out32, scratch, p, stream);
  if (variant == EdtVariant::kHull)
  {
    bool handled = false;
    const hipError_t err = LaunchPassYHull(in16, out32, p, stream, &handled);
    if (handled || err != hipSuccess) return err;
  }
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
  // (short lines: the default pipeline only -- the cross-check variant 3 keeps the sweeps on every length)
  // (The X pass keeps the sweeps beyond 64 rows even when a launch has few items: measured equal at 80 - 128 rows,
  // profiles/r5/short_vs_sweep.txt -- its rows pay for the final conversion either way.  The Y pass gains a third there.)
  if (variant == EdtVariant::kDefault && p.nx <= ShortLineRows())
    return LaunchPassXShortFinalizeRange(in32, sdf, minmax_enc, p, 0, -1, stream);
  if (IsSweepVariant(variant))
    return LaunchPassXSweepFinalizeRange(in32, sdf, minmax_enc, scratch, p, 0, -1, stream);
#ifdef VGT_HIP_TESTING
  if (variant == EdtVariant::kHull)
  {
    bool handled = false;
    const hipError_t err = LaunchPassXHullFinalize(in32, sdf, minmax_enc, p, stream, &handled);
    if (handled || err != hipSuccess) return err;
  }
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
  if (IsSweepVariant(variant)) return true;
#ifdef VGT_HIP_TESTING
  return variant == EdtVariant::kHull && HullPassesAreTiled(p);
#else
  (void)p;
  return false;
#endif
}

hipError_t LaunchPassXFinalizeRange(const int32_t* in32, float* sdf, uint32_t* minmax_enc, SweepScratch scratch,
                                    const SdfParams& p, EdtVariant variant, int64_t outer_begin, int64_t outer_count,
                                    hipStream_t stream)
{
  if (variant == EdtVariant::kDefault && p.nx <= ShortLineRows())
    return LaunchPassXShortFinalizeRange(in32, sdf, minmax_enc, p, outer_begin, outer_count, stream);
  if (IsSweepVariant(variant))
    return LaunchPassXSweepFinalizeRange(in32, sdf, minmax_enc, scratch, p, outer_begin, outer_count, stream);
#ifdef VGT_HIP_TESTING
  bool handled = false;
  const hipError_t err = LaunchPassXHullFinalizeRange(in32, sdf, minmax_enc, p, outer_begin, outer_count, stream, &handled);
  if (err != hipSuccess) return err;
  return handled ? hipSuccess : hipErrorInvalidValue;
#else
  return hipErrorInvalidValue;
#endif
}

#ifdef VGT_HIP_TESTING
std::atomic<int> g_short_line_rows{kShortLineRows};
int ShortLineRows() { return g_short_line_rows.load(); }
void SetShortLineRows(int rows)
{
  g_short_line_rows.store(rows < 0 ? 0 : (rows > kShortLineRowsFewItems ? kShortLineRowsFewItems : rows));
}
std::atomic<bool> g_sweep_hand_over{false};
bool SweepHandOver() { return g_sweep_hand_over.load(); }
void SetSweepHandOver(bool on) { g_sweep_hand_over.store(on); }
std::atomic<bool> g_sweep_coarse_hull{kSweepCoarseHullDefault};
bool SweepCoarseHull() { return g_sweep_coarse_hull.load(); }
void SetSweepCoarseHull(bool on) { g_sweep_coarse_hull.store(on); }
#else
bool SweepCoarseHull() { return kSweepCoarseHullDefault; }
int ShortLineRows() { return kShortLineRows; }
bool SweepHandOver() { return false; }
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
