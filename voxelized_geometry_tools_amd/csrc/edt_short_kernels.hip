// Line passes (Y and X) of the exact signed EDT for SHORT lines (at most 64 rows; 128 when a launch has few items) on
// gfx950: the whole line in registers,
// the 1-D transform by exhaustive search.
//
// Why a second formulation.  The lane-per-line sweeps (edt_sweep_kernels.hip) walk a line row by row, twice, through a
// stack whose state every row depends on: a wave needs about 0.4 - 0.8 us per row whatever the chip is doing, and an
// item of 64 rows has the fixed costs of an item of 1024 (work fetch, ring set-up, the sign words' round trip through
// memory).  On the grid sizes of the reference's own examples and tests (8^3 - 40^3) and on batches of such grids
// (vgt_hip_sdf_batch_dev) the passes were bound by that latency: 64 grids of 64^3 took the X pass 0.16 ms, four times what
// its rows cost at the large grids' rate.  The reference itself switches formulations by length -- lines of at most 8
// rows take a brute-force transform (signed_distance_field_generation.cpp:85-122, :229-248) -- and for the same reason:
// below some length the bookkeeping of Felzenszwalb-Huttenlocher costs more than the search it avoids.
//
// Here a lane holds its line's n <= 64 costs in registers (F[r], statically indexed: the loops over r are unrolled),
// and row q's result is min over r of F[r] + (q - r)^2, evaluated as q^2 + min over r of (F[r] + r^2 - 2 q r): stepping q
// subtracts a constant from each term, so a (q, r) pair costs 1.5 vector instructions (an add of a literal and half a
// v_min3) and nothing on the scalar unit; no LDS, no scratch, no second sweep.  Classes as in the sweeps: the envelope ignores them (cost |F[r]|), and the squared distance
// to the nearest row of the OTHER class, from the line's class bits (one 64-bit word per lane), is a second candidate.
// Same geometry as the sweeps (a wave = 64 neighbouring Z positions of one outer index, every row access one contiguous
// 256-B segment), same input and output encodings, same final conversion: the two formulations are interchangeable
// per pass, and the parity suite runs both on every short shape (the testing library can move the limit between them:
// vgt_hip_testing_set_short_line_rows).
#include "edt_device.hpp"
#include "edt_line_geom.hpp"

#include <type_traits>

namespace vgt
{
namespace
{
constexpr int32_t kNoSite = 0x60000000;  // cost of a row that is no site: above every real squared distance (< 2^30)
constexpr int kFarRows = 32768;          // "no row of the other class": kFarRows^2 is above every real squared distance

__device__ __forceinline__ uint32_t AbsDiffPlusOne(uint32_t a, uint32_t b_uniform)
{
  return (max(a, b_uniform) - min(a, b_uniform)) + 1u;
}

// One bit per row of a lane's line: a 64-bit word for lines of at most 64 rows, 128 bits beyond.
template <int NMAX>
struct RowBits
{
  using Word = typename std::conditional<(NMAX <= 64), uint64_t, unsigned __int128>::type;
};
__device__ __forceinline__ int HighestBit(uint64_t v) { return 63 - __clzll(static_cast<long long>(v)); }  // v != 0
__device__ __forceinline__ int LowestBit(uint64_t v) { return __ffsll(static_cast<long long>(v)) - 1; }
__device__ __forceinline__ int HighestBit(unsigned __int128 v)
{
  const uint64_t hi = static_cast<uint64_t>(v >> 64);
  return hi ? 64 + HighestBit(hi) : HighestBit(static_cast<uint64_t>(v));
}
__device__ __forceinline__ int LowestBit(unsigned __int128 v)
{
  const uint64_t lo = static_cast<uint64_t>(v);
  return lo ? LowestBit(lo) : 64 + LowestBit(static_cast<uint64_t>(v >> 64));
}
template <typename Word>
__device__ __forceinline__ Word LowRows(int n)  // bits [0, n)
{
  constexpr int kBits = static_cast<int>(sizeof(Word)) * 8;
  return (n >= kBits) ? ~static_cast<Word>(0) : ((static_cast<Word>(1) << n) - static_cast<Word>(1));
}

// Squared distance from row q to the nearest row whose class differs from q's, rows [0, n) of a line whose class bits
// are `cls` (bit r = row r is filled); kFarRows^2 when the line holds one class only.
template <typename Word>
__device__ __forceinline__ uint32_t OtherClassCandidate(Word cls, int q, int n)
{
  constexpr int kBits = static_cast<int>(sizeof(Word)) * 8;
  const bool mine = ((cls >> q) & static_cast<Word>(1)) != 0;
  const Word other = (mine ? ~cls : cls) & LowRows<Word>(n);
  const Word below = other & LowRows<Word>(q);
  const Word above = (q >= kBits - 1) ? static_cast<Word>(0) : (other >> (q + 1));
  int d = kFarRows;
  if (below != 0) d = q - HighestBit(below);
  if (above != 0) d = min(d, LowestBit(above) + 1);
  return static_cast<uint32_t>(d) * static_cast<uint32_t>(d);
}

// The envelope of a line, row by row: min over r of F[r] + (q - r)^2 = q^2 + min over r of H[r], H[r] = G[r] - 2 q r with
// G[r] = F[r] + r^2 -- and stepping q by one subtracts the CONSTANT 2 r from H[r].  So a row costs, per r, one add of a
// literal and half a v_min3, all on the vector unit: no per-pair scalar work (the scalar unit is shared by the CU's four
// SIMDs and was the bound of a first version that computed (q - r)^2 there: 4 scalar instructions per 3 vector ones).
// H stays within [-63^2, kNoSite + 63^2]: signed minima.
// H for row q0 (a wave may begin anywhere on the line: see `parts`)
template <int NMAX>
__device__ __forceinline__ void EnvelopeBegin(int32_t (&H)[NMAX], int q0)
{
#pragma unroll
  for (int r = 0; r < NMAX; r++) H[r] += r * r - 2 * r * q0;
}
// value of the envelope at the row H stands at, then on to the next row
template <int NMAX>
__device__ __forceinline__ uint32_t EnvelopeStep(int32_t (&H)[NMAX], int q)
{
  // (two chains: a lone wave -- one small grid -- would otherwise wait for every minimum before it issues the next;
  // spelled out: left to itself the compiler pairs the minima only now and then)
  int32_t best = 0x7fffffff, best2 = 0x7fffffff;
#pragma unroll
  for (int r = 0; r < NMAX; r += 4)
  {
    asm("v_min3_i32 %0, %0, %1, %2" : "+v"(best) : "v"(H[r]), "v"(H[r + 1]));
    if (r + 2 < NMAX) asm("v_min3_i32 %0, %0, %1, %2" : "+v"(best2) : "v"(H[r + 2]), "v"(H[r + 3]));
  }
#pragma unroll
  for (int r = 1; r < NMAX; r++) H[r] -= 2 * r;
  return static_cast<uint32_t>(min(best, best2) + q * q);
}

// The wave's extrema into minmax_enc[0 / 1] (ordered encodings), one atomic pair.
__device__ __forceinline__ void WaveExtrema(float lo_value, float hi_value, uint32_t* minmax_enc)
{
  uint32_t lo = 0xffffffffu, hi = 0u;
  if (lo_value <= hi_value)
  {
    lo = EncodeOrdered(lo_value);
    hi = EncodeOrdered(hi_value);
  }
  for (int off = kWaveSize / 2; off > 0; off >>= 1)
  {
    lo = min(lo, static_cast<uint32_t>(__shfl_xor(static_cast<int>(lo), off)));
    hi = max(hi, static_cast<uint32_t>(__shfl_xor(static_cast<int>(hi), off)));
  }
  if (threadIdx.x == 0)
  {
    atomicMin(&minmax_enc[0], lo);
    atomicMax(&minmax_enc[1], hi);
  }
}

// kFinal = false: Y pass, class records (pass 1, edt_record_kernels.hip) -> int32 (sign and magnitude: bit 31 = class,
// kInf32 = none).  kFinal = true: X pass, that int32 field -> float SDF + extrema.  One wave per item.
template <int NMAX, bool kFinal>
__global__ __launch_bounds__(kWaveSize) void ShortLinePassKernel(const void* __restrict__ in_raw, void* __restrict__ out_raw,
                                                                uint32_t* __restrict__ minmax_enc, const SweepGeom g)
{
  static_assert(NMAX % 4 == 0 && NMAX <= 128, "the line's class bits are one or two 64-bit words");
  using Word = typename RowBits<NMAX>::Word;
  const int lane = threadIdx.x;
  const int n = g.n;
  // g.groups waves share an item, each taking a contiguous share of its rows (every wave holds the whole line): a few
  // small grids leave most of the chip idle, and a lone wave issues at a fraction of a SIMD's rate
  const int parts = g.groups;
  const int item = static_cast<int>(blockIdx.x) / parts;
  const int part = static_cast<int>(blockIdx.x) - item * parts;
  const int q_begin = (n * part) / parts, q_end = (n * (part + 1)) / parts;
  const int outer = item / g.zsegs;
  const int z0 = (item - outer * g.zsegs) * kWaveSize;
  const int batch_index = outer / g.batch_outers;
  const int outer_in_grid = outer - batch_index * g.batch_outers;
  const int64_t outer_offset =
      static_cast<int64_t>(outer) * g.outer_stride + static_cast<int64_t>(batch_index) * g.batch_skip;
  // lanes beyond the grid repeat the grid's last line: same input, same result, stored to the same address
  const int zl = min(lane, g.nz - 1 - z0);
  int32_t F[NMAX];
  Word cls = 0;  // bit r: row r of this lane's line is filled
  if constexpr (!kFinal)
  {
    // ---- rows from class records: one vector load brings the records of rows 0..63 (lane l holds row l's; rows past
    // the line's end read the next line's records or the padding behind the buffer and are not used); a lane's distance
    // along Z is the smallest distance across the transitions around it (vgt_internal.hpp, ClassRecord) ----
    const ClassRecord* records = static_cast<const ClassRecord*>(in_raw) + static_cast<int64_t>(item) * n;
    using Raw = uint32_t __attribute__((ext_vector_type(4)));
    // (lines of more than 64 rows: a second block)
    const Raw blk0 = *(reinterpret_cast<const Raw*>(records) + lane);
    const Raw blk1 = (NMAX > 64) ? *(reinterpret_cast<const Raw*>(records) + 64 + lane) : blk0;
    const uint32_t xq = 2u * static_cast<uint32_t>(zl) + (kRecordBias - 1u);
#pragma unroll
    for (int r = 0; r < NMAX; r++)
    {
      F[r] = kNoSite;
      if (r < n)
      {
        const Raw& blk = (r < 64) ? blk0 : blk1;
        const int kLaneOfRow = r & 63;
        const uint32_t mask_lo = __builtin_amdgcn_readlane(blk.x, kLaneOfRow), above2 = __builtin_amdgcn_readlane(blk.w, kLaneOfRow);
        if (above2 == kRecordNoSite)
        {
          // pass 1's mark: the whole Z line holds one class (most rows of a sparse scene) -- no lane has a site in this
          // row, and every lane has the line's class
          if (mask_lo & 1u) cls |= static_cast<Word>(1) << r;
          continue;
        }
        const uint32_t mask_hi = __builtin_amdgcn_readlane(blk.y, kLaneOfRow), below2 = __builtin_amdgcn_readlane(blk.z, kLaneOfRow);
        const uint64_t m = (static_cast<uint64_t>(mask_hi) << 32) | mask_lo;
        // (bits past the end of the Z line repeat its last voxel, so bit `lane` is the class of voxel zl)
        if (__builtin_amdgcn_inverse_ballot_w64(m)) cls |= static_cast<Word>(1) << r;
        uint32_t f2 = min(AbsDiffPlusOne(xq, below2), AbsDiffPlusOne(xq, above2));
        uint64_t own = (m ^ (m >> 1)) & 0x7fffffffffffffffull;  // the word's own transitions (wave-uniform)
        while (own != 0ull)
        {
          const uint32_t t = 2u * static_cast<uint32_t>(__ffsll(static_cast<long long>(own)) - 1) + kRecordBias;
          f2 = min(f2, AbsDiffPlusOne(xq, t));
          own &= own - 1ull;
        }
        // (no transition anywhere on the Z line: f2 >= 2 kInf16, the row is no site for this lane)
        const int32_t f = static_cast<int32_t>(f2 >> 1);
        if (f2 < 2u * static_cast<uint32_t>(kInf16)) F[r] = __mul24(f, f);
      }
    }
  }
  else
  {
    // (straight code: rows past the line's end read its last row again and are then replaced by "no site")
    const int32_t* rows = static_cast<const int32_t*>(in_raw) + (outer_offset + z0 + zl);
    uint32_t v[NMAX];
#pragma unroll
    for (int r = 0; r < NMAX; r++)
      v[r] = static_cast<uint32_t>(__builtin_nontemporal_load(rows + static_cast<int64_t>(min(r, n - 1)) * g.row_stride));
#pragma unroll
    for (int r = 0; r < NMAX; r++)
    {
      const uint32_t value = (r < n) ? v[r] : static_cast<uint32_t>(kNoSite);
      F[r] = static_cast<int32_t>(min(value & 0x7fffffffu, static_cast<uint32_t>(kNoSite)));
      cls |= static_cast<Word>(value >> 31) << r;
    }
  }
  // does any lane's line change class?  (most waves of a sparse scene: no -- the candidates are then skipped)
  const Word rows_mask = LowRows<Word>(n);
  const bool mixed = ((cls & rows_mask) != 0) && ((cls & rows_mask) != rows_mask);
  const bool classes = __builtin_amdgcn_ballot_w64(mixed) != 0ull;

  float lo_value = INFINITY, hi_value = -INFINITY;
  EnvelopeBegin<NMAX>(F, q_begin);
  for (int q = q_begin; q < q_end; q++)
  {
    uint32_t best = EnvelopeStep<NMAX>(F, q);
    if (classes) best = min(best, OtherClassCandidate(cls, q, n));
    const uint32_t sign = static_cast<uint32_t>((cls >> q) & static_cast<Word>(1)) << 31;
    if constexpr (!kFinal)
    {
      const uint32_t d2 = (best >= static_cast<uint32_t>(kNoSite)) ? static_cast<uint32_t>(kInf32) : best;
      int32_t* row_out = static_cast<int32_t*>(out_raw) + (outer_offset + z0 + zl) + static_cast<int64_t>(q) * g.row_stride;
      __builtin_nontemporal_store(static_cast<int32_t>(d2 | sign), row_out);
    }
    else
    {
      int32_t d2 = (best >= static_cast<uint32_t>(kNoSite)) ? kInf32 : static_cast<int32_t>(best);
      if (g.add_virtual_border)
      {
        const int x = (g.pass_axis == 0) ? q : outer_in_grid + g.outer_begin;
        const int y = (g.pass_axis == 0) ? outer_in_grid + g.outer_begin : q;
        d2 = ClampToVirtualBorder(d2, x, y, z0 + zl + g.z_offset, g.nx, g.ny, g.nz_global);
      }
      const float dist = (d2 == kInf32) ? __uint_as_float(0x7f800000u) : SqrtTimesResolution(d2, g.resolution);
      const float value = __uint_as_float(__float_as_uint(dist) | sign);
      float* row_out = static_cast<float*>(out_raw) + (outer_offset + z0 + zl) + static_cast<int64_t>(q) * g.row_stride;
      __builtin_nontemporal_store(value, row_out);
      lo_value = fminf(lo_value, value);
      hi_value = fmaxf(hi_value, value);
    }
  }
  if constexpr (kFinal) WaveExtrema(lo_value, hi_value, minmax_enc + 2 * batch_index);
}

template <bool kFinal>
hipError_t LaunchShort(const void* in, void* out, uint32_t* minmax_enc, SweepGeom g, int64_t outer_count, hipStream_t stream)
{
  g.zsegs = (g.nz + kWaveSize - 1) / kWaveSize;
  const int64_t items = outer_count * g.zsegs;
  if (items <= 0) return hipSuccess;
  if (items > 0x7fffffffLL || g.n > kShortLineRowsFewItems) return hipErrorInvalidValue;
  g.items = static_cast<int>(items);
  g.outers = static_cast<int>(outer_count);
  if (g.batch_outers <= 0)
  {
    g.batch_outers = g.outers;  // one grid
    g.batch_skip = 0;
  }
  // waves per item: as many as it takes to put about two waves on every SIMD, at most one per 8 rows
  int64_t parts = 1;
  while (parts < 16 && items * parts * 2 <= 2048 && g.n >= 16 * parts) parts *= 2;
  g.groups = static_cast<int>(parts);
  const dim3 grid(static_cast<unsigned>(items * parts)), block(kWaveSize);
  if (g.n <= 8)
    hipLaunchKernelGGL((ShortLinePassKernel<8, kFinal>), grid, block, 0, stream, in, out, minmax_enc, g);
  else if (g.n <= 16)
    hipLaunchKernelGGL((ShortLinePassKernel<16, kFinal>), grid, block, 0, stream, in, out, minmax_enc, g);
  else if (g.n <= 24)
    hipLaunchKernelGGL((ShortLinePassKernel<24, kFinal>), grid, block, 0, stream, in, out, minmax_enc, g);
  else if (g.n <= 32)
    hipLaunchKernelGGL((ShortLinePassKernel<32, kFinal>), grid, block, 0, stream, in, out, minmax_enc, g);
  else if (g.n <= 48)
    hipLaunchKernelGGL((ShortLinePassKernel<48, kFinal>), grid, block, 0, stream, in, out, minmax_enc, g);
  else if (g.n <= 64)
    hipLaunchKernelGGL((ShortLinePassKernel<64, kFinal>), grid, block, 0, stream, in, out, minmax_enc, g);
  else if (g.n <= 96)
    hipLaunchKernelGGL((ShortLinePassKernel<96, kFinal>), grid, block, 0, stream, in, out, minmax_enc, g);
  else
    hipLaunchKernelGGL((ShortLinePassKernel<128, kFinal>), grid, block, 0, stream, in, out, minmax_enc, g);
  return hipGetLastError();
}
}  // namespace

// Y pass over class records, lines of at most kShortLineRows rows.  `records` must be followed by kRecordPadding
// readable records (the block load of a line's records covers 64 rows whatever the line's length).
hipError_t LaunchPassYShortRecords(const ClassRecord* records, int32_t* out32, const SdfParams& p, hipStream_t stream)
{
  int64_t outer_count = 0;
  const SweepGeom g = SweepGeometry(p, 1, &outer_count);
  return LaunchShort<false>(records, out32, nullptr, g, outer_count, stream);
}

// X pass + finalize over the Y positions [outer_begin, outer_begin + outer_count) (outer_count < 0: all, of every grid
// of a batch), lines of at most kShortLineRows rows; full-grid pointers and extents in `p`.
hipError_t LaunchPassXShortFinalizeRange(const int32_t* in32, float* sdf, uint32_t* minmax_enc, const SdfParams& p,
                                         int64_t outer_begin, int64_t outer_count_or_all, hipStream_t stream)
{
  int64_t outer_count = 0;
  SweepGeom g = SweepGeometry(p, 0, &outer_count);
  if (p.batch > 1)
  {
    if (outer_count_or_all >= 0 || p.batch * outer_count > 0x7fffffffLL) return hipErrorInvalidValue;
    g.batch_outers = static_cast<int>(outer_count);
    g.batch_skip = (p.nx - 1) * p.ny * p.nz;
    outer_count *= p.batch;
  }
  if (outer_count_or_all >= 0)
  {
    in32 += outer_begin * g.outer_stride;
    sdf += outer_begin * g.outer_stride;
    g.outer_begin = static_cast<int>(outer_begin);
    outer_count = outer_count_or_all;
  }
  return LaunchShort<true>(in32, sdf, minmax_enc, g, outer_count, stream);
}
}  // namespace vgt
