// LDS-tiled line passes (Y and X) of the exact signed EDT for gfx950 -- search on the monotone
// argmin (EdtVariant::kSearch; a second, independent formulation of the pass computed by
// edt_hull_kernels.hip, kept as an on-device cross-check).
//
// Tile: all n rows of the pass axis x W adjacent Z positions, resident in LDS as signed squared
// distances F[row][line] (bank = line, so lanes working on different lines never conflict).
//
// Per line and row q the pass computes min over sites o of (q-o)^2 + f(o).  With [a,b] the
// maximal same-class run around q (see edt_hull_kernels.hip)
//     out(q) = min( min_{o in [a,b], f(o) finite} (q-o)^2 + f(o),  (q-(a-1))^2,  ((b+1)-q)^2 ).
// The inner minimisation is evaluated through the property the lower envelope of
// signed_distance_field_generation.cpp:124-226 rests on: inside a run the leftmost argmin
// opt(q) is non-decreasing in q, so opt(q) lies between the argmins of already solved rows
// below and above.  Exact integer evaluations only; no stack, no division.
//
// One thread per (line, band of 32 rows); lanes of a wave are neighbouring lines.
//   1   sign word and smallest member value of every band
//   1b  per-word carries: nearest row of either class below / above the word (wave scans)
//   2   "band queries": opt() of the first row of every band (and of the last row of the line),
//       bisection over the bands, each query solved by the thread of its band
//   3   each thread sweeps its rows upwards: candidates between opt(q-1) and the next band's
//       query, windowed by (q-o)^2 <= incumbent and pruned per band by
//       gap^2 + min f(band) > incumbent; results stored directly (fused finalize in the X pass).
#include "edt_tile.hpp"

#include <cstdlib>

namespace vgt
{
#ifdef VGT_HULL_STATS
// Diagnostic build only: [0..3] cycles load / prep / band queries / rows, [4] workgroups,
// [5] total cycles, [6] row-phase evaluations, [7] band-query scorings, [8] max evaluations of
// one thread; X pass at +16.
__device__ unsigned long long g_dc_stats[32];
__device__ int g_dc_stats_base;
#define VGT_DC_ADD(i, v) \
  atomicAdd(&g_dc_stats[g_dc_stats_base + (i)], static_cast<unsigned long long>(v))
#define VGT_DC_MAX(i, v) \
  atomicMax(&g_dc_stats[g_dc_stats_base + (i)], static_cast<unsigned long long>(v))
#endif
namespace
{
constexpr uint64_t kNoBest = ~0ull;

template <int W>
struct DcTile
{
  int32_t* F;          // [n][W]
  uint64_t* best;      // [nq+1][W] packed (value << 16 | row) of the band queries; slot nq = last row
  uint32_t* S;         // [nq][W] sign bits (1 = negative class)
  int16_t* last_neg;   // [nq][W] highest negative row below the word, -1 if none
  int16_t* last_pos;
  int16_t* next_neg;   // [nq][W] lowest negative row above the word, n if none
  int16_t* next_pos;
  int32_t* bandmin;    // [nq][W] smallest member value of the band (kInf32 if none)
  int n;
  int nq;  // bands = mask words = band queries
  int w;

  __device__ __forceinline__ int32_t Mag(int row) const
  {
    const int32_t v = F[row * W + w];
    return v < 0 ? -v : v;
  }
  __device__ __forceinline__ int PrevOpp(int row, bool neg) const
  {
    const int j = row >> 5;
    const uint32_t s = S[j * W + w];
    const uint32_t m = (neg ? ~s : s) & LowMask(row & 31);
    if (m) return (j << 5) + 31 - __clz(static_cast<int>(m));
    return neg ? last_pos[j * W + w] : last_neg[j * W + w];
  }
  __device__ __forceinline__ int NextOpp(int row, bool neg) const
  {
    const int j = row >> 5;
    const uint32_t s = S[j * W + w];
    const uint32_t m = (neg ? ~s : s) & LowMask(n - (j << 5)) & ~LowMask((row & 31) + 1);
    if (m) return (j << 5) + __ffs(static_cast<int>(m)) - 1;
    return neg ? next_pos[j * W + w] : next_neg[j * W + w];
  }
};

__device__ __forceinline__ int OptOf(uint64_t packed)
{
  return (packed == kNoBest) ? -1 : static_cast<int>(packed & 0xffffull);
}

template <int W>
size_t DcTileBytes(int n)
{
  const size_t nq = static_cast<size_t>((n + kBandRows - 1) / kBandRows);
  // one extra band query: the last row of the line
  return static_cast<size_t>(n) * W * sizeof(int32_t) + (nq + 1) * W * sizeof(uint64_t) +
         nq * W * sizeof(uint32_t) + 4 * nq * W * sizeof(int16_t) + nq * W * sizeof(int32_t);
}

template <typename InT, typename OutT, bool kFinal, int W, int SW>
__global__ __launch_bounds__(1024) void DcPassKernel(const InT* __restrict__ in,
                                                    OutT* __restrict__ out,
                                                    uint32_t* __restrict__ minmax_enc,
                                                    const TileGeom g)
{
  extern __shared__ __align__(16) unsigned char smem[];
  const int n = g.n;
  const int nq = (n + kBandRows - 1) / kBandRows;
  const int mw = nq * W;
  DcTile<W> t;
  t.F = reinterpret_cast<int32_t*>(smem);
  t.best = reinterpret_cast<uint64_t*>(t.F + static_cast<size_t>(n) * W);
  t.S = reinterpret_cast<uint32_t*>(t.best + mw + W);
  t.last_neg = reinterpret_cast<int16_t*>(t.S + mw);
  t.last_pos = t.last_neg + mw;
  t.next_neg = t.last_pos + mw;
  t.next_pos = t.next_neg + mw;
  t.bandmin = reinterpret_cast<int32_t*>(t.next_pos + mw);
  t.n = n;
  t.nq = nq;
  t.w = threadIdx.x % W;

  const int outer = blockIdx.x / g.ztiles;
  const int z0 = (blockIdx.x % g.ztiles) * W;
  const int64_t base = static_cast<int64_t>(outer) * g.outer_stride + z0;
  const int band = threadIdx.x / W;
  const int z = z0 + t.w;
  const bool in_band = band < nq;
  const bool active = in_band && (z < g.nz);

#ifdef VGT_HULL_STATS
  const long long clk0 = clock64();
  unsigned int n_evals = 0;
  const unsigned int n_scores = 0;
#endif
  LoadTile<InT, W>(in, t.F, n, base, z0, g);
  __syncthreads();
#ifdef VGT_HULL_STATS
  const long long clk1 = clock64();
#endif

  // ---- 1. sign word and minimum member value of this band ----
  const int r0 = band * kBandRows;
  uint32_t sbits = 0;
  if (in_band)
  {
    int32_t bmin = kInf32;
#pragma unroll 8
    for (int k = 0; k < kBandRows; k++)
    {
      const int row = r0 + k;
      int32_t v = kInf32;
      if (row < n) v = t.F[row * W + t.w];
      if (v < 0) sbits |= 1u << k;
      bmin = min(bmin, v < 0 ? -v : v);
    }
    t.S[band * W + t.w] = sbits;
    t.bandmin[band * W + t.w] = bmin;
  }
  __syncthreads();

  // ---- 1b. per word: nearest row of either class below / above ----
  for (int tt = threadIdx.x; tt < W * SW; tt += blockDim.x)
  {
    const int line = tt / SW;
    const int j = tt % SW;
    const bool ok = j < nq;
    const uint32_t s = ok ? t.S[j * W + line] : 0u;
    const uint32_t p = ok ? (~s & LowMask(n - (j << 5))) : 0u;
    int hi_neg = s ? (j << 5) + 31 - __clz(static_cast<int>(s)) : -1;
    int hi_pos = p ? (j << 5) + 31 - __clz(static_cast<int>(p)) : -1;
    int lo_neg = s ? (j << 5) + __ffs(static_cast<int>(s)) - 1 : n;
    int lo_pos = p ? (j << 5) + __ffs(static_cast<int>(p)) - 1 : n;
    for (int d = 1; d < SW; d <<= 1)
    {
      const int un = __shfl_up(hi_neg, d, SW), up = __shfl_up(hi_pos, d, SW);
      const int dn = __shfl_down(lo_neg, d, SW), dp = __shfl_down(lo_pos, d, SW);
      if (j >= d)
      {
        hi_neg = max(hi_neg, un);
        hi_pos = max(hi_pos, up);
      }
      if (j + d < SW)
      {
        lo_neg = min(lo_neg, dn);
        lo_pos = min(lo_pos, dp);
      }
    }
    const int ex_hn = __shfl_up(hi_neg, 1, SW), ex_hp = __shfl_up(hi_pos, 1, SW);
    const int ex_ln = __shfl_down(lo_neg, 1, SW), ex_lp = __shfl_down(lo_pos, 1, SW);
    if (ok)
    {
      t.last_neg[j * W + line] = static_cast<int16_t>(j == 0 ? -1 : ex_hn);
      t.last_pos[j * W + line] = static_cast<int16_t>(j == 0 ? -1 : ex_hp);
      t.next_neg[j * W + line] = static_cast<int16_t>(j == SW - 1 ? n : ex_ln);
      t.next_pos[j * W + line] = static_cast<int16_t>(j == SW - 1 ? n : ex_lp);
    }
  }
  __syncthreads();
#ifdef VGT_HULL_STATS
  const long long clk2 = clock64();
#endif

  // Scores the candidate rows [lo, hi] (all inside the run of q) for row q and keeps the
  // leftmost argmin.  The two end points go first (argmins of already solved rows, i.e. strong
  // candidates); after that a candidate o can only matter if (q-o)^2 <= best, which bounds the
  // window, and whole bands are skipped when even their smallest member value cannot beat the
  // incumbent:  gap^2 + min f(band) > best.
  auto scan = [&](int q, int lo, int hi, int32_t& best, int& arg) {
    auto score = [&](int o, int32_t f) {
      const int d = q - o;
      const int32_t val = (f == kInf32) ? kInf32 : d * d + f;
      if (val < best || (val == best && val != kInf32 && o < arg))
      {
        best = val;
        arg = o;
      }
    };
    score(lo, t.Mag(lo));
    if (hi > lo) score(hi, t.Mag(hi));
    int wlo = lo + 1, whi = hi - 1;
    if (best != kInf32)
    {
      const int radius = static_cast<int>(sqrtf(static_cast<float>(best))) + 1;
      wlo = max(wlo, q - radius);
      whi = min(whi, q + radius);
    }
    int o = wlo;
    while (o <= whi)
    {
      const int bnd = o >> 5;
      const int bend = min(whi, (bnd << 5) + 31);
      const int gap = (q < o) ? (o - q) : ((q > bend) ? (q - bend) : 0);
      const int32_t bm = t.bandmin[bnd * W + t.w];
      if (bm == kInf32 || gap * gap + bm > best)
      {
        o = bend + 1;
        continue;
      }
#ifdef VGT_HULL_STATS
      n_evals += bend - o + 1;
#endif
      for (; o + 3 <= bend; o += 4)
      {
        const int32_t f0 = t.Mag(o), f1 = t.Mag(o + 1), f2 = t.Mag(o + 2), f3 = t.Mag(o + 3);
        score(o, f0);
        score(o + 1, f1);
        score(o + 2, f2);
        score(o + 3, f3);
      }
      for (; o <= bend; o++) score(o, t.Mag(o));
    }
  };

  // ---- 2. band queries (first row of every band, plus the last row of the line) in bisection
  // order over the bands; a query is solved by the thread of its band ----
  int pow2 = 1;
  while (pow2 < nq) pow2 <<= 1;
  bool first_level = true;
  for (int step = max(pow2 >> 1, 1); step >= 1; step >>= 1)
  {
    if (active)
    {
      // slots solved on this level by this thread: its own band start if the band index is an
      // odd multiple of step (0 and step on the first level), and the last row of the line
      // (slot nq, thread of the last band) on the first level
      for (int which = 0; which < 2; which++)
      {
        int slot;
        if (which == 0)
        {
          const bool mine = first_level ? (band == 0 || band == step)
                                        : ((band % (2 * step)) == step);
          if (!mine) continue;
          slot = band;
        }
        else
        {
          if (!(first_level && band == nq - 1)) continue;
          slot = nq;
        }
        const int q = (slot == nq) ? n - 1 : (slot << 5);
        const bool neg = (sbits >> (q - r0)) & 1u;
        const int a = t.PrevOpp(q, neg) + 1;
        const int e = t.NextOpp(q, neg) - 1;
        int lo = a, hi = e;
        bool none = false;
        if (!first_level)
        {
          const int bl = slot - step;  // solved on an earlier level
          if (a <= (bl << 5))
          {
            const int o = OptOf(t.best[bl * W + t.w]);
            if (o < 0) none = true;
            lo = max(a, o);
          }
          const int bh = slot + step;
          const int sh = (bh < nq) ? bh : nq;  // above the last band start: the last row
          const int qh = (bh < nq) ? (bh << 5) : n - 1;
          if (e >= qh)
          {
            const int o = OptOf(t.best[sh * W + t.w]);
            if (o < 0) none = true;
            hi = min(e, o);
          }
        }
        int32_t best = kInf32;
        int arg = -1;
        if (!none) scan(q, lo, hi, best, arg);
        t.best[slot * W + t.w] =
            (arg < 0) ? kNoBest
                      : ((static_cast<uint64_t>(static_cast<uint32_t>(best)) << 16) |
                         static_cast<uint32_t>(arg));
      }
    }
    first_level = false;
    __syncthreads();
    if (nq == 1) break;
  }
#ifdef VGT_HULL_STATS
  const long long clk3 = clock64();
#endif

  // ---- 3. rows 1..31 of this band, ascending: opt(q) >= opt(q-1) inside a run, and the
  // value of opt(q-1) at q is a near-optimal incumbent that keeps the search window tight ----
  uint32_t lo_enc = 0xffffffffu, hi_enc = 0u;
  if (active)
  {
    const int nrows = min(kBandRows, n - r0);
    const uint32_t valid = LowMask(nrows);
    const int carry_prev_neg = t.last_neg[band * W + t.w], carry_prev_pos = t.last_pos[band * W + t.w];
    const int carry_next_neg = t.next_neg[band * W + t.w], carry_next_pos = t.next_pos[band * W + t.w];
    const uint64_t b0 = t.best[band * W + t.w];
    const uint64_t b_last = t.best[nq * W + t.w];
    // upper anchor: the next band's first row, or the last row of the line in the last band
    const int top_row = (band + 1 < nq) ? r0 + kBandRows : n - 1;
    const int top_opt = (band + 1 < nq) ? OptOf(t.best[(band + 1) * W + t.w]) : OptOf(b_last);
    int prev_opt = OptOf(b0);
    for (int k = 0; k < nrows; k++)
    {
      const int q = r0 + k;
      const bool neg = (sbits >> k) & 1u;
      const uint32_t other = (neg ? ~sbits : sbits) & valid;
      const uint32_t below = other & LowMask(k);
      const uint32_t above = other & ~LowMask(k + 1);
      const int prev_opp =
          below ? r0 + 31 - __clz(static_cast<int>(below)) : (neg ? carry_prev_pos : carry_prev_neg);
      const int next_opp =
          above ? r0 + __ffs(static_cast<int>(above)) - 1 : (neg ? carry_next_pos : carry_next_neg);
      int32_t best = kInf32;
      if (k == 0)
        best = (b0 == kNoBest) ? kInf32 : static_cast<int32_t>(b0 >> 16);
      else if (q == n - 1)
      {
        best = (b_last == kNoBest) ? kInf32 : static_cast<int32_t>(b_last >> 16);
        prev_opt = OptOf(b_last);
      }
      else
      {
        const int run_a = prev_opp + 1, run_b = next_opp - 1;
        int lo = run_a, hi = run_b;
        bool none = false;
        if (run_a <= q - 1)  // the previous row is in the same run
        {
          if (prev_opt < 0) none = true;
          lo = max(run_a, prev_opt);
        }
        if (run_b >= top_row)  // so is the upper anchor
        {
          if (top_opt < 0) none = true;
          hi = min(run_b, top_opt);
        }
        int arg = -1;
        if (!none) scan(q, lo, hi, best, arg);
        prev_opt = arg;
      }
      if (prev_opp >= 0) best = min(best, (q - prev_opp) * (q - prev_opp));
      if (next_opp < n) best = min(best, (next_opp - q) * (next_opp - q));
      const int64_t idx = base + static_cast<int64_t>(q) * g.row_stride + t.w;
      if constexpr (kFinal)
      {
        const int x = (g.pass_axis == 0) ? q : outer;
        const int y = (g.pass_axis == 0) ? outer : q;
        const float v = FinalizeSdf(best, neg, x, y, z + g.z_offset, g.nx, g.ny, g.nz_global, g.resolution,
                                    g.add_virtual_border);
        out[idx] = v;
        const uint32_t enc = EncodeOrdered(v);
        lo_enc = min(lo_enc, enc);
        hi_enc = max(hi_enc, enc);
      }
      else
      {
        out[idx] = neg ? -best : best;
      }
    }
  }
  if constexpr (kFinal) BlockMinMax(lo_enc, hi_enc, minmax_enc);
#ifdef VGT_HULL_STATS
  VGT_DC_ADD(6, n_evals);
  VGT_DC_ADD(7, n_scores);
  VGT_DC_MAX(8, n_evals);
  __syncthreads();
  if (threadIdx.x == 0)
  {
    const long long clk4 = clock64();
    VGT_DC_ADD(0, clk1 - clk0);
    VGT_DC_ADD(1, clk2 - clk1);
    VGT_DC_ADD(2, clk3 - clk2);
    VGT_DC_ADD(3, clk4 - clk3);
    VGT_DC_ADD(4, 1);
    VGT_DC_ADD(5, clk4 - clk0);
  }
#endif
}

int DcLinesPerTile(int64_t n)
{
  const int64_t rows = (n + kBandRows - 1) / kBandRows * kBandRows;
  if (rows * 32 <= 32768) return 32;
  if (rows * 16 <= 32768) return 16;
  return 0;
}

template <typename InT, typename OutT, bool kFinal, int W, int SW>
hipError_t LaunchDc(const InT* in, OutT* out, uint32_t* minmax_enc, const TileGeom& g,
                    int64_t outer_count, hipStream_t stream)
{
  const int nq = (g.n + kBandRows - 1) / kBandRows;
  const size_t lds = DcTileBytes<W>(g.n);
  int threads = nq * W;
  threads = (threads + 63) / 64 * 64;
  auto kernel = DcPassKernel<InT, OutT, kFinal, W, SW>;
  hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds));
  if (err != hipSuccess) return err;
#ifdef VGT_HULL_STATS
  {
    const int stats_base = kFinal ? 16 : 0;
    err = hipMemcpyToSymbolAsync(HIP_SYMBOL(g_dc_stats_base), &stats_base, sizeof(int), 0,
                                 hipMemcpyHostToDevice, stream);
    if (err != hipSuccess) return err;
    (void)hipStreamSynchronize(stream);
  }
#endif
  const int64_t blocks = outer_count * g.ztiles;
  hipLaunchKernelGGL(kernel, dim3(static_cast<unsigned>(blocks)), dim3(threads), lds, stream, in,
                     out, minmax_enc, g);
  return hipGetLastError();
}

template <typename InT, typename OutT, bool kFinal>
hipError_t DispatchDc(const InT* in, OutT* out, uint32_t* minmax_enc, TileGeom g,
                      int64_t outer_count, hipStream_t stream, bool* handled)
{
  const int W = DcLinesPerTile(g.n);
  *handled = (W != 0);
  if (W == 0) return hipSuccess;
  g.ztiles = (g.nz + W - 1) / W;
  g.debug_skip = 0;
  constexpr int kVec = 16 / static_cast<int>(sizeof(InT));
  g.vector_io = (g.nz % kVec == 0) && (reinterpret_cast<uintptr_t>(in) % 16 == 0);
  if (outer_count * g.ztiles > 0x7fffffffLL)
  {
    *handled = false;
    return hipSuccess;
  }
  const int nq = (g.n + kBandRows - 1) / kBandRows;
  if (W == 32)
    return LaunchDc<InT, OutT, kFinal, 32, 32>(in, out, minmax_enc, g, outer_count, stream);
  if (nq <= 32)
    return LaunchDc<InT, OutT, kFinal, 16, 32>(in, out, minmax_enc, g, outer_count, stream);
  return LaunchDc<InT, OutT, kFinal, 16, 64>(in, out, minmax_enc, g, outer_count, stream);
}
}  // namespace

hipError_t LaunchPassYDc(const int16_t* in16, int32_t* out32, const SdfParams& p,
                         hipStream_t stream, bool* handled)
{
  TileGeom g{};
  g.n = static_cast<int>(p.ny);
  g.nz = static_cast<int>(p.nz);
  g.row_stride = p.nz;
  g.outer_stride = p.ny * p.nz;
  g.nx = static_cast<int>(p.nx);
  g.ny = static_cast<int>(p.ny);
  g.pass_axis = 1;
  g.resolution = p.resolution;
  g.add_virtual_border = p.add_virtual_border;
  g.z_offset = static_cast<int>(p.z_offset);
  g.nz_global = static_cast<int>(p.nz_global > 0 ? p.nz_global : p.nz);
  return DispatchDc<int16_t, int32_t, false>(in16, out32, nullptr, g, p.nx, stream, handled);
}

hipError_t LaunchPassXDcFinalize(const int32_t* in32, float* sdf, uint32_t* minmax_enc,
                                 const SdfParams& p, hipStream_t stream, bool* handled)
{
  TileGeom g{};
  g.n = static_cast<int>(p.nx);
  g.nz = static_cast<int>(p.nz);
  g.row_stride = p.ny * p.nz;
  g.outer_stride = p.nz;
  g.nx = static_cast<int>(p.nx);
  g.ny = static_cast<int>(p.ny);
  g.pass_axis = 0;
  g.resolution = p.resolution;
  g.add_virtual_border = p.add_virtual_border;
  g.z_offset = static_cast<int>(p.z_offset);
  g.nz_global = static_cast<int>(p.nz_global > 0 ? p.nz_global : p.nz);
  return DispatchDc<int32_t, float, true>(in32, sdf, minmax_enc, g, p.ny, stream, handled);
}
}  // namespace vgt

#ifdef VGT_HULL_STATS
// Diagnostic build only: read (and clear) the counters.
extern "C" int vgt_hip_debug_dc_stats(unsigned long long* out32, int reset)
{
  hipError_t err = hipDeviceSynchronize();
  if (err == hipSuccess)
    err = hipMemcpyFromSymbol(out32, HIP_SYMBOL(vgt::g_dc_stats), 32 * sizeof(unsigned long long));
  if (err == hipSuccess && reset)
  {
    unsigned long long zeros[32] = {0};
    err = hipMemcpyToSymbol(HIP_SYMBOL(vgt::g_dc_stats), zeros, sizeof(zeros));
  }
  return err == hipSuccess ? 0 : 2;
}
#endif
