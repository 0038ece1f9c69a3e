// LDS-tiled lower-envelope line passes (Y and X) of the exact signed EDT for gfx950.
//
// One workgroup owns a tile of W adjacent Z positions x ALL n rows of the pass axis (Y or X):
// W lines of n elements, resident in LDS as signed squared distances F[row][line] (bank = line,
// so a lane that works on one line never conflicts with its neighbours).  Rows are coalesced
// 4*W-byte segments in HBM for both axes, so no global transpose is needed; the tile is read
// once and the results are written once.
//
// Per line the pass computes, for every row q, min over sites o of (q-o)^2 + f(o), where the
// sites seen by a voxel of one class are the voxels of its own class with their value so far
// ("members", value f) and every voxel of the other class with value 0.  A zero-valued site
// shields everything behind it: with b_below / b_above the nearest rows of the other class on
// either side of q, every row r of the other class -- and every row of the own class beyond
// one -- can only offer (q-r)^2 + |F[r]| >= (q-b)^2.  So the classes can be ignored while the
// envelope is built:
//     out(q) = min( lower envelope over ALL rows with cost |F[r]| at q, (q-b_below)^2, (b_above-q)^2 ),
// i.e. ONE Felzenszwalb-Huttenlocher envelope per line plus the distance to the two bounding
// rows of the other class (when they exist), which the evaluation adds as candidates.
//
// Parallelisation inside a line: the line is cut into bands of 32 rows, one thread per
// (line, band); lanes of a wave are different lines (neighbouring Z) and different bands.  An
// envelope is a bit mask (bit = member survives), one 32-bit word per band, so "pop" clears a
// bit and stack neighbours are clz/ffs; a per-line summary word skips empty mask words.  Phases
// (workgroup barriers between them):
//   1a the band's 32 rows -> registers; sign word; the band's strongest member is published
//   1b per-word carries of the nearest row of either class below / above (wave scans, for the
//      evaluation's bounding rows) and, by a doubling scan over the published members, one SEED
//      per band and direction: a strong member of the bands below / above
//   1c branch-free prefilter over the registers (a member matched at its own row by a site on
//      each side never owns a row; chord test against the adjacent rows), then the stack
//      algorithm over the surviving candidates, with the seeds as virtual bottom / closing
//      elements (exact integer predicate)
//   2  the band hulls are joined at every band boundary at once (walks with atomic bit clears),
//      in rounds, until a round changes nothing
//   2b every surviving member computes the first row it owns once (float quotient estimate +
//      exact remainder repair); bands exchange the start of their first member; members that own
//      no row are dropped, the others set a "start" bit at that row
//   2c per-word prefix counts of survivors and of start bits (wave scans)
//   3  each thread evaluates its 32 rows: the owner of the first row is the k-th survivor, k =
//      number of start bits at or below it (rank / select); at every further start bit the owner
//      iterator moves to the next survivor.
// The kernel is VALU-issue bound, and a wave pays for its busiest lane: slow paths that only some
// lane takes run on almost every iteration, so the data-dependent loops keep their state (mask
// words, iterators) in registers and avoid the generic LDS bit searches wherever a wave would
// otherwise execute them all the time.  Tile width: see LinesPerTile.
// All predicates are integer-exact (64-bit cross multiplication / exact floor of a quotient).
#include "edt_tile.hpp"

#include <cstdlib>

// Phase-skipping knobs for timing experiments (VGT_HULL_SKIP=<bits>) exist only in builds with
// -DVGT_HULL_DEBUG: in the product build the tests are compile-time false, so no kernel argument
// is re-read inside the hot loops.
#ifdef VGT_HULL_DEBUG
#define VGT_SKIP(bit) (g.debug_skip & (bit))
#else
#define VGT_SKIP(bit) false
#endif

namespace vgt
{
#ifdef VGT_HULL_STATS
// Diagnostic build only (make STATS=1): per-phase cycle sums and operation counts.
// [0..4] cycles load/local/merge/starts/eval, [5] workgroups, [6] total cycles,
// [7] predicate tests, [8] local pops, [9] merge kills, [10] survivors after 2b,
// [12] longest merge walk (max)
__device__ unsigned long long g_hull_stats[32];  // [0..15] Y pass, [16..31] X pass
__device__ int g_hull_stats_base;
#define VGT_CLK_ADD(i, v) \
  atomicAdd(&g_hull_stats[g_hull_stats_base + (i)], static_cast<unsigned long long>(v))
#if VGT_HULL_STATS > 1  // event counters perturb the timings: separate level
#define VGT_STAT_ADD(i, v) VGT_CLK_ADD(i, v)
#define VGT_STAT_MAX(i, v) \
  atomicMax(&g_hull_stats[g_hull_stats_base + (i)], static_cast<unsigned long long>(v))
#else
#define VGT_STAT_ADD(i, v)
#define VGT_STAT_MAX(i, v)
#endif
#else
#define VGT_STAT_ADD(i, v)
#define VGT_STAT_MAX(i, v)
#endif
namespace
{
// Packed owners (kMagBits != 0): when every input magnitude is below 2^kMagBits and a row index fits the other
// 32 - kMagBits bits (22 + 10 for lines of up to 1024 rows, 21 + 11 up to 2048), the
// 32-bit LDS word of a row holds the row's magnitude and -- once the hull is final -- the position of the member
// whose ownership starts at that row.  The evaluation then needs no iterator over the member masks: at a start row
// the owner is read from the row's own word.
template <int W, int kMagBits>
struct Tile
{
  static constexpr bool kPacked = kMagBits != 0;
  static constexpr int32_t kMagMask = (1 << kMagBits) - 1;
  int32_t* F;          // [n][W]  signed squared value (sign = class, |.| = distance^2 or kInf32); packed
                       //         owners: magnitude in the low kMagBits, from phase 2b the owner that starts at
                       //         this row above them
  uint32_t* S;         // [nwords][W] sign bits (1 = negative class)
  uint32_t* A;         // [nwords][W] hull bits while merging
  uint32_t* A2;        // [nwords][W] members that own at least one row
  uint32_t* T;         // [nwords][W] start bits
  int16_t* last_neg;   // [nwords][W] highest negative row below the word, -1 if none
  int16_t* last_pos;   //             same for positive rows
  int16_t* next_neg;   // [nwords][W] lowest negative row above the word, n if none
  int16_t* next_pos;
  uint16_t* cumA;      // [nwords][W] survivors in lower words
  uint16_t* cumT;      // [nwords][W] start bits in lower words
  int16_t* seed_lo;    // [nwords][W] seed row entering the band from below (-1 none); aliases cumA
  int16_t* seed_hi;    //             ... from above; aliases cumT (both dead before 2c writes cumA/cumT)
  int16_t* first_start;  // [nwords][W] start row of the band's first member (-1 none); aliases cumA (2b only)
  uint8_t* min_member;  // [nwords][W] offset of the band's smallest member (255 = none)
  uint64_t* sumA;      // [W] words of A that may be non-empty (superset)
  uint64_t* sumA2;     // [W] words of A2 that are non-empty
  int n;
  int nwords;
  int w;  // this thread's line

  __device__ __forceinline__ int32_t Raw(int row) const { return F[row * W + w]; }
  __device__ __forceinline__ int32_t Mag(int row) const
  {
    const int32_t v = Raw(row);
    if constexpr (kPacked) return v & kMagMask;
    return v < 0 ? -v : v;
  }
  // highest set bit of mask M over rows [lo, r), or -1
  __device__ int PrevBit(const uint32_t* M, const uint64_t* sum, int r, int lo) const
  {
    if (r <= lo) return -1;
    const int j = (r - 1) >> 5;
    const int jlo = lo >> 5;
    uint32_t m = M[j * W + w] & LowMask(r - (j << 5));
    if (j == jlo) m &= ~LowMask(lo & 31);
    if (m) return (j << 5) + 31 - __clz(static_cast<int>(m));
    if (j == jlo) return -1;
    uint64_t cand = sum[w] & ((1ull << j) - 1ull) & ~((1ull << jlo) - 1ull);
    while (cand)
    {
      const int jj = 63 - __clzll(static_cast<long long>(cand));
      m = M[jj * W + w];
      if (jj == jlo) m &= ~LowMask(lo & 31);
      if (m) return (jj << 5) + 31 - __clz(static_cast<int>(m));
      cand &= ~(1ull << jj);
    }
    return -1;
  }
  // lowest set bit of mask M over rows (r, hi), or -1  (r may be -1)
  __device__ int NextBit(const uint32_t* M, const uint64_t* sum, int r, int hi) const
  {
    const int start = r + 1;
    if (start >= hi) return -1;
    const int j = start >> 5;
    const int jhi = (hi - 1) >> 5;
    uint32_t m = M[j * W + w] & ~LowMask(start & 31);
    if (j == jhi) m &= LowMask(hi - (j << 5));
    if (m) return (j << 5) + __ffs(static_cast<int>(m)) - 1;
    if (j == jhi) return -1;
    uint64_t cand = sum[w] & ~((2ull << j) - 1ull);
    if (jhi < 63) cand &= (2ull << jhi) - 1ull;
    while (cand)
    {
      const int jj = __ffsll(static_cast<long long>(cand)) - 1;
      m = M[jj * W + w];
      if (jj == jhi) m &= LowMask(hi - (jj << 5));
      if (m) return (jj << 5) + __ffs(static_cast<int>(m)) - 1;
      cand &= cand - 1ull;
    }
    return -1;
  }
};

// Site b (between a and c) never owns a point of the envelope: with G(v) = f(v) + v^2 the
// parabola of c overtakes b no later than b overtakes a,
//   (Gc - Gb) / (2 (c - b)) <= (Gb - Ga) / (2 (b - a)),
// cross-multiplied (denominators positive) so that it is exact in integers.  This is the pop
// test `s <= z[k]` of signed_distance_field_generation.cpp:193-197.
__device__ __forceinline__ bool Dominated(int32_t Ga, int a, int32_t Gb, int b, int32_t Gc, int c)
{
  VGT_STAT_ADD(7, 1);
  const int64_t lhs = static_cast<int64_t>(Gc - Gb) * static_cast<int64_t>(b - a);
  const int64_t rhs = static_cast<int64_t>(Gb - Ga) * static_cast<int64_t>(c - b);
  return lhs <= rhs;
}

// First row at which site h beats its predecessor p (p < h): smallest integer q with
// (q-h)^2 + f(h) < (q-p)^2 + f(p)  <=>  q > (Gh - Gp) / (2 (h - p)), i.e. floor(quotient) + 1.
// The callers clamp the result to the run, so it only has to be exact inside [-3, 32769]; outside
// it saturates.  Inside, the float estimate of the quotient is off by less than 2^-7 (|q| <= 2^15,
// relative error < 2^-22), so its floor is at most one away and one exact integer remainder
// check repairs it (no overflow: |q| * divisor < 2^15 * 2^13).
__device__ __forceinline__ int FirstOwnedRow(int32_t Gp, int p, int32_t Gh, int h)
{
  const int num = Gh - Gp;
  const int den = 2 * (h - p);  // 2 .. 4094 in the tiled kernel (n <= 2048)
  float qf = static_cast<float>(num) * __frcp_rn(static_cast<float>(den));
  qf = fminf(fmaxf(qf, -4.0f), 32768.0f);
  int q = static_cast<int>(floorf(qf));
  const int r = num - __mul24(q, den);
  q += (r >= den) ? 1 : 0;
  q -= (r < 0) ? 1 : 0;
  return q + 1;
}

using HullGeom = TileGeom;
constexpr bool kXBeforeY = false;  // see XBeforeY(): measured 15.13 vs 15.21 ms at 1024^3 (neutral), kept off
constexpr int kChordMaxSpacing = 4;  // spacings 1, 2, 4 (13.48 -> 13.28 ms at 1024^3 once the envelope ignores classes)

// Bytes of dynamic LDS for a tile of n rows x W lines.
template <int W>
size_t TileBytes(int n)
{
  const size_t nwords = static_cast<size_t>((n + kBandRows - 1) / kBandRows);
  return static_cast<size_t>(n) * W * sizeof(int32_t) + 4 * nwords * W * sizeof(uint32_t) +
         4 * nwords * W * sizeof(int16_t) + 2 * nwords * W * sizeof(uint16_t) +
         nwords * W * sizeof(uint8_t) + 2 * W * sizeof(uint64_t);
}

// SW = lanes used per line in the transposed scans (32 when a line has <= 32 words, else 64).
template <typename InT, typename OutT, bool kFinal, int W, int SW, int kMagBits>
__global__ __launch_bounds__(1024) void HullPassKernel(const InT* __restrict__ in,
                                                      OutT* __restrict__ out,
                                                      uint32_t* __restrict__ minmax_enc,
                                                      const HullGeom g)
{
  extern __shared__ __align__(16) unsigned char smem[];
  const int n = g.n;
  const int nwords = (n + kBandRows - 1) / kBandRows;
  const int mw = nwords * W;
  constexpr bool kPacked = kMagBits != 0;
  constexpr int32_t kMagMask = (1 << kMagBits) - 1;
  Tile<W, kMagBits> t;
  t.F = reinterpret_cast<int32_t*>(smem);
  t.sumA = reinterpret_cast<uint64_t*>(t.F + static_cast<size_t>(n) * W);
  t.sumA2 = t.sumA + W;
  t.S = reinterpret_cast<uint32_t*>(t.sumA2 + W);
  t.A = t.S + mw;
  t.A2 = t.A + mw;
  t.T = t.A2 + mw;
  t.last_neg = reinterpret_cast<int16_t*>(t.T + mw);
  t.last_pos = t.last_neg + mw;
  t.next_neg = t.last_pos + mw;
  t.next_pos = t.next_neg + mw;
  t.cumA = reinterpret_cast<uint16_t*>(t.next_pos + mw);
  t.cumT = t.cumA + mw;
  t.seed_lo = reinterpret_cast<int16_t*>(t.cumA);
  t.seed_hi = reinterpret_cast<int16_t*>(t.cumT);
  t.first_start = reinterpret_cast<int16_t*>(t.cumA);
  t.min_member = reinterpret_cast<uint8_t*>(t.cumT + mw);
  t.n = n;
  t.nwords = nwords;
  t.w = threadIdx.x % W;

  // Workgroups are dealt round-robin to the 8 XCDs (block b runs on XCD b % 8).  Within every
  // group of 8 tile rows give each XCD one whole row of z tiles, so that the tiles resident on
  // one XCD at a time are neighbours along z and their short row segments add up to long
  // contiguous rows in that XCD's L2 and in the DRAM pages behind it.
  int tile_id = blockIdx.x;
  {
    const int group = kNumXcd * g.ztiles;
    if (!(VGT_SKIP(64)) && tile_id < static_cast<int>(gridDim.x) / group * group)
    {
      const int local = tile_id % group;
      tile_id = tile_id - local + (local % kNumXcd) * g.ztiles + local / kNumXcd;
    }
  }
  const int outer = tile_id / g.ztiles;
  const int z0 = (tile_id % g.ztiles) * W;
  const int64_t base = static_cast<int64_t>(outer) * g.outer_stride + z0;
  const int band = threadIdx.x / W;
  const int z = z0 + t.w;
  const bool active = (band < nwords) && (z < g.nz);
#ifdef VGT_HULL_STATS
  const long long clk0 = clock64();
#endif

  // ---- 0. load the tile (coalesced rows, 16-byte chunks when aligned) ----
  LoadTile<InT, W>(in, t.F, n, base, z0, g);
  __syncthreads();
#ifdef VGT_HULL_STATS
  const long long clk1 = clock64();
#endif

  const int r0 = band * kBandRows;
  const int r1 = min(r0 + kBandRows, n);
  uint32_t sbits = 0;

  // ---- 1a. this band's rows into registers; sign word; the band's strongest member (smallest
  // value), published for the neighbouring bands ----
  int32_t fr[kBandRows];  // magnitudes, kInf32 = not a site (also rows past the end)
  uint32_t finite = 0;
  if (band < nwords)
  {
    // (lines beyond nz hold +kInf32 in the tile, so only rows past the end need a guard; a full
    // band reads its 32 rows at constant offsets from one address)
    int32_t raw[kBandRows];
    int32_t* column = t.F + r0 * W + t.w;
    if (r0 + kBandRows <= n)
    {
#pragma unroll
      for (int k = 0; k < kBandRows; k++) raw[k] = column[k * W];
    }
    else
    {
#pragma unroll
      for (int k = 0; k < kBandRows; k++) raw[k] = (r0 + k < n) ? column[k * W] : kInf32;
    }
    // sign word and "no site" word gathered from sign masks (no compares): kInf32 + 1 is the only sum that wraps
    uint32_t none = 0;
#pragma unroll
    for (int k = 0; k < kBandRows; k++)
    {
      const int32_t v = raw[k];
      const int32_t sign = v >> 31;
      sbits |= static_cast<uint32_t>(sign) & (1u << k);
      fr[k] = (v ^ sign) - sign;
      none |= static_cast<uint32_t>(static_cast<int32_t>(static_cast<uint32_t>(fr[k]) + 1u) >> 31) & (1u << k);
    }
    finite = ~none;
    if constexpr (kPacked)
    {
      // magnitudes only in the tile from here on (the signs are in S); rows without a site get clean upper bits too
      if (r0 + kBandRows <= n)
      {
#pragma unroll
        for (int k = 0; k < kBandRows; k++) column[k * W] = fr[k] & kMagMask;
      }
      else
      {
#pragma unroll
        for (int k = 0; k < kBandRows; k++)
          if (r0 + k < n) column[k * W] = fr[k] & kMagMask;
      }
    }
    // strongest member of the band (smallest value), for the seeds
    int32_t min_value = kInf32;
    int best = 255;
#pragma unroll
    for (int k = 0; k < kBandRows; k++)
    {
      if (fr[k] < min_value)
      {
        min_value = fr[k];
        best = k;
      }
    }
    t.S[band * W + t.w] = sbits;
    t.min_member[band * W + t.w] = static_cast<uint8_t>(best);
    t.T[band * W + t.w] = 0u;
  }
  __syncthreads();

#ifdef VGT_HULL_STATS
  const long long clk1a = clock64();
#endif
  // ---- 1b. per word: nearest row of either class below / above ----
  for (int tt = threadIdx.x; tt < W * SW; tt += blockDim.x)
  {
    const int line = tt / SW;
    const int j = tt % SW;
    const bool ok = j < nwords;
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
    // Seeds.  Band j+1 gets a strong member of the lower bands: doubling scan over the bands' published members,
    // each step keeping the candidate with the smaller value at the first row of band j+1.  Any row of the line is
    // a legal seed; the scan picks a good one, not necessarily the best.  Same from above.
    {
      const int off = ok ? t.min_member[j * W + line] : 255;
      const int member = (off != 255) ? (j << 5) + off : -1;
      int32_t member_f = 0;
      if (member >= 0)
      {
        const int32_t v = t.F[member * W + line];
        member_f = kPacked ? (v & kMagMask) : (v < 0 ? -v : v);
      }
      // upwards: boundary row R = first row of band j+1
      {
        const int R = (j + 1) << 5;
        int row = member;
        int32_t f = member_f;
        int32_t val = (row >= 0) ? f + Sq(R - row) : kInf32;
        for (int d = 1; d < SW; d <<= 1)
        {
          const int prow = __shfl_up(row, d, SW);
          const int32_t pf = __shfl_up(f, d, SW);
          if (j >= d && prow >= 0)
          {
            const int32_t pval = pf + Sq(R - prow);
            if (pval < val)
            {
              val = pval;
              row = prow;
              f = pf;
            }
          }
        }
        if (j + 1 < nwords) t.seed_lo[(j + 1) * W + line] = static_cast<int16_t>((VGT_SKIP(16)) ? -1 : row);
        if (j == 0) t.seed_lo[line] = -1;
      }
      // downwards: boundary row Q = last row of band j-1
      {
        const int Q = (j << 5) - 1;
        int row = member;
        int32_t f = member_f;
        int32_t val = (row >= 0) ? f + Sq(row - Q) : kInf32;
        for (int d = 1; d < SW; d <<= 1)
        {
          const int prow = __shfl_down(row, d, SW);
          const int32_t pf = __shfl_down(f, d, SW);
          if (j + d < SW && prow >= 0)
          {
            const int32_t pval = pf + Sq(prow - Q);
            if (pval < val)
            {
              val = pval;
              row = prow;
              f = pf;
            }
          }
        }
        if (ok && j >= 1) t.seed_hi[(j - 1) * W + line] = static_cast<int16_t>((VGT_SKIP(16)) ? -1 : row);
        if (j == nwords - 1) t.seed_hi[j * W + line] = -1;
      }
    }
  }
  __syncthreads();

#ifdef VGT_HULL_STATS
  const long long clk1b = clock64();
#endif
  // ---- 1c. hull of this band, members only; the stack restarts at every class change.
  // The run portions that continue into the neighbouring bands are seeded with the most
  // competitive published member on that side (a real site of the same run): most members that
  // only a far, strong site removes die here, in parallel, instead of in the merge levels. ----
#if VGT_HULL_STATS > 1
  int stat_iters = 0;
#endif
  if (band < nwords)
  {
    uint32_t abits = 0;
    if (z < g.nz && !(VGT_SKIP(8)) && finite)
    {
      // seeds chosen in 1b
      int seed_l = t.seed_lo[band * W + t.w], seed_r = t.seed_hi[band * W + t.w];
      const int32_t Gseed_l = (seed_l >= 0) ? t.Mag(seed_l) + Sq(seed_l) : 0;
      const int32_t Gseed_r = (seed_r >= 0) ? t.Mag(seed_r) + Sq(seed_r) : 0;

#ifdef VGT_HULL_STATS
      const long long clk_seeds = clock64();
#endif
      // Prefilter, branch-free over the register-resident rows: a member that is matched or
      // beaten AT ITS OWN ROW by a site on its left and by a site on its right never owns a row
      // (the left one wins everywhere below, the right one everywhere above), so it need not
      // enter the stack.
      uint32_t cand = finite;
      if (!(VGT_SKIP(32)))
      {
        // One carried site per direction: sweeping up, the carried site is the one that was
        // strictly better at its own row than the site carried before it; a member the carried
        // site matches at the member's own row is "covered from the left".  Same downwards.  The
        // carried sites are real rows of the line (or the seeds), so the kills are valid; the
        // choice of carried site is only a heuristic (the stack below is exact).
        constexpr int32_t kNone = 0x40000000;  // > any finite value (< 3 * 16384^2), no overflow when a square is added
        int32_t cf = (seed_l >= 0) ? Gseed_l - Sq(seed_l) : kNone;
        int cd = (seed_l >= 0) ? r0 - seed_l : 0;
        uint32_t covered_l = 0, covered_r = 0;
#pragma unroll
        for (int k = 0; k < kBandRows; k++)
        {
          const int32_t lv = cf + Sq(cd);
          const bool covered = lv <= fr[k];
          covered_l |= (covered ? 1u : 0u) << k;
          cf = covered ? cf : fr[k];
          cd = covered ? cd + 1 : 1;
        }
        cf = (seed_r >= 0) ? Gseed_r - Sq(seed_r) : kNone;
        cd = (seed_r >= 0) ? seed_r - (r0 + kBandRows - 1) : 0;
#pragma unroll
        for (int k = kBandRows - 1; k >= 0; k--)
        {
          const int32_t rv = cf + Sq(cd);
          const bool covered = rv <= fr[k];
          covered_r |= (covered ? 1u : 0u) << k;
          cf = covered ? cf : fr[k];
          cd = covered ? cd + 1 : 1;
        }
        cand &= ~(covered_l & covered_r);
        // Chord tests at fixed spacings: member k lies on or above the chord of the members d rows
        // below and above it <=> G(k-d) + G(k+d) <= 2 G(k)
        // <=> f(k-d) + f(k+d) + 2 d^2 <= 2 f(k): no multiplications for a constant d.
        if (!(VGT_SKIP(256)))
        {
          uint32_t above = 0;
#pragma unroll
          for (int d = 1; d <= kChordMaxSpacing; d <<= 1)
          {
            // The test as a sign bit, f(k-d) + f(k+d) + 2 d^2 - 1 - 2 f(k) < 0, shifted into the mask from the top
            // row down (one v_alignbit per test instead of compare + select + or).  Rows without a site wrap around;
            // they are masked below.
            uint32_t dom = 0;
#pragma unroll
            for (int k = kBandRows - 1 - d; k >= d; k--)
            {
              const uint32_t lo = static_cast<uint32_t>(fr[k - d]), hi = static_cast<uint32_t>(fr[k + d]);
              const uint32_t diff = lo + hi + (2u * d * d - 1u) - 2u * static_cast<uint32_t>(fr[k]);
              dom = __builtin_amdgcn_alignbit(dom, diff, 31);
            }
            dom <<= d;
            // both neighbours are members
            above |= dom & (finite << d) & (finite >> d);
          }
          cand &= ~above;
        }
      }

#ifdef VGT_HULL_STATS
      const long long clk_pref = clock64();
#if VGT_HULL_STATS == 1
      if (threadIdx.x == 64) { VGT_CLK_ADD(14, clk_seeds - clk1b); VGT_CLK_ADD(15, clk_pref - clk_seeds); VGT_CLK_ADD(12, clk_pref - clk1); }
#endif
#endif
      // stack over the remaining candidates: one predicate test or one push per iteration
      // (lanes never wait for each other's pops); the seed from below is the virtual bottom element
      int top = seed_l, sec = -1;
      const int bottom_seed = seed_l;
      int32_t Gtop = Gseed_l, Gsec = 0;
      auto pop = [&]() {
        VGT_STAT_ADD(8, 1);
        abits &= ~(1u << (top - r0));
        top = sec;
        Gtop = Gsec;
        sec = -1;
        if (top != bottom_seed)
        {
          const uint32_t m = abits & LowMask(top - r0);
          if (m)
          {
            sec = r0 + 31 - __clz(static_cast<int>(m));
            Gsec = t.Mag(sec) + Sq(sec);
          }
          else if (bottom_seed >= 0)
          {
            sec = bottom_seed;
            Gsec = Gseed_l;
          }
        }
      };
      VGT_STAT_ADD(11, __popc(finite));
      VGT_STAT_ADD(13, __popc(cand));
      uint32_t todo = cand;
      int k = todo ? __ffs(static_cast<int>(todo)) - 1 : kBandRows;
      int32_t fcur = (k < kBandRows) ? t.Mag(r0 + k) : 0;
      while (k < kBandRows)
      {
#if VGT_HULL_STATS > 1
        stat_iters++;
#endif
        const int r = r0 + k;
        const int32_t Gc = fcur + Sq(r);
        if (sec >= 0 && !(VGT_SKIP(4)) && Dominated(Gsec, sec, Gtop, top, Gc, r))
        {
          pop();
          continue;  // same candidate again
        }
        sec = top;
        Gsec = Gtop;
        top = r;
        Gtop = Gc;
        abits |= 1u << k;
        todo &= todo - 1u;
        k = todo ? __ffs(static_cast<int>(todo)) - 1 : kBandRows;
        if (k < kBandRows) fcur = t.Mag(r0 + k);
      }
      // the seed from above closes the band's hull
      if (seed_r >= 0)
        while (sec >= 0 && Dominated(Gsec, sec, Gtop, top, Gseed_r, seed_r)) pop();
    }
    else if (z < g.nz && (VGT_SKIP(8)))
    {
      // debug: every finite member survives the band phase
      for (int r = r0; r < r1; r++)
        if (t.Mag(r) != kInf32) abits |= 1u << (r - r0);
    }
    t.A[band * W + t.w] = abits;
#if VGT_HULL_STATS == 1
    if (threadIdx.x == 64) VGT_CLK_ADD(9, clock64() - clk1);
#endif
  }
#if VGT_HULL_STATS > 1
  {
    // [14] sum over lanes of stack-loop iterations, [15] sum over waves of the slowest lane's count
    int wave_max = stat_iters;
    for (int off = 32; off > 0; off >>= 1) wave_max = max(wave_max, __shfl_xor(wave_max, off));
    VGT_STAT_ADD(14, stat_iters);
    if ((threadIdx.x & 63) == 0) VGT_STAT_ADD(15, wave_max);
  }
#endif
  __syncthreads();

  // ---- 1d. summary of A: which mask words of a line are non-empty ----
  for (int tt = threadIdx.x; tt < W * SW; tt += blockDim.x)
  {
    const int line = tt / SW;
    const int j = tt % SW;
    const uint32_t a = (j < nwords) ? t.A[j * W + line] : 0u;
    const uint64_t any = __ballot(a != 0u);
    if (j == 0)
    {
      const int sh = (threadIdx.x & 63) / SW * SW;  // 0 or 32 when SW == 32
      t.sumA[line] = (SW == 64) ? any : ((any >> sh) & 0xffffffffull);
    }
  }
  __syncthreads();
#ifdef VGT_HULL_STATS
  const long long clk2 = clock64();
#endif

  // ---- 2. join the band hulls.  Every band boundary is a junction; its thread
  // removes, from the two hull ends that meet there, every member that the members across the
  // junction dominate (walking outwards as far as needed, across bands).  All junctions work at
  // once; removals only clear bits (LDS atomics) and a removal justified by ANY real site is
  // valid, so concurrent walks cannot corrupt the hull.  Rounds repeat until a whole round
  // removes nothing, at which point every junction has been verified convex on a stable state:
  // band hulls convex + every junction convex = the line's lower envelope. ----
  for (int round = 0; round < 4 * nwords + 8; round++)
  {
    int changed = 0;
    if (active && band > 0 && !(VGT_SKIP(1)))
    {
      {
        // walkers over the members below / above the junction: the current mask word is held in
        // a register (its bits already passed are cleared), the next non-empty word comes from
        // the per-line summary.  A word copy may miss kills made by other junctions meanwhile;
        // a killed member is still a real site, so a removal it justifies stays valid, and the
        // last round (no kills anywhere) sees stable words.
        int down_word = band - 1, up_word = band;
        uint32_t down_rest = t.A[down_word * W + t.w], up_rest = t.A[up_word * W + t.w];
        auto prev_member = [&]() -> int {
          while (down_rest == 0u)
          {
            const uint64_t below = t.sumA[t.w] & ((1ull << down_word) - 1ull);
            if (below == 0ull) return -1;
            down_word = 63 - __clzll(static_cast<long long>(below));
            down_rest = t.A[down_word * W + t.w];
          }
          const int bit = 31 - __clz(static_cast<int>(down_rest));
          down_rest &= ~(1u << bit);
          return (down_word << 5) + bit;
        };
        auto next_member = [&]() -> int {
          while (up_rest == 0u)
          {
            const uint64_t above = (up_word < 63) ? (t.sumA[t.w] & ~((2ull << up_word) - 1ull)) : 0ull;
            if (above == 0ull) return -1;
            up_word = __ffsll(static_cast<long long>(above)) - 1;
            up_rest = t.A[up_word * W + t.w];
          }
          const int bit = __ffs(static_cast<int>(up_rest)) - 1;
          up_rest &= up_rest - 1u;
          return (up_word << 5) + bit;
        };
        int i = prev_member();
        int j = next_member();
        if (i >= 0 && j >= 0)
        {
          int32_t Gi = t.Mag(i) + Sq(i);
          int32_t Gj = t.Mag(j) + Sq(j);
          int a = prev_member();
          int32_t Ga = (a >= 0) ? t.Mag(a) + Sq(a) : 0;
          int c = next_member();
          int32_t Gc = (c >= 0) ? t.Mag(c) + Sq(c) : 0;
          for (;;)
          {
            if (a >= 0 && Dominated(Ga, a, Gi, i, Gj, j))
            {
              atomicAnd(&t.A[(i >> 5) * W + t.w], ~(1u << (i & 31)));
              VGT_STAT_ADD(9, 1);
              changed = 1;
              i = a;
              Gi = Ga;
              a = prev_member();
              Ga = (a >= 0) ? t.Mag(a) + Sq(a) : 0;
              continue;
            }
            if (c >= 0 && Dominated(Gi, i, Gj, j, Gc, c))
            {
              atomicAnd(&t.A[(j >> 5) * W + t.w], ~(1u << (j & 31)));
              VGT_STAT_ADD(9, 1);
              changed = 1;
              j = c;
              Gj = Gc;
              c = next_member();
              Gc = (c >= 0) ? t.Mag(c) + Sq(c) : 0;
              continue;
            }
            break;
          }
        }
      }
    }
    if (!__syncthreads_or(changed)) break;
  }
#ifdef VGT_HULL_STATS
  const long long clk3 = clock64();
#endif

  // ---- 2b. first owned row of every hull member; drop members that own no row ----
  // A member owns [its start, the start of the next member of the line).  Every thread computes the start of each
  // of its members once (the predecessor of the band's first member is found in the bands below), publishes the
  // start of the band's first member, and after a barrier closes its last member with the published start of the
  // next member above.
  {
    uint32_t a2 = 0;
    int pend_row = -1, pend_start = 0;  // the band's last member: its end is the start of the next member above
    auto owns = [&](int member, int start, int end) {
      if (start < end)
      {
        a2 |= 1u << (member - r0);
        atomicOr(&t.T[(start >> 5) * W + t.w], 1u << (start & 31));
        if constexpr (kPacked)
          atomicOr(reinterpret_cast<uint32_t*>(&t.F[start * W + t.w]), static_cast<uint32_t>(member) << kMagBits);
        VGT_STAT_ADD(10, 1);
      }
    };
    if (band < nwords)
    {
      int first_start = -1;
      if (z < g.nz)
      {
        uint32_t bits = t.A[band * W + t.w];
        int prev = -1, prev_start = 0;
        int32_t Gprev = 0;
        while (bits)
        {
          const int h = r0 + __ffs(static_cast<int>(bits)) - 1;
          bits &= bits - 1u;
          const int32_t Gh = t.Mag(h) + Sq(h);
          int start_h;
          if (prev < 0)
          {
            // the band's first member: its predecessor is the nearest member below the band, if any
            start_h = 0;
            const int p = t.PrevBit(t.A, t.sumA, r0, 0);
            if (p >= 0) start_h = max(0, FirstOwnedRow(t.Mag(p) + Sq(p), p, Gh, h));
            first_start = min(start_h, n);
          }
          else
          {
            start_h = max(0, FirstOwnedRow(Gprev, prev, Gh, h));
            owns(prev, prev_start, min(n, start_h));
          }
          prev = h;
          prev_start = start_h;
          Gprev = Gh;
        }
        pend_row = prev;
        pend_start = prev_start;
      }
      t.first_start[band * W + t.w] = static_cast<int16_t>(first_start);
    }
    __syncthreads();
    if (band < nwords)
    {
      if (pend_row >= 0)
      {
        int end = n;
        // first member above this band (sumA may list words that have become empty)
        uint64_t cand = (band < 63) ? (t.sumA[t.w] & ~((2ull << band) - 1ull)) : 0ull;
        while (cand)
        {
          const int jj = __ffsll(static_cast<long long>(cand)) - 1;
          if (t.A[jj * W + t.w])
          {
            end = min(end, static_cast<int>(t.first_start[jj * W + t.w]));
            break;
          }
          cand &= cand - 1ull;
        }
        owns(pend_row, pend_start, end);
      }
      t.A2[band * W + t.w] = a2;
    }
  }
  __syncthreads();

  // ---- 2c. per word: survivors and start bits in lower words, summary of A2 (packed owners: only the summary
  // of the start words, kept in sumA2) ----
  if constexpr (kPacked)
  {
    for (int tt = threadIdx.x; tt < W * SW; tt += blockDim.x)
    {
      const int line = tt / SW;
      const int j = tt % SW;
      const uint32_t s = (j < nwords) ? t.T[j * W + line] : 0u;
      const uint64_t any = __ballot(s != 0u);
      if (j == 0)
      {
        const int sh = (threadIdx.x & 63) / SW * SW;
        t.sumA2[line] = (SW == 64) ? any : ((any >> sh) & 0xffffffffull);
      }
    }
  }
  else
  for (int tt = threadIdx.x; tt < W * SW; tt += blockDim.x)
  {
    const int line = tt / SW;
    const int j = tt % SW;
    const bool ok = j < nwords;
    const uint32_t a = ok ? t.A2[j * W + line] : 0u;
    const uint32_t s = ok ? t.T[j * W + line] : 0u;
    int ca = __popc(a), ct = __popc(s);
    for (int d = 1; d < SW; d <<= 1)
    {
      const int ua = __shfl_up(ca, d, SW), ut = __shfl_up(ct, d, SW);
      if (j >= d)
      {
        ca += ua;
        ct += ut;
      }
    }
    const uint64_t any = __ballot(a != 0u);
    if (ok)
    {
      t.cumA[j * W + line] = static_cast<uint16_t>(ca - __popc(a));
      t.cumT[j * W + line] = static_cast<uint16_t>(ct - __popc(s));
    }
    if (j == 0)
    {
      const int sh = (threadIdx.x & 63) / SW * SW;
      t.sumA2[line] = (SW == 64) ? any : ((any >> sh) & 0xffffffffull);
    }
  }
  __syncthreads();
#ifdef VGT_HULL_STATS
  const long long clk4 = clock64();
#endif

  // ---- 3. evaluate this band's rows, store ----
  uint32_t lo_enc = 0xffffffffu, hi_enc = 0u;
  float lo_value = INFINITY, hi_value = -INFINITY;
  if (active && !(VGT_SKIP(2)))
  {
    const uint32_t tw = t.T[band * W + t.w];
    // owner of r0: the k-th survivor of the line, k = number of start bits at or below r0
    int cur = -1;
    int32_t fcur = 0;
    // iterator over the line's owners (bits of A2): word index and the bits of that word above cur
    [[maybe_unused]] int owner_word = -1;
    [[maybe_unused]] uint32_t owner_rest = 0;
    // packed owners: OR-ed into the owner's value, kNoValue while the line has no member at all
    [[maybe_unused]] int32_t no_owner = 0;
    if constexpr (kPacked)
    {
      // the owner of r0 is written at the nearest start row at or below r0 (row 0 starts the first member's range)
      cur = r0;
      if (!(tw & 1u))
      {
        const uint64_t below = t.sumA2[t.w] & ((1ull << band) - 1ull);
        if (below)
        {
          const int j = 63 - __clzll(static_cast<long long>(below));
          const uint32_t tj = t.T[j * W + t.w];
          const int row = (j << 5) + 31 - __clz(static_cast<int>(tj));
          cur = static_cast<int>(static_cast<uint32_t>(t.Raw(row)) >> kMagBits);
        }
        else
        {
          no_owner = 0x60000000;  // = kNoValue below
        }
      }
    }
    else
    {
      int k = t.cumT[band * W + t.w] + static_cast<int>(tw & 1u);
      if (k > 0)
      {
        int lo = 0, hi = nwords - 1;  // largest word j with cumA[j] < k
        while (lo < hi)
        {
          const int mid = (lo + hi + 1) >> 1;
          if (t.cumA[mid * W + t.w] < k)
            lo = mid;
          else
            hi = mid - 1;
        }
        k -= t.cumA[lo * W + t.w];
        uint32_t x = t.A2[lo * W + t.w];
        int pos = 0;
#pragma unroll
        for (int width = 16; width >= 1; width >>= 1)
        {
          const int cnt = __popc(x & ((1u << width) - 1u));
          if (k > cnt)
          {
            k -= cnt;
            x >>= width;
            pos += width;
          }
        }
        cur = (lo << 5) + pos;
        fcur = t.Mag(cur);
        owner_word = lo;
        owner_rest = t.A2[lo * W + t.w] & ~LowMask(pos + 1);
      }
    }
    // Sentinels instead of validity selects: a missing owner has the value kNoValue, a missing bounding row lies
    // kFarRows rows away, so every candidate is a plain square-plus-value and anything at or above kRealLimit means
    // "no site at all".  An owner that belongs to an earlier run needs no check either: it lies at or beyond the
    // bounding row of the other class, whose candidate is never larger.
    constexpr int32_t kNoValue = 0x60000000;    // above every real squared distance, + 2047^2 stays below 2^31
    constexpr int32_t kFarRows = 36000;         // 36000^2 > 3 * 16384^2, (36000 + 2048)^2 < 2^31
    constexpr int32_t kRealLimit = 0x40000000;
    if (!kPacked && cur < 0)
    {
      cur = r0;
      fcur = kNoValue;
    }
    static_assert(kNoValue == 0x60000000, "no_owner above");
    [[maybe_unused]] const int32_t* column = t.F + r0 * W + t.w;
    bool neg = false;
    int prev_opp = -kFarRows, next_opp = n + kFarRows;
    const uint32_t row_mask = LowMask(r1 - r0);
    // bit k set: row k starts a run (row 0 always does)
    const uint32_t run_starts = ((sbits ^ (sbits << 1)) | 1u) & row_mask;
    const int wi = band * W + t.w;
    OutT* dst = out + (base + static_cast<int64_t>(r0) * g.row_stride + t.w);
    auto evaluate_row = [&](const int k) __attribute__((always_inline)) {
      {
        const int r = r0 + k;
        if constexpr (kPacked)
        {
          // a start row names its owner in its own word; the owner's value is gathered every row (cheaper than a
          // branch that some lane of the wave takes at nearly every row)
          const uint32_t word = static_cast<uint32_t>(column[k * W]);
          cur = ((tw >> k) & 1u) ? static_cast<int>(word >> kMagBits) : cur;
          fcur = t.Mag(cur) | no_owner;
        }
        else if (k > 0 && ((tw >> k) & 1u) && !(VGT_SKIP(4096)))
        {
          // every start bit has its owner: the next bit of A2 (sumA2 lists the non-empty words exactly)
          if (owner_rest == 0u)
          {
            uint64_t above = t.sumA2[t.w];
            if (owner_word >= 0) above &= ~((2ull << owner_word) - 1ull);
            owner_word = __ffsll(static_cast<long long>(above)) - 1;
            owner_rest = t.A2[owner_word * W + t.w];
          }
          cur = (owner_word << 5) + __ffs(static_cast<int>(owner_rest)) - 1;
          owner_rest &= owner_rest - 1u;
          fcur = t.Mag(cur);
        }
        if ((run_starts >> k) & 1u)
        {
          // bounding rows of the other class: inside the band from the sign word in registers,
          // outside it from the per-word carries of phase 1b
          neg = (sbits >> k) & 1u;
          int below = r - 1;
          if (k == 0) below = neg ? t.last_pos[wi] : t.last_neg[wi];
          const uint32_t other_above = (neg ? ~sbits : sbits) & row_mask & ~LowMask(k + 1);
          const int above = other_above ? r0 + __ffs(static_cast<int>(other_above)) - 1
                                        : static_cast<int>(neg ? t.next_pos[wi] : t.next_neg[wi]);
          prev_opp = (below >= 0) ? below : -kFarRows;
          next_opp = (above < n) ? above : n + kFarRows;
        }
        int32_t best = min(SqPlusAsm(r - cur, fcur), min(SqAsm(r - prev_opp), SqAsm(next_opp - r)));
        best = (best >= kRealLimit) ? kInf32 : best;
        if constexpr (kFinal)
        {
          const int x = (g.pass_axis == 0) ? r : outer + g.outer_begin;
          const int y = (g.pass_axis == 0) ? outer + g.outer_begin : r;
          if (g.add_virtual_border) best = ClampToVirtualBorder(best, x, y, z + g.z_offset, g.nx, g.ny, g.nz_global);
#ifdef VGT_HULL_DEBUG
          if (VGT_SKIP(2048))  // timing experiment: what a plain float conversion would cost (results are NOT exact)
            *dst = (neg ? -1.0f : 1.0f) * __fsqrt_rn(static_cast<float>(best)) * static_cast<float>(g.resolution);
          else
#endif
          {
            const float value = DistanceToSdf(best, neg, g.resolution);
            *dst = value;
            // extrema of the field (plain instructions: no NaN can occur, so no canonicalisation is needed)
            asm("v_min_f32 %0, %0, %1" : "+v"(lo_value) : "v"(value));
            asm("v_max_f32 %0, %0, %1" : "+v"(hi_value) : "v"(value));
          }
        }
        else
        {
          *dst = neg ? -best : best;
        }
        dst += g.row_stride;
      }
    };
    if ((n & (kBandRows - 1)) == 0)
    {
      // every band is full: no per-row guard
#pragma unroll
      for (int k = 0; k < kBandRows; k++) evaluate_row(k);
    }
    else
    {
#pragma unroll
      for (int k = 0; k < kBandRows; k++)
        if (k < r1 - r0) evaluate_row(k);
    }
  }
  if constexpr (kFinal)
  {
    if (lo_value <= hi_value)
    {
      lo_enc = EncodeOrdered(lo_value);
      hi_enc = EncodeOrdered(hi_value);
    }
    BlockMinMax(lo_enc, hi_enc, minmax_enc);
  }
#ifdef VGT_HULL_STATS
  __syncthreads();
  if (threadIdx.x == 0)
  {
    const long long clk5 = clock64();
    VGT_CLK_ADD(0, clk1 - clk0);
    VGT_CLK_ADD(11, clk1a - clk1);
    VGT_CLK_ADD(13, clk1b - clk1a);
    VGT_CLK_ADD(1, clk2 - clk1);
    VGT_CLK_ADD(2, clk3 - clk2);
    VGT_CLK_ADD(3, clk4 - clk3);
    VGT_CLK_ADD(4, clk5 - clk4);
    VGT_CLK_ADD(5, 1);
    VGT_CLK_ADD(6, clk5 - clk0);
  }
#endif
}

// Lines per tile for n rows.  The kernel needs ~120 VGPRs, so a CU holds at most 16 of its waves;
// what matters beyond that is how many workgroups share the CU (they are in different phases,
// which keeps the VALUs busy across barriers and load / store phases) without starving the row
// segments.  Measured (ms per SDF, D1): n = 512: 32 lines 2.08, 16 lines 2.19, 8 lines 2.16;
// n = 768: 8.97 / 9.52 / 6.88; n = 1024: 21.6 / 16.4 / 15.7 (4 lines: 17.5); n = 2048 slabs: - / 8.26 / 7.70.
// So: 32 lines (tile <= 64 KiB, two workgroups per CU) up to n = 512, 8 lines above (four or
// more workgroups per CU up to n = 1024, two at 2048; the XCD-aware tile order lets the L2 merge
// the short row segments of neighbouring tiles).  A line may have at most 64 mask words (one
// wave-wide scan), i.e. n <= 2048.
int LinesPerTile(int64_t n)
{
  const int64_t rows = (n + kBandRows - 1) / kBandRows * kBandRows;
  if (rows <= 512) return 32;
  if (rows <= 2048) return 8;
  return 0;
}

template <typename InT, typename OutT, bool kFinal, int W, int SW, int kMagBits = 0>
hipError_t LaunchHull(const InT* in, OutT* out, uint32_t* minmax_enc, const HullGeom& g,
                      int64_t outer_count, hipStream_t stream)
{
  const int nwords = (g.n + kBandRows - 1) / kBandRows;
  const size_t lds = TileBytes<W>(g.n);
  int threads = nwords * W;
  threads = (threads + 63) / 64 * 64;
  auto kernel = HullPassKernel<InT, OutT, kFinal, W, SW, kMagBits>;
  hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds));
  if (err != hipSuccess) return err;
#ifdef VGT_HULL_STATS
  {
    const int stats_base = kFinal ? 16 : 0;
    err = hipMemcpyToSymbolAsync(HIP_SYMBOL(g_hull_stats_base), &stats_base, sizeof(int), 0,
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
hipError_t DispatchHull(const InT* in, OutT* out, uint32_t* minmax_enc, HullGeom g,
                        int64_t outer_count, int64_t max_input, hipStream_t stream, bool* handled)
{
  int W = LinesPerTile(g.n);
#ifdef VGT_HULL_DEBUG  // tile-width experiments (diagnostic build only)
  static const int force_w = getenv("VGT_HULL_W") ? atoi(getenv("VGT_HULL_W")) : 0;
  if (force_w == 32 && g.n <= 1024) W = 32;
  if (force_w == 16 && g.n <= 2048) W = 16;
  if (force_w == 8 && g.n <= 2048) W = 8;
  if (force_w == 4 && g.n <= 2048) W = 4;
#endif
  *handled = (W != 0);
  if (W == 0) return hipSuccess;
  g.ztiles = (g.nz + W - 1) / W;
#ifdef VGT_HULL_DEBUG
  static const int debug_skip = getenv("VGT_HULL_SKIP") ? atoi(getenv("VGT_HULL_SKIP")) : 0;
  g.debug_skip = debug_skip;
#else
  g.debug_skip = 0;
#endif
  constexpr int kVec = 16 / static_cast<int>(sizeof(InT));  // shorter rows load whole rows (W elements)
  g.vector_io = (g.nz % kVec == 0) && (reinterpret_cast<uintptr_t>(in) % 16 == 0);
  if (outer_count * g.ztiles > 0x7fffffffLL)
  {
    *handled = false;
    return hipSuccess;
  }
  const int nwords = (g.n + kBandRows - 1) / kBandRows;
  // packed owners: the row index above the magnitude in one LDS word
  // (strictly below the all-ones magnitude, which stands for "no site" in the tile)
  bool packed22 = (g.n <= 1024) && (max_input < (1 << 22) - 1);
  bool packed21 = (g.n <= 2048) && (max_input < (1 << 21) - 1);
#ifdef VGT_HULL_DEBUG
  static const int no_packed = getenv("VGT_HULL_NO_PACKED") ? atoi(getenv("VGT_HULL_NO_PACKED")) : 0;
  if (no_packed) packed22 = packed21 = false;
#endif
  if (W == 32 && packed22)
    return LaunchHull<InT, OutT, kFinal, 32, 32, 22>(in, out, minmax_enc, g, outer_count, stream);
  if (W == 8 && nwords <= 32 && packed22)
    return LaunchHull<InT, OutT, kFinal, 8, 32, 22>(in, out, minmax_enc, g, outer_count, stream);
  if (W == 8 && nwords > 32 && packed21)
    return LaunchHull<InT, OutT, kFinal, 8, 64, 21>(in, out, minmax_enc, g, outer_count, stream);
  if (W == 32)
    return LaunchHull<InT, OutT, kFinal, 32, 32>(in, out, minmax_enc, g, outer_count, stream);
#ifdef VGT_HULL_DEBUG
  if (W == 4 && nwords <= 32)
    return LaunchHull<InT, OutT, kFinal, 4, 32>(in, out, minmax_enc, g, outer_count, stream);
  if (W == 4)
    return LaunchHull<InT, OutT, kFinal, 4, 64>(in, out, minmax_enc, g, outer_count, stream);
#endif
  if (W == 8 && nwords <= 32)
    return LaunchHull<InT, OutT, kFinal, 8, 32>(in, out, minmax_enc, g, outer_count, stream);
  if (W == 8)
    return LaunchHull<InT, OutT, kFinal, 8, 64>(in, out, minmax_enc, g, outer_count, stream);
  if (nwords <= 32)
    return LaunchHull<InT, OutT, kFinal, 16, 32>(in, out, minmax_enc, g, outer_count, stream);
  return LaunchHull<InT, OutT, kFinal, 16, 64>(in, out, minmax_enc, g, outer_count, stream);
}
}  // namespace

namespace
{
// Geometry of a line pass along X (axis 0) or Y (axis 1) of the [x][y][z] grid.
HullGeom PassGeometry(const SdfParams& p, int axis, int64_t* outer_count)
{
  HullGeom g{};
  g.nz = static_cast<int>(p.nz);
  g.nx = static_cast<int>(p.nx);
  g.ny = static_cast<int>(p.ny);
  g.pass_axis = axis;
  if (axis == 0)
  {
    g.n = static_cast<int>(p.nx);
    g.row_stride = p.ny * p.nz;
    g.outer_stride = p.nz;
    *outer_count = p.ny;
  }
  else
  {
    g.n = static_cast<int>(p.ny);
    g.row_stride = p.nz;
    g.outer_stride = p.ny * p.nz;
    *outer_count = p.nx;
  }
  g.resolution = p.resolution;
  g.add_virtual_border = p.add_virtual_border;
  g.z_offset = static_cast<int>(p.z_offset);
  g.nz_global = static_cast<int>(p.nz_global > 0 ? p.nz_global : p.nz);
  return g;
}

// The separable transform does not care which of X and Y goes first.  X lines are 4 MiB apart per
// row at 1024^3 (a different page for every row segment), Y lines 4 KiB: running X as the MIDDLE
// pass puts the hostile stride on its 2 + 4 bytes per voxel and leaves the final pass, which
// writes the float field, with the friendly one.  Only when both axes fit the tiled kernel (the
// fallback kernels are written for Y-then-X).
bool XBeforeY(const SdfParams& p)
{
  // both passes must be taken by the tiled kernel (see DispatchHull), or neither is swapped
  const int wx = LinesPerTile(p.nx), wy = LinesPerTile(p.ny);
  if (wx == 0 || wy == 0) return false;
  if (p.ny * ((p.nz + wx - 1) / wx) > 0x7fffffffLL || p.nx * ((p.nz + wy - 1) / wy) > 0x7fffffffLL) return false;
#ifdef VGT_HULL_DEBUG
  static const int order = getenv("VGT_HULL_ORDER") ? atoi(getenv("VGT_HULL_ORDER")) : -1;
  if (order >= 0) return order == 1;
#endif
  return kXBeforeY;
}
}  // namespace

hipError_t LaunchPassXHullFinalizeRange(const int32_t* in32, float* sdf, uint32_t* minmax_enc,
                                        const SdfParams& p, int64_t outer_begin, int64_t outer_count_or_all,
                                        hipStream_t stream, bool* handled);

hipError_t LaunchPassYHull(const int16_t* in16, int32_t* out32, const SdfParams& p,
                           hipStream_t stream, bool* handled)
{
  int64_t outer_count = 0;
  const HullGeom g = PassGeometry(p, XBeforeY(p) ? 0 : 1, &outer_count);
  // input: squared Z distances
  const int64_t nzg = p.nz_global > 0 ? p.nz_global : p.nz;
  const int64_t max_input = (nzg - 1) * (nzg - 1);
  return DispatchHull<int16_t, int32_t, false>(in16, out32, nullptr, g, outer_count, max_input, stream, handled);
}

hipError_t LaunchPassXHullFinalize(const int32_t* in32, float* sdf, uint32_t* minmax_enc,
                                   const SdfParams& p, hipStream_t stream, bool* handled)
{
  return LaunchPassXHullFinalizeRange(in32, sdf, minmax_enc, p, 0, -1, stream, handled);
}

bool HullPassesAreTiled(const SdfParams& p)
{
  const int wx = LinesPerTile(p.nx), wy = LinesPerTile(p.ny);
  if (wx == 0 || wy == 0 || XBeforeY(p)) return false;
  return p.ny * ((p.nz + wx - 1) / wx) <= 0x7fffffffLL && p.nx * ((p.nz + wy - 1) / wy) <= 0x7fffffffLL;
}

hipError_t LaunchPassXHullFinalizeRange(const int32_t* in32, float* sdf, uint32_t* minmax_enc,
                                        const SdfParams& p, int64_t outer_begin, int64_t outer_count_or_all,
                                        hipStream_t stream, bool* handled)
{
  int64_t outer_count = 0;
  HullGeom g = PassGeometry(p, XBeforeY(p) ? 1 : 0, &outer_count);
  if (outer_count_or_all >= 0)
  {
    // part of the outer axis (Y positions of the X pass): same strides, shifted base, fewer tiles
    in32 += outer_begin * g.outer_stride;
    sdf += outer_begin * g.outer_stride;
    g.outer_begin = static_cast<int>(outer_begin);
    outer_count = outer_count_or_all;
  }
  // input: squared distances within the plane of Z and the axis of the first line pass
  const int64_t nzg = p.nz_global > 0 ? p.nz_global : p.nz;
  const int64_t first = XBeforeY(p) ? p.nx : p.ny;
  const int64_t max_input = (nzg - 1) * (nzg - 1) + (first - 1) * (first - 1);
  return DispatchHull<int32_t, float, true>(in32, sdf, minmax_enc, g, outer_count, max_input, stream, handled);
}
}  // namespace vgt

#ifdef VGT_HULL_STATS
// Diagnostic build only: read (and clear) the counters.
extern "C" __attribute__((visibility("default"))) int vgt_hip_debug_hull_stats(unsigned long long* out32, int reset)
{
  hipError_t err = hipDeviceSynchronize();
  if (err == hipSuccess)
    err = hipMemcpyFromSymbol(out32, HIP_SYMBOL(vgt::g_hull_stats),
                              32 * sizeof(unsigned long long));
  if (err == hipSuccess && reset)
  {
    unsigned long long zeros[32] = {0};
    err = hipMemcpyToSymbol(HIP_SYMBOL(vgt::g_hull_stats), zeros, sizeof(zeros));
  }
  return err == hipSuccess ? 0 : 2;
}
#endif
