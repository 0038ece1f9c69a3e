// LDS-tiled lower-envelope line passes (Y and X) of the exact signed EDT for gfx950, second
// generation ("sequential bands").  Same tile and the same mathematics as edt_hull_kernels.hip --
// one workgroup owns all n rows of the pass axis x W adjacent Z positions, the tile sits in LDS as
// signed squared distances F[row][line], a line splits at class changes into runs and
//     out(q) = min( lower envelope of the run's members at q, (q-(a-1))^2, ((b+1)-q)^2 )
// -- but with far fewer instructions per voxel: the passes are VALU-issue bound on gfx950 (most
// integer instructions cost 4 cycles per wave), so this version keeps only what the algorithm
// needs and nothing that is per row unless it has to be.
//
// Threads are (line, band of 32 rows); an envelope is a 32-bit survivor mask per band.  Phases:
//   H  every thread walks its 32 rows once, in order, straight from LDS: sign bits, and the
//      classic stack construction of the band's envelope (restarting at class changes), the top
//      two entries in registers, "pop" = clear a mask bit.  Exact integer predicate
//      (64-bit cross multiplication) = the pop test `s <= z[k]` of
//      signed_distance_field_generation.cpp:193-197.
//   C  per line: carries of the nearest row of either class below / above every band (wave
//      scans) and the summary word of non-empty mask words.
//   J  the band envelopes are joined at every band boundary at once (walks with atomic bit
//      clears), in rounds, until a round changes nothing: band hulls convex + junctions convex =
//      the line's lower envelope.
//   E  every thread walks its 32 rows again and evaluates them against the envelope.  Along the
//      envelope the value of the members at a fixed row is unimodal (it falls up to the row's
//      owner and rises after it) and the owner moves monotonically with the row, so the walk keeps
//      (owner, next member) and advances while the next member is at least as good at the current
//      row: no intersection abscissae, no divisions, no rank / select structures.  The same
//      property finds the owner of the band's first row by a local walk from the nearest member.
//      The X pass finishes here: sqrt / resolution / sign / virtual border / min-max, coalesced
//      row stores.
#include "edt_tile.hpp"

#include <cstdlib>

namespace vgt
{
#ifdef VGT_HULL_DEBUG
// Diagnostic build only: per-phase cycle sums (workgroup clock at the phase barriers).
// [0] load [1] H [2] C [3] J [4] E [5] workgroups [6] junction rounds; +8 for the final (X) pass
__device__ unsigned long long g_seq_stats[16];
#define VGT_SEQ_CLK(var) const long long var = clock64()
#define VGT_SEQ_ADD(i, v) atomicAdd(&g_seq_stats[(kFinal ? 8 : 0) + (i)], static_cast<unsigned long long>(v))
#define VGT_SEQ_SKIP(bit) (g.debug_skip & (bit))
#else
#define VGT_SEQ_CLK(var)
#define VGT_SEQ_ADD(i, v)
#define VGT_SEQ_SKIP(bit) false
#endif
namespace
{
constexpr int32_t kNoMember = 0x60000000;   // value of "no member": above every real squared distance (<= 3 * 16384^2),
                                            // low enough that adding a row offset squared (< 2^23) cannot overflow
constexpr int32_t kFarBelow = -36000;       // row of "no voxel of the other class below": (q + 36000)^2 is above every
constexpr int32_t kFarAbove = 36000;        // real value and below 2^31 for q < 2048; above: n + 36000
constexpr int32_t kRealLimit = 0x40000000;  // results at or above this are "no site at all"

template <int W>
struct SeqTile
{
  int32_t* F;          // [n][W]  signed squared value (sign = class, |.| = distance^2 or kInf32)
  uint32_t* S;         // [nwords][W] sign bits (1 = negative class)
  uint32_t* A;         // [nwords][W] envelope bits
  int16_t* last_neg;   // [nwords][W] highest negative row below the word, -1 if none
  int16_t* last_pos;
  int16_t* next_neg;   // [nwords][W] lowest negative row above the word, n if none
  int16_t* next_pos;
  uint64_t* sumA;      // [W] words of A that may be non-empty (superset)
  int n;
  int nwords;
  int w;  // this thread's line

  __device__ __forceinline__ int32_t Mag(int row) const
  {
    const int32_t v = F[row * W + w];
    return v < 0 ? -v : v;
  }
  __device__ __forceinline__ uint32_t Valid(int j) const { return LowMask(n - (j << 5)); }
  __device__ __forceinline__ bool Neg(int row) const
  {
    return (S[(row >> 5) * W + w] >> (row & 31)) & 1u;
  }
  // nearest row of the OTHER class strictly below `row` (-1 if none)
  __device__ __forceinline__ int PrevOpp(int row, bool neg) const
  {
    const int j = row >> 5;
    const uint32_t s = S[j * W + w];
    const uint32_t m = (neg ? ~s : s) & LowMask(row & 31);
    if (m) return (j << 5) + 31 - __clz(static_cast<int>(m));
    return neg ? last_pos[j * W + w] : last_neg[j * W + w];
  }
  // nearest row of the OTHER class strictly above `row` (n if none)
  __device__ __forceinline__ int NextOpp(int row, bool neg) const
  {
    const int j = row >> 5;
    const uint32_t s = S[j * W + w];
    const uint32_t m = (neg ? ~s : s) & Valid(j) & ~LowMask((row & 31) + 1);
    if (m) return (j << 5) + __ffs(static_cast<int>(m)) - 1;
    return neg ? next_pos[j * W + w] : next_neg[j * W + w];
  }
  // highest set bit of A over rows [lo, r), or -1
  __device__ int PrevBit(int r, int lo) const
  {
    if (r <= lo) return -1;
    const int j = (r - 1) >> 5;
    const int jlo = lo >> 5;
    uint32_t m = A[j * W + w] & LowMask(r - (j << 5));
    if (j == jlo) m &= ~LowMask(lo & 31);
    if (m) return (j << 5) + 31 - __clz(static_cast<int>(m));
    if (j == jlo) return -1;
    uint64_t cand = sumA[w] & ((1ull << j) - 1ull) & ~((1ull << jlo) - 1ull);
    while (cand)
    {
      const int jj = 63 - __clzll(static_cast<long long>(cand));
      m = A[jj * W + w];
      if (jj == jlo) m &= ~LowMask(lo & 31);
      if (m) return (jj << 5) + 31 - __clz(static_cast<int>(m));
      cand &= ~(1ull << jj);
    }
    return -1;
  }
  // lowest set bit of A over rows (r, hi), or -1  (r may be -1)
  __device__ int NextBit(int r, int hi) const
  {
    const int start = r + 1;
    if (start >= hi) return -1;
    const int j = start >> 5;
    const int jhi = (hi - 1) >> 5;
    uint32_t m = A[j * W + w] & ~LowMask(start & 31);
    if (j == jhi) m &= LowMask(hi - (j << 5));
    if (m) return (j << 5) + __ffs(static_cast<int>(m)) - 1;
    if (j == jhi) return -1;
    uint64_t cand = sumA[w] & ~((2ull << j) - 1ull);
    if (jhi < 63) cand &= (2ull << jhi) - 1ull;
    while (cand)
    {
      const int jj = __ffsll(static_cast<long long>(cand)) - 1;
      m = A[jj * W + w];
      if (jj == jhi) m &= LowMask(hi - (jj << 5));
      if (m) return (jj << 5) + __ffs(static_cast<int>(m)) - 1;
      cand &= cand - 1ull;
    }
    return -1;
  }
};

// Site b (between a and c) never owns a point of the envelope: with G(v) = f(v) + v^2,
//   (Gc - Gb) (b - a) <= (Gb - Ga) (c - b)       (exact in 64-bit integers)
__device__ __forceinline__ bool SeqDominated(int32_t Ga, int a, int32_t Gb, int b, int32_t Gc, int c)
{
  const int64_t lhs = static_cast<int64_t>(Gc - Gb) * static_cast<int64_t>(b - a);
  const int64_t rhs = static_cast<int64_t>(Gb - Ga) * static_cast<int64_t>(c - b);
  return lhs <= rhs;
}

template <int W>
size_t SeqTileBytes(int n)
{
  const size_t nwords = static_cast<size_t>((n + kBandRows - 1) / kBandRows);
  return static_cast<size_t>(n) * W * sizeof(int32_t) + 2 * nwords * W * sizeof(uint32_t) +
         4 * nwords * W * sizeof(int16_t) + W * sizeof(uint64_t);
}

// SW = lanes used per line in the transposed scans (32 when a line has <= 32 words, else 64).
template <typename InT, typename OutT, bool kFinal, int W, int SW>
__global__ __launch_bounds__(1024) void SeqPassKernel(const InT* __restrict__ in,
                                                     OutT* __restrict__ out,
                                                     uint32_t* __restrict__ minmax_enc,
                                                     const TileGeom g)
{
  extern __shared__ __align__(16) unsigned char smem[];
  const int n = g.n;
  const int nwords = (n + kBandRows - 1) / kBandRows;
  const int mw = nwords * W;
  SeqTile<W> t;
  t.F = reinterpret_cast<int32_t*>(smem);
  t.sumA = reinterpret_cast<uint64_t*>(t.F + static_cast<size_t>(n) * W);
  t.S = reinterpret_cast<uint32_t*>(t.sumA + W);
  t.A = t.S + mw;
  t.last_neg = reinterpret_cast<int16_t*>(t.A + mw);
  t.last_pos = t.last_neg + mw;
  t.next_neg = t.last_pos + mw;
  t.next_pos = t.next_neg + mw;
  t.n = n;
  t.nwords = nwords;
  t.w = threadIdx.x % W;

  // XCD-aware tile order: within every group of 8 tile rows each XCD works on one whole row of z tiles
  // (neighbouring short row segments meet in that XCD's L2 and in the DRAM pages behind it).
  int tile_id = blockIdx.x;
  {
    const int group = kNumXcd * g.ztiles;
    if (tile_id < static_cast<int>(gridDim.x) / group * group)
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

  VGT_SEQ_CLK(clk0);
  LoadTile<InT, W>(in, t.F, n, base, z0, g);
  __syncthreads();
  VGT_SEQ_CLK(clk1);

  const int r0 = band * kBandRows;
  const int r1 = min(r0 + kBandRows, n);
  const int nrows = r1 - r0;
  uint32_t sbits = 0;

  // ---- H. sign word and envelope of this band ----
  if (band < nwords)
  {
    uint32_t abits = 0;
    int top = -1, sec = -1, run_bit0 = 0;
    int32_t Gtop = 0, Gsec = 0;
    bool run_neg = false;
    const int32_t* column = t.F + r0 * W + t.w;
    for (int k = 0; k < nrows; k++)
    {
      const int32_t v = column[k * W];
      const bool neg = v < 0;
      const int32_t f = neg ? -v : v;
      if (k == 0 || neg != run_neg)
      {
        run_neg = neg;
        run_bit0 = k;
        top = sec = -1;
      }
      sbits |= (neg ? 1u : 0u) << k;
      if (f != kInf32)
      {
        const int r = r0 + k;
        const int32_t G = f + Sq(r);
        while (sec >= 0 && SeqDominated(Gsec, sec, Gtop, top, G, r))
        {
          // pop: the entry below the old second becomes the new second
          abits &= ~(1u << (top - r0));
          top = sec;
          Gtop = Gsec;
          const uint32_t m = abits & LowMask(top - r0) & ~LowMask(run_bit0);
          sec = -1;
          if (m)
          {
            sec = r0 + 31 - __clz(static_cast<int>(m));
            Gsec = t.Mag(sec) + Sq(sec);
          }
        }
        sec = top;
        Gsec = Gtop;
        top = r;
        Gtop = G;
        abits |= 1u << k;
      }
    }
    t.S[band * W + t.w] = sbits;
    t.A[band * W + t.w] = (z < g.nz) ? abits : 0u;
  }
  __syncthreads();
  VGT_SEQ_CLK(clk2);

  // ---- C. per word: nearest row of either class below / above; summary of A ----
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
    const uint32_t a = ok ? t.A[j * W + line] : 0u;
    const uint64_t any = __ballot(a != 0u);
    if (j == 0)
    {
      const int sh = (threadIdx.x & 63) / SW * SW;  // 0 or 32 when SW == 32
      t.sumA[line] = (SW == 64) ? any : ((any >> sh) & 0xffffffffull);
    }
  }
  __syncthreads();
  VGT_SEQ_CLK(clk3);
#ifdef VGT_HULL_DEBUG
  int rounds_run = 0;
#endif

  // ---- J. join the band envelopes (see the header) ----
  for (int round = 0; round < 4 * nwords + 8; round++)
  {
    int changed = 0;
#ifdef VGT_HULL_DEBUG
    rounds_run++;
#endif
    if (active && band > 0 && !VGT_SEQ_SKIP(1))
    {
      const int R = r0;  // first row above the junction
      const bool neg = (sbits & 1u) != 0u;
      if (t.Neg(R - 1) == neg)  // the run continues across the junction
      {
        const int lo = t.PrevOpp(R, neg) + 1;
        const int hi = t.NextOpp(R - 1, neg);
        int down_word = band - 1, up_word = band;
        uint32_t down_rest = t.A[down_word * W + t.w], up_rest = t.A[up_word * W + t.w];
        auto prev_member = [&]() -> int {
          while (down_rest == 0u)
          {
            const uint64_t below = t.sumA[t.w] & ((1ull << down_word) - 1ull);
            if (below == 0ull) return -1;
            down_word = 63 - __clzll(static_cast<long long>(below));
            if (((down_word << 5) + 31) < lo) return -1;
            down_rest = t.A[down_word * W + t.w];
          }
          const int bit = 31 - __clz(static_cast<int>(down_rest));
          down_rest &= ~(1u << bit);
          const int row = (down_word << 5) + bit;
          return row >= lo ? row : -1;
        };
        auto next_member = [&]() -> int {
          while (up_rest == 0u)
          {
            const uint64_t above = (up_word < 63) ? (t.sumA[t.w] & ~((2ull << up_word) - 1ull)) : 0ull;
            if (above == 0ull) return -1;
            up_word = __ffsll(static_cast<long long>(above)) - 1;
            if ((up_word << 5) >= hi) return -1;
            up_rest = t.A[up_word * W + t.w];
          }
          const int bit = __ffs(static_cast<int>(up_rest)) - 1;
          up_rest &= up_rest - 1u;
          const int row = (up_word << 5) + bit;
          return row < hi ? row : -1;
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
            if (a >= 0 && SeqDominated(Ga, a, Gi, i, Gj, j))
            {
              atomicAnd(&t.A[(i >> 5) * W + t.w], ~(1u << (i & 31)));
              changed = 1;
              i = a;
              Gi = Ga;
              a = prev_member();
              Ga = (a >= 0) ? t.Mag(a) + Sq(a) : 0;
              continue;
            }
            if (c >= 0 && SeqDominated(Gi, i, Gj, j, Gc, c))
            {
              atomicAnd(&t.A[(j >> 5) * W + t.w], ~(1u << (j & 31)));
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

  VGT_SEQ_CLK(clk4);
  // ---- E. evaluate this band's rows against the envelope, store ----
  int32_t lo_key = kInf32, hi_key = -kInf32;
  if (active && !VGT_SEQ_SKIP(2))
  {
    const int wi = band * W + t.w;
    bool neg = false;
    int prev_opp = kFarBelow, next_opp = n + kFarAbove;
    int cur = -1, nxt = -1;
    int32_t fcur = kNoMember, fnxt = kNoMember;
    int run_end = n;  // one past the last row of the current run
    OutT* dst = out + (base + static_cast<int64_t>(r0) * g.row_stride + t.w);
    for (int k = 0; k < nrows; k++, dst += g.row_stride)
    {
      const int q = r0 + k;
      const bool sneg = (sbits >> k) & 1u;
      if (k == 0 || sneg != neg)
      {
        // a run begins (or the band enters one): its bounding rows of the other class, and the owner of q
        neg = sneg;
        int po, no;
        if (k == 0)
          po = neg ? t.last_pos[wi] : t.last_neg[wi];
        else
          po = q - 1;
        const uint32_t other_above = (neg ? ~sbits : sbits) & LowMask(nrows) & ~LowMask(k + 1);
        no = other_above ? r0 + __ffs(static_cast<int>(other_above)) - 1
                         : static_cast<int>(neg ? t.next_pos[wi] : t.next_neg[wi]);
        const int run_a = po + 1;
        run_end = no;
        prev_opp = (po >= 0) ? po : kFarBelow;
        next_opp = (no < n) ? no : n + kFarAbove;
        // owner of q among the run's members: start at the nearest member below q, step down while the
        // member below is at least as good at q (values along the envelope are unimodal at a fixed row)
        cur = t.PrevBit(q, run_a);
        if (cur >= 0)
        {
          fcur = t.Mag(cur);
          int32_t vcur = SqPlusAsm(q - cur, fcur);
          for (;;)
          {
            const int pp = t.PrevBit(cur, run_a);
            if (pp < 0) break;
            const int32_t fpp = t.Mag(pp);
            const int32_t vpp = SqPlusAsm(q - pp, fpp);
            if (vpp > vcur) break;
            cur = pp;
            fcur = fpp;
            vcur = vpp;
          }
          nxt = t.NextBit(cur, run_end);
        }
        else
        {
          cur = t.NextBit(q - 1, run_end);
          fcur = (cur >= 0) ? t.Mag(cur) : kNoMember;
          nxt = (cur >= 0) ? t.NextBit(cur, run_end) : -1;
          if (cur < 0) cur = q;
        }
        fnxt = (nxt >= 0) ? t.Mag(nxt) : kNoMember;
      }
      int32_t vcur = SqPlusAsm(q - cur, fcur);
      while (nxt >= 0)
      {
        const int32_t vnxt = SqPlusAsm(q - nxt, fnxt);
        if (vnxt > vcur) break;
        cur = nxt;
        fcur = fnxt;
        vcur = vnxt;
        nxt = t.NextBit(cur, run_end);
        fnxt = (nxt >= 0) ? t.Mag(nxt) : kNoMember;
      }
      int32_t best = min(vcur, min(SqAsm(q - prev_opp), SqAsm(next_opp - q)));
      best = (best >= kRealLimit) ? kInf32 : best;
      if constexpr (kFinal)
      {
        const int x = (g.pass_axis == 0) ? q : outer;
        const int y = (g.pass_axis == 0) ? outer : q;
        if (g.add_virtual_border) best = ClampToVirtualBorder(best, x, y, z + g.z_offset, g.nx, g.ny, g.nz_global);
        *dst = DistanceToSdf(best, neg, g.resolution);
        const int32_t key = neg ? -best : best;
        lo_key = min(lo_key, key);
        hi_key = max(hi_key, key);
      }
      else
      {
        *dst = neg ? -best : best;
      }
    }
  }
  if constexpr (kFinal)
  {
    uint32_t lo_enc = 0xffffffffu, hi_enc = 0u;
    if (lo_key <= hi_key)
    {
      lo_enc = EncodeOrdered(DistanceToSdf(lo_key < 0 ? -lo_key : lo_key, lo_key < 0, g.resolution));
      hi_enc = EncodeOrdered(DistanceToSdf(hi_key < 0 ? -hi_key : hi_key, hi_key < 0, g.resolution));
    }
    BlockMinMax(lo_enc, hi_enc, minmax_enc);
  }
#ifdef VGT_HULL_DEBUG
  __syncthreads();
  if (threadIdx.x == 0)
  {
    const long long clk5 = clock64();
    VGT_SEQ_ADD(0, clk1 - clk0);
    VGT_SEQ_ADD(1, clk2 - clk1);
    VGT_SEQ_ADD(2, clk3 - clk2);
    VGT_SEQ_ADD(3, clk4 - clk3);
    VGT_SEQ_ADD(4, clk5 - clk4);
    VGT_SEQ_ADD(5, 1);
    VGT_SEQ_ADD(6, rounds_run);
  }
#endif
}

// Lines per tile for n rows (tile = n * W * 4 bytes of LDS plus ~16 bytes per (band, line)).
int SeqLinesPerTile(int64_t n)
{
  const int64_t rows = (n + kBandRows - 1) / kBandRows * kBandRows;
  if (rows <= 512) return 32;
  if (rows <= 2048) return 8;
  return 0;
}

template <typename InT, typename OutT, bool kFinal, int W, int SW>
hipError_t LaunchSeq(const InT* in, OutT* out, uint32_t* minmax_enc, const TileGeom& g,
                     int64_t outer_count, hipStream_t stream)
{
  const int nwords = (g.n + kBandRows - 1) / kBandRows;
  const size_t lds = SeqTileBytes<W>(g.n);
  int threads = nwords * W;
  threads = (threads + 63) / 64 * 64;
  auto kernel = SeqPassKernel<InT, OutT, kFinal, W, SW>;
  hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds));
  if (err != hipSuccess) return err;
  const int64_t blocks = outer_count * g.ztiles;
  hipLaunchKernelGGL(kernel, dim3(static_cast<unsigned>(blocks)), dim3(threads), lds, stream, in,
                     out, minmax_enc, g);
  return hipGetLastError();
}

template <typename InT, typename OutT, bool kFinal>
hipError_t DispatchSeq(const InT* in, OutT* out, uint32_t* minmax_enc, TileGeom g,
                       int64_t outer_count, hipStream_t stream, bool* handled)
{
  int W = SeqLinesPerTile(g.n);
#ifdef VGT_HULL_DEBUG  // tile-width experiments (diagnostic build only)
  static const int force_w = getenv("VGT_HULL_W") ? atoi(getenv("VGT_HULL_W")) : 0;
  if (force_w == 32 && g.n <= 1024) W = 32;
  if (force_w == 16 && g.n <= 2048) W = 16;
  if (force_w == 8 && g.n <= 2048) W = 8;
#endif
  *handled = (W != 0);
  if (W == 0) return hipSuccess;
  g.ztiles = (g.nz + W - 1) / W;
  g.debug_skip = 0;
#ifdef VGT_HULL_DEBUG
  static const int debug_skip = getenv("VGT_SEQ_SKIP") ? atoi(getenv("VGT_SEQ_SKIP")) : 0;
  g.debug_skip = debug_skip;
#endif
  constexpr int kVec = 16 / static_cast<int>(sizeof(InT));
  g.vector_io = (g.nz % kVec == 0) && (reinterpret_cast<uintptr_t>(in) % 16 == 0);
  if (outer_count * g.ztiles > 0x7fffffffLL)
  {
    *handled = false;
    return hipSuccess;
  }
  const int nwords = (g.n + kBandRows - 1) / kBandRows;
  if (W == 32 && nwords <= 32)
    return LaunchSeq<InT, OutT, kFinal, 32, 32>(in, out, minmax_enc, g, outer_count, stream);
  if (W == 16 && nwords <= 32)
    return LaunchSeq<InT, OutT, kFinal, 16, 32>(in, out, minmax_enc, g, outer_count, stream);
  if (W == 16)
    return LaunchSeq<InT, OutT, kFinal, 16, 64>(in, out, minmax_enc, g, outer_count, stream);
  if (nwords <= 32)
    return LaunchSeq<InT, OutT, kFinal, 8, 32>(in, out, minmax_enc, g, outer_count, stream);
  return LaunchSeq<InT, OutT, kFinal, 8, 64>(in, out, minmax_enc, g, outer_count, stream);
}

// Geometry of a line pass along X (axis 0) or Y (axis 1) of the [x][y][z] grid.
TileGeom SeqPassGeometry(const SdfParams& p, int axis, int64_t* outer_count)
{
  TileGeom g{};
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
}  // namespace

hipError_t LaunchPassYSeq(const int16_t* in16, int32_t* out32, const SdfParams& p,
                          hipStream_t stream, bool* handled)
{
  int64_t outer_count = 0;
  const TileGeom g = SeqPassGeometry(p, 1, &outer_count);
  return DispatchSeq<int16_t, int32_t, false>(in16, out32, nullptr, g, outer_count, stream, handled);
}

hipError_t LaunchPassXSeqFinalize(const int32_t* in32, float* sdf, uint32_t* minmax_enc,
                                  const SdfParams& p, hipStream_t stream, bool* handled)
{
  int64_t outer_count = 0;
  const TileGeom g = SeqPassGeometry(p, 0, &outer_count);
  return DispatchSeq<int32_t, float, true>(in32, sdf, minmax_enc, g, outer_count, stream, handled);
}
}  // namespace vgt

#ifdef VGT_HULL_DEBUG
// Diagnostic build only: read (and clear) the per-phase counters of the sequential-band passes.
extern "C" int vgt_hip_debug_seq_stats(unsigned long long* out16, int reset)
{
  hipError_t err = hipDeviceSynchronize();
  if (err == hipSuccess)
    err = hipMemcpyFromSymbol(out16, HIP_SYMBOL(vgt::g_seq_stats), 16 * sizeof(unsigned long long));
  if (err == hipSuccess && reset)
  {
    unsigned long long zeros[16] = {0};
    err = hipMemcpyToSymbol(HIP_SYMBOL(vgt::g_seq_stats), zeros, sizeof(zeros));
  }
  return err == hipSuccess ? 0 : 2;
}
#endif
