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
// and every voxel of the other class with value 0.  A zero-valued site shields everything
// behind it, so a line splits into maximal same-class RUNS that are independent envelope
// problems: sites = members of the run + the one voxel of the other class on either side.
// Total work is one Felzenszwalb-Huttenlocher sweep per line for both classes together.
//
// Parallelisation inside a line: the line is cut into bands of 32 rows, one thread per
// (line, band); lanes of a wave are different lines (neighbouring Z), which keeps them in
// near lockstep.  The envelope is kept as a bit mask (bit = site survives), one 32-bit word
// per band, so "pop" is clearing a bit and stack neighbours are clz/ffs:
//   1. each thread builds the hull of its band (stack algorithm, exact integer predicate),
//   2. hulls of adjacent bands are merged pairwise (log2(bands) levels, tangent walk),
//   3. each thread walks the final hull to evaluate its 32 rows and stores them.
// The domination predicate is evaluated in 64-bit integers, never as a quotient.
#include "edt_device.hpp"

#include <cstdlib>

namespace vgt
{
#ifdef VGT_HULL_STATS
// Diagnostic build only (make STATS=1): per-phase cycle sums and operation counts.
// [0..4] cycles load/local/merge/eval/total (thread 0 of each workgroup), [5] workgroups,
// [6] mask words read, [7] predicate tests, [8] local pops, [9] merge kills,
// [10] bisection steps, [11] evaluation advances, [12] longest merge walk (max)
__device__ unsigned long long g_hull_stats[32];  // [0..15] Y pass, [16..31] X pass
__device__ int g_hull_stats_base;
#define VGT_STAT_ADD(i, v) atomicAdd(&g_hull_stats[g_hull_stats_base + (i)], static_cast<unsigned long long>(v))
#define VGT_STAT_MAX(i, v) atomicMax(&g_hull_stats[g_hull_stats_base + (i)], static_cast<unsigned long long>(v))
#define VGT_STAT_LOCAL(name) unsigned int name = 0
#define VGT_STAT_INC(name) (name)++
#else
#define VGT_STAT_ADD(i, v)
#define VGT_STAT_MAX(i, v)
#define VGT_STAT_LOCAL(name)
#define VGT_STAT_INC(name)
#endif
namespace
{
constexpr int kBandRows = 32;

template <int W>
struct Tile
{
  int32_t* F;   // [n][W]  signed squared value (sign = class, |.| = distance^2 or kInf32)
  uint32_t* S;  // [nwords][W] sign bits (1 = negative class)
  uint32_t* A;  // [nwords][W] survivor bits
  int n;
  int nwords;
  int w;  // this thread's line

  __device__ __forceinline__ int32_t Raw(int row) const { return F[row * W + w]; }
  __device__ __forceinline__ bool Neg(int row) const
  {
    return (S[(row >> 5) * W + w] >> (row & 31)) & 1u;
  }
  // Value of `row` as a site of a run of class `neg`: members carry their value, voxels of the
  // other class are zero-valued sites.
  __device__ __forceinline__ int32_t SiteF(int row, bool neg, bool& member) const
  {
    const int32_t v = Raw(row);
    member = (v < 0) == neg;
    return member ? (v < 0 ? -v : v) : 0;
  }
  // Rows that are sites for a run of class `neg`: survivors, plus every voxel of the other class.
  __device__ __forceinline__ uint32_t KWord(int j, bool neg) const
  {
#ifdef VGT_HULL_STATS
    atomicAdd(&g_hull_stats[g_hull_stats_base + 6], 1ull);
#endif
    const uint32_t s = S[j * W + w];
    const uint32_t a = A[j * W + w];
    const int tail = n - (j << 5);
    const uint32_t valid = (tail >= 32) ? ~0u : ((1u << tail) - 1u);
    return (a | (neg ? ~s : s)) & valid;
  }
  // highest site row in [lo, r), or -1
  __device__ int PrevK(int r, int lo, bool neg) const
  {
    if (r <= lo) return -1;
    int j = (r - 1) >> 5;
    const int jlo = lo >> 5;
    uint32_t m = KWord(j, neg);
    const int rb = r - (j << 5);
    if (rb < 32) m &= (1u << rb) - 1u;
    for (;;)
    {
      if (j == jlo) m &= ~((1u << (lo & 31)) - 1u);
      if (m) return (j << 5) + 31 - __clz(static_cast<int>(m));
      if (j == jlo) return -1;
      j--;
      m = KWord(j, neg);
    }
  }
  // lowest site row in (r, hi), or -1
  __device__ int NextK(int r, int hi, bool neg) const
  {
    const int start = r + 1;
    if (start >= hi) return -1;
    int j = start >> 5;
    const int jhi = (hi - 1) >> 5;
    uint32_t m = KWord(j, neg) & ~((1u << (start & 31)) - 1u);
    for (;;)
    {
      if (j == jhi)
      {
        const int hb = hi - (j << 5);
        if (hb < 32) m &= (1u << hb) - 1u;
      }
      if (m) return (j << 5) + __ffs(static_cast<int>(m)) - 1;
      if (j == jhi) return -1;
      j++;
      m = KWord(j, neg);
    }
  }
  // Neighbour in the site list of the run, restricted to the block [lo, hi) being worked on.
  // The voxel of the other class just outside the block belongs to the list of the block that
  // holds the adjacent end of the run.
  __device__ __forceinline__ int PrevSite(int r, int lo, bool neg) const
  {
    int p = PrevK(r, lo, neg);
    if (p < 0 && lo > 0 && ((Raw(lo - 1) < 0) != neg)) p = lo - 1;
    return p;
  }
  __device__ __forceinline__ int NextSite(int r, int hi, bool neg) const
  {
    int p = NextK(r, hi, neg);
    if (p < 0 && hi < n && ((Raw(hi) < 0) != neg)) p = hi;
    return p;
  }
  __device__ __forceinline__ void Kill(int row) { A[(row >> 5) * W + w] &= ~(1u << (row & 31)); }
};

// Site b (between a and c) never owns a point of the envelope: with G(v) = f(v) + v^2 the
// parabola of c overtakes b no later than b overtakes a,
//   (Gc - Gb) / (2 (c - b)) <= (Gb - Ga) / (2 (b - a)),
// cross-multiplied (denominators positive) so that it is exact in integers.  This is the pop
// test `s <= z[k]` of signed_distance_field_generation.cpp:193-197.
__device__ __forceinline__ bool Dominated(int32_t Ga, int a, int32_t Gb, int b, int32_t Gc, int c)
{
#ifdef VGT_HULL_STATS
  atomicAdd(&g_hull_stats[g_hull_stats_base + 7], 1ull);
#endif
  const int64_t lhs = static_cast<int64_t>(Gc - Gb) * static_cast<int64_t>(b - a);
  const int64_t rhs = static_cast<int64_t>(Gb - Ga) * static_cast<int64_t>(c - b);
  return lhs <= rhs;
}

struct HullGeom
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
  int debug_skip;        // timing experiments only (VGT_HULL_SKIP): 1 merges, 2 evaluation, 4 pops
};

template <typename InT, typename OutT, bool kFinal, int W>
__global__ __launch_bounds__(1024) void HullPassKernel(const InT* __restrict__ in,
                                                      OutT* __restrict__ out,
                                                      uint32_t* __restrict__ minmax_enc,
                                                      const HullGeom g)
{
  extern __shared__ __align__(16) unsigned char smem[];
  const int n = g.n;
  const int nwords = (n + kBandRows - 1) / kBandRows;
  Tile<W> t;
  t.F = reinterpret_cast<int32_t*>(smem);
  t.S = reinterpret_cast<uint32_t*>(t.F + static_cast<size_t>(n) * W);
  t.A = t.S + static_cast<size_t>(nwords) * W;
  t.n = n;
  t.nwords = nwords;
  t.w = threadIdx.x % W;

  const int outer = blockIdx.x / g.ztiles;
  const int z0 = (blockIdx.x % g.ztiles) * W;
  const int64_t base = static_cast<int64_t>(outer) * g.outer_stride + z0;
  const int band = threadIdx.x / W;
  const int z = z0 + t.w;
  const bool active = (band < nwords) && (z < g.nz);
#ifdef VGT_HULL_STATS
  const long long clk0 = clock64();
#endif

  // ---- load: rows are contiguous 4*W-byte (2*W for int16) segments; 16-byte chunks per
  // lane when the tile is full and aligned, several loads in flight per lane either way ----
  {
    constexpr int kVec = 16 / static_cast<int>(sizeof(InT));  // elements per 16-byte chunk
    constexpr int kChunksPerRow = W / kVec;
    constexpr int kBatch = 4;
    if (g.vector_io && (z0 + W <= g.nz))
    {
      using Chunk = __attribute__((__vector_size__(16))) int;
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
  __syncthreads();

#ifdef VGT_HULL_STATS
  const long long clk1 = clock64();
#endif
  const int r0 = band * kBandRows;
  const int r1 = min(r0 + kBandRows, n);
  uint32_t sbits = 0;

  // ---- 1. hull of this band ----
  if (active)
  {
    uint32_t abits = 0;
    bool neg = false, have_run = false;
    int top = -1, sec = -1, zero_bottom = -1, run_bit0 = 0;
    int32_t Gtop = 0, Gsec = 0;

    auto pop = [&]() {
      VGT_STAT_ADD(8, 1);
      abits &= ~(1u << (top - r0));
      top = sec;
      Gtop = Gsec;
      if (top == zero_bottom)
        sec = -1;
      else
      {
        const uint32_t m = abits & ((1u << (top - r0)) - 1u) & ~((1u << run_bit0) - 1u);
        if (m)
        {
          sec = r0 + 31 - __clz(static_cast<int>(m));
          const int32_t v = t.Raw(sec);
          Gsec = (v < 0 ? -v : v) + sec * sec;
        }
        else if (zero_bottom >= 0)
        {
          sec = zero_bottom;
          Gsec = zero_bottom * zero_bottom;
        }
        else
          sec = -1;
      }
    };

    for (int r = r0; r < r1; r++)
    {
      const int32_t v = t.Raw(r);
      const bool sneg = v < 0;
      const int32_t f = sneg ? -v : v;
      if (sneg) sbits |= 1u << (r - r0);
      if (!have_run || sneg != neg)
      {
        bool left_zero;
        if (have_run)
        {
          // the voxel at r (other class) closes the run that ends at r-1
          const int32_t Gc = r * r;
          while (sec >= 0 && Dominated(Gsec, sec, Gtop, top, Gc, r)) pop();
          left_zero = true;
        }
        else
          left_zero = (r0 > 0) && ((t.Raw(r0 - 1) < 0) != sneg);
        neg = sneg;
        have_run = true;
        sec = -1;
        run_bit0 = r - r0;
        if (left_zero)
        {
          top = r - 1;
          Gtop = (r - 1) * (r - 1);
          zero_bottom = r - 1;
        }
        else
        {
          top = -1;
          zero_bottom = -1;
        }
      }
      if (f != kInf32)
      {
        const int32_t Gc = f + r * r;
        while (sec >= 0 && !(g.debug_skip & 4) && Dominated(Gsec, sec, Gtop, top, Gc, r)) pop();
        sec = top;
        Gsec = Gtop;
        top = r;
        Gtop = Gc;
        abits |= 1u << (r - r0);
      }
    }
    if (r1 < n && ((t.Raw(r1) < 0) != neg))
    {
      const int32_t Gc = r1 * r1;
      while (sec >= 0 && Dominated(Gsec, sec, Gtop, top, Gc, r1)) pop();
    }
    t.S[band * W + t.w] = sbits;
    t.A[band * W + t.w] = abits;
  }
  __syncthreads();

#ifdef VGT_HULL_STATS
  const long long clk2 = clock64();
#endif
  // ---- 2. merge hulls of adjacent blocks, doubling the block size each level ----
  for (int half = 1; half < nwords; half <<= 1)
  {
    if (active && !(g.debug_skip & 1) && (band % (2 * half)) == half)
    {
      const int R = r0;  // first row of the right block
      const bool neg = t.Neg(R);
      if (t.Neg(R - 1) == neg)  // the run continues across the block boundary
      {
        const int lo = (band - half) * kBandRows;
        const int hi = min((band + half) * kBandRows, n);
        int i = t.PrevSite(R, lo, neg);
        int j = t.NextSite(R - 1, hi, neg);
        if (i >= 0 && j >= 0)
        {
          VGT_STAT_LOCAL(walk);
          bool mi, mj, ma, mc;
          int32_t Gi = t.SiteF(i, neg, mi) + i * i;
          int32_t Gj = t.SiteF(j, neg, mj) + j * j;
          int a = mi ? t.PrevSite(i, lo, neg) : -1;
          int32_t Ga = (a >= 0) ? t.SiteF(a, neg, ma) + a * a : 0;
          int c = mj ? t.NextSite(j, hi, neg) : -1;
          int32_t Gc = (c >= 0) ? t.SiteF(c, neg, mc) + c * c : 0;
          for (;;)
          {
            if (a >= 0 && Dominated(Ga, a, Gi, i, Gj, j))
            {
              t.Kill(i);
              VGT_STAT_INC(walk);
              i = a;
              Gi = Ga;
              mi = ma;
              a = mi ? t.PrevSite(i, lo, neg) : -1;
              Ga = (a >= 0) ? t.SiteF(a, neg, ma) + a * a : 0;
              continue;
            }
            if (c >= 0 && Dominated(Gi, i, Gj, j, Gc, c))
            {
              t.Kill(j);
              VGT_STAT_INC(walk);
              j = c;
              Gj = Gc;
              mj = mc;
              c = mj ? t.NextSite(j, hi, neg) : -1;
              Gc = (c >= 0) ? t.SiteF(c, neg, mc) + c * c : 0;
              continue;
            }
            break;
          }
#ifdef VGT_HULL_STATS
          VGT_STAT_ADD(9, walk);
          VGT_STAT_MAX(12, walk);
#endif
        }
      }
    }
    __syncthreads();
  }

#ifdef VGT_HULL_STATS
  const long long clk3 = clock64();
#endif
  // ---- 3. evaluate this band's rows against the final hull, store ----
  uint32_t lo_enc = 0xffffffffu, hi_enc = 0u;
  if (active && !(g.debug_skip & 2))
  {
    bool neg = false, have = false;
    int cur = -1, nxt = -1;
    int32_t fcur = 0, fnxt = 0;
    for (int r = r0; r < r1; r++)
    {
      const bool sneg = (sbits >> (r - r0)) & 1u;
      if (!have || sneg != neg)
      {
        neg = sneg;
        bool mcur = false;
        if (have)
        {
          cur = r - 1;  // the voxel of the other class right before the run
          fcur = 0;
        }
        else
        {
          // first row of the band: locate the owner of r in the hull of the run
          cur = t.PrevSite(r + 1, 0, neg);
          if (cur < 0) cur = t.NextSite(r, n, neg);
          if (cur >= 0)
          {
            // The owner o of row r satisfies (r-o)^2 <= value of ANY site at r, so it lies
            // within R rows of r.  Along the hull "the successor is strictly better at r" holds
            // exactly for the sites before the owner, so bisect on the row position.
            fcur = t.SiteF(cur, neg, mcur);
            const int32_t v0 = (r - cur) * (r - cur) + fcur;
            const int R = static_cast<int>(sqrtf(static_cast<float>(v0))) + 1;
            int lo = max(r - R, 0);
            int hi = min(r + R, n - 1);
            while (lo < hi)
            {
              VGT_STAT_ADD(10, 1);
              const int mid = (lo + hi) >> 1;
              bool successor_better = true;  // no site at or below mid: the owner is above
              const int h = t.PrevSite(mid + 1, 0, neg);
              if (h >= 0)
              {
                bool mh;
                const int32_t fh = t.SiteF(h, neg, mh);
                successor_better = false;
                if (mh || h < r)
                {
                  const int hn = t.NextSite(h, n, neg);
                  if (hn >= 0)
                  {
                    bool mhn;
                    const int32_t fhn = t.SiteF(hn, neg, mhn);
                    successor_better =
                        ((r - hn) * (r - hn) + fhn) < ((r - h) * (r - h) + fh);
                  }
                }
              }
              if (successor_better)
                lo = mid + 1;
              else
                hi = mid;
            }
            const int owner = t.PrevSite(lo + 1, 0, neg);
            if (owner >= 0)
            {
              cur = owner;
              fcur = t.SiteF(cur, neg, mcur);
            }
          }
        }
        have = true;
        nxt = -1;
        if (cur >= 0 && (cur < r || ((t.Raw(cur) < 0) == neg)))
        {
          nxt = t.NextSite(cur, n, neg);
          if (nxt >= 0)
          {
            bool mn;
            fnxt = t.SiteF(nxt, neg, mn);
          }
        }
      }
      int32_t best = kInf32;
      if (cur >= 0)
      {
        best = (r - cur) * (r - cur) + fcur;
        while (nxt >= 0)
        {
          const int32_t vn = (r - nxt) * (r - nxt) + fnxt;
          if (vn >= best) break;
          cur = nxt;
          VGT_STAT_ADD(11, 1);
          fcur = fnxt;
          best = vn;
          const bool member = (t.Raw(cur) < 0) == neg;
          nxt = member ? t.NextSite(cur, n, neg) : -1;
          if (nxt >= 0)
          {
            bool mn;
            fnxt = t.SiteF(nxt, neg, mn);
          }
        }
      }
      const int64_t idx = base + static_cast<int64_t>(r) * g.row_stride + t.w;
      if constexpr (kFinal)
      {
        const int x = (g.pass_axis == 0) ? r : outer;
        const int y = (g.pass_axis == 0) ? outer : r;
        const float v = FinalizeSdf(best, neg, x, y, z, g.nx, g.ny, g.nz, g.resolution,
                                    g.add_virtual_border);
        out[idx] = v;
        const uint32_t e = EncodeOrdered(v);
        lo_enc = min(lo_enc, e);
        hi_enc = max(hi_enc, e);
      }
      else
      {
        out[idx] = neg ? -best : best;
      }
    }
  }
  if constexpr (kFinal) BlockMinMax(lo_enc, hi_enc, minmax_enc);
#ifdef VGT_HULL_STATS
  __syncthreads();
  if (threadIdx.x == 0)
  {
    const long long clk4 = clock64();
    VGT_STAT_ADD(0, clk1 - clk0);
    VGT_STAT_ADD(1, clk2 - clk1);
    VGT_STAT_ADD(2, clk3 - clk2);
    VGT_STAT_ADD(3, clk4 - clk3);
    VGT_STAT_ADD(4, clk4 - clk0);
    VGT_STAT_ADD(5, 1);
  }
#endif
}

// Lines per tile for n rows: the F tile (n * W * 4 bytes) must fit in 128 KiB of LDS and the
// (line, band) threads in one workgroup.
int LinesPerTile(int64_t n)
{
  const int64_t rows = (n + kBandRows - 1) / kBandRows * kBandRows;
  if (rows * 32 <= 32768) return 32;
  if (rows * 16 <= 32768) return 16;
  if (rows * 8 <= 32768) return 8;
  return 0;
}

template <typename InT, typename OutT, bool kFinal, int W>
hipError_t LaunchHull(const InT* in, OutT* out, uint32_t* minmax_enc, const HullGeom& g,
                      int64_t outer_count, hipStream_t stream)
{
  const int nwords = (g.n + kBandRows - 1) / kBandRows;
  const size_t lds = static_cast<size_t>(g.n) * W * sizeof(int32_t) +
                     2 * static_cast<size_t>(nwords) * W * sizeof(uint32_t);
  int threads = nwords * W;
  threads = (threads + 63) / 64 * 64;
  auto kernel = HullPassKernel<InT, OutT, kFinal, W>;
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
                        int64_t outer_count, hipStream_t stream, bool* handled)
{
  const int W = LinesPerTile(g.n);
  *handled = (W != 0);
  if (W == 0) return hipSuccess;
  g.ztiles = (g.nz + W - 1) / W;
  static const int debug_skip = getenv("VGT_HULL_SKIP") ? atoi(getenv("VGT_HULL_SKIP")) : 0;
  g.debug_skip = debug_skip;
  constexpr int kVec = 16 / static_cast<int>(sizeof(InT));
  g.vector_io = (g.nz % kVec == 0) && (reinterpret_cast<uintptr_t>(in) % 16 == 0) && (W % kVec == 0);
  if (outer_count * g.ztiles > 0x7fffffffLL)
  {
    *handled = false;
    return hipSuccess;
  }
  switch (W)
  {
    case 32:
      return LaunchHull<InT, OutT, kFinal, 32>(in, out, minmax_enc, g, outer_count, stream);
    case 16:
      return LaunchHull<InT, OutT, kFinal, 16>(in, out, minmax_enc, g, outer_count, stream);
    default:
      return LaunchHull<InT, OutT, kFinal, 8>(in, out, minmax_enc, g, outer_count, stream);
  }
}
}  // namespace

hipError_t LaunchPassYHull(const int16_t* in16, int32_t* out32, const SdfParams& p,
                           hipStream_t stream, bool* handled)
{
  HullGeom g{};
  g.n = static_cast<int>(p.ny);
  g.nz = static_cast<int>(p.nz);
  g.row_stride = p.nz;
  g.outer_stride = p.ny * p.nz;
  g.nx = static_cast<int>(p.nx);
  g.ny = static_cast<int>(p.ny);
  g.pass_axis = 1;
  g.resolution = p.resolution;
  g.add_virtual_border = p.add_virtual_border;
  return DispatchHull<int16_t, int32_t, false>(in16, out32, nullptr, g, p.nx, stream, handled);
}

hipError_t LaunchPassXHullFinalize(const int32_t* in32, float* sdf, uint32_t* minmax_enc,
                                   const SdfParams& p, hipStream_t stream, bool* handled)
{
  HullGeom g{};
  g.n = static_cast<int>(p.nx);
  g.nz = static_cast<int>(p.nz);
  g.row_stride = p.ny * p.nz;
  g.outer_stride = p.nz;
  g.nx = static_cast<int>(p.nx);
  g.ny = static_cast<int>(p.ny);
  g.pass_axis = 0;
  g.resolution = p.resolution;
  g.add_virtual_border = p.add_virtual_border;
  return DispatchHull<int32_t, float, true>(in32, sdf, minmax_enc, g, p.ny, stream, handled);
}
}  // namespace vgt

#ifdef VGT_HULL_STATS
// Diagnostic build only: read (and clear) the counters.
extern "C" int vgt_hip_debug_hull_stats(unsigned long long* out32, int reset)
{
  hipError_t err = hipDeviceSynchronize();
  if (err == hipSuccess) err = hipMemcpyFromSymbol(out32, HIP_SYMBOL(vgt::g_hull_stats), 32 * sizeof(unsigned long long));
  if (err == hipSuccess && reset)
  {
    unsigned long long zeros[32] = {0};
    err = hipMemcpyToSymbol(HIP_SYMBOL(vgt::g_hull_stats), zeros, sizeof(zeros));
  }
  return err == hipSuccess ? 0 : 2;
}
#endif
