// Lane-per-line sweep passes (Y and X) of the exact signed EDT for gfx950.
//
// One wave owns 64 neighbouring Z positions of one outer index; a lane owns one line of the pass axis and walks
// it row by row, all lanes in lockstep, so every row access of the wave is one contiguous segment (256 B of
// int32 / float, 128 B of int16) for both axes: no tile, no transpose, no partial cache lines on the store side.
// Each lane runs the Felzenszwalb-Huttenlocher stack algorithm (signed_distance_field_generation.cpp:124-226) on
// its own line, in exact integer arithmetic:
//
//   * Classes are ignored while the envelope is built (see edt_hull_kernels.hip): out(q) = min(envelope over ALL
//     rows with cost |F[r]| at q, squared distance to the nearest row of the other class below / above q).
//   * A stack entry is (G = F + row^2, row) -- a point of the lower convex hull.  "The top never owns a point once
//     the new site is there" is the cross-multiplied test of the three points (the pop test `s <= z[k]` of
//     :193-197 without the division): (Gq - Gt)(rt - rs) - (Gt - Gs)(q - rt) < 0, two 64-bit multiply-adds.
//     A new site that cannot beat the top before the last row (Gq - Gt >= 2 (n-1) (q - rt)) is never pushed.
//   * Two sentinel entries at row 0 with values above every real one sit at the bottom of every stack (a site
//     T0 and, below it, V0 = (G(T0) + 1, row 0), which makes T0's own test come out false): the stack is never
//     empty, no depth checks in the loops, and a line without any site evaluates to "no site" by itself.
//   * The three topmost entries live in registers, so a pop never waits for memory unless it is the third in a
//     row; the stack itself is a ring of 32 entries per lane in LDS ([slot][lane]: conflict-free whatever the
//     lanes' depths), spilled to / refilled from a scratch buffer in chunks of 8 entries: 16-row boundaries check
//     the ring once for the whole wave; the evaluation prefetches its next chunk one boundary ahead.
//   * Entries are 32-bit words (22 bits of G, 10 bits of row) when the host can bound G below 2^22 - 2 on lines of
//     at most 1024 rows (both passes of a 1024^3 grid), otherwise 64-bit.
//   * Sweep 2 walks the line backwards: the owner of a row is found by comparing the two topmost members at that
//     row (values along the envelope are unimodal), (Gs - Gt) + 2 q (rt - rs) <= 0 pops.  The distances to the
//     bounding rows of the other class are running counters (sign bits of the line: one word per 32 rows, kept in
//     the scratch buffer between the sweeps); waves whose 64 lines hold one class only skip that part.
//   * X pass: fused sqrt / resolution / sign / virtual border / min-max as in the other variants.
#include "edt_device.hpp"

#include <type_traits>

namespace vgt
{
namespace
{
constexpr int kBand = 32;        // rows held in registers at a time
constexpr int kRing = 32;        // stack entries per lane resident in LDS
constexpr int kChunk = 8;        // entries per spill / refill
constexpr int kFar = 32768;      // "no row of the other class": kFar^2 is above every real squared distance

struct SweepGeom
{
  int n;                 // rows along the pass axis
  int nz;                // extent of the contiguous axis
  int zsegs;             // waves per outer index
  int nbands;            // ceil(n / 32)
  int chunks;            // spill chunks per lane
  int64_t row_stride;    // elements between consecutive rows
  int64_t outer_stride;  // elements between consecutive outer indices
  int nx, ny;
  int pass_axis;         // 0 = X pass (outer = y), 1 = Y pass (outer = x)
  double resolution;
  int add_virtual_border;
  int z_offset, nz_global;
  int outer_begin;
};

template <bool kPacked>
struct Codec;
template <>
struct Codec<true>
{
  using Entry = uint32_t;
  static constexpr int32_t kSentinelG = (1 << 22) - 2;  // T0; V0 = kSentinelG + 1; real G and real results below it
  static __device__ __forceinline__ Entry Pack(int32_t G, int row)
  {
    return (static_cast<uint32_t>(G) << 10) | static_cast<uint32_t>(row);
  }
  static __device__ __forceinline__ int32_t G(Entry e) { return static_cast<int32_t>(e >> 10); }
  static __device__ __forceinline__ int Row(Entry e) { return static_cast<int>(e & 1023u); }
  // a chunk of 8 entries <-> 32 contiguous bytes
  static __device__ __forceinline__ void StoreChunk(Entry* dst, const Entry (&e)[8])
  {
    reinterpret_cast<uint4*>(dst)[0] = make_uint4(e[0], e[1], e[2], e[3]);
    reinterpret_cast<uint4*>(dst)[1] = make_uint4(e[4], e[5], e[6], e[7]);
  }
  static __device__ __forceinline__ void LoadChunk(const Entry* src, Entry (&e)[8])
  {
    const uint4 a = reinterpret_cast<const uint4*>(src)[0], b = reinterpret_cast<const uint4*>(src)[1];
    e[0] = a.x; e[1] = a.y; e[2] = a.z; e[3] = a.w;
    e[4] = b.x; e[5] = b.y; e[6] = b.z; e[7] = b.w;
  }
};
template <>
struct Codec<false>
{
  using Entry = uint2;
  static constexpr int32_t kSentinelG = 1 << 30;  // 3 * 16383^2 < 2^30
  static __device__ __forceinline__ Entry Pack(int32_t G, int row)
  {
    return make_uint2(static_cast<uint32_t>(G), static_cast<uint32_t>(row));
  }
  static __device__ __forceinline__ int32_t G(Entry e) { return static_cast<int32_t>(e.x); }
  static __device__ __forceinline__ int Row(Entry e) { return static_cast<int>(e.y); }
  // a chunk of 8 entries <-> 64 contiguous bytes
  static __device__ __forceinline__ void StoreChunk(Entry* dst, const Entry (&e)[8])
  {
#pragma unroll
    for (int j = 0; j < 4; j++)
      reinterpret_cast<uint4*>(dst)[j] = make_uint4(e[2 * j].x, e[2 * j].y, e[2 * j + 1].x, e[2 * j + 1].y);
  }
  static __device__ __forceinline__ void LoadChunk(const Entry* src, Entry (&e)[8])
  {
#pragma unroll
    for (int j = 0; j < 4; j++)
    {
      const uint4 a = reinterpret_cast<const uint4*>(src)[j];
      e[2 * j] = make_uint2(a.x, a.y);
      e[2 * j + 1] = make_uint2(a.z, a.w);
    }
  }
};

__device__ __forceinline__ uint32_t LowBits(int bits)  // bits in [1, 32]
{
  return (bits >= 32) ? ~0u : ((1u << bits) - 1u);
}

// Magnitude of an input row value: squared distance so far, kInf32 when the row is no site.
__device__ __forceinline__ int32_t Magnitude(int16_t v)
{
  const int32_t a = v < 0 ? -static_cast<int32_t>(v) : static_cast<int32_t>(v);
  return (a == kInf16) ? kInf32 : __mul24(a, a);
}
__device__ __forceinline__ int32_t Magnitude(int32_t v) { return v < 0 ? -v : v; }

template <typename InT, typename OutT, bool kFinal, bool kPacked>
__global__ __launch_bounds__(kWaveSize, 4) void SweepPassKernel(const InT* __restrict__ in, OutT* __restrict__ out,
                                                            typename Codec<kPacked>::Entry* __restrict__ spill,
                                                            uint2* __restrict__ band_info,
                                                            uint32_t* __restrict__ minmax_enc, const SweepGeom g)
{
  using C = Codec<kPacked>;
  using Entry = typename C::Entry;
  constexpr int32_t kLimit = C::kSentinelG;  // values at or above: no site
  constexpr InT kNoRow = static_cast<InT>(sizeof(InT) == 2 ? kInf16 : kInf32);
  __shared__ Entry ring[kRing * kWaveSize];

  const int lane = threadIdx.x;
  const int item = blockIdx.x;
  const int outer = item / g.zsegs;
  const int z = (item - outer * g.zsegs) * kWaveSize + lane;
  const bool live = z < g.nz;
  const int n = g.n;
  const int nbands = g.nbands;
  const int64_t rstride = g.row_stride;
  const int64_t base = static_cast<int64_t>(outer) * g.outer_stride + z;
  Entry* const my_spill = spill + (static_cast<int64_t>(item) * g.chunks * kWaveSize + lane) * kChunk;
  uint2* const my_info = band_info + static_cast<int64_t>(item) * nbands * kWaveSize + lane;
  constexpr int64_t kChunkStride = static_cast<int64_t>(kWaveSize) * kChunk;  // entries between chunks of a lane

  // ---- stack state: entries [0, depth); [0, lo) live in the spill buffer, [lo, depth) in the ring ----
  int depth = 3, lo = 0;
  int32_t Gt = C::kSentinelG, Gs = C::kSentinelG + 1, G3 = 0;  // top, second, third
  int rt = 0, rsec = 0, r3 = 0;
  int A = 0;        // rt - rsec
  int32_t nB = 1;   // Gs - Gt
  ring[0 * kWaveSize + lane] = C::Pack(0, 0);  // never looked at: keeps "third" inside the stack
  ring[1 * kWaveSize + lane] = C::Pack(C::kSentinelG + 1, 0);
  ring[2 * kWaveSize + lane] = C::Pack(C::kSentinelG, 0);

  auto ring_at = [&](int index) -> Entry& { return ring[((index & (kRing - 1)) << 6) + lane]; };
  auto store_chunk = [&](int first) {
    Entry buf[kChunk];
#pragma unroll
    for (int j = 0; j < kChunk; j++) buf[j] = ring_at(first + j);
    C::StoreChunk(my_spill + static_cast<int64_t>(first >> 3) * kChunkStride, buf);
  };
  auto load_chunk = [&](int first, Entry (&buf)[kChunk]) {
    C::LoadChunk(my_spill + static_cast<int64_t>(first >> 3) * kChunkStride, buf);
  };
  auto commit_chunk = [&](int first, const Entry (&buf)[kChunk]) {
#pragma unroll
    for (int j = 0; j < kChunk; j++) ring_at(first + j) = buf[j];
  };

  // =====================================================================================================
  // Sweep 1: build the envelope.
  // =====================================================================================================
  uint32_t any_transition = 0;
  {
    auto load_row = [&](int q) -> InT { return (live && q < n) ? in[base + static_cast<int64_t>(q) * rstride] : kNoRow; };
    auto pop = [&]() {
      depth--;
      Gt = Gs;
      rt = rsec;
      Gs = G3;
      rsec = r3;
      const int index = depth - 3;
      if (index < lo)
      {
        // the entry that becomes the third one is in the spill buffer: bring its chunk back (rare: a long run of pops)
        Entry buf[kChunk];
        lo -= kChunk;
        load_chunk(lo, buf);
        commit_chunk(lo, buf);
      }
      const Entry e = ring_at(index);
      G3 = C::G(e);
      r3 = C::Row(e);
      A = rt - rsec;
      nB = Gs - Gt;
    };

    InT nxt[kBand];
#pragma unroll
    for (int k = 0; k < kBand; k++) nxt[k] = load_row(k);
    int din = kFar;          // distance from the row below this band to the nearest row of the other class below it
    uint32_t prev_bit = 0;   // class of the row below this band
    const int32_t n2m = 2 * (n - 1);
    for (int b = 0; b < nbands; b++)
    {
      InT cur[kBand];
#pragma unroll
      for (int k = 0; k < kBand; k++) cur[k] = nxt[k];
      if (b + 1 < nbands)
      {
#pragma unroll
        for (int k = 0; k < kBand; k++) nxt[k] = load_row((b + 1) * kBand + k);
      }
      uint32_t sw = 0;
      const int r0 = b * kBand;
#pragma unroll
      for (int k = 0; k < kBand; k++)
      {
        if (k % 16 == 0)
        {
          // at most 16 pushes until the next check: make room for them
          while (__any(depth - lo > kRing - 16))
          {
            if (depth - lo > kRing - 16)
            {
              store_chunk(lo);
              lo += kChunk;
            }
          }
        }
        const int q = r0 + k;
        const InT v = cur[k];
        sw = (sw >> 1) | ((v < 0) ? 0x80000000u : 0u);
        const int32_t f = Magnitude(v);
        if (f < kLimit)
        {
          const int32_t G = f + __mul24(q, q);
          int32_t dG = G - Gt;
          int dr = q - rt;
          while (static_cast<int64_t>(dG) * A + static_cast<int64_t>(nB) * dr < 0)
          {
            pop();
            dG = G - Gt;
            dr = q - rt;
          }
          if (dG < __mul24(n2m, dr))
          {
            ring_at(depth) = C::Pack(G, q);
            G3 = Gs;
            r3 = rsec;
            Gs = Gt;
            rsec = rt;
            Gt = G;
            rt = q;
            A = dr;
            nB = -dG;
            depth++;
          }
        }
      }
      // per band: sign word, distance carry for the evaluation's downward counters
      const int valid = min(kBand, n - r0);
      const uint32_t valid_mask = LowBits(valid);
      uint32_t xdn = (sw ^ ((sw << 1) | prev_bit)) & valid_mask;  // bit k: row k differs from the row below it
      if (b == 0) xdn &= ~1u;
      if (live) my_info[static_cast<int64_t>(b) * kWaveSize] = make_uint2(sw, static_cast<uint32_t>(din));
      any_transition |= xdn;
      din = xdn ? (valid - (31 - __clz(static_cast<int>(xdn)))) : min(din + valid, kFar);
      prev_bit = (sw >> (valid - 1)) & 1u;
    }
  }

  // =====================================================================================================
  // Sweep 2: evaluate, last row first.
  // =====================================================================================================
  const bool classes = __any(any_transition != 0u);
  float lo_value = INFINITY, hi_value = -INFINITY;
  {
    int rt2 = 2 * rt;
    Entry pf[kChunk];
    bool pf_valid = false;
    auto refill_step = [&]() {
      if (pf_valid)
      {
        lo -= kChunk;
        commit_chunk(lo, pf);
        pf_valid = false;
      }
      if (lo > 0 && depth - lo <= kRing - kChunk - 4)
      {
        load_chunk(lo - kChunk, pf);
        pf_valid = true;
      }
    };
    auto pop = [&]() {
      depth--;
      Gt = Gs;
      rt = rsec;
      rt2 = 2 * rt;
      Gs = G3;
      rsec = r3;
      const int index = depth - 3;
      if (index < lo)
      {
        if (!pf_valid) load_chunk(lo - kChunk, pf);
        lo -= kChunk;
        commit_chunk(lo, pf);
        pf_valid = false;
      }
      const Entry e = ring_at(index);
      G3 = C::G(e);
      r3 = C::Row(e);
      A = rt - rsec;
      nB = Gs - Gt;
    };

    auto run = [&](auto with_classes) {
      constexpr bool kClasses = decltype(with_classes)::value;
      int dn = kFar;             // distance from the row above the current one to ... (running, see below)
      uint32_t above_bit0 = 0;   // class of the first row of the band above
      uint2 next_info = make_uint2(0u, 0u);
      if (live) next_info = my_info[static_cast<int64_t>(nbands - 1) * kWaveSize];
      for (int b = nbands - 1; b >= 0; b--)
      {
        const uint2 info = next_info;
        if (b > 0 && live) next_info = my_info[static_cast<int64_t>(b - 1) * kWaveSize];
        const uint32_t sw = info.x;
        const int r0 = b * kBand;
        const int valid = min(kBand, n - r0);
        [[maybe_unused]] int dp[kBand];
        [[maybe_unused]] uint32_t xup = 0;
        if constexpr (kClasses)
        {
          const uint32_t prev_bit = (b > 0) ? (next_info.x >> 31) : (sw & 1u);
          const uint32_t xdn = sw ^ ((sw << 1) | prev_bit);   // bit k: row k differs from the row below it
          xup = sw ^ ((sw >> 1) | (above_bit0 << 31));        // bit k: row k differs from the row above it
          if (b == nbands - 1) xup &= ~(1u << (valid - 1));   // nothing above the last row
          int d = static_cast<int>(info.y);
#pragma unroll
          for (int k = 0; k < kBand; k++)
          {
            d = ((xdn >> k) & 1u) ? 1 : d + 1;
            dp[k] = d;
          }
        }
#pragma unroll
        for (int k = kBand - 1; k >= 0; k--)
        {
          if (k < valid)
          {
            if (k % 16 == 15) refill_step();
            const int q = r0 + k;
            const int q2 = 2 * q;
            while (__mul24(A, q2) + nB <= 0) pop();
            uint32_t best = static_cast<uint32_t>(Gt + __mul24(q - rt2, q));
            if constexpr (kClasses)
            {
              dn = ((xup >> k) & 1u) ? 1 : dn + 1;
              const uint32_t dm = static_cast<uint32_t>(min(dp[k], dn));
              best = min(best, dm * dm);
            }
            const bool neg = (sw >> k) & 1u;
            const int32_t d2 = (best >= static_cast<uint32_t>(kLimit)) ? kInf32 : static_cast<int32_t>(best);
            if (live)
            {
              const int64_t idx = base + static_cast<int64_t>(q) * rstride;
              if constexpr (kFinal)
              {
                int32_t clamped = d2;
                if (g.add_virtual_border)
                {
                  const int x = (g.pass_axis == 0) ? q : outer + g.outer_begin;
                  const int y = (g.pass_axis == 0) ? outer + g.outer_begin : q;
                  clamped = ClampToVirtualBorder(d2, x, y, z + g.z_offset, g.nx, g.ny, g.nz_global);
                }
                const float value = DistanceToSdf(clamped, neg, g.resolution);
                out[idx] = value;
                asm("v_min_f32 %0, %0, %1" : "+v"(lo_value) : "v"(value));
                asm("v_max_f32 %0, %0, %1" : "+v"(hi_value) : "v"(value));
              }
              else
              {
                out[idx] = neg ? -d2 : d2;
              }
            }
          }
        }
        above_bit0 = sw & 1u;
        if constexpr (kClasses) dn = min(dn, kFar);
      }
    };
    if (classes)
      run(std::true_type{});
    else
      run(std::false_type{});
  }
  if constexpr (kFinal)
  {
    uint32_t lo_enc = 0xffffffffu, hi_enc = 0u;
    if (lo_value <= hi_value)
    {
      lo_enc = EncodeOrdered(lo_value);
      hi_enc = EncodeOrdered(hi_value);
    }
    BlockMinMax(lo_enc, hi_enc, minmax_enc);
  }
}

int64_t SpillChunks(int64_t n) { return (n + 3 + kChunk - 1) / kChunk + 1; }

template <typename InT, typename OutT, bool kFinal>
hipError_t LaunchSweep(const InT* in, OutT* out, void* scratch, uint32_t* minmax_enc, SweepGeom g, int64_t outer_count,
                       int64_t max_input, hipStream_t stream)
{
  g.zsegs = (g.nz + kWaveSize - 1) / kWaveSize;
  g.nbands = (g.n + kBand - 1) / kBand;
  g.chunks = static_cast<int>(SpillChunks(g.n));
  const int64_t items = outer_count * g.zsegs;
  if (items <= 0) return hipSuccess;
  if (items > 0x7fffffffLL) return hipErrorInvalidValue;
  // scratch: spill chunks (8 bytes per entry reserved), then one (sign word, carry) pair per band and lane
  char* bytes = static_cast<char*>(scratch);
  const size_t spill_bytes = static_cast<size_t>(items) * g.chunks * kWaveSize * kChunk * sizeof(uint2);
  uint2* info = reinterpret_cast<uint2*>(bytes + spill_bytes);
  const int64_t rows = g.n - 1;
  const bool packed = (g.n <= 1024) && (max_input + rows * rows < Codec<true>::kSentinelG);
  const dim3 grid(static_cast<unsigned>(items)), block(kWaveSize);
  if (packed)
    hipLaunchKernelGGL((SweepPassKernel<InT, OutT, kFinal, true>), grid, block, 0, stream, in, out,
                       reinterpret_cast<uint32_t*>(bytes), info, minmax_enc, g);
  else
    hipLaunchKernelGGL((SweepPassKernel<InT, OutT, kFinal, false>), grid, block, 0, stream, in, out,
                       reinterpret_cast<uint2*>(bytes), info, minmax_enc, g);
  return hipGetLastError();
}

SweepGeom SweepGeometry(const SdfParams& p, int axis, int64_t* outer_count)
{
  SweepGeom g{};
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

// Scratch of the sweep passes for a grid: the larger of the two passes' needs.
size_t SweepPassScratchBytes(int64_t nx, int64_t ny, int64_t nz)
{
  const int64_t zsegs = (nz + kWaveSize - 1) / kWaveSize;
  auto pass_bytes = [&](int64_t n, int64_t outer) {
    const int64_t items = outer * zsegs;
    const int64_t nbands = (n + kBand - 1) / kBand;
    return static_cast<size_t>(items) * (SpillChunks(n) * kWaveSize * kChunk * sizeof(uint2) +
                                         nbands * kWaveSize * sizeof(uint2));
  };
  const size_t y = pass_bytes(ny, nx), x = pass_bytes(nx, ny);
  return (y > x ? y : x) + 256;
}

hipError_t LaunchPassYSweep(const int16_t* in16, int32_t* out32, void* scratch, const SdfParams& p, hipStream_t stream)
{
  int64_t outer_count = 0;
  const SweepGeom g = SweepGeometry(p, 1, &outer_count);
  const int64_t nzg = p.nz_global > 0 ? p.nz_global : p.nz;
  const int64_t max_input = (nzg - 1) * (nzg - 1);
  return LaunchSweep<int16_t, int32_t, false>(in16, out32, scratch, nullptr, g, outer_count, max_input, stream);
}

hipError_t LaunchPassXSweepFinalize(const int32_t* in32, float* sdf, uint32_t* minmax_enc, void* scratch,
                                    const SdfParams& p, hipStream_t stream)
{
  int64_t outer_count = 0;
  const SweepGeom g = SweepGeometry(p, 0, &outer_count);
  const int64_t nzg = p.nz_global > 0 ? p.nz_global : p.nz;
  const int64_t max_input = (nzg - 1) * (nzg - 1) + (p.ny - 1) * (p.ny - 1);
  return LaunchSweep<int32_t, float, true>(in32, sdf, scratch, minmax_enc, g, outer_count, max_input, stream);
}
}  // namespace vgt
