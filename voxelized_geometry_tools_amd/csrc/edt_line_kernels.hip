// Line-sweep passes (Y and X) of the exact signed EDT for gfx950: one lane per line, all lanes
// of a wave walking the pass axis in lockstep.
//
// Lanes of a wave are neighbouring Z positions, so every row access of the wave is one
// contiguous segment (256 B of int32, 128 B of int16) for both axes -- no tile, no LDS, no
// workgroup barrier, full occupancy.  Each lane runs the Felzenszwalb-Huttenlocher stack
// algorithm (signed_distance_field_generation.cpp:124-226) on its own line, with two changes
// that keep it exact in integers and cheap:
//   * stack entries carry the FIRST ROW the site owns (an integer) instead of the real-valued
//     intersection z[k]; "site c overtakes the top before the top's first row" is then the
//     division-free test  num < start(top) * den  with  num = G(c) - G(top), den = 2 (c - top),
//     and sites that would own no integer row are dropped at once;
//   * the stack lives in an HBM scratch array shaped like the grid (entry k of the run that
//     starts at row a sits at row a + k <= current row), 8 bytes per entry.  Only the top entry
//     is kept in registers; pushes write through, pops re-read.  Neighbouring lanes have similar
//     stack depths, so these accesses fall into a few cache lines, and entries that are popped
//     soon never leave L2.
// A line splits at class changes into runs (see edt_hull_kernels.hip): the envelope is built per
// run over its members; the voxels of the other class that bound the run are separate
// candidates.  Sweep 1 builds the stacks and leaves, per run, (end row, stack depth) in the
// output slot of the run's first row plus one sign bit per row; sweep 2 walks rows again, reads
// that record when a run starts, advances along the stack whenever the next entry's first row is
// reached, and writes the result (fused sqrt / resolution / sign / virtual border / min-max in
// the X pass).
#include "edt_device.hpp"

namespace vgt
{
namespace
{
struct __attribute__((aligned(8))) StackEntry
{
  int32_t f;       // value of the site
  uint16_t v;      // row of the site
  uint16_t start;  // first row the site owns
};

struct LineGeom
{
  int n;                 // rows along the pass axis
  int nz;                // lines per outer index (= extent of the contiguous axis)
  int64_t num_lines;
  int64_t row_stride;    // elements between consecutive rows
  int64_t outer_stride;  // elements between consecutive outer indices
  int nx, ny;
  int pass_axis;         // 0 = X pass (outer = y), 1 = Y pass (outer = x)
  double resolution;
  int add_virtual_border;
  int z_offset, nz_global;
};

constexpr int kLineBlock = 256;
constexpr int kRowChunk = 8;  // rows whose loads are issued together

template <typename InT, typename OutT, bool kFinal>
__global__ __launch_bounds__(kLineBlock) void LinePassKernel(const InT* __restrict__ in,
                                                            OutT* __restrict__ out,
                                                            StackEntry* __restrict__ stack,
                                                            uint32_t* __restrict__ signw,
                                                            uint32_t* __restrict__ minmax_enc,
                                                            const LineGeom g)
{
  const int64_t line = static_cast<int64_t>(blockIdx.x) * kLineBlock + threadIdx.x;
  const int n = g.n;
  uint32_t lo_enc = 0xffffffffu, hi_enc = 0u;
  if (line < g.num_lines)
  {
    const int outer = static_cast<int>(line / g.nz);
    const int z = static_cast<int>(line - static_cast<int64_t>(outer) * g.nz);
    const int64_t base = static_cast<int64_t>(outer) * g.outer_stride + z;
    const int64_t rs = g.row_stride;
    uint32_t* out_raw = reinterpret_cast<uint32_t*>(out);

    // ---- sweep 1: build the stacks ----
    // The two topmost entries live in registers and are written to the scratch array only when
    // they sink to third place (or when the run closes), so an entry that is popped again soon
    // never touches memory; a pop refills the second register from memory while the next test
    // already runs on the old second entry.  A member that is matched or beaten at its own row
    // by the current top (a site on its left) and by a member 1, 2 or 4 rows ahead (a site on
    // its right, same run) can never own a row and is not pushed at all.
    {
      bool neg = false;
      int run_a = 0, depth = 0;
      int v1 = 0, s1 = 0, v2 = 0, s2 = 0;  // top / second: row, first owned row
      int32_t f1 = 0, f2 = 0;              //               value
      bool dirty2 = false;                 // the second entry has not been written yet
      uint32_t sw = 0;
      int32_t cur[kRowChunk], nxt[kRowChunk];
#pragma unroll
      for (int j = 0; j < kRowChunk; j++)
        nxt[j] = (j < n) ? ToSignedSquare(in[base + static_cast<int64_t>(j) * rs]) : kInf32;

      auto store_entry = [&](int index, int32_t f, int v, int st) {
        StackEntry e;
        e.f = f;
        e.v = static_cast<uint16_t>(v);
        e.start = static_cast<uint16_t>(st);
        stack[base + static_cast<int64_t>(run_a + index) * rs] = e;
      };
      auto close_run = [&](int end_row) {
        if (depth >= 2 && dirty2) store_entry(depth - 2, f2, v2, s2);
        if (depth >= 1) store_entry(depth - 1, f1, v1, s1);
        out_raw[base + static_cast<int64_t>(run_a) * rs] =
            static_cast<uint32_t>(end_row) | (static_cast<uint32_t>(depth) << 16);
      };

      for (int q0 = 0; q0 < n; q0 += kRowChunk)
      {
#pragma unroll
        for (int j = 0; j < kRowChunk; j++) cur[j] = nxt[j];
#pragma unroll
        for (int j = 0; j < kRowChunk; j++)
        {
          const int q = q0 + kRowChunk + j;
          nxt[j] = (q < n) ? ToSignedSquare(in[base + static_cast<int64_t>(q) * rs]) : kInf32;
        }
        // class bits of the 16-row window [q0, q0 + 16)
        uint32_t sgn = 0;
#pragma unroll
        for (int j = 0; j < kRowChunk; j++)
        {
          if (cur[j] < 0) sgn |= 1u << j;
          if (nxt[j] < 0) sgn |= 1u << (j + kRowChunk);
        }
#pragma unroll
        for (int j = 0; j < kRowChunk; j++)
        {
          const int q = q0 + j;
          if (q >= n) break;
          const int32_t v = cur[j];
          const bool sneg = v < 0;
          const int32_t f = sneg ? -v : v;
          if (sneg) sw |= 1u << (q & 31);
          if ((q & 31) == 31 || q == n - 1)
          {
            signw[static_cast<int64_t>(q >> 5) * g.num_lines + line] = sw;
            sw = 0;
          }
          if (q == 0 || sneg != neg)
          {
            if (q > 0) close_run(q);  // the run [run_a, q) is complete
            neg = sneg;
            run_a = q;
            depth = 0;
            dirty2 = false;
          }
          if (f == kInf32) continue;
          // own-row test
          bool right_beaten = false;
          const uint32_t other = (sneg ? ~sgn : sgn) >> j;  // bit d set: row q+d is of the other class
#pragma unroll
          for (int d = 1; d <= 4; d <<= 1)
          {
            const int32_t u = (j + d < kRowChunk) ? cur[j + d] : nxt[j + d - kRowChunk];
            const int32_t fu = u < 0 ? -u : u;
            const bool same_run = (other & ((2u << d) - 1u)) == 0u;
            if (same_run && fu != kInf32 && fu + d * d <= f) right_beaten = true;
          }
          if (right_beaten && depth > 0 && (q - v1) * (q - v1) + f1 <= f) continue;

          int32_t num = 0, den = 1;
          while (depth > 0)
          {
            const int dq = q - v1;
            num = (f - f1) + dq * (q + v1);  // G(q) - G(v1)
            den = 2 * dq;
            if (num >= s1 * den) break;  // the top keeps its first row
            // pop: the second entry becomes the top; refill the second from memory
            depth--;
            v1 = v2;
            f1 = f2;
            s1 = s2;
            dirty2 = false;
            if (depth >= 2)
            {
              const StackEntry e = stack[base + static_cast<int64_t>(run_a + depth - 2) * rs];
              v2 = e.v;
              f2 = e.f;
              s2 = e.start;
            }
          }
          int start = run_a;
          if (depth > 0)
          {
            if (num >= n * den) continue;  // would take over beyond the last row: owns nothing
            // exact floor(num / den): 0 <= num < n * den <= 2^29, quotient < 2^14, so the float
            // estimate is off by at most one
            int quo = static_cast<int>(__fdividef(static_cast<float>(num), static_cast<float>(den)));
            const int rem = num - quo * den;
            quo += (rem >= den) ? 1 : ((rem < 0) ? -1 : 0);
            start = quo + 1;  // > s1 because num >= s1 * den
          }
          // push: the old second sinks to third place and must be in memory from now on
          if (depth >= 2 && dirty2) store_entry(depth - 2, f2, v2, s2);
          v2 = v1;
          f2 = f1;
          s2 = s1;
          dirty2 = (depth >= 1);
          v1 = q;
          f1 = f;
          s1 = start;
          depth++;
        }
      }
      close_run(n);
    }

    // ---- sweep 2: evaluate ----
    {
      bool neg = false;
      int prev_opp = -1, next_opp = n;
      int k = 0, kend = 0;  // stack entries of the current run: rows [k, kend)
      int vc = 0, vn = 0, sn = 0;
      int32_t fc = 0, fn = 0;
      bool have_cur = false;
      uint32_t sw = 0;
      for (int q = 0; q < n; q++)
      {
        if ((q & 31) == 0) sw = signw[static_cast<int64_t>(q >> 5) * g.num_lines + line];
        const bool sneg = (sw >> (q & 31)) & 1u;
        const int64_t idx = base + static_cast<int64_t>(q) * rs;
        if (q == 0 || sneg != neg)
        {
          neg = sneg;
          prev_opp = q - 1;
          const uint32_t info = out_raw[idx];
          next_opp = static_cast<int>(info & 0xffffu);
          const int depth = static_cast<int>(info >> 16);
          k = q;
          kend = q + depth;
          have_cur = depth > 0;
          if (depth > 0)
          {
            const StackEntry e = stack[idx];
            vc = e.v;
            fc = e.f;
          }
          if (depth > 1)
          {
            const StackEntry e = stack[idx + rs];
            vn = e.v;
            fn = e.f;
            sn = e.start;
          }
        }
        while (k + 1 < kend && sn <= q)
        {
          k++;
          vc = vn;
          fc = fn;
          if (k + 1 < kend)
          {
            const StackEntry e = stack[base + static_cast<int64_t>(k + 1) * rs];
            vn = e.v;
            fn = e.f;
            sn = e.start;
          }
        }
        int32_t best = kInf32;
        if (have_cur) best = (q - vc) * (q - vc) + fc;
        if (prev_opp >= 0) best = min(best, (q - prev_opp) * (q - prev_opp));
        if (next_opp < n) best = min(best, (next_opp - q) * (next_opp - q));
        if constexpr (kFinal)
        {
          const int x = (g.pass_axis == 0) ? q : outer;
          const int y = (g.pass_axis == 0) ? outer : q;
          const float val = FinalizeSdf(best, neg, x, y, z + g.z_offset, g.nx, g.ny, g.nz_global,
                                        g.resolution, g.add_virtual_border);
          out[idx] = val;
          const uint32_t enc = EncodeOrdered(val);
          lo_enc = min(lo_enc, enc);
          hi_enc = max(hi_enc, enc);
        }
        else
        {
          out[idx] = neg ? -best : best;
        }
      }
    }
  }
  if constexpr (kFinal) BlockMinMax(lo_enc, hi_enc, minmax_enc);
}

template <typename InT, typename OutT, bool kFinal>
hipError_t LaunchLine(const InT* in, OutT* out, void* scratch, uint32_t* minmax_enc, LineGeom g,
                      const SdfParams& p, hipStream_t stream)
{
  const int64_t nvox = p.nx * p.ny * p.nz;
  StackEntry* stack = static_cast<StackEntry*>(scratch);
  uint32_t* signw = reinterpret_cast<uint32_t*>(stack + nvox);
  g.resolution = p.resolution;
  g.add_virtual_border = p.add_virtual_border;
  g.nx = static_cast<int>(p.nx);
  g.ny = static_cast<int>(p.ny);
  g.z_offset = static_cast<int>(p.z_offset);
  g.nz_global = static_cast<int>(p.nz_global > 0 ? p.nz_global : p.nz);
  const int64_t blocks = (g.num_lines + kLineBlock - 1) / kLineBlock;
  hipLaunchKernelGGL((LinePassKernel<InT, OutT, kFinal>), dim3(static_cast<unsigned>(blocks)),
                     dim3(kLineBlock), 0, stream, in, out, stack, signw, minmax_enc, g);
  return hipGetLastError();
}
}  // namespace

// Scratch needed by the line-sweep passes: one 8-byte stack entry per voxel plus one sign bit per
// voxel rounded up to whole words per line.
size_t LinePassScratchBytes(int64_t nx, int64_t ny, int64_t nz)
{
  const int64_t nvox = nx * ny * nz;
  const int64_t words_y = ((ny + 31) / 32) * (nx * nz);
  const int64_t words_x = ((nx + 31) / 32) * (ny * nz);
  const int64_t words = words_y > words_x ? words_y : words_x;
  return static_cast<size_t>(nvox) * sizeof(StackEntry) + static_cast<size_t>(words) * 4 + 256;
}

hipError_t LaunchPassYLine(const int16_t* in16, int32_t* out32, void* scratch, const SdfParams& p,
                           hipStream_t stream)
{
  LineGeom g{};
  g.n = static_cast<int>(p.ny);
  g.nz = static_cast<int>(p.nz);
  g.num_lines = p.nx * p.nz;
  g.row_stride = p.nz;
  g.outer_stride = p.ny * p.nz;
  g.pass_axis = 1;
  return LaunchLine<int16_t, int32_t, false>(in16, out32, scratch, nullptr, g, p, stream);
}

hipError_t LaunchPassXLineFinalize(const int32_t* in32, float* sdf, uint32_t* minmax_enc,
                                   void* scratch, const SdfParams& p, hipStream_t stream)
{
  LineGeom g{};
  g.n = static_cast<int>(p.nx);
  g.nz = static_cast<int>(p.nz);
  g.num_lines = p.ny * p.nz;
  g.row_stride = p.ny * p.nz;
  g.outer_stride = p.nz;
  g.pass_axis = 0;
  return LaunchLine<int32_t, float, true>(in32, sdf, scratch, minmax_enc, g, p, stream);
}
}  // namespace vgt
