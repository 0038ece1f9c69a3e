// CPU check of the lane-per-line sweep kernels (csrc/edt_sweep_kernels.hip, the default EDT line passes): the kernel source is compiled by
// g++ against tests/cpp/hip_shim and run one lane at a time, on random lines, against a brute-force evaluation of
//     out(q) = min( min_r (q-r)^2 + |F[r]|,  min over rows r of the other class (q-r)^2 ),
// i.e. the per-line contract of the Y and X passes.  It exercises what the GPU parity tests
// cannot reach cheaply: deep stacks (ring spills / refills in both sweeps), every ring / band size the kernel can be
// built with (-DVGT_SWEEP_BAND / _RING / _CHUNK / _RING_WIDE / _CHUNK_WIDE), packed and 64-bit entries, partial waves, partial bands, the virtual
// border and the final conversion.  The GPU tests (tests/test_gpu_sdf.py) pin the same kernels to the oracle.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <random>
#include <vector>

#define BlockMinMax BlockMinMaxOnDevice
#include "../../voxelized_geometry_tools_amd/csrc/edt_device.hpp"
#undef BlockMinMax
namespace vgt
{
void SetLastError(const std::string&) {}
inline void BlockMinMax(uint32_t lo, uint32_t hi, uint32_t* minmax_enc)
{
  minmax_enc[0] = std::min(minmax_enc[0], lo);
  minmax_enc[1] = std::max(minmax_enc[1], hi);
}
int ShortLineOverride() { return -1; }  // (edt_kernels.hip defines it for the libraries; the launchers here never ask)
}  // namespace vgt
#include "../../voxelized_geometry_tools_amd/csrc/edt_sweep_kernels.hip"

namespace
{
struct Case
{
  int nx, ny, nz;
  int mode;      // value distribution
  int p_site;    // percent of rows that are sites
  int p_flip;    // percent chance of a class change per row
  bool border;
};

int64_t BruteRow(const std::vector<int64_t>& f, const std::vector<uint8_t>& neg, int n, int q)
{
  int64_t best = INT64_MAX;
  for (int r = 0; r < n; r++)
  {
    const int64_t d = static_cast<int64_t>(q - r) * (q - r);
    if (f[r] >= 0) best = std::min(best, d + f[r]);
    if (neg[r] != neg[q]) best = std::min(best, d);
  }
  return best;
}

int failures = 0;

// Class records (csrc/vgt_internal.hpp) of a class volume, by the definition: a plain restatement for the test, independent
// of the device kernel that writes them (csrc/edt_record_kernels.hip, checked on the GPU through the SDF parity tests).
std::vector<vgt::ClassRecord> RecordsOf(const std::vector<uint8_t>& cls, int nx, int ny, int nz, bool mark_no_site)
{
  const int nwords = static_cast<int>(vgt::RecordWords(nz));
  std::vector<vgt::ClassRecord> rec(static_cast<size_t>(nx) * nwords * ny + vgt::kRecordPadding);
  for (auto& r : rec) r = vgt::ClassRecord{0xdeadbeefu, 0xdeadbeefu, 12345u, 54321u};  // (padding: never used)
  for (int x = 0; x < nx; x++)
    for (int y = 0; y < ny; y++)
    {
      const uint8_t* line = &cls[(static_cast<size_t>(x) * ny + y) * nz];
      bool any = false;
      for (int z = 0; z + 1 < nz; z++) any = any || line[z] != line[z + 1];
      for (int w = 0; w < nwords; w++)
      {
        vgt::ClassRecord r{0, 0, vgt::kRecordNoneBelow, vgt::kRecordNoneAbove};
        for (int k = 0; k < 64; k++)
        {
          const int z = std::min(64 * w + k, nz - 1);
          if (line[z]) (k < 32 ? r.mask_lo : r.mask_hi) |= 1u << (k & 31);
        }
        for (int t = 64 * w - 1; t >= 0; t--)
          if (line[t] != line[t + 1])
          {
            r.below2 = static_cast<uint32_t>(2 * (t - 64 * w)) + vgt::kRecordBias;
            break;
          }
        for (int t = 64 * w + 63; t + 1 < nz; t++)
          if (line[t] != line[t + 1])
          {
            r.above2 = static_cast<uint32_t>(2 * (t - 64 * w)) + vgt::kRecordBias;
            break;
          }
        if (mark_no_site && !any) r.above2 = vgt::kRecordNoSite;
        rec[(static_cast<size_t>(x) * nwords + w) * ny + y] = r;
      }
    }
  return rec;
}

// Y pass of the default pipeline: class records in, int32 out, against the brute-force transform of the distances along Z.
// c.mode: 0 = independent voxels (p_flip = percent filled), 1 = runs along Z (p_flip = percent chance of a class change per
// voxel), 2 = runs along Z, and a share of the lines (100 - p_site percent) holds one class only; c.border: records
// without the one-class marks (what a slab sees before its neighbours' carries).
void CheckYRecords(const Case& c, std::mt19937& rng)
{
  const int nx = c.nx, ny = c.ny, nz = c.nz;
  const int64_t total = static_cast<int64_t>(nx) * ny * nz;
  std::vector<uint8_t> cls(total);
  for (int x = 0; x < nx; x++)
    for (int y = 0; y < ny; y++)
    {
      uint8_t* line = &cls[(static_cast<size_t>(x) * ny + y) * nz];
      int cur = rng() & 1;
      const bool flat = c.mode == 2 && static_cast<int>(rng() % 100) >= c.p_site;
      for (int z = 0; z < nz; z++)
      {
        if (c.mode == 0)
          cur = static_cast<int>(rng() % 100) < c.p_flip;
        else if (!flat && static_cast<int>(rng() % 100) < c.p_flip)
          cur ^= 1;
        line[z] = static_cast<uint8_t>(cur);
      }
    }
  const std::vector<vgt::ClassRecord> rec = RecordsOf(cls, nx, ny, nz, !c.border);
  std::vector<int32_t> out(total, 12345);
  vgt::SdfParams p{};
  p.nx = nx; p.ny = ny; p.nz = nz;
  p.resolution = 0.01;
  // (c.border also: a scratch that holds ONE workgroup's slot -- the launch must make do with what it is given)
  std::vector<unsigned char> scratch(c.border ? vgt::SweepPassScratchBytes(1, ny, std::min(nz, 64))
                                              : vgt::SweepPassScratchBytes(nx, ny, nz));
  vgt::LaunchPassYSweepRecords(rec.data(), out.data(), vgt::SweepScratch{scratch.data(), scratch.size()}, p, nullptr);
  std::vector<int64_t> f(ny);
  std::vector<uint8_t> neg(ny);
  for (int x = 0; x < nx; x++)
    for (int z = 0; z < nz; z++)
    {
      for (int y = 0; y < ny; y++)
      {
        const uint8_t* line = &cls[(static_cast<size_t>(x) * ny + y) * nz];
        neg[y] = line[z];
        int64_t d = -1;
        for (int k = 1; k < nz; k++)
          if ((z - k >= 0 && line[z - k] != line[z]) || (z + k < nz && line[z + k] != line[z]))
          {
            d = k;
            break;
          }
        f[y] = d < 0 ? -1 : d * d;
      }
      for (int y = 0; y < ny; y++)
      {
        int64_t want = BruteRow(f, neg, ny, y);
        want = (want == INT64_MAX) ? vgt::kInf32 : want;
        const int64_t signed_want = neg[y] ? -want : want;
        const int32_t raw = out[(static_cast<int64_t>(x) * ny + y) * nz + z];
        const int32_t got = (raw < 0) ? -(raw & 0x7fffffff) : raw;
        if (got != signed_want)
        {
          if (failures++ < 10)
            std::printf("Y (records) MISMATCH shape %dx%dx%d mode %d line (x=%d,z=%d) row %d: got %d want %lld\n", nx, ny, nz,
                        c.mode, x, z, y, got, static_cast<long long>(signed_want));
        }
      }
    }
}

// X-pass shaped run: lines along x, int32 input (signed squared YZ distance), float output + extrema.
void CheckX(const Case& c, std::mt19937& rng)
{
  const int nx = c.nx, ny = c.ny, nz = c.nz;
  const int64_t total = static_cast<int64_t>(nx) * ny * nz;
  const int64_t max_f = static_cast<int64_t>(nz - 1) * (nz - 1) + static_cast<int64_t>(ny - 1) * (ny - 1);
  std::vector<int32_t> in(total);
  for (int y = 0; y < ny; y++)
    for (int z = 0; z < nz; z++)
    {
      int cls = rng() & 1;
      for (int x = 0; x < nx; x++)
      {
        if (static_cast<int>(rng() % 100) < c.p_flip) cls ^= 1;
        int64_t v;
        switch (c.mode)
        {
          case 0: v = 1 + rng() % 3; break;
          case 1: v = 1 + rng() % std::max<int64_t>(1, max_f); break;
          case 2: v = 1 + static_cast<int64_t>(x - nx / 2) * (x - nx / 2) % std::max<int64_t>(1, max_f); break;
          case 3: v = std::max<int64_t>(1, max_f - static_cast<int64_t>(x) * x / 4); break;
          case 4:
          {
            // a distance-like field, as the X pass meets on sparse scenes: the squared distance to the nearest of a few
            // sites of the line's plane, each (x - sx)^2 + offset -- smooth, locally convex, popped by sites far away
            v = INT64_MAX;
            for (int site = 0; site < 5; site++)
            {
              const int64_t sx = (static_cast<int64_t>(y) * 7919 + z * 104729 + site * 31337) % nx;
              const int64_t offset = ((y * 13 + z * 7 + site * 101) % 97) * (max_f / 200 + 1);
              v = std::min(v, (x - sx) * (x - sx) / 3 + offset + 1);
            }
            break;
          }
          default: v = 1; break;
        }
        v = std::min<int64_t>(std::max<int64_t>(v, 1), std::max<int64_t>(max_f, 1));
        if (static_cast<int>(rng() % 100) >= c.p_site) v = vgt::kInf32;
        // the X sweep's input format: sign and magnitude (bit 31 = class)
        in[(static_cast<int64_t>(x) * ny + y) * nz + z] =
            static_cast<int32_t>(static_cast<uint32_t>(v) | (cls ? 0x80000000u : 0u));
      }
    }
  std::vector<float> out(total, 12345.0f);
  vgt::SdfParams p{};
  p.nx = nx; p.ny = ny; p.nz = nz;
  p.resolution = 0.37;
  p.add_virtual_border = c.border ? 1 : 0;
  std::vector<unsigned char> scratch(vgt::SweepPassScratchBytes(nx, ny, nz));
  uint32_t minmax[2] = {0xffffffffu, 0u};
  vgt::LaunchPassXSweepFinalize(in.data(), out.data(), minmax, vgt::SweepScratch{scratch.data(), scratch.size()}, p, nullptr);
  std::vector<int64_t> f(nx);
  std::vector<uint8_t> neg(nx);
  float lo = INFINITY, hi = -INFINITY;
  for (int y = 0; y < ny; y++)
    for (int z = 0; z < nz; z++)
    {
      for (int x = 0; x < nx; x++)
      {
        const int32_t v = in[(static_cast<int64_t>(x) * ny + y) * nz + z];
        neg[x] = v < 0;
        const int64_t a = v & 0x7fffffff;
        f[x] = (a == vgt::kInf32) ? -1 : a;
      }
      for (int x = 0; x < nx; x++)
      {
        int64_t d2 = BruteRow(f, neg, nx, x);
        if (c.border)
        {
          int64_t b = INT64_MAX;
          if (nx > 1) b = std::min<int64_t>(b, std::min(x + 1, nx - x));
          if (ny > 1) b = std::min<int64_t>(b, std::min(y + 1, ny - y));
          if (nz > 1) b = std::min<int64_t>(b, std::min(z + 1, nz - z));
          if (b != INT64_MAX) d2 = std::min(d2, b * b);
        }
        float want = (d2 == INT64_MAX) ? INFINITY : static_cast<float>(std::sqrt(static_cast<double>(d2)) * p.resolution);
        if (neg[x]) want = -want;
        lo = std::min(lo, want);
        hi = std::max(hi, want);
        const float got = out[(static_cast<int64_t>(x) * ny + y) * nz + z];
        if (std::memcmp(&got, &want, 4) != 0)
        {
          if (failures++ < 10)
            std::printf("X MISMATCH shape %dx%dx%d mode %d border %d line (y=%d,z=%d) row %d: got %g want %g\n", nx, ny, nz,
                        c.mode, c.border, y, z, x, got, want);
        }
      }
    }
  const float got_lo = vgt::DecodeOrdered(minmax[0]), got_hi = vgt::DecodeOrdered(minmax[1]);
  if (got_lo != lo || got_hi != hi)
  {
    if (failures++ < 10) std::printf("X EXTREMA shape %dx%dx%d: got (%g, %g) want (%g, %g)\n", nx, ny, nz, got_lo, got_hi, lo, hi);
  }
}
}  // namespace

int main(int argc, char** argv)
{
  const int rounds = argc > 1 ? std::atoi(argv[1]) : 1;
  std::mt19937 rng(4242);
  int cases = 0;
  for (int round = 0; round < rounds; round++)
  {
    // {nx, ny, nz, mode, p_site (mode 2: percent of lines with class changes), p_flip, border (no one-class marks)}
    const Case record_cases[] = {
        {2, 1, 3, 0, 0, 50, false},      {3, 2, 5, 1, 0, 30, false},     {2, 33, 7, 0, 0, 10, true},
        {1, 300, 64, 1, 0, 3, false},    {2, 300, 65, 0, 0, 1, false},   {1, 1024, 130, 2, 30, 2, false},
        {1, 1024, 70, 0, 0, 2, false},   {2, 700, 128, 2, 10, 5, false}, {1, 1500, 3, 1, 0, 20, false},
        {1, 2050, 2, 0, 0, 50, false},   {3, 97, 66, 1, 0, 40, true},    {2, 64, 64, 2, 0, 5, false},
        {1, 513, 200, 2, 50, 1, true},   {1, 999, 2, 0, 0, 99, false},   {1, 1030, 257, 1, 0, 1, false},
        {1, 77, 1, 0, 0, 30, false},     {2, 40, 64, 0, 0, 0, false},    {1, 260, 191, 0, 0, 3, false},
    };
    for (const Case& c : record_cases)
    {
      CheckYRecords(c, rng);
      cases++;
    }
    const Case x_cases[] = {
        {1, 2, 3, 0, 100, 10, false},    {2, 3, 5, 0, 100, 50, true},    {33, 2, 7, 0, 100, 0, false},
        {300, 1, 4, 0, 100, 0, false},   {300, 2, 3, 1, 100, 2, true},   {1024, 1, 2, 0, 100, 0, false},
        {1024, 2, 67, 1, 30, 1, false},  {700, 2, 5, 2, 100, 0, true},   {1500, 1, 3, 0, 100, 1, false},
        {2050, 1, 2, 3, 100, 0, false},  {97, 3, 66, 1, 60, 20, true},   {64, 2, 64, 0, 0, 5, false},
        {513, 1, 3, 3, 100, 0, false},   {999, 1, 2, 0, 100, 100, true}, {1024, 3, 3, 3, 100, 0, false},
        {64, 2, 64, 0, 0, 5, true},       {256, 2, 9, 1, 100, 0, false},  {257, 1, 64, 2, 100, 3, true},
        {600, 2, 3, 1, 20, 1, false},     {1000, 1, 5, 3, 100, 0, true},  {1024, 1, 64, 2, 100, 0, false},
        {1024, 90, 2, 4, 100, 0, false},  {800, 60, 3, 4, 97, 1, true},   {770, 50, 2, 4, 100, 2, false},
        {1000, 2, 64, 1, 100, 1, false},
    };
    for (const Case& c : x_cases)
    {
      CheckX(c, rng);
      cases++;
    }
  }
  std::printf("%d cases, %d mismatches (band %d; 32-bit entries: ring %d, chunk %d; 64-bit entries: ring %d, chunk %d)\n", cases,
              failures, vgt::kBand, vgt::RingShape<true>::kRing, vgt::RingShape<true>::kChunk, vgt::RingShape<false>::kRing,
              vgt::RingShape<false>::kChunk);
  return failures ? 1 : 0;
}
