// Pass 1 of the default EDT pipeline for gfx950: the grid's classes as CLASS RECORDS (vgt_internal.hpp).
//
// The reference's first 1-D transform runs on binary data ({0, +inf} per class,
// signed_distance_field_generation.hpp:57-74 + signed_distance_field_generation.cpp:258-391), where it degenerates
// to "distance to the nearest voxel of the other class along the line".  Here that pass is never materialised per
// voxel: this kernel reads the occupancy once (4 B per voxel, the only per-voxel traffic of the pass), applies the
// is_filled predicate (occupancy_map.hpp:181-205) and writes 16 bytes per 64 voxels -- the class bits of the word
// and the nearest class change below and above it.  The Y pass (edt_sweep_kernels.hip) turns a record into its 64
// lanes' distances along Z with a handful of instructions, so the 2 + 2 B per voxel of a stored distance field and
// its re-read are gone (6 -> 4.25 B per voxel for this pass, 6 -> 4.25 for the Y pass).
//
// Geometry: every load instruction of a wave covers 64 consecutive voxels of one Z line (256 B of float), whatever
// nz is; a wave takes L = 64 / W lines at a time (W = words per line rounded up to a power of two) so that all
// L x W loads are in flight before the first vote, and then lane (j, w) owns word w of line j: the votes
// (v_cmp -> scalar register pair -> v_writelane) transpose "64 voxels in 64 lanes" into "one word per lane", and
// the nearest transitions outside each word are a segmented prefix-max / suffix-min over the W lanes of a line.
// Lines longer than 4096 voxels take a slower one-line-per-wave path with the same result.
#include "edt_device.hpp"

#ifndef VGT_REC_EXP
#define VGT_REC_EXP 0
#endif
namespace vgt
{
namespace
{
constexpr int kRecordBlock = 256;
constexpr int kRecordWaves = kRecordBlock / kWaveSize;
constexpr int kNoTransitionBelow = -(1 << 20);
constexpr int kNoTransitionAbove = 1 << 20;

// is_filled (occupancy_map.hpp:181-205): occupancy > 0.5, or == 0.5 when unknown cells count as filled.
template <bool kUnknownFilled>
__device__ __forceinline__ bool IsFilledCell(float occupancy)
{
  return kUnknownFilled ? (occupancy >= 0.5f) : (occupancy > 0.5f);
}
template <bool kUnknownFilled>
__device__ __forceinline__ bool IsFilledCell(uint8_t mask)
{
  return mask != 0;
}

// value[lane `target`] = a wave-uniform word (v_writelane_b32 with a constant lane; this compiler has no builtin for it)
__device__ __forceinline__ void WriteLane(uint32_t& value, uint32_t word_uniform, int target)
{
  // (the word comes straight from a vector compare: the wait states between a vector instruction's scalar result and
  // its use by another vector instruction are the compiler's job everywhere else, inside an asm statement they are ours)
  asm("s_nop 3\n\tv_writelane_b32 %0, %1, %2" : "+v"(value) : "s"(word_uniform), "n"(target));
}

__device__ __forceinline__ uint32_t EncodeBelow(int t, int word_begin)
{
  return (t == kNoTransitionBelow) ? kRecordNoneBelow : static_cast<uint32_t>(2 * (t - word_begin)) + kRecordBias;
}
__device__ __forceinline__ uint32_t EncodeAbove(int t, int word_begin)
{
  return (t == kNoTransitionAbove) ? kRecordNoneAbove : static_cast<uint32_t>(2 * (t - word_begin)) + kRecordBias;
}

__device__ __forceinline__ uint16_t SummaryField(bool filled, int other_local, int z_offset)
{
  return static_cast<uint16_t>((filled ? kSlabFilledBit : 0u) |
                               (other_local < 0 ? kSlabNone : static_cast<uint16_t>(other_local + z_offset)));
}

// W words per line (a power of two, 1..64), L = 64 / W lines per wave and step; lane = j * W + w.
template <typename InT, int W, bool kUnknownFilled>
__global__ __launch_bounds__(kRecordBlock) void ClassRecordKernel(const InT* __restrict__ in,
                                                                 ClassRecord* __restrict__ records,
                                                                 int64_t num_lines, int ny, int nz, int nwords,
                                                                 SlabLineSummary* __restrict__ summary, int z_offset,
                                                                 int mark_no_site)
{
  constexpr int L = kWaveSize / W;
  const int lane = threadIdx.x & (kWaveSize - 1);
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) / kWaveSize);  // (told to be uniform)
  const int j = lane / W, w = lane % W;
  const int64_t num_steps = (num_lines + L - 1) / L;
  for (int64_t step = static_cast<int64_t>(blockIdx.x) * kRecordWaves + wave; step < num_steps;
       step += static_cast<int64_t>(gridDim.x) * kRecordWaves)
  {
    const int64_t line0 = step * L;
    const int lines_here = static_cast<int>(min(static_cast<int64_t>(L), num_lines - line0));
    // ---- every load of the step before the first vote: L x W wave loads of 64 consecutive voxels.  All of them
    // unconditional (straight code, one wait): lines past the last one and words past the line's last repeat the last
    // one, lanes past the end of the line repeat its last voxel (no transition there) ----
    InT v[L][W];
#pragma unroll
    for (int jj = 0; jj < L; jj++)
    {
      const InT* line_in = in + (line0 + min(jj, lines_here - 1)) * nz;
#pragma unroll
      for (int c = 0; c < W; c++)
      {
        const int z = min(min(c, nwords - 1) * kWaveSize + lane, nz - 1);
#if VGT_REC_EXP & 2
        v[jj][c] = line_in[z];
#else
        v[jj][c] = __builtin_nontemporal_load(line_in + z);
#endif
      }
    }
    // ---- votes: word (jj, c) goes to lane jj * W + c ----
    uint32_t mlo = 0, mhi = 0;
#pragma unroll
    for (int jj = 0; jj < L; jj++)
    {
#pragma unroll
      for (int c = 0; c < W; c++)
      {
        const uint64_t m = __builtin_amdgcn_ballot_w64(IsFilledCell<kUnknownFilled>(v[jj][c]));
        WriteLane(mlo, static_cast<uint32_t>(m), jj * W + c);
        WriteLane(mhi, static_cast<uint32_t>(m >> 32), jj * W + c);
      }
    }
    const bool valid = j < lines_here && w < nwords;
    const uint64_t m = (static_cast<uint64_t>(mhi) << 32) | mlo;
    // ---- transitions of my word: inside it, and between its last voxel and the next word's first ----
    const uint32_t next_bit0 = static_cast<uint32_t>(__shfl_down(static_cast<int>(mlo & 1u), 1, W));
    const bool boundary = valid && (w + 1 < nwords) && (((mhi >> 31) & 1u) != next_bit0);
    const uint64_t inner = (m ^ (m >> 1)) & 0x7fffffffffffffffull;
    const uint64_t transitions = valid ? (inner | (boundary ? 0x8000000000000000ull : 0ull)) : 0ull;
    const int word_begin = w * kWaveSize;
    int last_t = kNoTransitionBelow, first_t = kNoTransitionAbove;
    if (transitions)
    {
      last_t = word_begin + 63 - __clzll(static_cast<long long>(transitions));
      first_t = word_begin + __ffsll(static_cast<long long>(transitions)) - 1;
    }
    // ---- last transition in the words below me / first transition in the words above me: segmented scans ----
    int upto = last_t;   // inclusive prefix maximum over the words of my line
    int from = first_t;  // inclusive suffix minimum
#pragma unroll
    for (int d = 1; d < W; d *= 2)
    {
      const int a = __shfl_up(upto, d, W);
      const int b = __shfl_down(from, d, W);
      if (w >= d) upto = max(upto, a);
      if (w + d < W) from = min(from, b);
    }
    int below = __shfl_up(upto, 1, W);
    if (w == 0) below = kNoTransitionBelow;
    int above = __shfl_down(from, 1, W);
    if (w == W - 1) above = kNoTransitionAbove;
    if (boundary) above = word_begin + 63;
    // (the whole line: every lane of the line's group reads the totals)
    const int line_last = __shfl(upto, W - 1, W);
    const int line_first = __shfl(from, 0, W);
    if (valid)
    {
      const uint32_t line = static_cast<uint32_t>(line0) + static_cast<uint32_t>(j);  // (lines < 2^28)
      const int64_t x = line / static_cast<uint32_t>(ny);
      const int y = static_cast<int>(line - static_cast<uint32_t>(x) * static_cast<uint32_t>(ny));
      ClassRecord r;
      r.mask_lo = mlo;
      r.mask_hi = mhi;
      r.below2 = EncodeBelow(below, word_begin);
      r.above2 = EncodeAbove(above, word_begin);
      if (mark_no_site && line_last == kNoTransitionBelow) r.above2 = kRecordNoSite;
      using Raw = uint32_t __attribute__((ext_vector_type(4)));
      Raw raw;
      raw.x = r.mask_lo;
      raw.y = r.mask_hi;
      raw.z = r.below2;
      raw.w = r.above2;
#if VGT_REC_EXP & 1
      if (raw.x == 0x12345u)
#endif
      *reinterpret_cast<Raw*>(records + ((x * nwords + w) * ny + y)) = raw;
    }
    if (summary)
    {
      // class of the line's first and last voxel, first / last voxel of the other class (slab-local; -1: none)
      const uint32_t first_class = static_cast<uint32_t>(__shfl(static_cast<int>(mlo & 1u), 0, W));
      const int last_word = nwords - 1;
      const int last_bit = (nz - 1) & 63;
      const uint32_t my_last = static_cast<uint32_t>((m >> last_bit) & 1ull);
      const uint32_t last_class = static_cast<uint32_t>(__shfl(static_cast<int>(my_last), last_word, W));
      if (j < lines_here && w == 0)
      {
        SlabLineSummary s;
        s.first = SummaryField(first_class != 0u, line_first == kNoTransitionAbove ? -1 : line_first + 1, z_offset);
        s.last = SummaryField(last_class != 0u, line_last == kNoTransitionBelow ? -1 : line_last, z_offset);
        summary[line0 + j] = s;
      }
    }
  }
}

// Lines of more than 4096 voxels (up to kMaxExtent = 256 words): one line per wave, the words' masks through LDS, the
// carries by one lane.  Same records; not a fast path.
template <typename InT, bool kUnknownFilled>
__global__ __launch_bounds__(kRecordBlock) void ClassRecordLongLineKernel(const InT* __restrict__ in,
                                                                         ClassRecord* __restrict__ records,
                                                                         int64_t num_lines, int ny, int nz, int nwords,
                                                                         SlabLineSummary* __restrict__ summary,
                                                                         int z_offset, int mark_no_site)
{
  constexpr int kMaxWords = static_cast<int>(kMaxExtent / kWaveSize);
  __shared__ uint64_t s_mask[kRecordWaves][kMaxWords];
  __shared__ int32_t s_below[kRecordWaves][kMaxWords];
  __shared__ int32_t s_above[kRecordWaves][kMaxWords];
  __shared__ int32_t s_line[kRecordWaves][2];
  const int lane = threadIdx.x & (kWaveSize - 1);
  const int wave = threadIdx.x / kWaveSize;
  volatile uint64_t* mask = s_mask[wave];
  volatile int32_t* below = s_below[wave];
  volatile int32_t* above = s_above[wave];
  volatile int32_t* ends = s_line[wave];
  for (int64_t line = static_cast<int64_t>(blockIdx.x) * kRecordWaves + wave; line < num_lines;
       line += static_cast<int64_t>(gridDim.x) * kRecordWaves)
  {
    for (int c = 0; c < nwords; c++)
    {
      const int z = min(c * kWaveSize + lane, nz - 1);
      const uint64_t m = __builtin_amdgcn_ballot_w64(IsFilledCell<kUnknownFilled>(in[line * nz + z]));
      if (lane == 0) mask[c] = m;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0)
    {
      int last = kNoTransitionBelow;
      for (int c = 0; c < nwords; c++)
      {
        below[c] = last;
        const uint64_t m = mask[c];
        uint64_t t = (m ^ (m >> 1)) & 0x7fffffffffffffffull;
        if (c + 1 < nwords && ((m >> 63) & 1ull) != (mask[c + 1] & 1ull)) t |= 0x8000000000000000ull;
        if (t) last = c * kWaveSize + 63 - __clzll(static_cast<long long>(t));
      }
      ends[1] = last;
      int first = kNoTransitionAbove;
      for (int c = nwords - 1; c >= 0; c--)
      {
        const uint64_t m = mask[c];
        const bool boundary = c + 1 < nwords && ((m >> 63) & 1ull) != (mask[c + 1] & 1ull);
        above[c] = boundary ? c * kWaveSize + 63 : first;
        uint64_t t = (m ^ (m >> 1)) & 0x7fffffffffffffffull;
        if (boundary) t |= 0x8000000000000000ull;
        if (t) first = c * kWaveSize + __ffsll(static_cast<long long>(t)) - 1;
      }
      ends[0] = first;
    }
    __builtin_amdgcn_wave_barrier();
    const int64_t x = line / ny;
    const int y = static_cast<int>(line - x * ny);
    const int line_first = ends[0], line_last = ends[1];
    for (int c = lane; c < nwords; c += kWaveSize)
    {
      const uint64_t m = mask[c];
      ClassRecord r;
      r.mask_lo = static_cast<uint32_t>(m);
      r.mask_hi = static_cast<uint32_t>(m >> 32);
      r.below2 = EncodeBelow(below[c], c * kWaveSize);
      r.above2 = EncodeAbove(above[c], c * kWaveSize);
      if (mark_no_site && line_last == kNoTransitionBelow) r.above2 = kRecordNoSite;
      records[(x * nwords + c) * ny + y] = r;
    }
    if (summary && lane == 0)
    {
      const bool first_class = (mask[0] & 1ull) != 0ull;
      const bool last_class = ((mask[nwords - 1] >> ((nz - 1) & 63)) & 1ull) != 0ull;
      SlabLineSummary s;
      s.first = SummaryField(first_class, line_first == kNoTransitionAbove ? -1 : line_first + 1, z_offset);
      s.last = SummaryField(last_class, line_last == kNoTransitionBelow ? -1 : line_last, z_offset);
      summary[line] = s;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// Multi-GPU: the nearest transitions outside the slab, from the other slabs' carries, go into the records of the words
// that have none inside the slab (0.25 B per voxel read, less written; the Y pass never knows about slabs).
__global__ __launch_bounds__(256) void SlabRecordFixupKernel(ClassRecord* __restrict__ records,
                                                            const SlabLineCarry* __restrict__ carries, int64_t nx,
                                                            int ny, int nwords, int z_offset)
{
  const int64_t total = nx * nwords * static_cast<int64_t>(ny);
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
  {
    ClassRecord r = records[i];
    const bool none_below = r.below2 == kRecordNoneBelow, none_above = r.above2 == kRecordNoneAbove;
    if (!none_below && !none_above) continue;
    const int y = static_cast<int>(i % ny);
    const int64_t xw = i / ny;
    const int w = static_cast<int>(xw % nwords);
    const int64_t x = xw / nwords;
    const SlabLineCarry c = carries[x * ny + y];
    const int word_begin = w * kWaveSize;
    bool changed = false;
    if (none_below)
    {
      // no transition between the slab's first voxel and this word: the word's first voxel has the class of the slab's
      // first voxel, and the nearest voxel of the other class below is the other slabs' (global z; -1: none)
      const bool filled = (r.mask_lo & 1u) != 0u;
      const int other = filled ? c.prev_free : c.prev_filled;
      if (other >= 0)
      {
        r.below2 = EncodeBelow(other - z_offset, word_begin);
        changed = true;
      }
    }
    if (none_above)
    {
      // the word's last voxel has the class of the slab's last voxel (bits past the end of the line repeat it)
      const bool filled = (r.mask_hi >> 31) != 0u;
      const int other = filled ? c.next_free : c.next_filled;
      if (other >= 0)
      {
        // the voxel before `other` differs from it: a transition at other - 1
        r.above2 = EncodeAbove(other - 1 - z_offset, word_begin);
        changed = true;
      }
    }
    // a line that holds one class in every slab: no voxel of the word has a site
    const uint64_t m = (static_cast<uint64_t>(r.mask_hi) << 32) | r.mask_lo;
    const bool uniform = (m ^ (m >> 1)) << 1 == 0ull;
    if (r.below2 == kRecordNoneBelow && r.above2 == kRecordNoneAbove && uniform)
    {
      r.above2 = kRecordNoSite;
      changed = true;
    }
    if (changed) records[i] = r;
  }
}

int RecordGridFor(int64_t steps)
{
  const int64_t blocks = (steps + kRecordWaves - 1) / kRecordWaves;
  const int64_t cap = 256 * 32;
  return static_cast<int>(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
}

template <typename InT, bool kUnknownFilled>
hipError_t LaunchRecords(const InT* in, ClassRecord* records, const SdfParams& p, SlabLineSummary* summary,
                         hipStream_t stream)
{
  const int64_t lines = p.nx * p.ny;
  const int nz = static_cast<int>(p.nz), ny = static_cast<int>(p.ny);
  const int nwords = static_cast<int>(RecordWords(p.nz));
  const int z_offset = static_cast<int>(p.z_offset);
  // a line without a transition holds one class: without other slabs that is the whole story
  const int mark_no_site = summary == nullptr ? 1 : 0;
#define VGT_RECORD_CASE(W)                                                                                          \
  hipLaunchKernelGGL((ClassRecordKernel<InT, W, kUnknownFilled>), dim3(RecordGridFor((lines + 64 / W - 1) / (64 / W))), \
                     dim3(kRecordBlock), 0, stream, in, records, lines, ny, nz, nwords, summary, z_offset,          \
                     mark_no_site)
  if (nwords <= 1)
    VGT_RECORD_CASE(1);
  else if (nwords <= 2)
    VGT_RECORD_CASE(2);
  else if (nwords <= 4)
    VGT_RECORD_CASE(4);
  else if (nwords <= 8)
    VGT_RECORD_CASE(8);
  else if (nwords <= 16)
    VGT_RECORD_CASE(16);
  else if (nwords <= 32)
    VGT_RECORD_CASE(32);
  else if (nwords <= 64)
    VGT_RECORD_CASE(64);
  else
    hipLaunchKernelGGL((ClassRecordLongLineKernel<InT, kUnknownFilled>), dim3(RecordGridFor(lines)), dim3(kRecordBlock),
                       0, stream, in, records, lines, ny, nz, nwords, summary, z_offset, mark_no_site);
#undef VGT_RECORD_CASE
  return hipGetLastError();
}
}  // namespace

hipError_t LaunchClassRecordsFromOccupancy(const float* occupancy, ClassRecord* records, const SdfParams& p,
                                           SlabLineSummary* summary, hipStream_t stream)
{
  if (p.unknown_is_filled) return LaunchRecords<float, true>(occupancy, records, p, summary, stream);
  return LaunchRecords<float, false>(occupancy, records, p, summary, stream);
}

hipError_t LaunchClassRecordsFromMask(const uint8_t* mask, ClassRecord* records, const SdfParams& p,
                                      SlabLineSummary* summary, hipStream_t stream)
{
  return LaunchRecords<uint8_t, false>(mask, records, p, summary, stream);
}

hipError_t LaunchSlabRecordFixup(ClassRecord* records, const SlabLineCarry* carries, const SdfParams& p,
                                 hipStream_t stream)
{
  const int nwords = static_cast<int>(RecordWords(p.nz));
  const int64_t total = p.nx * nwords * p.ny;
  const int64_t blocks = (total + 255) / 256;
  const int grid = static_cast<int>(blocks < 1 ? 1 : (blocks > 256 * 32 ? 256 * 32 : blocks));
  hipLaunchKernelGGL(SlabRecordFixupKernel, dim3(grid), dim3(256), 0, stream, records, carries, p.nx,
                     static_cast<int>(p.ny), nwords, static_cast<int>(p.z_offset));
  return hipGetLastError();
}
}  // namespace vgt
