// Lane-per-line sweep passes (Y and X) of the exact signed EDT for gfx950.
//
// One wave owns 64 neighbouring Z positions of one outer index; a lane owns one line of the pass axis and walks
// it row by row, all lanes in lockstep, so every row access of the wave is one contiguous segment (256 B of
// int32 / float) for both axes: no tile, no transpose, no partial cache lines on the store side.
// Each lane runs the Felzenszwalb-Huttenlocher stack algorithm (signed_distance_field_generation.cpp:124-226) on
// its own line, in exact integer arithmetic:
//
//   * Classes are ignored while the envelope is built: out(q) = min(envelope over ALL rows with cost |F[r]| at q, squared
//     distance to the nearest row of the other class below / above q) -- a row of the other class is a zero-cost site, and
//     a row of the own class can only offer its cost, which is a distance to the other class too.
//   * A stack entry is (G = F + row^2, row) -- a point of the lower convex hull.  "The top never owns a point once
//     the new site is there" is the cross-multiplied test of the three points (the pop test `s <= z[k]` of
//     :193-197 without the division): (Gq - Gt)(rt - rs) - (Gt - Gs)(q - rt) < 0, two 64-bit multiply-adds.
//     A new site that cannot beat the top before the last row (Gq - Gt >= 2 (n-1) (q - rt)) is never pushed.
//   * Two sentinel entries at row 0 with values above every real one sit at the bottom of every stack (a site
//     T0 and, below it, V0 = (G(T0) + 1, row 0), which makes T0's own test come out false): the stack is never
//     empty, no depth checks in the loops, and a line without any site evaluates to "no site" by itself.
//   * Registers hold the top, the second entry as differences to the top and the third entry as it came from the
//     ring; the stack itself is a ring of kRing entries per lane in LDS ([slot][lane]: conflict-free whatever the
//     lanes' depths), spilled to / refilled from a scratch buffer in chunks of kChunk entries.  The ring is checked
//     once per kChunk rows for the whole wave (a lane-by-lane check would run on every row because SOME lane
//     always needs it); chunks that will be needed soon are requested one check ahead.
//   * Entries are 32-bit words (22 bits of G, 10 bits of row) when the host can bound G below 2^22 - 2 on lines of
//     at most 1024 rows (both passes of a 1024^3 grid), otherwise 64-bit.
//   * Sweep 2 walks the line backwards: the owner of a row is found by comparing the two topmost members at that
//     row (values along the envelope are unimodal), (Gs - Gt) + 2 q (rt - rs) <= 0 pops.  The distances to the
//     bounding rows of the other class are running counters fed by the line's sign bits (one word per 32 rows, kept
//     in the scratch buffer between the sweeps); waves whose 64 lines hold one class only skip that part.
//   * X pass: fused sqrt / resolution / sign / virtual border / min-max; on bands whose inputs are all below 512 the
//     conversion comes from an exact table in LDS (dense scenes).
//   * The Y pass of the default pipeline does not read a distance field: its rows come as CLASS RECORDS (pass 1,
//     edt_record_kernels.hip; vgt_internal.hpp) -- 64 rows per vector load, two blocks ahead -- and a lane's distance
//     along Z is the minimum over the transitions around it, one v_sad_u32 each.  Rows whose lines hold one class only
//     are marked by pass 1 and cost a few scalar instructions per band of 16.
//   * A site on or above the segment between its two neighbour rows' hull points is not on the lower hull and never
//     touches the stack (X pass).  A row that is no site is a hull
//     point far above every real one there (kNoSiteG): the same two comparisons handle it, no validity flags.
//   * What the listing taught (profiles/r4/experiments.md): a row pays for exec-mask regions and scalar flag logic more
//     than for vector instructions; a chunk of the ring is addressed ONCE (it never wraps), which turns its eight LDS
//     accesses into four paired instructions; conditions of rare paths are combined bitwise, not with || and &&.
//
//   * What rounds 5 and 6 measured and did not keep (profiles/r5, r6/experiments.md): a coarse hull in front of the X sweep,
//     handing lower halves of second sweeps to idle workgroups, a packed-word form of sweep 1 (fewer instructions per row,
//     hardly less time: a row costs its dependent chain, not its instruction count).
//
// The kernels are bound by instruction issue and by the latency of their dependent chains, not by HBM: rows are
// processed kBand at a time (registers), the code below keeps rare paths (refills, exact final conversion) out of
// the straight-line code, and register and LDS use are sized for kWaves waves per SIMD.
#include "edt_device.hpp"
#include "edt_line_geom.hpp"

#include <cstdlib>
#include <type_traits>

namespace vgt
{
namespace
{
#ifndef VGT_SWEEP_BAND
#define VGT_SWEEP_BAND 16
#endif
#ifndef VGT_SWEEP_RING
#define VGT_SWEEP_RING 32
#endif
#ifndef VGT_SWEEP_CHUNK
#define VGT_SWEEP_CHUNK 8
#endif
#ifndef VGT_SWEEP_WAVES
#define VGT_SWEEP_WAVES 4
#endif
#ifndef VGT_SWEEP_GROUP
#define VGT_SWEEP_GROUP 2  // rows of the X pass's sweep 2 whose final conversions form one block of code (1, 2 or 4)
#endif
constexpr int kBand = VGT_SWEEP_BAND;    // rows held in registers at a time: 8, 16 or 32
constexpr int kWord = 32;                // rows per sign word
#ifndef VGT_SWEEP_RING_WIDE
#define VGT_SWEEP_RING_WIDE 16
#endif
#ifndef VGT_SWEEP_CHUNK_WIDE
#define VGT_SWEEP_CHUNK_WIDE 4
#endif
// Stack entries per lane resident in LDS (a power of two) and entries per spill / refill (= rows between two checks
// of the ring), for 32-bit and for 64-bit entries: the ring of either takes 8 KiB of LDS per wave.
#ifndef VGT_SWEEP_GROUPS
#define VGT_SWEEP_GROUPS 8
#endif
constexpr int kSweepGroups = VGT_SWEEP_GROUPS;  // work counters (= XCDs of an MI355X)
constexpr int kCounterStride = 32;              // ints between two counters (128 bytes)
constexpr int kMaxSweepSlots = 4096;
template <bool kPacked>
struct RingShape
{
  static constexpr int kRing = kPacked ? VGT_SWEEP_RING : VGT_SWEEP_RING_WIDE;
  static constexpr int kChunk = kPacked ? VGT_SWEEP_CHUNK : VGT_SWEEP_CHUNK_WIDE;
  static_assert(kBand % kChunk == 0 && kRing >= 4 * kChunk && (kRing & (kRing - 1)) == 0 && kChunk % 4 == 0, "sizes");
};
constexpr int kFar = 32768;              // "no row of the other class": kFar^2 is above every real squared distance
static_assert(kWord % kBand == 0 && kBand % 2 == 0, "sizes");


template <bool kPacked>
struct Codec;
template <>
struct Codec<true>
{
  using Entry = uint32_t;
  static constexpr int32_t kSentinelG = (1 << 22) - 3;  // T0; V0, W0 = kSentinelG + 1, + 2; real G and real results below it
  static __device__ __forceinline__ Entry Pack(int32_t G, int row)
  {
    return (static_cast<uint32_t>(G) << 10) | static_cast<uint32_t>(row);
  }
  static __device__ __forceinline__ int32_t G(Entry e) { return static_cast<int32_t>(e >> 10); }
  static __device__ __forceinline__ int Row(Entry e) { return static_cast<int>(e & 1023u); }
  // "not loaded": no entry looks like it (the largest G belongs to row 0)
  static __device__ __forceinline__ Entry Unknown() { return ~0u; }
  static __device__ __forceinline__ bool IsUnknown(Entry e) { return e == ~0u; }
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
  static __device__ __forceinline__ Entry Unknown() { return make_uint2(0u, ~0u); }
  static __device__ __forceinline__ bool IsUnknown(Entry e) { return e.y == ~0u; }
};

#ifndef VGT_SWEEP_NT
#define VGT_SWEEP_NT 3  // streaming accesses: 1 = row loads, 2 = row stores, 4 = sign words, 8 = spill chunks
#endif
// 16-byte and 8-byte scratch accesses, streaming or not
__device__ __forceinline__ void StoreQuad(uint4* p, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
#if defined(VGT_HOST_EMULATION) || !(VGT_SWEEP_NT & 8)
  *p = make_uint4(a, b, c, d);
#else
  using Raw = uint32_t __attribute__((ext_vector_type(4)));
  Raw r;
  r.x = a; r.y = b; r.z = c; r.w = d;
  __builtin_nontemporal_store(r, reinterpret_cast<Raw*>(p));
#endif
}
__device__ __forceinline__ uint4 LoadQuad(const uint4* p)
{
#if defined(VGT_HOST_EMULATION) || !(VGT_SWEEP_NT & 8)
  return *p;
#else
  using Raw = uint32_t __attribute__((ext_vector_type(4)));
  const Raw r = __builtin_nontemporal_load(reinterpret_cast<const Raw*>(p));
  return make_uint4(r.x, r.y, r.z, r.w);
#endif
}
__device__ __forceinline__ void StorePair(uint2* p, uint32_t a, uint32_t b)
{
#if defined(VGT_HOST_EMULATION) || !(VGT_SWEEP_NT & 4)
  *p = make_uint2(a, b);
#else
  using Raw = uint32_t __attribute__((ext_vector_type(2)));
  Raw r;
  r.x = a; r.y = b;
  __builtin_nontemporal_store(r, reinterpret_cast<Raw*>(p));
#endif
}
__device__ __forceinline__ uint2 LoadPair(const uint2* p)
{
#if defined(VGT_HOST_EMULATION) || !(VGT_SWEEP_NT & 4)
  return *p;
#else
  using Raw = uint32_t __attribute__((ext_vector_type(2)));
  const Raw r = __builtin_nontemporal_load(reinterpret_cast<const Raw*>(p));
  return make_uint2(r.x, r.y);
#endif
}

// A chunk of N entries <-> contiguous bytes of the spill buffer, as 16-byte accesses.
template <int N>
__device__ __forceinline__ void StoreChunk(uint32_t* dst, const uint32_t (&e)[N])
{
#pragma unroll
  for (int j = 0; j < N / 4; j++)
    StoreQuad(reinterpret_cast<uint4*>(dst) + j, e[4 * j], e[4 * j + 1], e[4 * j + 2], e[4 * j + 3]);
}
template <int N>
__device__ __forceinline__ void LoadChunk(const uint32_t* src, uint32_t (&e)[N])
{
#pragma unroll
  for (int j = 0; j < N / 4; j++)
  {
    const uint4 a = LoadQuad(reinterpret_cast<const uint4*>(src) + j);
    e[4 * j] = a.x;
    e[4 * j + 1] = a.y;
    e[4 * j + 2] = a.z;
    e[4 * j + 3] = a.w;
  }
}
template <int N>
__device__ __forceinline__ void StoreChunk(uint2* dst, const uint2 (&e)[N])
{
#pragma unroll
  for (int j = 0; j < N / 2; j++)
    StoreQuad(reinterpret_cast<uint4*>(dst) + j, e[2 * j].x, e[2 * j].y, e[2 * j + 1].x, e[2 * j + 1].y);
}
template <int N>
__device__ __forceinline__ void LoadChunk(const uint2* src, uint2 (&e)[N])
{
#pragma unroll
  for (int j = 0; j < N / 2; j++)
  {
    const uint4 a = LoadQuad(reinterpret_cast<const uint4*>(src) + j);
    e[2 * j] = make_uint2(a.x, a.y);
    e[2 * j + 1] = make_uint2(a.z, a.w);
  }
}

// A pointer into global memory that every lane holds the same value of, told to the compiler (scalar registers, scalar
// address arithmetic; the address space is spelled out because a pointer rebuilt from integers would be a flat one).
#ifdef VGT_HOST_EMULATION
#define VGT_GLOBAL
#else
#define VGT_GLOBAL __attribute__((address_space(1)))
#endif
template <typename T>
__device__ __forceinline__ VGT_GLOBAL T* UniformPointer(const VGT_GLOBAL T* p)
{
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v));
  const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32));
  return reinterpret_cast<VGT_GLOBAL T*>((static_cast<uint64_t>(hi) << 32) | lo);
}
template <typename T>
__device__ __forceinline__ VGT_GLOBAL T* GlobalPointer(T* p)
{
#ifdef VGT_HOST_EMULATION
  return p;
#else
  return reinterpret_cast<VGT_GLOBAL T*>(reinterpret_cast<uint64_t>(p));
#endif
}

// a * b + c with 24-bit signed a and b, b the same in every lane (a scalar register): one v_mad_i32_i24 (the compiler
// prefers a 32-bit multiply and an add, or a 64-bit multiply-add that ties up a register pair).
__device__ __forceinline__ int32_t Mad24Uniform(int32_t a, int32_t b_uniform, int32_t c)
{
#ifdef VGT_HOST_EMULATION
  return a * b_uniform + c;
#else
  int32_t d;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b_uniform), "v"(c));
  return d;
#endif
}

// A copy by an instruction of its own.  Sweep 2's rare pop moves its register pairs one place up (second <- third); written
// as plain assignments the compiler renames the registers instead and pays for it with two copies per ROW on the path
// that does not pop.
__device__ __forceinline__ int32_t FreshCopy(int32_t v)
{
#ifdef VGT_HOST_EMULATION
  return v;
#else
  int32_t d;
  asm("v_mov_b32 %0, %1" : "=v"(d) : "v"(v));
  return d;
#endif
}

// The lane's element of a row of 4-byte values: wave-uniform base + 32-bit byte offset, the scalar-base addressing mode.
// The empty asm keeps the zero-extension of the offset next to the access: hoisted out of the loop it reaches instruction
// selection as a 64-bit register, and every access takes a 64-bit vector add and a vector address pair.  (Not for the 16-bit
// loads of the cross-check Y pass: they lose their sign extension to a separate instruction this way.)
// Stores pass the offset through the asm IN PLACE (one variable for all of them: the old value is dead, so no copy); loads
// pin a copy, which costs a v_mov per load -- and measures 3.5 % FASTER on the X pass than the sixteen loads of a band
// back to back (profiles/r4/experiments.md).
template <bool kInPlace>
__device__ __forceinline__ uint32_t PinnedOffset(uint32_t& byte_offset)
{
#ifndef VGT_HOST_EMULATION
  if constexpr (kInPlace)
    asm volatile("" : "+v"(byte_offset));
  else
  {
    uint32_t copy = byte_offset;
    asm volatile("" : "+v"(copy));
    return copy;
  }
#endif
  return byte_offset;
}
template <typename T>
__device__ __forceinline__ const VGT_GLOBAL T* LaneAddress(const VGT_GLOBAL T* row, uint32_t& byte_offset)
{
  return reinterpret_cast<const VGT_GLOBAL T*>(reinterpret_cast<const VGT_GLOBAL unsigned char*>(row) +
                                                PinnedOffset<false>(byte_offset));
}
template <typename T>
__device__ __forceinline__ VGT_GLOBAL T* LaneAddress(VGT_GLOBAL T* row, uint32_t& byte_offset)
{
  return reinterpret_cast<VGT_GLOBAL T*>(reinterpret_cast<VGT_GLOBAL unsigned char*>(row) +
                                         PinnedOffset<true>(byte_offset));
}

// |a - b| + 1 with b the same in every lane.  Written so that the compiler selects ONE v_sad_u32 with a scalar operand
// (max - min + c is its pattern for that instruction) and, unlike with an asm statement, keeps the wait states between a
// vector instruction that writes a scalar register (v_readlane) and a vector instruction that reads it.
__device__ __forceinline__ uint32_t AbsDiffPlusOne(uint32_t a, uint32_t b_uniform)
{
  return (max(a, b_uniform) - min(a, b_uniform)) + 1u;
}
// all ones in the lanes whose bit of a wave-uniform 64-bit mask is set, else zero: one v_cndmask with the mask as the
// condition (a scalar register pair)
__device__ __forceinline__ uint32_t SpreadLaneMask(uint64_t mask_uniform, [[maybe_unused]] int lane)
{
#ifdef VGT_HOST_EMULATION
  return ((mask_uniform >> lane) & 1ull) ? ~0u : 0u;
#else
  return __builtin_amdgcn_inverse_ballot_w64(mask_uniform) ? ~0u : 0u;
#endif
}

__device__ __forceinline__ uint32_t LowBits(int bits)  // bits in [0, 32]
{
  return (bits >= 32) ? ~0u : ((1u << bits) - 1u);
}

// Rows are read once and written once per pass: streaming accesses (no reuse to protect in the caches).
#if defined(VGT_HOST_EMULATION) || !(VGT_SWEEP_NT & 1)
#define VGT_STREAM_LOAD(p) (*(p))
#else
#define VGT_STREAM_LOAD(p) __builtin_nontemporal_load(p)
#endif
#if defined(VGT_HOST_EMULATION) || !(VGT_SWEEP_NT & 2)
#define VGT_STREAM_STORE(v, p) (*(p) = (v))
#else
#define VGT_STREAM_STORE(v, p) __builtin_nontemporal_store(v, p)
#endif

// Host emulation (tests/cpp/sweep_emulation.cc compiles this file with g++ and runs the lanes one by one): no GPU asm.
#ifdef VGT_HOST_EMULATION
#define VGT_COLD_PATH()
#define VGT_MIN_F32(acc, v) acc = fminf(acc, v)
#define VGT_MAX_F32(acc, v) acc = fmaxf(acc, v)
#define VGT_MIN3_F32(acc, a, b) acc = fminf(acc, fminf(a, b))
#define VGT_MAX3_F32(acc, a, b) acc = fmaxf(acc, fmaxf(a, b))
#else
#define VGT_COLD_PATH() asm volatile("; rare path")
// plain instructions: no NaN can occur, so no canonicalisation is needed
#define VGT_MIN_F32(acc, v) asm("v_min_f32 %0, %0, %1" : "+v"(acc) : "v"(v))
#define VGT_MAX_F32(acc, v) asm("v_max_f32 %0, %0, %1" : "+v"(acc) : "v"(v))
#define VGT_MIN3_F32(acc, a, b) asm("v_min3_f32 %0, %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b))
#define VGT_MAX3_F32(acc, a, b) asm("v_max3_f32 %0, %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b))
#endif


// The wave's extrema (none when lo > hi) into minmax_enc[0 / 1], ordered encodings: one atomic pair per call.
__device__ __forceinline__ void WaveMinMax(float lo_value, float hi_value, uint32_t* minmax_enc)
{
  uint32_t lo = 0xffffffffu, hi = 0u;
  if (lo_value <= hi_value)
  {
    lo = EncodeOrdered(lo_value);
    hi = EncodeOrdered(hi_value);
  }
#ifdef VGT_HOST_EMULATION
  minmax_enc[0] = min(minmax_enc[0], lo);
  minmax_enc[1] = max(minmax_enc[1], hi);
#else
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
#endif
}

// kPlain (X pass only): no virtual border and a resolution inside the fast conversion's range.
template <typename InT, typename OutT, bool kFinal, bool kPacked, bool kPlain>
__global__ __launch_bounds__(kWaveSize, VGT_SWEEP_WAVES) void SweepPassKernel(const InT* __restrict__ in,
                                                                            OutT* __restrict__ out,
                                                                            unsigned char* __restrict__ spill,
                                                                            uint2* __restrict__ word_info,
                                                                            uint32_t* __restrict__ minmax_enc,
                                                                            int* __restrict__ work_counter,
                                                                            const SweepGeom g)
{
  using C = Codec<kPacked>;
  using Entry = typename C::Entry;
  // the Y pass reads class records, the X pass the Y pass's int32 field
  constexpr bool kRecords = std::is_same<InT, ClassRecord>::value;
  static_assert(!(kRecords && kFinal), "records feed the Y pass");
  constexpr int kRing = RingShape<kPacked>::kRing;
  constexpr int kChunk = RingShape<kPacked>::kChunk;
  constexpr int32_t kLimit = C::kSentinelG;  // values at or above: no site
  constexpr int kEntryBytes = static_cast<int>(sizeof(Entry));
  constexpr int kShift = kPacked ? 8 : 9;                   // log2(bytes of one ring slot = 64 lanes x entry)
  constexpr uint32_t kSlot = 1u << kShift;
  constexpr uint32_t kRingMask = (kRing - 1) << kShift;     // slot bits of a ring byte address
  constexpr int kChunkBytes = kChunk * kEntryBytes;
  constexpr uint32_t kChunkSlots = static_cast<uint32_t>(kChunk) << kShift;
  __shared__ Entry ring[kRing * kWaveSize];
  unsigned char* const ring_bytes = reinterpret_cast<unsigned char*>(ring);

  const int lane = threadIdx.x;
  const int n = g.n;
  const int64_t rstride = g.row_stride;
  // The workgroup (one wave) owns one slot of the scratch buffer -- spilled stack entries and sign words -- for all
  // the lines it works on, so the scratch is sized by the number of workgroups in flight, not by the grid.
  unsigned char* const own_spill = spill + static_cast<int64_t>(blockIdx.x) * g.chunks * (kWaveSize * kChunkBytes);
  uint2* const own_info = word_info + static_cast<int64_t>(blockIdx.x) * (g.nwords + 1) * kWaveSize;
  const uint32_t lane_entry = static_cast<uint32_t>(lane) * kEntryBytes;  // byte offset of this lane inside a ring slot
  const uint32_t lane_chunk = static_cast<uint32_t>(lane) * kChunkBytes;  // ... inside a row of spill chunks
  float lo_value = INFINITY, hi_value = -INFINITY;
  // The plain X pass with 32-bit entries: the final conversion float(sqrt(double(d2)) * resolution) of SMALL squared
  // distances comes from a table of kSmallD2 exact values in LDS, filled once per workgroup.  A dozen dependent
  // instructions per row (reciprocal square root, fp64 steps) become one LDS read -- on the bands where sweep 1 has seen
  // that every lane's inputs, hence every result, are below kSmallD2: the rows of a dense scene (every voxel a few cells
  // from the other class), the neighbourhood of obstacles on a sparse one.  (`small_bands`: one bit per band of the line.)
  constexpr bool kUseTable = kFinal && kPlain && kPacked && (1024 / kBand <= 64);
  constexpr uint32_t kSmallD2 = 512;
  [[maybe_unused]] float* small_table = nullptr;
  if constexpr (kUseTable)
  {
    __shared__ float small_table_storage[kSmallD2];
    small_table = small_table_storage;
#ifdef VGT_HOST_EMULATION
    for (uint32_t d2 = 0; d2 < kSmallD2; d2++)  // (lanes run one after the other here: each fills all of it)
#else
    for (uint32_t d2 = static_cast<uint32_t>(lane); d2 < kSmallD2; d2 += kWaveSize)
#endif
      small_table[d2] = SqrtTimesResolutionExact(static_cast<int32_t>(d2), g.resolution);
    __syncthreads();
  }
#ifdef VGT_HOST_EMULATION
  int emulated_round = 0;
#endif
  [[maybe_unused]] auto fetch_item = [&]() -> int {
    int fetched = 0;
#ifndef VGT_HOST_EMULATION
    // Workgroups with the same blockIdx modulo 8 share an XCD and its L2 (the dispatcher deals them round robin): with
    // g.groups = 8 there is one counter per such group, dealing whole rows of the grid -- the segments of one contiguous
    // row are then written through one L2 at about the same time instead of through eight.  A speed choice only.
    const int group = static_cast<int>(blockIdx.x) % g.groups;
    if (lane == 0) fetched = atomicAdd(work_counter + group * kCounterStride, 1);
    fetched = __builtin_amdgcn_readfirstlane(fetched);
    const int dealt = fetched / g.zsegs;
    fetched = (dealt * g.groups + group) * g.zsegs + (fetched - dealt * g.zsegs);
#endif
    return fetched;
  };
  for (;;)
  {
  // ---- next unit of work: 64 neighbouring lines of one outer index ----
  int item = 0;
  unsigned char* wave_spill = own_spill;
  uint2* wave_info = own_info;
#ifdef VGT_HOST_EMULATION
  item = static_cast<int>(blockIdx.x + gridDim.x * emulated_round++);  // (lanes run one after the other: fixed deal)
#else
  item = fetch_item();
#endif
  if (item >= g.items) break;
  const int outer = item / g.zsegs;
  const int z0 = (item - outer * g.zsegs) * kWaveSize;
  // (a batch of grids: which grid, and the outer index inside it -- one grid: 0 and `outer`)
  const bool batched = g.batch_outers != g.outers;
  const int batch_index = batched ? outer / g.batch_outers : 0;
  const int outer_in_grid = outer - batch_index * g.batch_outers;
  const int64_t outer_offset = static_cast<int64_t>(outer) * g.outer_stride + static_cast<int64_t>(batch_index) * g.batch_skip;
  // Lanes beyond the grid repeat the last line of the grid: same input, same result, stored to the same address.
  const uint32_t zl = static_cast<uint32_t>(min(lane, g.nz - 1 - z0));
  [[maybe_unused]] uint32_t lane_bytes = zl * 4u;  // the lane's byte offset inside a row of 4-byte values (see LaneAddress)
  // (records: [outer][64-voxel segment][row], one per row of this item)
  const InT* const wave_in = kRecords ? in + static_cast<int64_t>(item) * n : in + (outer_offset + z0);
  OutT* const wave_out = out + (outer_offset + z0);

  // ---- stack state.  Entries [0, depth): [0, lo) live in the spill buffer, [lo, depth) in the ring (slot = index
  // mod kRing).  D and L are depth and lo times the slot size, so that (D & kRingMask) | lane_entry is the ring
  // address of entry `depth` and L + lane_chunk the spill address of the chunk that starts at entry `lo`.  Registers
  // hold the top (Gt, rt), the second entry as differences to the top (A = rt - rs >= 0, nB = Gs - Gt) and the third
  // as it came from the ring (e3, decoded when it moves up): a pop is a few additions and only waits for the ring
  // when it is the second pop in a row.  Three row-0 sentinels T0 > V0 > W0 (in stack order; G = S, S + 1, S + 2) and
  // a never-decoded entry 0 keep a fourth entry under every top: sweep 2 looks two entries down. ----
  uint32_t D = 4u << kShift, L = 0;
  int32_t Gt = C::kSentinelG, nB = 1;
  int rt = 0, A = 0;
  Entry e3 = C::Pack(C::kSentinelG + 2, 0);
  auto ring_ref = [&](uint32_t scaled_index) -> Entry& {
    return *reinterpret_cast<Entry*>(ring_bytes + ((scaled_index & kRingMask) | lane_entry));
  };
  // The ring slots of the chunk that starts at entry `first` (scaled like D and L; a multiple of kChunk, as every lo is):
  // contiguous in the ring -- its size is a multiple of the chunk's -- so ONE address serves the chunk's entries, as
  // base[j * kWaveSize] (an immediate offset of the LDS instruction; addressed entry by entry each costs two instructions).
  // (32-bit entries only: the paired 64-bit LDS accesses the compiler makes of it measure 1 % slower on config 5 than
  // entry-by-entry addresses, which the 64-bit kind keeps.)
  auto chunk_in_ring = [&](uint32_t scaled_first) -> Entry* {
    return reinterpret_cast<Entry*>(ring_bytes + ((scaled_first & kRingMask) | lane_entry));
  };
  auto chunk_slot = [&](Entry* slots, uint32_t scaled_first, int j) -> Entry& {
    if constexpr (kPacked)
      return slots[j * kWaveSize];
    else
      return ring_ref(scaled_first + (static_cast<uint32_t>(j) << kShift));
  };
  static_assert(kRing % kChunk == 0, "a chunk must not wrap inside the ring");
  auto spill_ptr = [&](uint32_t scaled_first) -> Entry* {
    return reinterpret_cast<Entry*>(wave_spill + (scaled_first + lane_chunk));
  };
  ring_ref(0u << kShift) = C::Pack(0, 0);  // never decoded: keeps "fourth" inside the stack
  ring_ref(1u << kShift) = C::Pack(C::kSentinelG + 2, 0);
  ring_ref(2u << kShift) = C::Pack(C::kSentinelG + 1, 0);
  ring_ref(3u << kShift) = C::Pack(C::kSentinelG, 0);

  // chunks requested from the spill buffer at the last check (a refill_now of the same chunk drops them)
  Entry pf0[kChunk], pf1[kChunk];
  int pf_count = 0;
  // the chunk that ends below entry lo comes back from the spill buffer (slow path: a run of pops reached it)
  auto refill_now = [&]() {
    L -= kChunkSlots;
    const Entry* src = spill_ptr(L);
    Entry* const slots = chunk_in_ring(L);
#pragma unroll 1
    for (int j = 0; j < kChunk; j += 2)
    {
      const Entry a = src[j], b = src[j + 1];
      chunk_slot(slots, L, j) = a;
      chunk_slot(slots, L, j + 1) = b;
    }
    pf_count = 0;
  };
  // top <- second <- third <- ring (the sentinels are never popped, so the third always exists).  Sweep 1 pops only on behalf
  // of a site that is then pushed, and a push leaves the third entry in its register: a row's FIRST pop finds it there
  // (`third_in_register`), every further pop of the row reads the ring -- neither needs a test.
  auto pop = [&](auto third_in_register) {
    Gt += nB;
    rt -= A;
    if constexpr (decltype(third_in_register)::value)
    {
#ifdef VGT_HOST_EMULATION
      assert(!C::IsUnknown(e3));
#endif
    }
    else
    {
#ifdef VGT_HOST_EMULATION
      assert(C::IsUnknown(e3));
#endif
      if (__builtin_expect(D - 3 * kSlot < L, 0)) refill_now();
      e3 = ring_ref(D - 3 * kSlot);
    }
    A = rt - C::Row(e3);
    nB = C::G(e3) - Gt;
    D -= kSlot;
    e3 = C::Unknown();
  };
  auto commit = [&](const Entry (&buf)[kChunk]) {
    L -= kChunkSlots;
    Entry* const slots = chunk_in_ring(L);
#pragma unroll
    for (int j = 0; j < kChunk; j++) chunk_slot(slots, L, j) = buf[j];
  };

  // =====================================================================================================
  // Sweep 1: build the envelope.
  // =====================================================================================================
  uint32_t any_transition = 0;
  [[maybe_unused]] uint64_t small_bands = 0;  // (kUseTable) bit b: every input of band b, in every lane, is below kSmallD2
  {
    // every kChunk rows: the ring must have room for kChunk pushes
    auto check_ring = [&]() {
      // Two separate things.  Rare: a ring that pops have nearly emptied, and the chunk requested for it at the last check
      // (kept apart and hinted cold, so that what it does to pf0's registers stays off the common path).
      const uint32_t resident = D - L;
      // (bitwise, not short-circuit: the compiler turns `||` / `&&` of these comparisons into nested exec-mask regions)
      const bool refilling = static_cast<int>(pf_count != 0) |
                             (static_cast<int>(L != 0) & static_cast<int>(resident <= kChunkSlots));
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(refilling) != 0ull, 0))
      {
        VGT_COLD_PATH();
        // a chunk requested at the last check goes into the ring (the ring had at most kChunk entries then and
        // has taken at most kChunk more)
        if (pf_count != 0)
        {
          commit(pf0);
          pf_count = 0;
        }
        // a ring that pops have nearly emptied asks for the chunk below it before a pop has to wait for it
        if (static_cast<int>(L != 0) & static_cast<int>(D - L <= kChunkSlots))
        {
          LoadChunk(spill_ptr(L - kChunkSlots), pf0);
          pf_count = 1;
        }
      }
      // Common (in the X pass at nearly every check, for some lanes): a ring without room for kChunk more pushes spills
      // its oldest chunk.
      while (__builtin_amdgcn_ballot_w64(D - L > (static_cast<uint32_t>(kRing - kChunk) << kShift)) != 0ull)
      {
        if (D - L > (static_cast<uint32_t>(kRing - kChunk) << kShift))
        {
          Entry buf[kChunk];
          Entry* const slots = chunk_in_ring(L);
#pragma unroll
          for (int j = 0; j < kChunk; j++) buf[j] = chunk_slot(slots, L, j);
          StoreChunk(spill_ptr(L), buf);
          L += kChunkSlots;
        }
      }
    };

    int din = kFar;          // distance from the row below this word to the nearest row of the other class below it
    uint32_t prev_bit = 0;   // class of the row below this word
    uint32_t sw = 0;         // sign bits of the word being swept
    const int32_t n2m = 2 * (n - 1);
    // A site at row q, as the hull point G = cost + q^2.  It is pushed when it beats the top before the last row
    // (G - Gt < 2 (n - 1) (q - rt)); a site that does not cannot pop the top either (the top beat ITS predecessor before
    // the last row when it was pushed, so its interval ends before the new site's would begin), and a site that pops
    // the top beats the entry below it even earlier: the one comparison against the top decides the push, and the pop
    // tests -- two 64-bit multiply-adds and a 64-bit compare each -- are only paid by sites that will be pushed.
    // (`wanted`: what the caller knows against the site, folded into the same test -- one exec-mask region per row, not two)
    auto site = [&](int q, int32_t G, bool wanted) {
      int32_t dG = G - Gt;
      int dr = q - rt;
      // (no __builtin_assume(dr >= 0) here: knowing the sign, the compiler multiplies unsigned and corrects -- three
      // multiply-adds and three moves for the first hull test of a row instead of two multiply-adds)
      if (static_cast<int>(wanted) & static_cast<int>(dG < static_cast<int32_t>(__umul24(n2m, dr))))
      {
        if (static_cast<int64_t>(dG) * A + static_cast<int64_t>(nB) * dr < 0)
        {
          // (one pop is the common case: it is laid out as straight code, further pops out of line)
          pop(std::true_type{});
          dG = G - Gt;
          dr = q - rt;
          if (__builtin_expect(static_cast<int64_t>(dG) * A + static_cast<int64_t>(nB) * dr < 0, 0))
          {
            do
            {
              pop(std::false_type{});
              dG = G - Gt;
              dr = q - rt;
            } while (static_cast<int64_t>(dG) * A + static_cast<int64_t>(nB) * dr < 0);
          }
        }
        ring_ref(D) = C::Pack(G, q);
        e3 = C::Pack(Gt + nB, rt - A);
        A = dr;
        nB = -dG;
        Gt = G;
        rt = q;
        D += kSlot;
      }
#ifndef VGT_HOST_EMULATION
      // (Opaque: otherwise the compiler forwards "the next row's G - Gt is its rise over this row where this row was pushed"
      // through the merge and pays for the one saved subtraction with an else-region -- three scalar instructions -- per row.)
      asm volatile("" : "+v"(Gt));
#endif
    };
    // the band [r0, r0 + kBand) has been swept: its sign bits join the word, a complete word goes to the scratch
    auto band_done = [&](int r0, uint32_t bits) {
      const int sub = r0 & (kWord - 1);
      sw |= (bits >> (32 - kBand)) << sub;
      if (sub + kBand == kWord || r0 + kBand >= n)
      {
        // the word is complete: sign word and distance carry for the evaluation's downward counters
        const int word_first = r0 - sub;
        const int valid = min(kWord, n - word_first);
        uint32_t xdn = sw ^ ((sw << 1) | prev_bit);  // bit k: row k differs from the row below it (rows past the end: 0)
        xdn &= LowBits(valid);
        if (word_first == 0) xdn &= ~1u;
        StorePair(wave_info + static_cast<int64_t>(word_first / kWord) * kWaveSize + lane, sw, static_cast<uint32_t>(din));
        any_transition |= xdn;
        din = xdn ? (valid - (31 - __clz(static_cast<int>(xdn)))) : min(din + valid, kFar);
        prev_bit = (sw >> (valid - 1)) & 1u;
        sw = 0;
      }
    };
    if constexpr (kRecords)
    {
      // ---- Y pass of the default pipeline: the rows come as class records, one scalar load per row.  A lane's distance
      // along Z to the other class is the smallest distance across the transitions around it: the two the record names
      // (nearest below / above the word) and the word's own.  |xq - t2| + 1 is twice that distance for the transition
      // encoded as t2 on either side (vgt_internal.hpp): one v_sad_u32 per transition, a min3, a shift. ----
      // The records of 64 rows are ONE vector load (16 bytes per lane: lane l holds the record of row b + l), two blocks
      // ahead of the block at work, so nothing of the record stream is ever waited for and the scalar unit's loads --
      // which can only be waited for all at once, together with the LDS -- stay out of the loop.  A block's "no voxel
      // of these 64 has a site" marks and, for those rows, classes are two votes; a row that does have sites gets its
      // record into scalar registers with four v_readlane.
      constexpr int kRows = 4;     // rows per group: the unit of the "no site" short cut inside a band
      constexpr int kBlock = 64;   // rows per record block
      static_assert(kBand % kRows == 0 && kChunk % kRows == 0 && kBlock % kBand == 0, "sizes");
      static_assert(3 * kBlock <= kRecordPadding, "reads past the end stay inside the padding");
      const uint32_t xq = 2u * zl + (kRecordBias - 1u);
      [[maybe_unused]] auto load_block = [&](int first_row) -> uint4 {
#ifdef VGT_HOST_EMULATION
        return make_uint4(0u, 0u, 0u, 0u);
#else
        using Raw = uint32_t __attribute__((ext_vector_type(4)));
        const Raw r = *(reinterpret_cast<const Raw*>(wave_in + first_row) + lane);
        return make_uint4(r.x, r.y, r.z, r.w);
#endif
      };
      // bit l: row first_row + l of the block carries the one-class mark / is of the filled class (meaningful with the mark)
      auto block_marks = [&]([[maybe_unused]] const uint4& blk, [[maybe_unused]] int first_row, uint64_t& marks,
                             uint64_t& filled) {
#ifdef VGT_HOST_EMULATION
        marks = filled = 0;
        for (int l = 0; l < kBlock; l++)
        {
          // (rows past the line's end: whatever the padding holds, never used)
          const ClassRecord r = wave_in[first_row + l];
          if (r.above2 == kRecordNoSite) marks |= 1ull << l;
          if (r.mask_lo & 1u) filled |= 1ull << l;
        }
#else
        marks = __builtin_amdgcn_ballot_w64(blk.w == kRecordNoSite);
        filled = __builtin_amdgcn_ballot_w64((blk.x & 1u) != 0u);
#endif
      };
      auto record_of = [&]([[maybe_unused]] const uint4& blk, [[maybe_unused]] int first_row, int index) -> uint4 {
#ifdef VGT_HOST_EMULATION
        const ClassRecord r = wave_in[first_row + index];
        return make_uint4(r.mask_lo, r.mask_hi, r.below2, r.above2);
#else
        return make_uint4(__builtin_amdgcn_readlane(blk.x, index), __builtin_amdgcn_readlane(blk.y, index),
                          __builtin_amdgcn_readlane(blk.z, index), __builtin_amdgcn_readlane(blk.w, index));
#endif
      };
      auto row = [&](int q, const uint4& rec, uint32_t& bits) {
        const uint64_t m = (static_cast<uint64_t>(rec.y) << 32) | rec.x;
        bits = __builtin_amdgcn_alignbit(SpreadLaneMask(m, lane), bits, 1);  // shifted in from the top
        // The two transitions the record names, then the word's own (most words have none: one scalar test; two are the common
        // case of a word that has any -- in and out of an obstacle -- and sit inline, more in a loop out of line).
        uint32_t f2 = min(AbsDiffPlusOne(xq, rec.z), AbsDiffPlusOne(xq, rec.w));
        // (a word of one class is all zeros or all ones: its own sign bit, spread)
        if (__builtin_expect(static_cast<int64_t>(m) != (static_cast<int64_t>(m) >> 63), 0))
        {
          uint64_t own = (m ^ (m >> 1)) & 0x7fffffffffffffffull;
          const uint32_t t1 = 2u * static_cast<uint32_t>(__ffsll(static_cast<long long>(own)) - 1) + kRecordBias;
          own &= own - 1ull;
          uint32_t t2 = t1;
          if (own)
          {
            t2 = 2u * static_cast<uint32_t>(__ffsll(static_cast<long long>(own)) - 1) + kRecordBias;
            own &= own - 1ull;
          }
          f2 = min(f2, min(AbsDiffPlusOne(xq, t1), AbsDiffPlusOne(xq, t2)));
          if (__builtin_expect(own != 0ull, 0))
          {
            VGT_COLD_PATH();
            do
            {
              const uint32_t t = 2u * static_cast<uint32_t>(__ffsll(static_cast<long long>(own)) - 1) + kRecordBias;
              f2 = min(f2, AbsDiffPlusOne(xq, t));
              own &= own - 1ull;
            } while (own != 0ull);
          }
        }
        // (a lane without any transition around it: f = kInf16 - 1 or more, its square is never pushed -- `wanted` says so)
        const int32_t f = static_cast<int32_t>(f2 >> 1);
        site(q, __mul24(f, f) + q * q, f2 < 2u * static_cast<uint32_t>(kInf16));
      };
      uint4 blk0 = load_block(0), blk1 = load_block(kBlock), blk2 = load_block(2 * kBlock);
      uint64_t marks = 0, filled = 0;
      constexpr uint32_t kBandMask = kBand >= 32 ? ~0u : ((1u << (kBand % 32)) - 1u);
      for (int r0 = 0; r0 < n; r0 += kBand)
      {
        __builtin_assume(r0 >= 0 && r0 < 16384);
        const int in_block = r0 & (kBlock - 1);
        const int block_first = r0 - in_block;
        if (in_block == 0)
        {
          if (r0 != 0)
          {
            blk0 = blk1;
            blk1 = blk2;
            blk2 = load_block(r0 + 2 * kBlock);
          }
          block_marks(blk0, r0, marks, filled);
        }
        const uint32_t band_marks = static_cast<uint32_t>(marks >> in_block) & kBandMask;
        const uint32_t band_filled = static_cast<uint32_t>(filled >> in_block) & kBandMask;
        uint32_t bits;  // sign bits of this band, in the top kBand bits
        if (band_marks == kBandMask && r0 + kBand <= n)
        {
          // A band whose lines hold one class only (marked by pass 1): no lane has a site and a row's lanes share their
          // class -- most bands of a sparse scene -- costs a few scalar instructions.
          bits = band_filled << (32 - kBand);
        }
        else
        {
          bits = 0;
#pragma unroll
          for (int k0 = 0; k0 < kBand; k0 += kRows)
          {
            const int q0 = r0 + k0;
            if (k0 % kChunk == 0) check_ring();
            if (((band_marks >> k0) & 0xfu) == 0xfu && q0 + kRows <= n)
              bits = __builtin_amdgcn_alignbit((band_filled >> k0) & 0xfu, bits, kRows);
            else
            {
#pragma unroll
              for (int k = 0; k < kRows; k++)
              {
                if (q0 + k < n)
                  row(q0 + k, record_of(blk0, block_first, in_block + k0 + k), bits);
                else
                  bits >>= 1;  // (keeps the rows of a partial band at their bit positions)
              }
            }
          }
        }
        band_done(r0, bits);
      }
    }
    else
    {
      static_assert(kRecords || kFinal, "the X pass reads the Y pass's field");
      // Rows are loaded through a wave-uniform pointer that steps by the row stride, plus the lane's 32-bit offset: the
      // address of a row costs two scalar adds and no vector register (the readfirstlane keeps the compiler from folding
      // the lane offset into a per-lane 64-bit base, which costs a register pair and a 64-bit vector add per row).
      auto load_band = [&](int32_t (&dst)[kBand], int first_row) {
        const VGT_GLOBAL InT* row_in = UniformPointer(GlobalPointer(wave_in + static_cast<int64_t>(first_row) * rstride));
        if (first_row + kBand <= n)
        {
  #pragma unroll
          for (int k = 0; k < kBand; k++)
          {
            dst[k] = static_cast<int32_t>(VGT_STREAM_LOAD(LaneAddress(row_in, lane_bytes)));
            row_in = UniformPointer(row_in + rstride);
          }
        }
        else
        {
          // the last, partial band: rows past the end repeat the last row (not used)
  #pragma unroll
          for (int k = 0; k < kBand; k++)
          {
            dst[k] = static_cast<int32_t>(VGT_STREAM_LOAD(LaneAddress(row_in, lane_bytes)));
            if (first_row + k + 1 < n) row_in = UniformPointer(row_in + rstride);
          }
        }
      };
      int32_t nxt[kBand];
      load_band(nxt, 0);
      // Hull point of a row.  A row that is no site gets G = kNoSiteG + q^2: above every real hull point by more than any
      // site can make up before the last row (site()'s first test rejects it: kNoSiteG - G > 2 (n - 1)^2 for every real G of
      // either entry kind) and small enough that differences of two hull points never overflow.  So no row needs a "valid"
      // flag: the neighbour test below and site() see an ordinary, hopeless point.
      constexpr int32_t kNoSiteG = 0x60000000;
      auto decode = [&](int32_t v, int q, int32_t& G) {
        // the Y sweep hands its squared distances to the X sweep as sign and magnitude (bit 31 = class), which one `and`
        // takes apart; they are below kLimit, or kInf32
        const int32_t f = static_cast<int32_t>(min(static_cast<uint32_t>(v) & 0x7fffffffu, static_cast<uint32_t>(kNoSiteG)));
        G = f + q * q;
      };
      // A site whose hull point lies on or above the segment between its two NEIGHBOUR rows' points (2 G(q) >= G(q - 1) +
      // G(q + 1), both sites) is not on the lower hull whatever else the line holds: it never touches the stack.  Along a
      // stretch where G is concave -- the common case: every site would pop its predecessor and be popped by its
      // successor -- that is every row but the ends.
      // (with "no site" as a huge hull point the test needs no flags: next to such a neighbour the site is kept, and a row
      // that is no site itself is dropped here or rejected by site())
      int32_t G_prev = kNoSiteG, G_cur = 0;
      decode(nxt[0], 0, G_cur);
      for (int r0 = 0; r0 < n; r0 += kBand)
      {
        __builtin_assume(r0 >= 0 && r0 < 16384);
        int32_t cur[kBand];
#pragma unroll
        for (int k = 0; k < kBand; k++) cur[k] = nxt[k];
        if (r0 + kBand < n) load_band(nxt, r0 + kBand);
        if constexpr (kUseTable)
        {
          // (a row's result is at most its own input: the row itself is a candidate)
          uint32_t any = 0u;
#pragma unroll
          for (int k = 0; k < kBand; k++) any |= static_cast<uint32_t>(cur[k]);
          if (r0 + kBand <= n && __builtin_amdgcn_ballot_w64((any & 0x7fffffffu) >= kSmallD2) == 0ull)
            small_bands |= 1ull << (r0 / kBand);
        }
        uint32_t bits = 0;  // sign bits of this band
        auto rows = [&](auto guarded) {
          constexpr bool kGuard = decltype(guarded)::value;
#pragma unroll
          for (int k = 0; k < kBand; k++)
          {
            if (k % kChunk == 0) check_ring();
            if (kGuard && r0 + k >= n) bits <<= 1;  // (keeps the rows of a partial band at their bit positions)
            if (!kGuard || r0 + k < n)
            {
              const int q = r0 + k;
              // the class (bit 31 of either input format) is shifted in from the bottom by ONE instruction, (bits : v) >> 31;
              // the band's bits come out in reverse order and are turned round once per band below
              bits = __builtin_amdgcn_alignbit(bits, static_cast<uint32_t>(cur[k]), 31);
              int32_t G_next = kNoSiteG + (q + 1) * (q + 1);  // (past the last row: no site)
              // (every row of a full band but its last has a successor: no test)
              if ((!kGuard && k + 1 < kBand) || q + 1 < n) decode(k + 1 < kBand ? cur[k + 1] : nxt[0], q + 1, G_next);
              site(q, G_cur, G_cur - G_prev < G_next - G_cur);
              G_prev = G_cur;
              G_cur = G_next;
            }
          }
        };
        if (r0 + kBand <= n)
          rows(std::false_type{});
        else
          rows(std::true_type{});
        band_done(r0, __builtin_bitreverse32(bits));  // (row k of the band: bit kBand - 1 - k -> bit 32 - kBand + k)
      }
    }
  }

  // =====================================================================================================
  // Sweep 2: evaluate, last row first.
  // =====================================================================================================
  const bool classes = __builtin_amdgcn_ballot_w64(any_transition != 0u) != 0ull;
  const bool any_empty = __builtin_amdgcn_ballot_w64(D == (4u << kShift)) != 0ull;  // a line without any site
  // No site on any of the wave's 64 lines and no class change along them (more than half of the Y pass's items on a sparse
  // scene: the slice holds no voxel whose Z line changes class): every row's result is "no voxel of the other class",
  // +-infinity by the line's class, and the evaluation is a fill from the sign words.  (The virtual border turns "none" into a finite distance: then the
  // general evaluation runs.)
  bool all_empty = !classes && __builtin_amdgcn_ballot_w64(D != (4u << kShift)) == 0ull;
  if constexpr (kFinal && !kPlain) all_empty = all_empty && !g.add_virtual_border;
  if (all_empty)
  {
    const int nwords = g.nwords;
    VGT_GLOBAL OutT* row_out = UniformPointer(GlobalPointer(wave_out + static_cast<int64_t>(n - 1) * rstride));
    uint2 info = LoadPair(wave_info + static_cast<int64_t>(nwords - 1) * kWaveSize + lane);
    for (int w = nwords - 1; w >= 0; w--)
    {
      const uint32_t sw = info.x;
      if (w > 0) info = LoadPair(wave_info + static_cast<int64_t>(w - 1) * kWaveSize + lane);
      const int valid = min(kWord, n - w * kWord);
      if constexpr (kFinal)
      {
        const uint32_t rows_mask = LowBits(valid);
        if (sw & rows_mask) lo_value = -INFINITY;
        if (~sw & rows_mask) hi_value = INFINITY;
      }
      for (int k = valid - 1; k >= 0; k--)
      {
        const uint32_t sign = (sw >> k) << 31;
        if constexpr (kFinal)
        {
          VGT_STREAM_STORE(__uint_as_float(0x7f800000u | sign), LaneAddress(row_out, lane_bytes));
        }
        else
        {
          // (sign and magnitude, like the general evaluation)
          VGT_STREAM_STORE(static_cast<OutT>(static_cast<uint32_t>(kInf32) | sign), LaneAddress(row_out, lane_bytes));
        }
        row_out = UniformPointer(row_out - rstride);
      }
    }
  }
  else
  {
    // Every kStep rows the chunks requested at the last step go into the ring and up to two more are requested
    // when the ring has room (between two steps the ring only shrinks).
    pf_count = 0;  // (sweep 1 may leave a request behind: dropped)
    auto refill_step = [&]() {
      // (bitwise, not short-circuit: see check_ring)
      const bool refilling =
          static_cast<int>(pf_count != 0) |
          (static_cast<int>(L != 0) & static_cast<int>(D - L <= (static_cast<uint32_t>(kRing - kChunk) << kShift)));
      if (__builtin_amdgcn_ballot_w64(refilling) != 0ull)
      {
        if (pf_count > 0) commit(pf0);
        if (pf_count > 1) commit(pf1);
        pf_count = 0;
        const uint32_t resident = D - L;  // (after the commits)
        if (static_cast<int>(L != 0) & static_cast<int>(resident <= (static_cast<uint32_t>(kRing - kChunk) << kShift)))
        {
          LoadChunk(spill_ptr(L - kChunkSlots), pf0);
          pf_count = 1;
          if (static_cast<int>(L != kChunkSlots) &
              static_cast<int>(resident <= (static_cast<uint32_t>(kRing - 2 * kChunk) << kShift)))
          {
            LoadChunk(spill_ptr(L - 2 * kChunkSlots), pf1);
            pf_count = 2;
          }
        }
      }
    };
    constexpr int kStep = (kChunk < 8) ? 8 : kChunk;  // rows between two refill steps

    if (C::IsUnknown(e3))  // (sweep 1 loads the third entry lazily)
    {
      if (D - 3 * kSlot < L) refill_now();
      e3 = ring_ref(D - 3 * kSlot);
    }
    // Sweep 2 looks TWO entries down: a row's value is the better of the top and the second entry, so a lane HAS to
    // pop only when the third entry has caught up with the second -- and when some lane of the wave has to, every
    // lane whose second entry is already the better one pops along.  A wave goes through the pop code a quarter as
    // often as when every lane pops as soon as it can (tools/sim/lane_sweep_sim.c).  Registers: second -> third as
    // differences (A2 = rs - r3 >= 0, nB2 = G3 - Gs) and the fourth entry as it came from the ring (e4).
    int A2 = (rt - A) - C::Row(e3);
    int32_t nB2 = C::G(e3) - (Gt + nB);
    if (__builtin_expect(D - 4 * kSlot < L, 0)) refill_now();
    Entry e4 = ring_ref(D - 4 * kSlot);
    auto pop_down = [&]() {
      Gt += nB;
      rt -= A;
      A = FreshCopy(A2);
      nB = FreshCopy(nB2);
      A2 = (rt - A) - C::Row(e4);
      nB2 = C::G(e4) - (Gt + nB);
      D -= kSlot;
      if (__builtin_expect(D - 4 * kSlot < L, 0)) refill_now();
      e4 = ring_ref(D - 4 * kSlot);
    };

    int dn = kFar;             // distance from the row above the current one to the nearest row of the other class above
    uint32_t above_bit0 = 0;   // class of the first row of the word above
    // sign words: this word's and the next lower word's are in registers, the one below that is on its way
    const int nwords = g.nwords;
    uint2 info = LoadPair(wave_info + static_cast<int64_t>(nwords - 1) * kWaveSize + lane);
    uint2 info_below = make_uint2(0u, 0u);
    if (nwords > 1) info_below = LoadPair(wave_info + static_cast<int64_t>(nwords - 2) * kWaveSize + lane);
    uint2 info_next = make_uint2(0u, 0u);
    uint32_t xdn_word = 0, xup_word = 0;
    VGT_GLOBAL OutT* row_out = UniformPointer(GlobalPointer(wave_out + static_cast<int64_t>(n - 1) * rstride));  // row being evaluated
    const int last_band = (n - 1) / kBand * kBand;
    for (int r0 = last_band; r0 >= 0; r0 -= kBand)
    {
      __builtin_assume(r0 >= 0 && r0 < 16384);
      const int sub = r0 & (kWord - 1);
      if (sub + kBand == kWord || r0 == last_band)
      {
        // a new word starts (from its top)
        const int w = r0 / kWord;
        if (r0 != last_band)
        {
          above_bit0 = info.x & 1u;
          info = info_below;
          info_below = info_next;
        }
        if (w > 1) info_next = LoadPair(wave_info + static_cast<int64_t>(w - 2) * kWaveSize + lane);
        if (classes)
        {
          const uint32_t sw = info.x;
          const int valid = min(kWord, n - w * kWord);
          const uint32_t prev_bit = (w > 0) ? (info_below.x >> (kWord - 1)) : (sw & 1u);
          xdn_word = sw ^ ((sw << 1) | prev_bit);                    // bit k: row k differs from the row below it
          xup_word = sw ^ ((sw >> 1) | (above_bit0 << (kWord - 1)));  // bit k: row k differs from the row above it
          if (r0 == last_band) xup_word &= ~(1u << (valid - 1));      // nothing above the last row
        }
      }
      const uint32_t swb = info.x >> sub;  // bit k: class of row r0 + k
      // distances to the nearest row of the other class below, for 8 rows at a time (dp[k & 7]): counted upwards from
      // the carry of the word and the class changes below the group
      int dp[8];
      uint32_t xup = 0;
      if (classes) xup = xup_word >> sub;
      // (two copies of a full band's code, with and without the class-change candidates: no test per row)
      auto rows = [&](auto guarded, auto with_classes, auto from_table) {
        constexpr bool kGuard = decltype(guarded)::value;
        constexpr bool kClasses = decltype(with_classes)::value;
        constexpr bool kTable = decltype(from_table)::value;  // every result of this band is below kSmallD2
        constexpr int kGroup = kGuard ? 1 : VGT_SWEEP_GROUP;  // rows finished together (X pass)
        [[maybe_unused]] float& lo_acc = lo_value;  // (named here: the uses below sit in code that depends on kGroup)
        [[maybe_unused]] float& hi_acc = hi_value;
        [[maybe_unused]] uint32_t grp_best[kGroup];
        [[maybe_unused]] int32_t grp_sign[kGroup];
#pragma unroll
        for (int k = kBand - 1; k >= 0; k--)
        {
          if (k % kStep == kStep - 1) refill_step();
          if (kClasses && k % 8 == 7)
          {
            const int first = sub + k - 7;  // position of the group's first row in the word
            const uint32_t below = xdn_word & LowBits(first);
            int d = below ? (first - (31 - __clz(static_cast<int>(below)))) : static_cast<int>(info.y) + first;
            const uint32_t xdn = xdn_word >> first;
#pragma unroll
            for (int j = 0; j < 8; j++)
            {
              d = ((xdn >> j) & 1u) ? 1 : d + 1;
              dp[j] = d;
            }
          }
          if (!kGuard || r0 + k < n)
          {
            const int q = r0 + k;
            const int q2 = 2 * q;
            // t1 = value of the second entry at row q minus the top's, t2 = the third's minus the second's
            // (rows and row differences are below 2^14: plain unsigned 24-bit multiplies)
            __builtin_assume(A >= 0 && A < 16384 && A2 >= 0 && A2 < 16384 && rt >= 0 && rt < 16384);
            int32_t t1 = Mad24Uniform(A, q2, nB);
            int32_t t2 = Mad24Uniform(A2, q2, nB2);
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(t2 <= 0) != 0ull, 0))
            {
              do
              {
                if (t1 <= 0)
                {
                  pop_down();
                  __builtin_assume(A2 >= 0 && A2 < 16384);
                  t1 = t2;
                  t2 = Mad24Uniform(A2, q2, nB2);
                }
              } while (__builtin_amdgcn_ballot_w64(t2 <= 0) != 0ull);
              __builtin_assume(rt >= 0 && rt < 16384);
            }
            // the better of the top (Gt + q^2 - 2 q rt) and the second entry at q
            uint32_t best = static_cast<uint32_t>(Mad24Uniform(rt, -q2, Gt) + q * q + min(t1, 0));
            if constexpr (kClasses)
            {
              dn = ((xup >> k) & 1u) ? 1 : dn + 1;
              const uint32_t dm = static_cast<uint32_t>(min(dp[k & 7], dn));
              best = min(best, dm * dm);
            }
            const int32_t sign = __builtin_amdgcn_sbfe(static_cast<int32_t>(swb), k, 1);  // -1 on the negative class
            if constexpr (kFinal)
            {
              // The conversions of kGroup rows run as ONE block of code: each is a chain of a dozen dependent
              // instructions (reciprocal square root, fp64 steps), and a wave-uniform branch per row would keep the
              // compiler from overlapping the chains of neighbouring rows.  The rare-path test is taken once per group.
              grp_best[k % kGroup] = best;
              grp_sign[k % kGroup] = sign;
              if (k % kGroup == 0)
              {
                float dist[kGroup];
                uint32_t d2s[kGroup];
                bool redo[kGroup], none[kGroup];
                bool any_rare = false;
                if constexpr (kTable)
                {
#pragma unroll
                  for (int j = kGroup - 1; j >= 0; j--)
                  {
#ifdef VGT_HOST_EMULATION
                    assert(grp_best[j] < kSmallD2);
#endif
                    dist[j] = small_table[grp_best[j]];
                    redo[j] = none[j] = false;
                    d2s[j] = 0;
                  }
                }
                else
                {
#pragma unroll
                for (int j = kGroup - 1; j >= 0; j--)  // row k + j
                {
                  uint32_t d2 = grp_best[j];
                  bool exact = false;
                  if constexpr (!kPlain)
                  {
                    if (g.add_virtual_border)
                    {
                      const int qq = q + j;
                      const int x = (g.pass_axis == 0) ? qq : outer_in_grid + g.outer_begin;
                      const int y = (g.pass_axis == 0) ? outer_in_grid + g.outer_begin : qq;
                      const int32_t site = (d2 >= static_cast<uint32_t>(kLimit)) ? kInf32 : static_cast<int32_t>(d2);
                      const int32_t clamped = ClampToVirtualBorder(site, x, y, z0 + static_cast<int>(zl) + g.z_offset, g.nx,
                                                                   g.ny, g.nz_global);
                      d2 = (clamped == kInf32) ? static_cast<uint32_t>(kLimit) : static_cast<uint32_t>(clamped);
                    }
                    exact = !(g.resolution > 1.0e-30 && g.resolution < 1.0e30);
                  }
                  bool unsure;
                  dist[j] = FastSqrtTimesResolution(d2, g.resolution, unsure);
                  // no voxel of the other class anywhere: only on lines whose stack holds nothing but the sentinels, and
                  // waves with such a line run the copy of the band code that also has the class candidates
                  none[j] = kClasses && d2 >= static_cast<uint32_t>(kLimit);
                  redo[j] = unsure || exact;
                  d2s[j] = d2;
                  any_rare = any_rare || redo[j] || none[j];
                }
                }
                if (!kTable && __builtin_expect(__builtin_amdgcn_ballot_w64(any_rare) != 0ull, 0))
                {
                  VGT_COLD_PATH();  // keeps the block out of the straight-line code
#pragma unroll
                  for (int j = 0; j < kGroup; j++)
                  {
                    if (redo[j]) dist[j] = SqrtTimesResolutionExact(static_cast<int32_t>(d2s[j]), g.resolution);
                    if (none[j]) dist[j] = __uint_as_float(0x7f800000u);
                  }
                }
                float value[kGroup];
                VGT_GLOBAL OutT* store_at = row_out;  // row k + kGroup - 1
#pragma unroll
                for (int j = kGroup - 1; j >= 0; j--)
                {
                  value[j] = __uint_as_float(__float_as_uint(dist[j]) | (static_cast<uint32_t>(grp_sign[j]) & 0x80000000u));
                  VGT_STREAM_STORE(value[j], LaneAddress(store_at, lane_bytes));
                  store_at = UniformPointer(store_at - rstride);
                }
                row_out = store_at;
                // extrema: two rows per instruction where rows come in pairs
#pragma unroll
                for (int j = kGroup - 1; j >= 0; j -= 2)
                {
                  if (j >= 1)
                  {
                    VGT_MIN3_F32(lo_acc, value[j], value[j - 1]);
                    VGT_MAX3_F32(hi_acc, value[j], value[j - 1]);
                  }
                  else
                  {
                    VGT_MIN_F32(lo_acc, value[j]);
                    VGT_MAX_F32(hi_acc, value[j]);
                  }
                }
              }
            }
            else
            {
              // ("no site" only on lines without any: the waves that have one run the band copy with kClasses)
              const int32_t d2 =
                  (kClasses && best >= static_cast<uint32_t>(kLimit)) ? kInf32 : static_cast<int32_t>(best);
              // (sign and magnitude, not two's complement: see sweep 1 of the X pass)
              VGT_STREAM_STORE(static_cast<OutT>(d2 | (sign & static_cast<int32_t>(0x80000000u))), LaneAddress(row_out, lane_bytes));
              row_out = UniformPointer(row_out - rstride);
            }
          }
        }
      };
      bool small_band = false;
      if constexpr (kUseTable) small_band = ((small_bands >> (r0 / kBand)) & 1ull) != 0ull;
      if (r0 + kBand > n)
        rows(std::true_type{}, std::true_type{}, std::false_type{});  // (the partial band: one copy, the candidates are "far" without classes)
      else if (small_band)
      {
        // (a band of small results has sites in every lane: `any_empty` does not concern it)
        if constexpr (kUseTable)
        {
          if (classes)
            rows(std::false_type{}, std::true_type{}, std::true_type{});
          else
            rows(std::false_type{}, std::false_type{}, std::true_type{});
        }
      }
      else if (classes || any_empty)
        rows(std::false_type{}, std::true_type{}, std::false_type{});
      else
        rows(std::false_type{}, std::false_type{}, std::false_type{});
      if (classes) dn = min(dn, kFar);
    }
  }
  if constexpr (kFinal)
  {
    // a batch of grids: every grid has its own extrema, so a workgroup hands them over item by item
    if (batched)
    {
      WaveMinMax(lo_value, hi_value, minmax_enc + 2 * batch_index);
      lo_value = INFINITY;
      hi_value = -INFINITY;
    }
  }
  }  // next unit of work
  // (the workgroup is one wave: no LDS for the reduction -- with the ring and the table of small distances a workgroup's
  // LDS is exactly a sixteenth of the CU's 160 KiB)
  if constexpr (kFinal) WaveMinMax(lo_value, hi_value, minmax_enc);
}

// Workgroups per launch = slots of the scratch buffer: what the 256 CUs x 16 waves of an MI355X hold at four waves per
// SIMD (more workgroups than that find nothing different to do: 4096, 5120 and 6144 measured the same).
#ifdef VGT_HOST_EMULATION
constexpr int64_t kSweepSlots = 3;  // (the CPU test wants slots that are used again and again)
#else
constexpr int64_t kSweepSlots = kMaxSweepSlots;
#endif
// Head of the scratch buffer: the work counters, each on its own cache line, cleared before every launch.
constexpr size_t kCounterBytes = (kSweepGroups * kCounterStride * sizeof(int) + 255) / 256 * 256;

int64_t SpillChunks(int64_t n, int chunk) { return (n + 4 + chunk - 1) / chunk + 1; }

// Entries of a line of n rows fit 32 bits when every G = F + row^2 stays below the sentinels.
bool PackedEntries(int64_t n, int64_t max_input)
{
  return n <= 1024 && max_input + (n - 1) * (n - 1) < Codec<true>::kSentinelG;
}

// Scratch of one slot: the spill chunks of its 64 lines (every entry of a full-depth stack) and one (sign word, carry) pair
// per 32 rows and lane.
size_t SlotScratchBytes(int64_t n, bool packed)
{
  const int64_t nwords = (n + kWord - 1) / kWord;
  const size_t spill = packed ? SpillChunks(n, RingShape<true>::kChunk) * kWaveSize * RingShape<true>::kChunk * sizeof(uint32_t)
                              : SpillChunks(n, RingShape<false>::kChunk) * kWaveSize * RingShape<false>::kChunk * sizeof(uint2);
  return spill + static_cast<size_t>(nwords + 1) * kWaveSize * sizeof(uint2);
}

// Scratch of one pass: the work counters and the slots of the workgroups in flight, for the entry kind the extents call
// for.  The size only decides how many workgroups a launch can use (LaunchSweep fits its slots into what it is given): a
// slab whose global Z extent needs 64-bit entries where the local extents promised 32-bit ones runs with fewer slots.
size_t PassScratchBytes(int64_t n, int64_t items, int64_t max_input)
{
  const int64_t slots = items < kSweepSlots ? items : kSweepSlots;
  const size_t planned = static_cast<size_t>(slots) * SlotScratchBytes(n, PackedEntries(n, max_input));
  const size_t one_wide = SlotScratchBytes(n, false);
  return kCounterBytes + (planned > one_wide ? planned : one_wide);
}

template <typename InT, typename OutT, bool kFinal>
hipError_t LaunchSweep(const InT* in, OutT* out, void* scratch, size_t scratch_bytes, uint32_t* minmax_enc, SweepGeom g,
                       int64_t outer_count, int64_t max_input, hipStream_t stream)
{
  g.zsegs = (g.nz + kWaveSize - 1) / kWaveSize;
  g.nwords = (g.n + kWord - 1) / kWord;
  const int64_t items = outer_count * g.zsegs;
  if (items <= 0) return hipSuccess;
  if (items > 0x7fffffffLL) return hipErrorInvalidValue;
  g.items = static_cast<int>(items);
  g.outers = static_cast<int>(outer_count);
  if (g.batch_outers <= 0)
  {
    g.batch_outers = g.outers;  // one grid
    g.batch_skip = 0;
  }
  const bool packed = PackedEntries(g.n, max_input);
  int64_t slots = items < kSweepSlots ? items : kSweepSlots;
  {
    // as many slots as the scratch holds (see PassScratchBytes)
    const size_t per_slot = SlotScratchBytes(g.n, packed);
    const int64_t fit = scratch_bytes > kCounterBytes ? static_cast<int64_t>((scratch_bytes - kCounterBytes) / per_slot) : 0;
    if (fit < 1) return hipErrorInvalidValue;
    if (fit < slots) slots = fit;
  }
  // (measured with streaming row accesses: -1.7 % on the Y pass, -0.7 % on the X pass at 1024^3, -1 % at 2048 rows)
  g.groups = static_cast<int>(slots < kSweepGroups ? slots : kSweepGroups);
  const int chunk = packed ? RingShape<true>::kChunk : RingShape<false>::kChunk;
  g.chunks = static_cast<int>(SpillChunks(g.n, chunk));
  // scratch: work counter | spill chunks of every slot | one (sign word, carry) pair per 32 rows, lane and slot
  char* bytes = static_cast<char*>(scratch);
  int* counter = reinterpret_cast<int*>(bytes);
  unsigned char* spill = reinterpret_cast<unsigned char*>(bytes + kCounterBytes);
  const size_t spill_bytes =
      static_cast<size_t>(slots) * g.chunks * kWaveSize * chunk * (packed ? sizeof(uint32_t) : sizeof(uint2));
  uint2* info = reinterpret_cast<uint2*>(bytes + kCounterBytes + spill_bytes);
#ifdef VGT_HOST_EMULATION
  *counter = 0;
#endif
  const dim3 grid(static_cast<unsigned>(slots)), block(kWaveSize);
  // the plain X pass: no virtual border, resolution inside the range of the fast final conversion
  const bool general = kFinal && (g.add_virtual_border || !(g.resolution > 1.0e-30 && g.resolution < 1.0e30));
#ifndef VGT_HOST_EMULATION
  {
    // the work counters start at zero
    const hipError_t err = hipMemsetAsync(counter, 0, kSweepGroups * kCounterStride * sizeof(int), stream);
    if (err != hipSuccess) return err;
  }
#endif
#define VGT_LAUNCH_SWEEP(PACKED, PLAIN)                                                                             \
  hipLaunchKernelGGL((SweepPassKernel<InT, OutT, kFinal, PACKED, PLAIN>), grid, block, 0, stream, in, out, spill, info, \
                     minmax_enc, counter, g)
  if (packed && general)
    VGT_LAUNCH_SWEEP(true, !kFinal);
  else if (packed)
    VGT_LAUNCH_SWEEP(true, true);
  else if (general)
    VGT_LAUNCH_SWEEP(false, !kFinal);
  else
    VGT_LAUNCH_SWEEP(false, true);
#undef VGT_LAUNCH_SWEEP
  return hipGetLastError();
}


// Largest magnitudes the passes can meet: squared Z distances in the Y pass, plus squared Y distances in the X pass.
int64_t MaxInputY(const SdfParams& p)
{
  const int64_t nzg = p.nz_global > 0 ? p.nz_global : p.nz;
  return (nzg - 1) * (nzg - 1);
}
int64_t MaxInputX(const SdfParams& p) { return MaxInputY(p) + (p.ny - 1) * (p.ny - 1); }
}  // namespace

// Scratch of the sweep passes for a grid: the larger of the two passes' needs (they use the same bytes, one after
// the other).
size_t SweepPassScratchBytes(int64_t nx, int64_t ny, int64_t nz, int64_t batch)
{
  const int64_t zsegs = (nz + kWaveSize - 1) / kWaveSize;
  const int64_t max_y = (nz - 1) * (nz - 1), max_x = max_y + (ny - 1) * (ny - 1);
  const size_t y = PassScratchBytes(ny, batch * nx * zsegs, max_y), x = PassScratchBytes(nx, batch * ny * zsegs, max_x);
  return (y > x ? y : x) + 256;
}

// Y pass of the default pipeline: class records (pass 1, edt_record_kernels.hip) of p.nx slices -> int32.
// `records` must be followed by kRecordPadding readable records (vgt_internal.hpp: the block loads run up to 191 rows
// past a line's end; what they fetch there is never used).
hipError_t LaunchPassYSweepRecords(const ClassRecord* records, int32_t* out32, SweepScratch scratch, const SdfParams& p,
                                   hipStream_t stream)
{
#ifndef VGT_HOST_EMULATION  // (the CPU emulation runs the sweeps on every length)
  if (p.ny <= ShortLineLimit(p.nx * ((p.nz + kWaveSize - 1) / kWaveSize))) return LaunchPassYShortRecords(records, out32, p, stream);
#endif
  int64_t outer_count = 0;
  const SweepGeom g = SweepGeometry(p, 1, &outer_count);
  return LaunchSweep<ClassRecord, int32_t, false>(records, out32, scratch.ptr, scratch.bytes, nullptr, g, outer_count,
                                                  MaxInputY(p), stream);
}

// X pass over the Y positions [outer_begin, outer_begin + outer_count) of the grid (outer_count < 0: all of them):
// full-grid pointers and extents in `p`.
hipError_t LaunchPassXSweepFinalizeRange(const int32_t* in32, float* sdf, uint32_t* minmax_enc, SweepScratch scratch,
                                         const SdfParams& p, int64_t outer_begin, int64_t outer_count_or_all,
                                         hipStream_t stream)
{
  int64_t outer_count = 0;
  SweepGeom g = SweepGeometry(p, 0, &outer_count);
  if (p.batch > 1)
  {
    // p.batch grids of p.nx x p.ny x p.nz, one after the other in both fields: the outer indices of grid b are
    // [b * ny, (b + 1) * ny), its lines begin (nx - 1) * ny * nz elements further on per grid than y * nz alone says
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
  return LaunchSweep<int32_t, float, true>(in32, sdf, scratch.ptr, scratch.bytes, minmax_enc, g, outer_count,
                                           MaxInputX(p), stream);
}

hipError_t LaunchPassXSweepFinalize(const int32_t* in32, float* sdf, uint32_t* minmax_enc, SweepScratch scratch,
                                    const SdfParams& p, hipStream_t stream)
{
  return LaunchPassXSweepFinalizeRange(in32, sdf, minmax_enc, scratch, p, 0, -1, stream);
}

}  // namespace vgt


