// Internal declarations shared by the HIP translation units of libvgt_hip.so.
// Not part of the public ABI (that is include/vgt_hip.h).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include <string>

namespace vgt
{
// Sets the calling thread's vgt_hip_last_error() message (for translation units other than
// vgt_hip_capi.hip that implement parts of the C ABI).
void SetLastError(const std::string& message);
}  // namespace vgt
struct vgt_hip_ctx;
namespace vgt
{
// The stream a context enqueues on (vgt_hip_set_stream's, or its own), for translation units that add work
// which later calls on the context must see (vgt_hipx_multi.hip).
hipStream_t ContextStream(const vgt_hip_ctx* ctx);

// Intermediate encodings of the signed Euclidean distance transform.
//  pass 1 (Z scan)  -> int16: +d for a free voxel, -d for a filled voxel, d = distance in
//                      voxels along Z to the nearest voxel of the OTHER class,
//                      |value| == kInf16 when the line holds no such voxel.
//  pass 2 (Y pass)  -> int32: +-(squared distance in the YZ plane), kInf32 when none (two's complement between the
//                      brute-force passes, sign and magnitude between the sweep passes: the Y and X passes of one
//                      extraction are always of the same variant).
//  pass 3 (X pass)  -> float SDF.
constexpr int16_t kInf16 = 32767;
constexpr int32_t kInf32 = 0x7fffffff;
// Largest supported extent per axis: keeps every squared distance (<= 3*kMaxExtent^2)
// and every parabola value below 2^31.
constexpr int64_t kMaxExtent = 16384;

struct SdfParams
{
  int64_t nx, ny, nz;  // extents of the grid held by this device (a Z slab when partitioned)
  double resolution;
  int unknown_is_filled;
  int add_virtual_border;
  // Z-slab partitioning (multi-GPU): this device holds global z in [z_offset, z_offset + nz) of a
  // grid with nz_global voxels along Z.  Single device: z_offset = 0, nz_global = nz.
  int64_t z_offset = 0;
  int64_t nz_global = 0;
  // A batch of `batch` equal grids, one after the other in every buffer (vgt_hip_sdf_batch_dev): pass 1 and the Y pass
  // see it as one grid of batch * nx slices (their lines never leave a slice), only the X pass -- whose lines run along
  // x -- and the extrema need to know.
  int64_t batch = 1;
};

// Pass 1 of the default pipeline (edt_record_kernels.hip): the binarised grid as CLASS RECORDS, one 16-byte record per
// 64-voxel word of a Z line, laid out [x][word][y] so that the records a Y-pass wave walks through (one x, one word,
// every y) are contiguous.  A record holds the classes of its 64 voxels and where the nearest class change outside the
// word is; the Y pass derives every voxel's distance along Z from it (0.25 B per voxel instead of a 2-byte distance).
// A class change ("transition") at position t means class(t) != class(t + 1): for every voxel above t (up to the next
// transition) the nearest voxel of the other class below is t, for every voxel at or below t the nearest one above is t + 1.
//   mask:    bit k = voxel 64 w + k is filled; bits past the end of the line repeat the line's last voxel
//   below2:  last transition below the word (t < 64 w), as 2 (t - 64 w) + kRecordBias; kRecordNoneBelow when none
//   above2:  first transition at or after the word's last voxel (t >= 64 w + 63), same encoding; kRecordNoneAbove when
//            none; kRecordNoSite when the whole line -- all slabs of it -- holds one class only (below2 is then
//            kRecordNoneBelow and the word has no transition of its own): no voxel of the word has a distance along Z
// With xq = 2 lane - 1 + kRecordBias, |xq - t2| + 1 is TWICE the distance from voxel `lane` to the other class across
// the transition encoded as t2, whichever side it lies on (one v_sad_u32); the "none" encodings give >= 2 kInf16.
struct ClassRecord
{
  uint32_t mask_lo, mask_hi, below2, above2;
};
constexpr uint32_t kRecordBias = 1u << 17;
constexpr uint32_t kRecordNoneBelow = 0u;
constexpr uint32_t kRecordNoneAbove = 2u * kRecordBias;
constexpr uint32_t kRecordNoSite = 0xfffffff0u;
inline int64_t RecordWords(int64_t nz) { return (nz + 63) / 64; }
// records of a grid (+ padding: the Y pass loads whole 64-row blocks three blocks ahead, so it FETCHES records up to
// 3 * 64 - 1 = 191 rows past the end of the line it works on -- the next lines' records, a later chunk's not yet written
// ones, or this padding, which nobody initialises.  It never USES them: every row is guarded by `row < n`.  Whoever hands
// LaunchPassYSweepRecords a record buffer must keep the padding behind it, or the fetch leaves the allocation.)
constexpr int kRecordPadding = 256;
inline size_t ClassRecordBytes(int64_t nx, int64_t ny, int64_t nz)
{
  return (static_cast<size_t>(nx) * static_cast<size_t>(RecordWords(nz)) * static_cast<size_t>(ny) + kRecordPadding) *
         sizeof(ClassRecord);
}

// Per-line summary of a Z slab, exchanged between devices: 4 bytes.  A slab's first voxel is filled or free, so of
// "first filled" and "first free" one is the slab's first voxel; the record keeps the class of the first voxel and the
// position of the first voxel of the OTHER class (the same for the last voxel):
//   bit 15 = 1: the slab's first / last voxel is filled;  bits 0-14: global z of the first / last voxel of the other
//   class inside the slab, kSlabNone when the slab holds one class only.
// The slabs follow SlabRange (equal shares, earlier slabs take the remainder), which tells the reader where they begin.
struct SlabLineSummary
{
  uint16_t first, last;
};
constexpr uint16_t kSlabFilledBit = 0x8000u;
constexpr uint16_t kSlabNone = 0x7fffu;
// [z0, z0 + count) of slab `rank` of `world` along a Z axis of nz voxels.
inline void SlabRange(int64_t nz, int world, int rank, int64_t* z0, int64_t* count)
{
  const int64_t share = nz / world, extra = nz % world;
  *z0 = rank * share + (rank < extra ? rank : extra);
  *count = share + (rank < extra ? 1 : 0);
}
// Per-line carries derived from the other slabs' summaries: nearest filled / free voxel of the
// line below this slab (largest global z) and above it (smallest global z), -1 when absent.
struct SlabLineCarry
{
  int16_t prev_filled, next_filled, prev_free, next_free;
};

// kDefault: class records (pass 1, edt_record_kernels.hip) + lane-per-line sweep passes, Felzenszwalb-Huttenlocher
// stacks with their tops in LDS (edt_sweep_kernels.hip; any extent).  kBruteForce (testing library only): the one
// independent cross-check -- an int16 Z scan and a pruned outward search per voxel straight from HBM (edt_kernels.hip).
// Both exact.  (Earlier rounds' other pipelines -- LDS-tiled envelopes, the int16-fed sweeps -- are in the git history
// and profiles/r2 ... r5/experiments.md.)
enum class EdtVariant : int { kDefault = 0, kBruteForce = 1 };

// --- launchers (edt_kernels.hip).  All asynchronous on `stream`. ---
// Z scan: occupancy (float) or mask (u8) -> int16 signed 1-D distance.
// `summary` (optional) receives one SlabLineSummary per (x, y) line.
hipError_t LaunchScanZFromOccupancy(const float* occupancy, int16_t* out16, const SdfParams& p,
                                    SlabLineSummary* summary, hipStream_t stream);
hipError_t LaunchScanZFromMask(const uint8_t* mask, int16_t* out16, const SdfParams& p,
                                SlabLineSummary* summary, hipStream_t stream);
// Multi-GPU: folds the carries of the other slabs into the slab-local pass-1 distances, in place.
hipError_t LaunchFinalizeCheck(int64_t first, int64_t count, double resolution,
                               unsigned long long* result_dev, hipStream_t stream);
hipError_t LaunchSlabCarries(const SlabLineSummary* summaries, int world, int rank, int64_t lines, int64_t nz_global,
                             SlabLineCarry* carries, hipStream_t stream);
hipError_t LaunchSlabFixup(int16_t* io16, const SlabLineCarry* carries, const SdfParams& p,
                           hipStream_t stream);
// Pass 1 of the default pipeline (edt_record_kernels.hip): occupancy (float) or mask (u8) -> class records
// (ClassRecordBytes bytes).  `summary` (optional, multi-GPU) receives one SlabLineSummary per (x, y) line; lines that
// hold one class are then marked by LaunchSlabRecordFixup, which folds the other slabs' carries into the records.
hipError_t LaunchClassRecordsFromOccupancy(const float* occupancy, ClassRecord* records, const SdfParams& p,
                                           SlabLineSummary* summary, hipStream_t stream);
hipError_t LaunchClassRecordsFromMask(const uint8_t* mask, ClassRecord* records, const SdfParams& p,
                                      SlabLineSummary* summary, hipStream_t stream);
hipError_t LaunchSlabRecordFixup(ClassRecord* records, const SlabLineCarry* carries, const SdfParams& p,
                                 hipStream_t stream);
// Scratch of the sweep passes (work counters, spilled stack entries and sign words of the workgroups in flight): the
// launchers use as many workgroups as its size allows, see SweepPassScratchBytes.
struct SweepScratch
{
  void* ptr;
  size_t bytes;
};
// Y pass of the default pipeline: class records -> int32 signed squared distance (edt_sweep_kernels.hip).
hipError_t LaunchPassYSweepRecords(const ClassRecord* records, int32_t* out32, SweepScratch scratch, const SdfParams& p,
                                   hipStream_t stream);
// Y pass: int16 -> int32 signed squared distance.
// `scratch`: SweepPassScratchBytes bytes (part of the SDF workspace).
hipError_t LaunchPassY(const int16_t* in16, int32_t* out32, SweepScratch scratch, const SdfParams& p,
                       EdtVariant variant, hipStream_t stream);
// X pass + finalize: int32 -> float SDF, min/max folded into minmax_enc (2 x uint32,
// order-preserving encoding, must be pre-initialised by InitMinMax).
hipError_t LaunchPassXFinalize(const int32_t* in32, float* sdf, uint32_t* minmax_enc,
                               SweepScratch scratch, const SdfParams& p, EdtVariant variant,
                               hipStream_t stream);
// Lines of at most kShortLineRows rows take the short-line kernels (edt_short_kernels.hip: the whole line in registers,
// exhaustive search) instead of the sweeps; same encodings, same results.  ShortLineRows() is the limit in force for
// the X pass: kShortLineRows, or what a testing build was told (vgt_hip_testing_set_short_line_rows; 0 = sweeps for
// every length); ShortLineLimit(items) the Y pass's.
constexpr int kShortLineRows = 64;
// ... and lines of up to kShortLineRowsFewItems rows when the launch has so few items (at most kFewLineItems bundles of
// 64 lines) that the sweeps would leave most of the chip idle: the search costs O(rows^2) but splits over the waves.
constexpr int kShortLineRowsFewItems = 128;
constexpr int64_t kFewLineItems = 1024;
// What a testing build was told (vgt_hip_testing_set_short_line_rows), -1 = nothing (the product library: always).
int ShortLineOverride();
inline int ShortLineRows()
{
  const int told = ShortLineOverride();
  return told >= 0 ? told : kShortLineRows;
}
inline int ShortLineLimit(int64_t items)
{
  const int told = ShortLineOverride();
  if (told >= 0) return told;  // (a testing build was told: that limit, whatever the item count)
  return items <= kFewLineItems ? kShortLineRowsFewItems : kShortLineRows;
}
#ifdef VGT_HIP_TESTING
void SetShortLineRows(int rows);  // 0 ... kShortLineRowsFewItems; negative: back to the defaults
#endif
hipError_t LaunchPassYShortRecords(const ClassRecord* records, int32_t* out32, const SdfParams& p, hipStream_t stream);
hipError_t LaunchPassXShortFinalizeRange(const int32_t* in32, float* sdf, uint32_t* minmax_enc, const SdfParams& p,
                                         int64_t outer_begin, int64_t outer_count, hipStream_t stream);
// For callers that pipeline parts of a grid: the Y pass treats X slices independently (call LaunchPassY with nx =
// slices of a contiguous part), and the X pass can be launched over a range of Y positions (full-grid pointers and
// extents in `p`; outer_count < 0: the whole axis).  LinePassesTakeRanges: whether `variant` supports that for `p`.
bool LinePassesTakeRanges(const SdfParams& p, EdtVariant variant);
hipError_t LaunchPassXFinalizeRange(const int32_t* in32, float* sdf, uint32_t* minmax_enc, SweepScratch scratch,
                                    const SdfParams& p, EdtVariant variant, int64_t outer_begin, int64_t outer_count,
                                    hipStream_t stream);
// Scratch for the lane-per-line sweep passes (edt_sweep_kernels.hip): work counter, spilled stack entries and sign
// words of the workgroups in flight.
size_t SweepPassScratchBytes(int64_t nx, int64_t ny, int64_t nz, int64_t batch = 1);
// `count` pairs of ordered encodings (one per grid of a batch)
hipError_t LaunchInitMinMax(uint32_t* minmax_enc, hipStream_t stream, int64_t count = 1);
hipError_t LaunchDecodeMinMax(const uint32_t* minmax_enc, float* minmax_out, hipStream_t stream, int64_t count = 1);

// --- launchers (cell_kernels.hip): map types whose cells carry an object id ---
hipError_t LaunchCellMask(const void* cells_dev, int64_t num_cells, int cell_bytes, int object_id_offset,
                          int mode, const uint32_t* objects_dev, int num_objects, int unknown_is_filled,
                          uint8_t* mask_dev, hipStream_t stream);
// masks_dev[b * num_cells + i] = cell i is filled and belongs to object ids_dev[b]: the byte masks of a batch of
// per-object extractions, one pass over the cells.
hipError_t LaunchCellObjectMasks(const void* cells_dev, int64_t num_cells, int cell_bytes, int object_id_offset,
                                 const uint32_t* ids_dev, int num_ids, int unknown_is_filled, uint8_t* masks_dev,
                                 hipStream_t stream);
// Distinct object ids > 0 in one pass: table_dev = 2^table_log2 zeroed uint32 slots, ids_dev = room for as many ids,
// count_overflow_dev = {number of ids, 1 if the table overflowed} (zeroed by the caller).
hipError_t LaunchDistinctObjectIds(const void* cells_dev, int64_t num_cells, int cell_bytes, int object_id_offset,
                                   uint32_t* table_dev, int table_log2, uint32_t* ids_dev, uint32_t* count_overflow_dev,
                                   hipStream_t stream);
hipError_t LaunchNextObjectId(const void* cells_dev, int64_t num_cells, int cell_bytes, int object_id_offset,
                              uint32_t after, uint32_t* result_dev, hipStream_t stream);
hipError_t LaunchCombineFreeAndNamed(const float* free_sdf_dev, const float* named_sdf_dev, int64_t num_cells,
                                     float* out_dev, uint32_t* minmax_enc, hipStream_t stream);

// SDF consumer: grid-aligned (or rotated) coarse gradient of every voxel, 3 doubles per voxel.
hipError_t LaunchCoarseGradient(const float* sdf_dev, int64_t nx, int64_t ny, int64_t nz, double resolution,
                                int enable_edge_gradients, const double* rotation_host, double* gradient_dev,
                                uint8_t* has_value_dev, hipStream_t stream);

// SDF consumers, batched queries (3 doubles per query point): trilinear distance estimate and fine gradient.
// grid_from_world_host: 16 doubles column-major (InverseOriginTransform) or nullptr = identity.
hipError_t LaunchEstimateDistance(const float* sdf_dev, int64_t nx, int64_t ny, int64_t nz, double resolution,
                                  const double* grid_from_world_host, const double* queries_dev, int64_t num_queries,
                                  double* distance_dev, uint8_t* has_value_dev, hipStream_t stream);
hipError_t LaunchFineGradient(const float* sdf_dev, int64_t nx, int64_t ny, int64_t nz, double resolution,
                              const double* grid_from_world_host, const double* queries_dev, int64_t num_queries,
                              double nominal_window_size, double* gradient_dev, uint8_t* has_value_dev,
                              uint32_t* window_too_large_dev, hipStream_t stream);

// SignedDistanceField::ComputeLocalExtremaMap: 3 doubles per voxel (grid-frame location of the extremum the
// voxel's gradient chain ends at, +inf when it leaves the grid).  scratch_dev: LocalExtremaScratchBytes bytes.
size_t LocalExtremaScratchBytes(int64_t num_cells);
hipError_t LaunchLocalExtremaMap(const float* sdf_dev, int64_t nx, int64_t ny, int64_t nz, double resolution,
                                 const double* rotation_host, double* extrema_dev, void* scratch_dev,
                                 hipStream_t stream);

// --- launchers (voxelizer_kernels.hip) ---
struct RaycastGridF32
{
  float max_range;
  float xform[16];
  float voxel_size, inverse_voxel_size;
  float grid_size[3];
  int32_t counts[3];
};
struct RaycastGridF64
{
  double max_range;
  double xform[16];
  double voxel_size, inverse_voxel_size;
  double grid_size[3];
  int32_t counts[3];
};
// point_stride = floats between consecutive points (3 for packed xyz; PointCloud2: point_step / 4).
// scratch_dev (optional, RaycastScratchBytes(num_points) bytes): lets large clouds be ordered by ray
// direction and counted per workgroup in LDS (same counts, far fewer global atomics).
size_t RaycastScratchBytes(int64_t num_points);
hipError_t LaunchRaycastF32(const float* points_dev, int64_t num_points, int64_t point_stride,
                            const RaycastGridF32& g, int32_t* tracking_dev, int threads_per_block,
                            void* scratch_dev, size_t scratch_bytes, hipStream_t stream);
hipError_t LaunchRaycastF64(const double* points_dev, int64_t num_points, const RaycastGridF64& g,
                            int32_t* tracking_dev, int threads_per_block, void* scratch_dev, size_t scratch_bytes,
                            hipStream_t stream);
// dst[i] += src[i] over `count` int32 counts (both 16-byte aligned): sums the private tracking grids of a split cloud.
hipError_t LaunchAccumulateCounts(int32_t* dst_dev, const int32_t* src_dev, int64_t count, hipStream_t stream);
hipError_t LaunchFilter(const int32_t* tracking_dev, int64_t num_cells, int32_t num_grids,
                        double percent_seen_free, int32_t outlier_points_threshold,
                        int32_t num_cameras_seen_free, bool ratio_in_double, float* occupancy_dev,
                        int threads_per_block, hipStream_t stream);
}  // namespace vgt
