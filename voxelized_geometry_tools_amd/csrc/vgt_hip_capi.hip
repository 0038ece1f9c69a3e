// extern "C" surface of libvgt_hip.so (declared in include/vgt_hip.h): contexts, handles,
// host<->device staging and the sequencing of the kernels.  No kernel code here.
#include "../../include/vgt_hip.h"

#include "vgt_internal.hpp"
#include "host_pages.hpp"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <vector>
#include <new>
#include <string>
#include <thread>
#ifdef __linux__
#include <sys/mman.h>
#endif
#include <vector>

// Host point clouds are uploaded and raycast on one of a few "upload lanes" (own stream + own staging
// buffer), so that concurrent RaycastPoints calls of the reference's parallel cloud dispatch
// (S/device_pointcloud_voxelization.cpp:147-149) overlap: cloud i+1's H2D copy runs beside cloud
// i's kernel, and kernels into different tracking grids run side by side.
struct UploadLane
{
  std::mutex mutex;
  hipStream_t stream = nullptr;
  hipEvent_t after_ctx = nullptr;  // orders the lane behind what the context's stream holds (grid zeroing)
  void* stage = nullptr;
  size_t stage_bytes = 0;
  // Page-locked host buffer the caller's cloud is copied to first: a 12-MB cloud from pageable memory goes through the
  // runtime's bounce buffers in 5 - 12 ms, page-locking the caller's buffer per call serialises in the driver when
  // several clouds arrive at once; a copy into a buffer that is locked once costs about a millisecond, in parallel
  // across the lanes, and the DMA 0.2 ms.
  void* pinned = nullptr;
  size_t pinned_bytes = 0;
};
constexpr int kUploadLanes = 4;

struct vgt_hip_ctx
{
  int device = -1;
  int threads_per_block = 256;
  int raycast_threads = 0;       // 0: the raycast kernels choose their own workgroup sizes (no HIP_THREADS_PER_BLOCK given)
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::mutex mutex;              // serialises enqueues + the staging buffer
  UploadLane lanes[kUploadLanes];
  std::atomic<unsigned> next_lane{0};
  void* ray_scratch = nullptr;   // sort scratch of raycasts of device-resident clouds (context stream)
  size_t ray_scratch_bytes = 0;
  float* minmax_out = nullptr;   // 2 floats (device) for host-facing SDF calls
  vgt::EdtVariant variant = vgt::EdtVariant::kDefault;
  // deferred per-kernel timing (vgt_hip_timing_start / _stop): 8 events per SDF call
  std::vector<hipEvent_t> timing_events;
  std::vector<uint8_t> timing_kind;  // 1 = single-device call, 2 = slab begin + finish
  int timing_slots = 0;
  int timing_used = 0;
  // Device buffers of the host-pointer SDF entry points (input, field, workspace), kept across calls
  // and grown on demand: a caller that extracts fields repeatedly pays for hipMalloc / hipFree once.
  // vgt_hip_trim() gives them back.
  void* sdf_in = nullptr;
  size_t sdf_in_bytes = 0;
  void* sdf_out = nullptr;
  size_t sdf_out_bytes = 0;
  void* sdf_ws = nullptr;
  size_t sdf_ws_bytes = 0;
  // Page-locked staging ring of the batched downloads (DownloadToHostArrays): kStagingSlots slots of kStagingSlotBytes,
  // allocated by the first such call, kept until vgt_hip_trim / vgt_hip_destroy; one event per slot.
  void* host_staging = nullptr;
  std::vector<hipEvent_t> staging_events;
  // Handles created from this context (grids, filter grids, cell grids) point back at it.  A
  // context destroyed while handles are alive releases its device resources at once but keeps this
  // struct until the last handle is gone, so handle destructors never touch freed memory.
  std::atomic<int> children{0};
  std::atomic<bool> destroyed{false};
  // Device buffers of destroyed tracking-grid / filter-grid handles, kept for the next handle of the same size (the
  // voxelizer allocates both per call, S/cuda_voxelization_helpers.cu:641-658,701-708; a hipMalloc + hipFree pair of
  // 128 MiB costs more than the raycast it serves).  Guarded by `mutex`; vgt_hip_trim and vgt_hip_destroy free them.
  struct PooledBuffer
  {
    void* ptr;
    size_t bytes;
  };
  std::vector<PooledBuffer> pool;
  size_t pool_bytes = 0;
  bool pool_closed = false;  // set by vgt_hip_destroy: handles destroyed later free their buffers themselves
  // Z-slab calls: which slab a carries buffer was computed for (vgt_hip_sdf_slab_carries_dev decodes the gathered
  // summaries with the slab ranges of vgt_hip_sdf_slab_range; vgt_hip_sdf_slab_finish_dev refuses carries that were
  // made for another slab than the one it is given).  Guarded by `mutex`.
  struct SlabNote
  {
    int64_t z_offset, nz_local, nz_global;
  };
  std::map<const void*, SlabNote> slab_notes;
  // Copy streams and events of the pipelined host-pointer SDF extraction (SdfFromHostPipelined)
  hipStream_t copy_in = nullptr;
  hipStream_t copy_out = nullptr;
  std::vector<hipEvent_t> pipeline_events;
};

struct vgt_hip_grids
{
  vgt_hip_ctx* ctx = nullptr;
  int device = -1;
  int32_t* dev = nullptr;
  int64_t num_cells = 0;
  int32_t num_grids = 0;
};

struct vgt_hip_filter
{
  vgt_hip_ctx* ctx = nullptr;
  int device = -1;
  float* dev = nullptr;
  int64_t num_cells = 0;
  // vgt_hip_filter_grid_create_deferred: the upload runs on the context's copy stream; `uploaded` is recorded behind it
  // and the caller's array stays page-locked (`pin`, a ScopedHostPin) until a call has waited for the copy
  hipEvent_t uploaded = nullptr;
  bool upload_pending = false;
  void* pin = nullptr;
};

// Device copy of a grid of cell records (occupancy + optional object id) and the buffers the
// per-object SDF extractions reuse.
struct vgt_hip_cells
{
  vgt_hip_ctx* ctx = nullptr;
  int device = -1;
  int64_t nx = 0, ny = 0, nz = 0;
  int cell_bytes = 0;
  int object_id_offset = -1;
  void* records = nullptr;      // [num_cells] records of cell_bytes bytes
  uint8_t* mask = nullptr;      // [num_cells] predicate result, the Z scan's input
  float* sdf = nullptr;         // [num_cells]
  float* sdf_named = nullptr;   // [num_cells], allocated by the first free-and-named extraction
  void* workspace = nullptr;
  size_t workspace_bytes = 0;
  uint32_t* objects = nullptr;  // object list of the current call
  size_t objects_capacity = 0;
  uint32_t* scalar = nullptr;   // two uint32: result of a reduction + found flag
  // vgt_hip_cells_object_sdfs: masks, fields (+ extrema) and workspace of a batch of per-object extractions (grow-only)
  void* batch_masks = nullptr;
  size_t batch_masks_bytes = 0;
  void* batch_sdf = nullptr;
  size_t batch_sdf_bytes = 0;
  void* batch_ws = nullptr;
  size_t batch_ws_bytes = 0;
};

namespace
{
thread_local std::string g_last_error;

int Fail(int code, const std::string& msg)
{
  g_last_error = msg;
  return code;
}

}  // namespace
namespace vgt
{
void SetLastError(const std::string& message) { g_last_error = message; }
hipStream_t ContextStream(const vgt_hip_ctx* ctx) { return ctx->stream; }
}  // namespace vgt
namespace
{
int FailHip(const char* what, hipError_t err)
{
  g_last_error = std::string("[") + what + "] HIP error [" + hipGetErrorString(err) + "]";
  return VGT_HIP_ERR_RUNTIME;
}

// A handle is being created from / destroyed after its context (see vgt_hip_ctx::children).
void AdoptChild(vgt_hip_ctx* ctx) { ctx->children.fetch_add(1); }
// Waits for the work that may still use the handle's buffers, then drops the handle's reference.
void ReleaseChild(vgt_hip_ctx* ctx, int device)
{
  if (!ctx) return;
  (void)hipSetDevice(device);
  if (ctx->destroyed.load())
    (void)hipDeviceSynchronize();  // the context's stream is gone (vgt_hip_destroy drained it)
  else
    (void)hipStreamSynchronize(ctx->stream);
  if (ctx->children.fetch_sub(1) == 1 && ctx->destroyed.load()) delete ctx;
}

#define VGT_TRY_HIP(expr, what)                          \
  do                                                     \
  {                                                      \
    const hipError_t vgt_err_ = (expr);                  \
    if (vgt_err_ != hipSuccess) return FailHip(what, vgt_err_); \
  } while (0)

size_t AlignUp(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Handle buffers come from / go back to the context's pool (at most kPoolLimit bytes are kept).
constexpr size_t kPoolLimit = size_t{4} << 30;
hipError_t PoolAllocate(vgt_hip_ctx* ctx, void** ptr, size_t bytes)
{
  {
    std::lock_guard<std::mutex> lock(ctx->mutex);
    for (size_t i = 0; i < ctx->pool.size(); i++)
      if (ctx->pool[i].bytes == bytes)
      {
        *ptr = ctx->pool[i].ptr;
        ctx->pool_bytes -= bytes;
        ctx->pool.erase(ctx->pool.begin() + static_cast<std::ptrdiff_t>(i));
        return hipSuccess;
      }
  }
  return hipMalloc(ptr, bytes);
}
// (the caller has made sure no work uses the buffer any more)
void PoolRelease(vgt_hip_ctx* ctx, void* ptr, size_t bytes)
{
  if (!ptr) return;
  if (ctx && !ctx->destroyed.load())
  {
    std::lock_guard<std::mutex> lock(ctx->mutex);
    if (!ctx->pool_closed && ctx->pool_bytes + bytes <= kPoolLimit && ctx->pool.size() < 16)
    {
      ctx->pool.push_back({ptr, bytes});
      ctx->pool_bytes += bytes;
      return;
    }
  }
  (void)hipFree(ptr);
}
void FreePool(vgt_hip_ctx* ctx)
{
  for (const auto& b : ctx->pool) (void)hipFree(b.ptr);
  ctx->pool.clear();
  ctx->pool_bytes = 0;
}

// Grow-only device buffer.
hipError_t Reserve(void** ptr, size_t* have, size_t need)
{
  if (*have >= need && *ptr) return hipSuccess;
  if (*ptr) (void)hipFree(*ptr);
  *ptr = nullptr;
  *have = 0;
  const hipError_t err = hipMalloc(ptr, need);
  if (err == hipSuccess) *have = need;
  return err;
}

void FreeCachedSdfBuffers(vgt_hip_ctx* ctx)
{
  for (void** p : {&ctx->sdf_in, &ctx->sdf_out, &ctx->sdf_ws, &ctx->ray_scratch})
  {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
  }
  ctx->sdf_in_bytes = ctx->sdf_out_bytes = ctx->sdf_ws_bytes = ctx->ray_scratch_bytes = 0;
  if (ctx->host_staging) (void)hipHostFree(ctx->host_staging);
  ctx->host_staging = nullptr;
  for (hipEvent_t e : ctx->staging_events)
    if (e) (void)hipEventDestroy(e);
  ctx->staging_events.clear();
}

// Page-locks a caller-owned host range for the duration of a call, unless it already is pinned
// (hipHostMalloc / hipHostRegister by the caller): copies from pageable memory go through the
// runtime's bounce buffers at a fraction of the PCIe rate and are not asynchronous.  Best effort:
// when the driver refuses (limits, odd mappings) the copy simply takes the pageable path.
// Two threads may hand the same array to two contexts at once: the registration is shared through a
// process-wide table of the ranges THIS library page-locked, with a use count, so the range stays locked until the
// last call that relies on it has finished.
class ScopedHostPin
{
public:
  ScopedHostPin(const void* ptr, size_t bytes)
  {
    if (!ptr || bytes < (size_t{1} << 20)) return;  // small copies: registration costs more than it saves
    std::lock_guard<std::mutex> lock(TableLock());
    auto& table = Table();
    auto found = table.find(ptr);
    if (found != table.end())
    {
      if (found->second.bytes >= bytes)
      {
        found->second.users++;
        shared_ = ptr;
      }
      return;  // (a larger request on a locked range: the copy takes whatever path the runtime picks)
    }
    hipPointerAttribute_t attr{};
    if (hipPointerGetAttributes(&attr, ptr) == hipSuccess && attr.type == hipMemoryTypeHost) return;  // the caller's pin
    (void)hipGetLastError();
    if (hipHostRegister(const_cast<void*>(ptr), bytes, hipHostRegisterDefault) == hipSuccess)
    {
      table[ptr] = Entry{bytes, 1};
      shared_ = ptr;
    }
    else
      (void)hipGetLastError();
  }
  ~ScopedHostPin()
  {
    if (!shared_) return;
    std::lock_guard<std::mutex> lock(TableLock());
    auto& table = Table();
    auto found = table.find(shared_);
    if (found != table.end() && --found->second.users == 0)
    {
      (void)hipHostUnregister(const_cast<void*>(shared_));
      table.erase(found);
    }
  }
  ScopedHostPin(const ScopedHostPin&) = delete;
  ScopedHostPin& operator=(const ScopedHostPin&) = delete;

private:
  struct Entry
  {
    size_t bytes;
    int users;
  };
  static std::mutex& TableLock()
  {
    static std::mutex lock;
    return lock;
  }
  static std::map<const void*, Entry>& Table()
  {
    static auto* table = new std::map<const void*, Entry>();
    return *table;
  }
  const void* shared_ = nullptr;
};

struct SdfWorkspace
{
  vgt::ClassRecord* records;  // pass-1 result of the default pipeline ...
  int16_t* t16;               // ... or of the cross-check pipelines (one of the two, the other is null)
  int32_t* t32;
  uint32_t* minmax_enc;
  vgt::SweepScratch sweep_scratch;  // work counters, spilled stack entries and sign words of the line passes
  size_t bytes;
};

// (batch > 1: `batch` grids of nx x ny x nz one after the other -- records and the intermediate field of batch * nx
// slices, a pair of extrema per grid, the line passes' scratch for the batch's items; default pipeline only)
SdfWorkspace CarveWorkspace(void* base, int64_t nx, int64_t ny, int64_t nz, vgt::EdtVariant variant, int64_t batch = 1)
{
  SdfWorkspace ws;
  const size_t n = static_cast<size_t>(batch * nx * ny * nz);
  size_t off = 0;
  ws.records = nullptr;
  ws.t16 = nullptr;
  if (variant == vgt::EdtVariant::kDefault)
  {
    ws.records = reinterpret_cast<vgt::ClassRecord*>(static_cast<char*>(base) + off);
    off = AlignUp(off + vgt::ClassRecordBytes(batch * nx, ny, nz), 256);
  }
  else
  {
    ws.t16 = reinterpret_cast<int16_t*>(static_cast<char*>(base) + off);
    off = AlignUp(off + n * sizeof(int16_t), 256);
  }
  ws.t32 = reinterpret_cast<int32_t*>(static_cast<char*>(base) + off);
  off = AlignUp(off + n * sizeof(int32_t), 256);
  ws.minmax_enc = reinterpret_cast<uint32_t*>(static_cast<char*>(base) + off);
  off += AlignUp(static_cast<size_t>(batch) * 2 * sizeof(uint32_t), 256);
  ws.sweep_scratch.ptr = static_cast<char*>(base) + off;
  ws.sweep_scratch.bytes = vgt::SweepPassScratchBytes(nx, ny, nz, batch);
  off = AlignUp(off + ws.sweep_scratch.bytes, 256);
  ws.bytes = off;
  return ws;
}

// Pass 1 and the Y pass of `slices` X slices that start `first_slice` slices into the grid (the whole grid: 0, p.nx;
// `p` describes the part: p.nx = slices).  `summary`: per-line slab summaries of the part (multi-GPU) or null.
template <typename InT>
hipError_t LaunchPassOne(const InT* input_dev, const SdfWorkspace& ws, const vgt::SdfParams& part, int64_t first_slice,
                         vgt::SlabLineSummary* summary, hipStream_t s)
{
  const int64_t voxel_offset = first_slice * part.ny * part.nz;
  if (ws.records)
  {
    vgt::ClassRecord* records = ws.records + first_slice * vgt::RecordWords(part.nz) * part.ny;
    if constexpr (std::is_same<InT, float>::value)
      return vgt::LaunchClassRecordsFromOccupancy(input_dev + voxel_offset, records, part, summary, s);
    else
      return vgt::LaunchClassRecordsFromMask(input_dev + voxel_offset, records, part, summary, s);
  }
#ifdef VGT_HIP_TESTING
  if constexpr (std::is_same<InT, float>::value)
    return vgt::LaunchScanZFromOccupancy(input_dev + voxel_offset, ws.t16 + voxel_offset, part, summary, s);
  else
    return vgt::LaunchScanZFromMask(input_dev + voxel_offset, ws.t16 + voxel_offset, part, summary, s);
#else
  return hipErrorInvalidValue;  // (no records: a cross-check variant, not part of this build)
#endif
}
hipError_t LaunchPassTwo(const SdfWorkspace& ws, const vgt::SdfParams& part, int64_t first_slice, vgt::EdtVariant variant,
                         hipStream_t s)
{
  const int64_t voxel_offset = first_slice * part.ny * part.nz;
  if (ws.records)
    return vgt::LaunchPassYSweepRecords(ws.records + first_slice * vgt::RecordWords(part.nz) * part.ny,
                                        ws.t32 + voxel_offset, ws.sweep_scratch, part, s);
#ifdef VGT_HIP_TESTING
  return vgt::LaunchPassY(ws.t16 + voxel_offset, ws.t32 + voxel_offset, ws.sweep_scratch, part, variant, s);
#else
  (void)variant;
  return hipErrorInvalidValue;
#endif
}

int CheckSdfShape(int64_t nx, int64_t ny, int64_t nz, double resolution)
{
  if (nx <= 0 || ny <= 0 || nz <= 0)
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "grid extents must be positive");
  if (nx > vgt::kMaxExtent || ny > vgt::kMaxExtent || nz > vgt::kMaxExtent)
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "grid extent exceeds 16384 voxels on an axis");
  if (!(resolution > 0.0) || !std::isfinite(resolution))
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "Grid must have uniform, positive resolution");
  return VGT_HIP_OK;
}

// Events of the current timing slot (nullptr when no session is active or it is full).
hipEvent_t* TimingSlot(vgt_hip_ctx* ctx)
{
  if (ctx->timing_slots == 0 || ctx->timing_used >= ctx->timing_slots) return nullptr;
  return &ctx->timing_events[static_cast<size_t>(ctx->timing_used) * 8];
}

// Enqueues the three passes.  events (optional) = 4 recorded events bracketing the kernels.
template <typename InT>
int RunSdfPipeline(vgt_hip_ctx* ctx, const InT* input_dev, const vgt::SdfParams& p, float* sdf_dev,
                   void* workspace_dev, size_t workspace_bytes, float* minmax_dev,
                   hipEvent_t* events)
{
  if (p.batch > 1 && ctx->variant != vgt::EdtVariant::kDefault)
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "batches run on the default EDT pipeline only");
  const SdfWorkspace ws = CarveWorkspace(workspace_dev, p.nx, p.ny, p.nz, ctx->variant, p.batch);
  if (workspace_dev == nullptr || workspace_bytes < ws.bytes)
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "SDF workspace too small");
  hipStream_t s = ctx->stream;
  // A batch of grids (one after the other in every buffer) is ONE grid of batch * nx slices to pass 1 and to the Y
  // pass -- their lines never leave a slice -- and p.batch grids to the X pass, whose lines run along x.
  vgt::SdfParams slices = p;
  slices.nx = p.nx * p.batch;
  slices.batch = 1;
  VGT_TRY_HIP(vgt::LaunchInitMinMax(ws.minmax_enc, s, p.batch), "init min/max");
  if (events) VGT_TRY_HIP(hipEventRecord(events[0], s), "event record");
  VGT_TRY_HIP(LaunchPassOne<InT>(input_dev, ws, slices, 0, nullptr, s), "pass 1");
  if (events) VGT_TRY_HIP(hipEventRecord(events[1], s), "event record");
  VGT_TRY_HIP(LaunchPassTwo(ws, slices, 0, ctx->variant, s), "Y pass");
  if (events) VGT_TRY_HIP(hipEventRecord(events[2], s), "event record");
  VGT_TRY_HIP(vgt::LaunchPassXFinalize(ws.t32, sdf_dev, ws.minmax_enc, ws.sweep_scratch, p, ctx->variant, s), "X pass");
  if (events) VGT_TRY_HIP(hipEventRecord(events[3], s), "event record");
  if (minmax_dev) VGT_TRY_HIP(vgt::LaunchDecodeMinMax(ws.minmax_enc, minmax_dev, s, p.batch), "min/max");
  return VGT_HIP_OK;
}

// Limits of a batch: the stacked grid's lines and voxels must stay inside what pass 1 and the sweeps index with.
int CheckBatch(int64_t batch, int64_t nx, int64_t ny, int64_t nz)
{
  if (batch <= 0) return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "batch must be positive");
  const int64_t zsegs = (nz + 63) / 64;
  if (batch > (int64_t{1} << 20) || batch * nx * ny >= (int64_t{1} << 28) || batch * nx * zsegs > 0x7fffffffLL ||
      batch * ny * zsegs > 0x7fffffffLL)
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "batch too large: batch * nx * ny must stay below 2^28 lines");
  return VGT_HIP_OK;
}

// Many device fields into many host arrays the caller has just allocated (the per-object fields of a tagged map, the
// fields of a batch of maps).  Page-locking such arrays costs more than the copy: their pages have never been touched,
// so the registration faults every one of them in, on one thread, and at the end they are unlocked again (32 fields of
// 128^3: 21 ms for 268 MB, a third of the link).  Instead the fields cross the link into a ring of page-locked slots
// that the context keeps, and a few host threads copy every slot's content on into the caller's arrays -- touching the
// arrays' pages in parallel -- while the next slot fills.  `fields`: device pointer and host pointer per array, `bytes`
// each.  The caller holds ctx->mutex; `s` has everything the fields depend on enqueued.
constexpr size_t kStagingSlotBytes = size_t{8} << 20;
constexpr int kStagingSlots = 4;
struct HostArrayCopy
{
  const void* device;
  void* host;
};
// The context's page-locked ring exists (allocated by the first call that wants it).
hipError_t EnsureStaging(vgt_hip_ctx* ctx)
{
  if (ctx->host_staging) return hipSuccess;
  const hipError_t err = hipHostMalloc(&ctx->host_staging, kStagingSlots * kStagingSlotBytes, hipHostMallocDefault);
  if (err != hipSuccess)
  {
    ctx->host_staging = nullptr;
    (void)hipGetLastError();
  }
  return err;
}
// One trip of the ring: a slot's worth of bytes that is either part of ONE array (arrays larger than a slot) or SEVERAL
// whole arrays whose device sides lie one behind the other (small arrays: one DMA and one hand-shake per slot instead of
// one per array -- 64 maps of 64^3 are 8 trips, not 64).
struct RingPiece
{
  size_t first, count;  // arrays [first, first + count)
  size_t offset;        // into the array (count == 1)
  size_t bytes;         // of the whole piece
};
std::vector<RingPiece> RingPieces(const std::vector<HostArrayCopy>& arrays, size_t bytes)
{
  std::vector<RingPiece> pieces;
  if (bytes >= kStagingSlotBytes)
  {
    for (size_t a = 0; a < arrays.size(); a++)
      for (size_t off = 0; off < bytes; off += kStagingSlotBytes)
        pieces.push_back(RingPiece{a, 1, off, std::min(kStagingSlotBytes, bytes - off)});
    return pieces;
  }
  const size_t per_slot = kStagingSlotBytes / bytes;
  for (size_t a = 0; a < arrays.size();)
  {
    size_t count = 1;
    while (count < per_slot && a + count < arrays.size() &&
           static_cast<const char*>(arrays[a + count].device) == static_cast<const char*>(arrays[a].device) + count * bytes)
      count++;
    pieces.push_back(RingPiece{a, count, 0, count * bytes});
    a += count;
  }
  return pieces;
}
// bytes [begin, end) of a piece between its slot and the caller's arrays
void CopyPieceRange(const std::vector<HostArrayCopy>& arrays, size_t bytes, const RingPiece& p, char* slot, size_t begin,
                    size_t end, bool to_host)
{
  while (begin < end)
  {
    const size_t k = p.count == 1 ? 0 : begin / bytes;              // array of the piece
    const size_t in_array = p.count == 1 ? p.offset + begin : begin - k * bytes;
    const size_t array_end = p.count == 1 ? end : std::min(end, (k + 1) * bytes);
    char* const host = static_cast<char*>(arrays[p.first + k].host) + in_array;
    if (to_host)
      std::memcpy(host, slot + begin, array_end - begin);
    else
      std::memcpy(slot + begin, host, array_end - begin);
    begin = array_end;
  }
}

// direction: true = device -> host arrays (the fields of a batch), false = host arrays -> device (its inputs)
hipError_t MoveThroughRing(vgt_hip_ctx* ctx, const std::vector<HostArrayCopy>& arrays, size_t bytes, hipStream_t s, bool to_host)
{
  if (arrays.empty() || bytes == 0) return hipSuccess;
  {
    const hipError_t err = EnsureStaging(ctx);
    if (err != hipSuccess) return err;
  }
  while (static_cast<int>(ctx->staging_events.size()) < kStagingSlots)
  {
    hipEvent_t e = nullptr;
    const hipError_t err = hipEventCreateWithFlags(&e, hipEventDisableTiming);
    if (err != hipSuccess) return err;
    ctx->staging_events.push_back(e);
  }
  const std::vector<RingPiece> pieces = RingPieces(arrays, bytes);
  const unsigned cap = to_host ? 16u : 8u;
  const int workers = static_cast<int>(std::max(1u, std::min(cap, std::thread::hardware_concurrency() / 2)));
#ifdef __linux__
  // (fresh arrays of several MiB: ask for huge pages where the kernel gives them on request -- 2 MiB per fault instead of
  // 4 KiB; a hint, any answer is fine)
  if (to_host && bytes >= (size_t{4} << 20))
    for (const HostArrayCopy& f : arrays)
    {
      const uintptr_t begin = (reinterpret_cast<uintptr_t>(f.host) + 4095) & ~uintptr_t{4095};
      const uintptr_t end = (reinterpret_cast<uintptr_t>(f.host) + bytes) & ~uintptr_t{4095};
      if (end > begin) (void)madvise(reinterpret_cast<void*>(begin), end - begin, MADV_HUGEPAGE);
    }
#endif
  // Download: dma[i] = the copy of piece i into its slot is ENQUEUED (its event recorded); host[i] = workers done with it.
  // Upload: host[i] = workers have filled the slot; dma[i] = its copy to the device is enqueued (event recorded).
  std::vector<std::atomic<int>> dma(pieces.size()), host(pieces.size());
  for (auto& a : dma) a.store(0);
  for (auto& c : host) c.store(0);
  std::atomic<int> failed{0};
  char* const staging = static_cast<char*>(ctx->host_staging);
  const int device = ctx->device;
  std::vector<std::thread> pool;
  for (int w = 0; w < workers; w++)
    pool.emplace_back([&, w]() {
      (void)hipSetDevice(device);
      for (size_t i = 0; i < pieces.size(); i++)
      {
        const int slot = static_cast<int>(i % kStagingSlots);
        // the DMA this piece's slot has to wait for: its own (download) or the one that emptied the slot (upload)
        const bool wait_for_dma = to_host || i >= static_cast<size_t>(kStagingSlots);
        if (wait_for_dma)
        {
          const size_t which = to_host ? i : i - kStagingSlots;
          while (dma[which].load(std::memory_order_acquire) == 0)
          {
            if (failed.load()) return;
            std::this_thread::yield();
          }
          if (hipEventSynchronize(ctx->staging_events[static_cast<size_t>(slot)]) != hipSuccess)
          {
            failed.store(1);
            return;
          }
        }
        const RingPiece& p = pieces[i];
        // (slices on page boundaries: two workers never fault the same page of a fresh array)
        const size_t pages = (p.bytes + 4095) / 4096;
        const size_t begin = std::min(p.bytes, pages * static_cast<size_t>(w) / workers * 4096);
        const size_t end = std::min(p.bytes, pages * static_cast<size_t>(w + 1) / workers * 4096);
        if (end > begin) CopyPieceRange(arrays, bytes, p, staging + static_cast<size_t>(slot) * kStagingSlotBytes, begin, end, to_host);
        host[i].fetch_add(1, std::memory_order_release);
      }
    });
  hipError_t err = hipSuccess;
  for (size_t i = 0; i < pieces.size() && err == hipSuccess; i++)
  {
    const int slot = static_cast<int>(i % kStagingSlots);
    // the host side this DMA has to wait for: the workers that emptied the slot (download) or filled it (upload)
    if (to_host ? i >= static_cast<size_t>(kStagingSlots) : true)
    {
      const size_t which = to_host ? i - kStagingSlots : i;
      while (host[which].load(std::memory_order_acquire) < workers && !failed.load()) std::this_thread::yield();
    }
    if (failed.load()) break;
    const RingPiece& p = pieces[i];
    char* const in_slot = staging + static_cast<size_t>(slot) * kStagingSlotBytes;
    char* const on_device = static_cast<char*>(const_cast<void*>(arrays[p.first].device)) + p.offset;
    err = to_host ? hipMemcpyAsync(in_slot, on_device, p.bytes, hipMemcpyDeviceToHost, s)
                  : hipMemcpyAsync(on_device, in_slot, p.bytes, hipMemcpyHostToDevice, s);
    if (err == hipSuccess) err = hipEventRecord(ctx->staging_events[static_cast<size_t>(slot)], s);
    if (err == hipSuccess) dma[i].store(1, std::memory_order_release);
  }
  if (err != hipSuccess) failed.store(1);
  for (auto& th : pool) th.join();
  if (to_host)
  {
    const hipError_t sync = hipStreamSynchronize(s);
    if (err == hipSuccess) err = sync;
  }
  if (err == hipSuccess && failed.load()) err = hipErrorUnknown;
  return err;
}
// Many device fields into many host arrays (synchronises `s`) / many host arrays that hold data into device buffers (at
// return every copy is ENQUEUED on `s`; the ring is only reused by work enqueued on `s` later).  Page-locking a 1 MiB
// array for one copy costs ~0.15 ms (register + unregister) where the copy itself takes 0.02 -- 64 maps of 64^3 spent
// 10 ms there.
hipError_t DownloadToHostArrays(vgt_hip_ctx* ctx, const std::vector<HostArrayCopy>& fields, size_t bytes, hipStream_t s)
{
  return MoveThroughRing(ctx, fields, bytes, s, true);
}
hipError_t UploadFromHostArrays(vgt_hip_ctx* ctx, const std::vector<HostArrayCopy>& arrays, size_t bytes, hipStream_t s)
{
  return MoveThroughRing(ctx, arrays, bytes, s, false);
}

// The largest group of a batch that one launch can take: CheckBatch's limits (2^20 grids, 2^28 lines, 32-bit item counts)
// and `device_bytes` of buffers at `bytes_per_grid` each.  At least 1.
int64_t BatchGroup(int64_t batch, int64_t nx, int64_t ny, int64_t nz, size_t device_bytes, size_t bytes_per_grid)
{
  const int64_t zsegs = (nz + 63) / 64;
  int64_t group = batch;
  const int64_t limits[] = {int64_t{1} << 20, ((int64_t{1} << 28) - 1) / (nx * ny), 0x7fffffffLL / (nx * zsegs),
                            0x7fffffffLL / (ny * zsegs), static_cast<int64_t>(device_bytes / (bytes_per_grid + 1))};
  for (const int64_t limit : limits)
    if (group > limit) group = limit;
  return group < 1 ? 1 : group;
}

// Large grids through the host-pointer entry points: the three phases of a call -- upload, kernels, download --
// overlap.  The grid is uploaded in X chunks (contiguous) on a copy stream and every chunk is scanned along Z and
// swept along Y as soon as it has arrived; the X pass then runs over ranges of Y, and every finished range
// (nx pieces of ny_range * nz floats) goes back on a second copy stream while the next range is computed.  Only
// with the tiled line passes (the others need whole axes per launch).  The caller holds ctx->mutex.
constexpr int kPipelineChunks = 8;
std::atomic<int64_t> g_host_pipeline_min_voxels{int64_t{1} << 27};

template <typename InT>
bool CanPipelineFromHost(const vgt_hip_ctx* ctx, const vgt::SdfParams& p)
{
  // smallest grid that is pipelined: 2^27 voxels (testing builds: vgt_hip_testing_set_host_pipeline_min_voxels lowers
  // it so that small grids take this path, a negative value turns the pipeline off)
  const int64_t min_voxels = g_host_pipeline_min_voxels.load();
  if (min_voxels < 0 || !vgt::LinePassesTakeRanges(p, ctx->variant)) return false;
  return p.nx >= 4 * kPipelineChunks && p.ny >= 4 * kPipelineChunks && p.nx * p.ny * p.nz >= min_voxels;
}

template <typename InT>
int SdfFromHostPipelined(vgt_hip_ctx* ctx, const InT* input_host, InT* in_dev, const vgt::SdfParams& p,
                         float* sdf_dev, float* sdf_host, const std::function<void()>& before_downloads)
{
  const SdfWorkspace ws = CarveWorkspace(ctx->sdf_ws, p.nx, p.ny, p.nz, ctx->variant);
  if (ctx->sdf_ws_bytes < ws.bytes) return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "SDF workspace too small");
  if (!ctx->copy_in) VGT_TRY_HIP(hipStreamCreateWithFlags(&ctx->copy_in, hipStreamNonBlocking), "create stream");
  if (!ctx->copy_out) VGT_TRY_HIP(hipStreamCreateWithFlags(&ctx->copy_out, hipStreamNonBlocking), "create stream");
  while (ctx->pipeline_events.size() < static_cast<size_t>(2 * kPipelineChunks + 1))
  {
    hipEvent_t e = nullptr;
    VGT_TRY_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming), "create event");
    ctx->pipeline_events.push_back(e);
  }
  hipEvent_t* const uploaded = ctx->pipeline_events.data();
  hipEvent_t* const computed = uploaded + kPipelineChunks;
  hipEvent_t const start = ctx->pipeline_events[2 * kPipelineChunks];
  hipStream_t s = ctx->stream;
  // the copy streams begin after whatever the context's stream still has to do with these buffers
  VGT_TRY_HIP(hipEventRecord(start, s), "event record");
  VGT_TRY_HIP(hipStreamWaitEvent(ctx->copy_in, start, 0), "order the upload stream");
  VGT_TRY_HIP(hipStreamWaitEvent(ctx->copy_out, start, 0), "order the download stream");
  const int64_t plane = p.ny * p.nz;
  auto x_begin = [&](int c) { return p.nx * c / kPipelineChunks; };
  auto y_begin = [&](int c) { return p.ny * c / kPipelineChunks; };
  for (int c = 0; c < kPipelineChunks; c++)
  {
    const int64_t off = x_begin(c) * plane, count = (x_begin(c + 1) - x_begin(c)) * plane;
    VGT_TRY_HIP(hipMemcpyAsync(in_dev + off, input_host + off, static_cast<size_t>(count) * sizeof(InT),
                               hipMemcpyHostToDevice, ctx->copy_in),
                "copy occupancy to device");
    VGT_TRY_HIP(hipEventRecord(uploaded[c], ctx->copy_in), "event record");
  }
  VGT_TRY_HIP(vgt::LaunchInitMinMax(ws.minmax_enc, s), "init min/max");
  for (int c = 0; c < kPipelineChunks; c++)
  {
    vgt::SdfParams part = p;
    part.nx = x_begin(c + 1) - x_begin(c);
    VGT_TRY_HIP(hipStreamWaitEvent(s, uploaded[c], 0), "wait for a chunk of the upload");
    VGT_TRY_HIP(LaunchPassOne<InT>(in_dev, ws, part, x_begin(c), nullptr, s), "pass 1");
    VGT_TRY_HIP(LaunchPassTwo(ws, part, x_begin(c), ctx->variant, s), "Y pass");
  }
  // (everything up to here is enqueued and on its way: the moment to make the output array ready for the downloads)
  before_downloads();
  const size_t pitch = static_cast<size_t>(plane) * sizeof(float);
  for (int c = 0; c < kPipelineChunks; c++)
  {
    const int64_t y0 = y_begin(c), rows = y_begin(c + 1) - y0;
    VGT_TRY_HIP(vgt::LaunchPassXFinalizeRange(ws.t32, sdf_dev, ws.minmax_enc, ws.sweep_scratch, p, ctx->variant, y0, rows, s),
                "X pass");
    VGT_TRY_HIP(hipEventRecord(computed[c], s), "event record");
    VGT_TRY_HIP(hipStreamWaitEvent(ctx->copy_out, computed[c], 0), "wait for a range of the field");
    VGT_TRY_HIP(hipMemcpy2DAsync(sdf_host + y0 * p.nz, pitch, sdf_dev + y0 * p.nz, pitch,
                                 static_cast<size_t>(rows * p.nz) * sizeof(float), static_cast<size_t>(p.nx),
                                 hipMemcpyDeviceToHost, ctx->copy_out),
                "copy SDF to host");
  }
  VGT_TRY_HIP(vgt::LaunchDecodeMinMax(ws.minmax_enc, ctx->minmax_out, s), "min/max");
  return VGT_HIP_OK;
}

template <typename InT>
int SdfFromHost(vgt_hip_ctx* ctx, const InT* input_host, const vgt::SdfParams& p, float* sdf_host,
                float* out_min, float* out_max)
{
  if (!ctx || !input_host || !sdf_host)
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  const int rc = CheckSdfShape(p.nx, p.ny, p.nz, p.resolution);
  if (rc != VGT_HIP_OK) return rc;
  VGT_TRY_HIP(hipSetDevice(ctx->device), "set device");
  const size_t nvox = static_cast<size_t>(p.nx * p.ny * p.nz);
  const size_t ws_bytes = CarveWorkspace(nullptr, p.nx, p.ny, p.nz, ctx->variant).bytes;
  // The output is usually an array the caller has just allocated: its pages are faulted in by a few threads while the
  // input is page-locked and uploaded, and it is page-locked only when the downloads are about to be enqueued.
  vgt::HostRangePopulator fresh_out(sdf_host, nvox * sizeof(float));
  const ScopedHostPin pin_in(input_host, nvox * sizeof(InT));
  std::unique_ptr<ScopedHostPin> pin_out;
  const auto lock_output = [&]() {
    fresh_out.Wait();
    if (!pin_out) pin_out.reset(new ScopedHostPin(sdf_host, nvox * sizeof(float)));
  };
  std::lock_guard<std::mutex> lock(ctx->mutex);
  hipError_t err = Reserve(&ctx->sdf_in, &ctx->sdf_in_bytes, nvox * sizeof(InT));
  if (err == hipSuccess) err = Reserve(&ctx->sdf_out, &ctx->sdf_out_bytes, nvox * sizeof(float));
  if (err == hipSuccess) err = Reserve(&ctx->sdf_ws, &ctx->sdf_ws_bytes, ws_bytes);
  if (err != hipSuccess)
  {
    FreeCachedSdfBuffers(ctx);
    return FailHip("allocate SDF buffers", err);
  }
  InT* in_dev = static_cast<InT*>(ctx->sdf_in);
  float* sdf_dev = static_cast<float*>(ctx->sdf_out);
  hipStream_t s = ctx->stream;
  int result = VGT_HIP_OK;
  if (CanPipelineFromHost<InT>(ctx, p))
  {
    if (!ctx->minmax_out) return Fail(VGT_HIP_ERR_RUNTIME, "context has no extrema buffer");
    result = SdfFromHostPipelined<InT>(ctx, input_host, in_dev, p, sdf_dev, sdf_host, lock_output);
    float mm[2] = {0.0f, 0.0f};
    if (result == VGT_HIP_OK)
    {
      err = hipMemcpyAsync(mm, ctx->minmax_out, sizeof(mm), hipMemcpyDeviceToHost, s);
      if (err != hipSuccess) result = FailHip("copy extrema to host", err);
    }
    // drain all three streams whatever happened: the caller's arrays are about to be unpinned
    const hipError_t e1 = hipStreamSynchronize(ctx->copy_in);
    const hipError_t e2 = hipStreamSynchronize(s);
    const hipError_t e3 = hipStreamSynchronize(ctx->copy_out);
    if (result == VGT_HIP_OK && (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess))
      result = FailHip("pipelined SDF extraction", e1 != hipSuccess ? e1 : (e2 != hipSuccess ? e2 : e3));
    if (result == VGT_HIP_OK)
    {
      if (out_min) *out_min = mm[0];
      if (out_max) *out_max = mm[1];
    }
    return result;
  }
  // Small maps (the sizes of the reference's own examples and tests; up to 512 KiB) never cross the link as copies: the map
  // is copied into a page-locked slot of the context's ring and the kernels work ON the ring -- pass 1 reads the map from
  // host memory, the X pass writes the field to it, the extrema behind the field -- so a call is one memcpy in, five
  // dispatches, one wait, one memcpy out.  No DMA dispatch in either direction, no pageable copy that the runtime would
  // stage itself, no third copy for two floats: 40^3 0.115 ms per blocking call in round 5, 0.078 - 0.085 with a DMA each
  // way between the ring and the device buffers, 0.060 - 0.072 now; 16^3 0.066 / 0.048 / 0.037
  // (profiles/r6/experiments.md).  A kernel reads host memory slowly, so not beyond 512 KiB (1 MiB: 0.37 ms against
  // 0.13 - 0.36 with page-locked arrays, which is what larger maps get below).
  const size_t in_bytes = nvox * sizeof(InT), out_bytes = nvox * sizeof(float);
  if (in_bytes <= (size_t{1} << 19) && out_bytes + 2 * sizeof(float) <= kStagingSlotBytes && EnsureStaging(ctx) == hipSuccess)
  {
    char* const up = static_cast<char*>(ctx->host_staging);
    char* const down = up + kStagingSlotBytes;
    std::memcpy(up, input_host, in_bytes);
    float* const field = reinterpret_cast<float*>(down);
    result = RunSdfPipeline<InT>(ctx, reinterpret_cast<const InT*>(up), p, field, ctx->sdf_ws, ctx->sdf_ws_bytes,
                                 field + nvox, nullptr);
    err = hipStreamSynchronize(s);  // (whatever happened: the ring is reused by the next call)
    if (result == VGT_HIP_OK && err != hipSuccess) result = FailHip("small-map SDF extraction", err);
    if (result == VGT_HIP_OK)
    {
      std::memcpy(sdf_host, down, out_bytes);
      float mm[2];
      std::memcpy(mm, down + out_bytes, sizeof(mm));
      if (out_min) *out_min = mm[0];
      if (out_max) *out_max = mm[1];
    }
    return result;
  }
  err = hipMemcpyAsync(in_dev, input_host, nvox * sizeof(InT), hipMemcpyHostToDevice, s);
  if (err != hipSuccess)
    result = FailHip("copy occupancy to device", err);
  else
    result = RunSdfPipeline<InT>(ctx, in_dev, p, sdf_dev, ctx->sdf_ws, ctx->sdf_ws_bytes, ctx->minmax_out, nullptr);
  if (result == VGT_HIP_OK)
  {
    float mm[2] = {0.0f, 0.0f};
    lock_output();  // (upload and kernels are on their way)
    err = hipMemcpyAsync(sdf_host, sdf_dev, nvox * sizeof(float), hipMemcpyDeviceToHost, s);
    if (err == hipSuccess)
      err = hipMemcpyAsync(mm, ctx->minmax_out, sizeof(mm), hipMemcpyDeviceToHost, s);
    if (err == hipSuccess) err = hipStreamSynchronize(s);
    if (err != hipSuccess)
      result = FailHip("copy SDF to host", err);
    else
    {
      if (out_min) *out_min = mm[0];
      if (out_max) *out_max = mm[1];
    }
  }
  else
    (void)hipStreamSynchronize(s);
  return result;
}

int CheckRaycastArgs(const vgt_hip_ctx* ctx, const vgt_hip_grids* grids, size_t grid_index,
                     const void* points, int64_t num_points, const void* xform, int32_t nx,
                     int32_t ny, int32_t nz)
{
  if (!ctx || !grids || !xform) return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  if (grids->ctx != ctx)
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "tracking grids belong to another context");
  if (grid_index >= static_cast<size_t>(grids->num_grids))
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "tracking grid index out of range");
  if (num_points < 0 || (num_points > 0 && !points))
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "invalid point buffer");
  if (nx <= 0 || ny <= 0 || nz <= 0 ||
      static_cast<int64_t>(nx) * ny * nz != grids->num_cells)
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "voxel counts do not match the tracking grids");
  return VGT_HIP_OK;
}

// Releases the upload lanes' device resources (the caller has made sure nothing runs on them).
void FreeUploadLanes(vgt_hip_ctx* ctx, bool destroy_streams)
{
  for (UploadLane& lane : ctx->lanes)
  {
    std::lock_guard<std::mutex> lock(lane.mutex);
    if (lane.stream) (void)hipStreamSynchronize(lane.stream);
    if (lane.stage) (void)hipFree(lane.stage);
    lane.stage = nullptr;
    lane.stage_bytes = 0;
    if (lane.pinned) (void)hipHostFree(lane.pinned);
    lane.pinned = nullptr;
    lane.pinned_bytes = 0;
    if (destroy_streams)
    {
      if (lane.after_ctx) (void)hipEventDestroy(lane.after_ctx);
      if (lane.stream) (void)hipStreamDestroy(lane.stream);
      lane.after_ctx = nullptr;
      lane.stream = nullptr;
    }
  }
}

// Uploads `bytes` from host memory on a free upload lane, runs `launch(device_copy, lane_stream)` behind the
// copy and waits for it: the host buffer may be released by the caller as soon as this returns
// (S/cuda_voxelization_helpers.cu:676-699 frees its device copy at scope exit).  Calls on different lanes
// overlap; work queued on the context's stream before the call (the zeroing of new tracking grids) is
// ordered before the lane's work by an event, work queued after it follows the return of this call.
template <typename Launch>
int UploadAndRun(vgt_hip_ctx* ctx, const void* host, size_t bytes, size_t scratch_bytes, Launch launch)
{
  // the first idle lane (a caller that raycasts one cloud after the other keeps using lane 0, whose stream and staging
  // buffer exist; concurrent callers spread over the lanes), or, when all are busy, the next one in turn
  UploadLane* chosen = nullptr;
  std::unique_lock<std::mutex> lane_lock;
  for (int i = 0; i < kUploadLanes && !chosen; i++)
  {
    std::unique_lock<std::mutex> attempt(ctx->lanes[i].mutex, std::try_to_lock);
    if (attempt.owns_lock())
    {
      chosen = &ctx->lanes[i];
      lane_lock = std::move(attempt);
    }
  }
  if (!chosen)
  {
    chosen = &ctx->lanes[ctx->next_lane.fetch_add(1) % kUploadLanes];
    lane_lock = std::unique_lock<std::mutex>(chosen->mutex);
  }
  UploadLane& lane = *chosen;
  if (!lane.stream)
  {
    VGT_TRY_HIP(hipStreamCreateWithFlags(&lane.stream, hipStreamNonBlocking), "create upload stream");
    VGT_TRY_HIP(hipEventCreateWithFlags(&lane.after_ctx, hipEventDisableTiming), "create upload event");
  }
  // staging copy of the cloud, followed by the kernel's scratch (256-byte aligned)
  const size_t scratch_at = AlignUp(bytes, 256);
  if (lane.stage_bytes < scratch_at + scratch_bytes)
  {
    VGT_TRY_HIP(hipStreamSynchronize(lane.stream), "drain before regrowing staging buffer");
    if (lane.stage) VGT_TRY_HIP(hipFree(lane.stage), "free staging buffer");
    lane.stage = nullptr;
    lane.stage_bytes = 0;
    const size_t total = scratch_at + scratch_bytes;
    const size_t want = AlignUp(total + total / 4, 1 << 20);
    VGT_TRY_HIP(hipMalloc(&lane.stage, want), "allocate staging buffer");
    lane.stage_bytes = want;
  }
  {
    std::lock_guard<std::mutex> lock(ctx->mutex);
    VGT_TRY_HIP(hipEventRecord(lane.after_ctx, ctx->stream), "record event");
  }
  VGT_TRY_HIP(hipStreamWaitEvent(lane.stream, lane.after_ctx, 0), "order upload lane");
  const void* source = host;
  {
    hipPointerAttribute_t attr{};
    const bool caller_locked = hipPointerGetAttributes(&attr, host) == hipSuccess && attr.type == hipMemoryTypeHost;
    (void)hipGetLastError();
    if (!caller_locked && bytes >= (size_t{1} << 16))
    {
      if (lane.pinned_bytes < bytes)
      {
        if (lane.pinned) (void)hipHostFree(lane.pinned);
        lane.pinned = nullptr;
        lane.pinned_bytes = 0;
        const size_t want = AlignUp(bytes + bytes / 4, 1 << 20);
        if (hipHostMalloc(&lane.pinned, want, hipHostMallocDefault) == hipSuccess)
          lane.pinned_bytes = want;
        else
          (void)hipGetLastError();  // (no page-locked memory to be had: the copy takes the pageable path)
      }
      if (lane.pinned_bytes >= bytes)
      {
        // (the previous use of the buffer was waited for at the end of its call)  Piece by piece, so that the DMA of one
        // piece runs while the host copies the next: a 12 MB cloud is 0.4 ms of memcpy and 0.2 ms of DMA -- 0.45 ms this
        // way instead of 0.6.
        constexpr size_t kPiece = size_t{2} << 20;
        for (size_t off = 0; off < bytes; off += kPiece)
        {
          const size_t piece = std::min(kPiece, bytes - off);
          std::memcpy(static_cast<char*>(lane.pinned) + off, static_cast<const char*>(host) + off, piece);
          VGT_TRY_HIP(hipMemcpyAsync(static_cast<char*>(lane.stage) + off, static_cast<char*>(lane.pinned) + off, piece,
                                     hipMemcpyHostToDevice, lane.stream),
                      "Failed to copy points to the device");
        }
        source = nullptr;  // (uploaded)
      }
    }
  }
  if (source)
    VGT_TRY_HIP(hipMemcpyAsync(lane.stage, source, bytes, hipMemcpyHostToDevice, lane.stream),
                "Failed to copy points to the device");
  VGT_TRY_HIP(launch(lane.stage, static_cast<char*>(lane.stage) + scratch_at, lane.stream),
              "Failed to dispatch raycast kernel");
  VGT_TRY_HIP(hipStreamSynchronize(lane.stream), "raycast");
  return VGT_HIP_OK;
}
// Shared body of the batched SDF queries: `run(sdf_dev, queries_dev, out_dev, has_dev, flag_dev)` enqueues the
// kernel.  Host variant: uploads field and queries, downloads the results.  out_doubles = doubles per query.
template <typename Run>
int SdfQueriesHost(vgt_hip_ctx* ctx, const float* sdf_host, int64_t nx, int64_t ny, int64_t nz, double resolution,
                   const double* query_xyz_host, int64_t num_queries, int out_doubles, double* out_host,
                   uint8_t* has_value_host, const char* what, Run run)
{
  if (!ctx || !sdf_host || !out_host || num_queries < 0 || (num_queries > 0 && !query_xyz_host))
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  const int rc = CheckSdfShape(nx, ny, nz, resolution);
  if (rc != VGT_HIP_OK) return rc;
  if (num_queries == 0) return VGT_HIP_OK;
  VGT_TRY_HIP(hipSetDevice(ctx->device), "set device");
  const size_t n = static_cast<size_t>(nx * ny * nz), q = static_cast<size_t>(num_queries);
  float* sdf_dev = nullptr;
  double* queries_dev = nullptr;
  double* out_dev = nullptr;
  uint8_t* has_dev = nullptr;
  uint32_t* flag_dev = nullptr;
  uint32_t flag = 0;
  hipError_t err = hipMalloc(reinterpret_cast<void**>(&sdf_dev), n * sizeof(float));
  if (err == hipSuccess) err = hipMalloc(reinterpret_cast<void**>(&queries_dev), q * 3 * sizeof(double));
  if (err == hipSuccess) err = hipMalloc(reinterpret_cast<void**>(&out_dev), q * out_doubles * sizeof(double));
  if (err == hipSuccess) err = hipMalloc(reinterpret_cast<void**>(&has_dev), q);
  if (err == hipSuccess) err = hipMalloc(reinterpret_cast<void**>(&flag_dev), 256);
  if (err == hipSuccess)
  {
    std::lock_guard<std::mutex> lock(ctx->mutex);
    hipStream_t s = ctx->stream;
    err = hipMemcpyAsync(sdf_dev, sdf_host, n * sizeof(float), hipMemcpyHostToDevice, s);
    if (err == hipSuccess)
      err = hipMemcpyAsync(queries_dev, query_xyz_host, q * 3 * sizeof(double), hipMemcpyHostToDevice, s);
    if (err == hipSuccess) err = hipMemsetAsync(flag_dev, 0, sizeof(uint32_t), s);
    if (err == hipSuccess) err = run(sdf_dev, queries_dev, out_dev, has_dev, flag_dev, s);
    if (err == hipSuccess)
      err = hipMemcpyAsync(out_host, out_dev, q * out_doubles * sizeof(double), hipMemcpyDeviceToHost, s);
    if (err == hipSuccess && has_value_host) err = hipMemcpyAsync(has_value_host, has_dev, q, hipMemcpyDeviceToHost, s);
    if (err == hipSuccess) err = hipMemcpyAsync(&flag, flag_dev, sizeof(flag), hipMemcpyDeviceToHost, s);
    const hipError_t sync = hipStreamSynchronize(s);
    if (err == hipSuccess) err = sync;
  }
  for (void* p : {static_cast<void*>(sdf_dev), static_cast<void*>(queries_dev), static_cast<void*>(out_dev),
                  static_cast<void*>(has_dev), static_cast<void*>(flag_dev)})
    if (p) (void)hipFree(p);
  VGT_TRY_HIP(err, what);
  if (flag) return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "Window size for fine gradient is too large for SDF");
  return VGT_HIP_OK;
}
}  // namespace

extern "C" {

int vgt_hip_abi_version(void) { return VGT_HIP_ABI_VERSION; }

const char* vgt_hip_last_error(void) { return g_last_error.c_str(); }

int vgt_hip_device_count(int* count)
{
  if (!count) return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  *count = 0;
  int n = 0;
  const hipError_t err = hipGetDeviceCount(&n);
  if (err != hipSuccess) return FailHip("Failed to get device count", err);
  *count = n;
  return VGT_HIP_OK;
}

int vgt_hip_device_name(int device, char* buffer, size_t buffer_size)
{
  if (!buffer || buffer_size == 0) return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  hipDeviceProp_t prop;
  std::memset(&prop, 0, sizeof(prop));
  VGT_TRY_HIP(hipGetDeviceProperties(&prop, device), "Failed to get device properties");
  std::snprintf(buffer, buffer_size, "%s", prop.name);
  return VGT_HIP_OK;
}

int vgt_hip_create(int device, int threads_per_block, vgt_hip_ctx** out_ctx)
{
  if (!out_ctx) return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  *out_ctx = nullptr;
  int count = 0;
  const hipError_t cerr = hipGetDeviceCount(&count);
  if (cerr != hipSuccess || count <= 0)
  {
    g_last_error = "no usable HIP device (libvgt_hip has no CPU fallback)";
    return VGT_HIP_ERR_UNAVAILABLE;
  }
  if (device < 0 || device >= count)
  {
    g_last_error = "HIP_DEVICE = " + std::to_string(device) + " out of range for " +
                   std::to_string(count) + " devices";
    return VGT_HIP_ERR_UNAVAILABLE;
  }
  const int raycast_threads = threads_per_block > 0 ? threads_per_block : 0;
  if (threads_per_block <= 0) threads_per_block = 256;
  if (threads_per_block > 1024 || (threads_per_block % 64) != 0)
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT,
                "threads_per_block must be a multiple of 64 and at most 1024");
  VGT_TRY_HIP(hipSetDevice(device), "Failed to set device");
  vgt_hip_ctx* ctx = new (std::nothrow) vgt_hip_ctx();
  if (!ctx) return Fail(VGT_HIP_ERR_RUNTIME, "out of host memory");
  ctx->device = device;
  ctx->threads_per_block = threads_per_block;
  ctx->raycast_threads = raycast_threads;
  hipError_t err = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
  if (err == hipSuccess) err = hipMalloc(reinterpret_cast<void**>(&ctx->minmax_out), 256);
  if (err != hipSuccess)
  {
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return FailHip("create context", err);
  }
  ctx->stream = ctx->own_stream;
  *out_ctx = ctx;
  return VGT_HIP_OK;
}

void vgt_hip_destroy(vgt_hip_ctx* ctx)
{
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  FreeUploadLanes(ctx, true);
  if (ctx->minmax_out) (void)hipFree(ctx->minmax_out);
  FreeCachedSdfBuffers(ctx);
  {
    std::lock_guard<std::mutex> lock(ctx->mutex);
    ctx->pool_closed = true;
    FreePool(ctx);
  }
  for (hipEvent_t e : ctx->timing_events)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : ctx->pipeline_events)
    if (e) (void)hipEventDestroy(e);
  ctx->pipeline_events.clear();
  if (ctx->copy_in) (void)hipStreamDestroy(ctx->copy_in);
  if (ctx->copy_out) (void)hipStreamDestroy(ctx->copy_out);
  ctx->copy_in = nullptr;
  ctx->copy_out = nullptr;
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  ctx->minmax_out = nullptr;
  ctx->timing_events.clear();
  ctx->own_stream = nullptr;
  ctx->stream = nullptr;
  ctx->destroyed.store(true);
  if (ctx->children.load() == 0) delete ctx;  // otherwise the last handle's destroy frees it
}

int vgt_hip_trim(vgt_hip_ctx* ctx)
{
  if (!ctx) return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  {
    std::lock_guard<std::mutex> lock(ctx->mutex);
    VGT_TRY_HIP(hipSetDevice(ctx->device), "set device");
    VGT_TRY_HIP(hipStreamSynchronize(ctx->stream), "drain stream");
    FreeCachedSdfBuffers(ctx);
    FreePool(ctx);
  }
  // Lock order: an upload lane's mutex is taken BEFORE the context's (UploadAndRun holds its lane while it
  // records an event under ctx->mutex), so the lanes are released after ctx->mutex has been dropped; a raycast that
  // is in flight on a lane simply finishes first (FreeUploadLanes takes each lane's mutex and drains its stream).
  FreeUploadLanes(ctx, false);
  return VGT_HIP_OK;
}

int vgt_hip_set_stream(vgt_hip_ctx* ctx, void* hip_stream)
{
  if (!ctx) return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  std::lock_guard<std::mutex> lock(ctx->mutex);
  VGT_TRY_HIP(hipSetDevice(ctx->device), "set device");
  VGT_TRY_HIP(hipStreamSynchronize(ctx->stream), "drain stream");
  ctx->stream = static_cast<hipStream_t>(hip_stream);  // NULL = HIP's legacy default stream
  return VGT_HIP_OK;
}

int vgt_hip_reset_stream(vgt_hip_ctx* ctx)
{
  if (!ctx) return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  std::lock_guard<std::mutex> lock(ctx->mutex);
  VGT_TRY_HIP(hipSetDevice(ctx->device), "set device");
  VGT_TRY_HIP(hipStreamSynchronize(ctx->stream), "drain stream");
  ctx->stream = ctx->own_stream;
  return VGT_HIP_OK;
}

int vgt_hip_synchronize(vgt_hip_ctx* ctx)
{
  if (!ctx) return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  VGT_TRY_HIP(hipSetDevice(ctx->device), "set device");
  VGT_TRY_HIP(hipStreamSynchronize(ctx->stream), "synchronize");
  {
    // deferred filter-grid uploads run on the copy stream: "all work of the context has finished" includes them
    std::lock_guard<std::mutex> lock(ctx->mutex);
    if (ctx->copy_in) VGT_TRY_HIP(hipStreamSynchronize(ctx->copy_in), "synchronize the copy stream");
  }
  return VGT_HIP_OK;
}

int vgt_hip_device_of(const vgt_hip_ctx* ctx) { return ctx ? ctx->device : -1; }

#ifdef VGT_HIP_TESTING
/* Testing builds only (libvgt_hip_testing.so; declared in vgt_hip.h under VGT_HIP_TESTING). */
int vgt_hip_set_edt_variant(vgt_hip_ctx* ctx, int variant)
{
  if (!ctx || variant < 0 || variant > 1) return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "invalid EDT variant");
  ctx->variant = static_cast<vgt::EdtVariant>(variant);
  return VGT_HIP_OK;
}

int vgt_hip_debug_finalize_check(vgt_hip_ctx* ctx, int64_t first_d2, int64_t count, double resolution,
                                 uint64_t* mismatches, uint64_t* first_mismatch)
{
  if (!ctx || !mismatches || !first_mismatch || first_d2 < 0 || count <= 0 ||
      first_d2 + count > (int64_t{1} << 31))
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "invalid finalize-check range");
  VGT_TRY_HIP(hipSetDevice(ctx->device), "set device");
  unsigned long long* result = nullptr;
  VGT_TRY_HIP(hipMalloc(&result, 2 * sizeof(unsigned long long)), "allocate check result");
  const unsigned long long init[2] = {0ull, ~0ull};
  hipError_t err = hipMemcpyAsync(result, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream);
  if (err == hipSuccess) err = vgt::LaunchFinalizeCheck(first_d2, count, resolution, result, ctx->stream);
  unsigned long long host[2] = {0ull, ~0ull};
  if (err == hipSuccess)
    err = hipMemcpyAsync(host, result, sizeof(host), hipMemcpyDeviceToHost, ctx->stream);
  if (err == hipSuccess) err = hipStreamSynchronize(ctx->stream);
  (void)hipFree(result);
  VGT_TRY_HIP(err, "finalize check");
  *mismatches = host[0];
  *first_mismatch = host[1];
  return VGT_HIP_OK;
}

int vgt_hip_testing_set_host_pipeline_min_voxels(int64_t min_voxels)
{
  g_host_pipeline_min_voxels.store(min_voxels);
  return VGT_HIP_OK;
}

int vgt_hip_testing_set_short_line_rows(int rows)
{
  vgt::SetShortLineRows(rows);
  return VGT_HIP_OK;
}

size_t vgt_hip_testing_class_record_bytes(int64_t nx, int64_t ny, int64_t nz)
{
  if (nx <= 0 || ny <= 0 || nz <= 0) return 0;
  return vgt::ClassRecordBytes(nx, ny, nz);
}

int vgt_hip_testing_class_records_dev(vgt_hip_ctx* ctx, const float* occupancy_dev, int64_t nx, int64_t ny, int64_t nz,
                                      int unknown_is_filled, int64_t z_offset, void* records_dev, void* summary_dev)
{
  if (!ctx || !occupancy_dev || !records_dev) return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  const int rc = CheckSdfShape(nx, ny, nz, 1.0);
  if (rc != VGT_HIP_OK) return rc;
  VGT_TRY_HIP(hipSetDevice(ctx->device), "set device");
  vgt::SdfParams p{nx, ny, nz, 1.0, unknown_is_filled ? 1 : 0, 0};
  p.z_offset = z_offset;
  std::lock_guard<std::mutex> lock(ctx->mutex);
  VGT_TRY_HIP(vgt::LaunchClassRecordsFromOccupancy(occupancy_dev, static_cast<vgt::ClassRecord*>(records_dev), p,
                                                   static_cast<vgt::SlabLineSummary*>(summary_dev), ctx->stream),
              "pass 1");
  VGT_TRY_HIP(hipStreamSynchronize(ctx->stream), "synchronize");
  return VGT_HIP_OK;
}
#endif  // VGT_HIP_TESTING

/* ------------------------------ tracking grids ------------------------------ */

int vgt_hip_tracking_grids_create(vgt_hip_ctx* ctx, int64_t num_cells, int32_t num_grids,
                                  vgt_hip_grids** out_grids)
{
  if (!ctx || !out_grids) return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  *out_grids = nullptr;
  // zero-element buffers are an error, as in the reference (cuda_voxelization_helpers.cu:457-460)
  if (num_cells <= 0 || num_grids <= 0)
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "num_elements must be > 0");
  VGT_TRY_HIP(hipSetDevice(ctx->device), "set device");
  vgt_hip_grids* g = new (std::nothrow) vgt_hip_grids();
  if (!g) return Fail(VGT_HIP_ERR_RUNTIME, "out of host memory");
  g->ctx = ctx;
  g->device = ctx->device;
  g->num_cells = num_cells;
  g->num_grids = num_grids;
  const size_t bytes = static_cast<size_t>(num_cells) * num_grids * 2 * sizeof(int32_t);
  hipError_t err = PoolAllocate(ctx, reinterpret_cast<void**>(&g->dev), bytes);
  if (err == hipSuccess)
  {
    std::lock_guard<std::mutex> lock(ctx->mutex);
    err = hipMemsetAsync(g->dev, 0, bytes, ctx->stream);
  }
  if (err != hipSuccess)
  {
    if (g->dev) (void)hipFree(g->dev);
    delete g;
    return FailHip("Failed to allocate tracking grids", err);
  }
  AdoptChild(ctx);
  *out_grids = g;
  return VGT_HIP_OK;
}

void vgt_hip_tracking_grids_destroy(vgt_hip_grids* grids)
{
  if (!grids) return;
  vgt_hip_ctx* const ctx = grids->ctx;
  const size_t bytes = static_cast<size_t>(grids->num_cells) * grids->num_grids * 2 * sizeof(int32_t);
  // (the buffer goes back to the pool before the handle's reference, which may be the context's last, is dropped)
  (void)hipSetDevice(grids->device);
  if (ctx->destroyed.load())
    (void)hipDeviceSynchronize();
  else
    (void)hipStreamSynchronize(ctx->stream);
  PoolRelease(ctx, grids->dev, bytes);
  ReleaseChild(ctx, grids->device);
  delete grids;
}

int64_t vgt_hip_tracking_grids_num_cells(const vgt_hip_grids* grids)
{
  return grids ? grids->num_cells : 0;
}
int32_t vgt_hip_tracking_grids_num_grids(const vgt_hip_grids* grids)
{
  return grids ? grids->num_grids : 0;
}
int64_t vgt_hip_tracking_grids_offset(const vgt_hip_grids* grids, size_t grid_index)
{
  if (!grids || grid_index >= static_cast<size_t>(grids->num_grids)) return -1;
  return static_cast<int64_t>(grid_index) * grids->num_cells * 2;
}
void* vgt_hip_tracking_grids_dev_ptr(const vgt_hip_grids* grids, size_t grid_index)
{
  if (!grids || grid_index >= static_cast<size_t>(grids->num_grids)) return nullptr;
  return grids->dev + static_cast<int64_t>(grid_index) * grids->num_cells * 2;
}

int vgt_hip_tracking_grids_clear(vgt_hip_ctx* ctx, vgt_hip_grids* grids)
{
  if (!ctx || !grids || grids->ctx != ctx)
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "invalid tracking grids");
  VGT_TRY_HIP(hipSetDevice(ctx->device), "set device");
  std::lock_guard<std::mutex> lock(ctx->mutex);
  const size_t bytes = static_cast<size_t>(grids->num_cells) * grids->num_grids * 2 * sizeof(int32_t);
  VGT_TRY_HIP(hipMemsetAsync(grids->dev, 0, bytes, ctx->stream), "clear tracking grids");
  return VGT_HIP_OK;
}

/* --------------------------------- raycast ---------------------------------- */

int vgt_hip_raycast_points_f32_dev(vgt_hip_ctx* ctx, vgt_hip_grids* grids, size_t grid_index,
                                   const float* points_xyz_dev, int64_t num_points,
                                   float max_range, const float* grid_pointcloud_transform,
                                   float voxel_size, float inverse_voxel_size, float grid_x_size,
                                   float grid_y_size, float grid_z_size, int32_t num_x_voxels,
                                   int32_t num_y_voxels, int32_t num_z_voxels)
{
  const int rc = CheckRaycastArgs(ctx, grids, grid_index, points_xyz_dev, num_points,
                                  grid_pointcloud_transform, num_x_voxels, num_y_voxels,
                                  num_z_voxels);
  if (rc != VGT_HIP_OK) return rc;
  if (num_points == 0) return VGT_HIP_OK;
  VGT_TRY_HIP(hipSetDevice(ctx->device), "set device");
  vgt::RaycastGridF32 g;
  g.max_range = max_range;
  std::memcpy(g.xform, grid_pointcloud_transform, sizeof(g.xform));
  g.voxel_size = voxel_size;
  g.inverse_voxel_size = inverse_voxel_size;
  g.grid_size[0] = grid_x_size;
  g.grid_size[1] = grid_y_size;
  g.grid_size[2] = grid_z_size;
  g.counts[0] = num_x_voxels;
  g.counts[1] = num_y_voxels;
  g.counts[2] = num_z_voxels;
  int32_t* tracking = static_cast<int32_t*>(vgt_hip_tracking_grids_dev_ptr(grids, grid_index));
  std::lock_guard<std::mutex> lock(ctx->mutex);
  // stream order protects the scratch: the next call's kernels queue behind this one's
  const size_t scratch_bytes = vgt::RaycastScratchBytes(num_points);
  if (scratch_bytes > ctx->ray_scratch_bytes)
  {
    VGT_TRY_HIP(hipStreamSynchronize(ctx->stream), "drain before regrowing raycast scratch");
    if (ctx->ray_scratch) (void)hipFree(ctx->ray_scratch);
    ctx->ray_scratch = nullptr;
    ctx->ray_scratch_bytes = 0;
    VGT_TRY_HIP(hipMalloc(&ctx->ray_scratch, scratch_bytes + scratch_bytes / 4), "allocate raycast scratch");
    ctx->ray_scratch_bytes = scratch_bytes + scratch_bytes / 4;
  }
  VGT_TRY_HIP(vgt::LaunchRaycastF32(points_xyz_dev, num_points, 3, g, tracking, ctx->raycast_threads,
                                    ctx->ray_scratch, ctx->ray_scratch_bytes, ctx->stream),
              "Failed to dispatch raycast kernel");
  return VGT_HIP_OK;
}

int vgt_hip_raycast_points_f32(vgt_hip_ctx* ctx, vgt_hip_grids* grids, size_t grid_index,
                               const float* points_xyz_host, int64_t num_points, float max_range,
                               const float* grid_pointcloud_transform, float voxel_size,
                               float inverse_voxel_size, float grid_x_size, float grid_y_size,
                               float grid_z_size, int32_t num_x_voxels, int32_t num_y_voxels,
                               int32_t num_z_voxels)
{
  const int rc = CheckRaycastArgs(ctx, grids, grid_index, points_xyz_host, num_points,
                                  grid_pointcloud_transform, num_x_voxels, num_y_voxels,
                                  num_z_voxels);
  if (rc != VGT_HIP_OK) return rc;
  if (num_points == 0) return VGT_HIP_OK;
  VGT_TRY_HIP(hipSetDevice(ctx->device), "set device");
  vgt::RaycastGridF32 g;
  g.max_range = max_range;
  std::memcpy(g.xform, grid_pointcloud_transform, sizeof(g.xform));
  g.voxel_size = voxel_size;
  g.inverse_voxel_size = inverse_voxel_size;
  g.grid_size[0] = grid_x_size;
  g.grid_size[1] = grid_y_size;
  g.grid_size[2] = grid_z_size;
  g.counts[0] = num_x_voxels;
  g.counts[1] = num_y_voxels;
  g.counts[2] = num_z_voxels;
  int32_t* tracking = static_cast<int32_t*>(vgt_hip_tracking_grids_dev_ptr(grids, grid_index));
  const size_t bytes = static_cast<size_t>(num_points) * 3 * sizeof(float);
  const int threads = ctx->raycast_threads;
  const size_t scratch_bytes = vgt::RaycastScratchBytes(num_points);
  return UploadAndRun(ctx, points_xyz_host, bytes, scratch_bytes,
                      [&](const void* points_dev, void* scratch, hipStream_t stream) {
                        return vgt::LaunchRaycastF32(static_cast<const float*>(points_dev), num_points, 3, g, tracking,
                                                     threads, scratch, scratch_bytes, stream);
                      });
}

int vgt_hip_raycast_pointcloud2_f32(vgt_hip_ctx* ctx, vgt_hip_grids* grids, size_t grid_index,
                                    const uint8_t* cloud_data_host, int64_t num_points, int64_t point_step,
                                    int64_t xyz_offset, float max_range,
                                    const float* grid_pointcloud_transform, float voxel_size,
                                    float inverse_voxel_size, float grid_x_size, float grid_y_size,
                                    float grid_z_size, int32_t num_x_voxels, int32_t num_y_voxels,
                                    int32_t num_z_voxels)
{
  const int rc = CheckRaycastArgs(ctx, grids, grid_index, cloud_data_host, num_points,
                                  grid_pointcloud_transform, num_x_voxels, num_y_voxels, num_z_voxels);
  if (rc != VGT_HIP_OK) return rc;
  // x, y, z are three consecutive FLOAT32 fields (pointcloud_voxelization_ros_interface.cpp:49-78);
  // the kernel reads them in place, which needs them 4-byte aligned inside the buffer
  if (xyz_offset < 0 || point_step < xyz_offset + 12)
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "PointCloud does not have sequential xyz fields");
  if (point_step % 4 != 0 || xyz_offset % 4 != 0)
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "point_step and the xyz offset must be multiples of 4");
  if (num_points == 0) return VGT_HIP_OK;
  VGT_TRY_HIP(hipSetDevice(ctx->device), "set device");
  vgt::RaycastGridF32 g;
  g.max_range = max_range;
  std::memcpy(g.xform, grid_pointcloud_transform, sizeof(g.xform));
  g.voxel_size = voxel_size;
  g.inverse_voxel_size = inverse_voxel_size;
  g.grid_size[0] = grid_x_size;
  g.grid_size[1] = grid_y_size;
  g.grid_size[2] = grid_z_size;
  g.counts[0] = num_x_voxels;
  g.counts[1] = num_y_voxels;
  g.counts[2] = num_z_voxels;
  int32_t* tracking = static_cast<int32_t*>(vgt_hip_tracking_grids_dev_ptr(grids, grid_index));
  const size_t bytes = static_cast<size_t>(num_points) * static_cast<size_t>(point_step);
  const int threads = ctx->raycast_threads;
  const size_t scratch_bytes = vgt::RaycastScratchBytes(num_points);
  return UploadAndRun(ctx, cloud_data_host, bytes, scratch_bytes,
                      [&](const void* cloud_dev, void* scratch, hipStream_t stream) {
                        const float* first =
                            reinterpret_cast<const float*>(static_cast<const uint8_t*>(cloud_dev) + xyz_offset);
                        return vgt::LaunchRaycastF32(first, num_points, point_step / 4, g, tracking, threads, scratch,
                                                     scratch_bytes, stream);
                      });
}

int vgt_hip_raycast_points_f64(vgt_hip_ctx* ctx, vgt_hip_grids* grids, size_t grid_index,
                               const double* points_xyz_host, int64_t num_points,
                               double max_range, const double* grid_pointcloud_transform,
                               double voxel_size, double inverse_voxel_size, double grid_x_size,
                               double grid_y_size, double grid_z_size, int32_t num_x_voxels,
                               int32_t num_y_voxels, int32_t num_z_voxels)
{
  const int rc = CheckRaycastArgs(ctx, grids, grid_index, points_xyz_host, num_points,
                                  grid_pointcloud_transform, num_x_voxels, num_y_voxels,
                                  num_z_voxels);
  if (rc != VGT_HIP_OK) return rc;
  if (num_points == 0) return VGT_HIP_OK;
  VGT_TRY_HIP(hipSetDevice(ctx->device), "set device");
  vgt::RaycastGridF64 g;
  g.max_range = max_range;
  std::memcpy(g.xform, grid_pointcloud_transform, sizeof(g.xform));
  g.voxel_size = voxel_size;
  g.inverse_voxel_size = inverse_voxel_size;
  g.grid_size[0] = grid_x_size;
  g.grid_size[1] = grid_y_size;
  g.grid_size[2] = grid_z_size;
  g.counts[0] = num_x_voxels;
  g.counts[1] = num_y_voxels;
  g.counts[2] = num_z_voxels;
  int32_t* tracking = static_cast<int32_t*>(vgt_hip_tracking_grids_dev_ptr(grids, grid_index));
  const size_t bytes = static_cast<size_t>(num_points) * 3 * sizeof(double);
  const int threads = ctx->raycast_threads;
  const size_t scratch_bytes = vgt::RaycastScratchBytes(num_points);
  return UploadAndRun(ctx, points_xyz_host, bytes, scratch_bytes,
                      [&](const void* points_dev, void* scratch, hipStream_t stream) {
                        return vgt::LaunchRaycastF64(static_cast<const double*>(points_dev), num_points, g, tracking,
                                                     threads, scratch, scratch_bytes, stream);
                      });
}

/* --------------------------------- filter ----------------------------------- */

int vgt_hip_filter_grid_create(vgt_hip_ctx* ctx, int64_t num_cells, const float* occupancy_host,
                               vgt_hip_filter** out_filter)
{
  if (!ctx || !out_filter) return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  *out_filter = nullptr;
  if (num_cells <= 0) return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "num_elements must be > 0");
  if (!occupancy_host) return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "to_copy cannot be nullptr");
  VGT_TRY_HIP(hipSetDevice(ctx->device), "set device");
  vgt_hip_filter* f = new (std::nothrow) vgt_hip_filter();
  if (!f) return Fail(VGT_HIP_ERR_RUNTIME, "out of host memory");
  f->ctx = ctx;
  f->device = ctx->device;
  f->num_cells = num_cells;
  const size_t bytes = static_cast<size_t>(num_cells) * sizeof(float);
  hipError_t err = PoolAllocate(ctx, reinterpret_cast<void**>(&f->dev), bytes);
  if (err == hipSuccess)
  {
    // the caller's array is page-locked for the copy (a pageable 64 MiB copy is staged at a fifth of the link rate)
    const ScopedHostPin pin(occupancy_host, bytes);
    std::lock_guard<std::mutex> lock(ctx->mutex);
    err = hipMemcpyAsync(f->dev, occupancy_host, bytes, hipMemcpyHostToDevice, ctx->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(ctx->stream);
  }
  if (err != hipSuccess)
  {
    if (f->dev) (void)hipFree(f->dev);
    delete f;
    return FailHip("Failed to prepare filter grid", err);
  }
  AdoptChild(ctx);
  *out_filter = f;
  return VGT_HIP_OK;
}

namespace
{
// The filter kernel / a download is about to use the grid on the context's stream: order it behind a deferred upload.
static hipError_t OrderBehindUpload(vgt_hip_ctx* ctx, const vgt_hip_filter* filter)
{
  if (!filter->upload_pending) return hipSuccess;
  return hipStreamWaitEvent(ctx->stream, filter->uploaded, 0);
}
// The host has waited for work that was ordered behind the upload (or for the upload itself): the caller's array is free.
static void UploadHasFinished(vgt_hip_filter* filter)
{
  filter->upload_pending = false;
  delete static_cast<ScopedHostPin*>(filter->pin);
  filter->pin = nullptr;
}
}  // namespace

int vgt_hip_filter_grid_create_deferred(vgt_hip_ctx* ctx, int64_t num_cells, const float* occupancy_host,
                                        vgt_hip_filter** out_filter)
{
  if (!ctx || !out_filter) return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  *out_filter = nullptr;
  if (num_cells <= 0) return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "num_elements must be > 0");
  if (!occupancy_host) return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "to_copy cannot be nullptr");
  VGT_TRY_HIP(hipSetDevice(ctx->device), "set device");
  vgt_hip_filter* f = new (std::nothrow) vgt_hip_filter();
  if (!f) return Fail(VGT_HIP_ERR_RUNTIME, "out of host memory");
  f->ctx = ctx;
  f->device = ctx->device;
  f->num_cells = num_cells;
  const size_t bytes = static_cast<size_t>(num_cells) * sizeof(float);
  hipError_t err = PoolAllocate(ctx, reinterpret_cast<void**>(&f->dev), bytes);
  if (err == hipSuccess) err = hipEventCreateWithFlags(&f->uploaded, hipEventDisableTiming);
  if (err == hipSuccess)
  {
    f->pin = new (std::nothrow) ScopedHostPin(occupancy_host, bytes);  // (nullptr: the copy runs from pageable memory)
    std::lock_guard<std::mutex> lock(ctx->mutex);
    if (!ctx->copy_in) err = hipStreamCreateWithFlags(&ctx->copy_in, hipStreamNonBlocking);
    // (a pooled buffer may still be read by work queued on the context's stream: the copy starts behind it)
    if (err == hipSuccess) err = hipEventRecord(f->uploaded, ctx->stream);
    if (err == hipSuccess) err = hipStreamWaitEvent(ctx->copy_in, f->uploaded, 0);
    if (err == hipSuccess) err = hipMemcpyAsync(f->dev, occupancy_host, bytes, hipMemcpyHostToDevice, ctx->copy_in);
    if (err == hipSuccess) err = hipEventRecord(f->uploaded, ctx->copy_in);
    f->upload_pending = err == hipSuccess;
  }
  if (err != hipSuccess)
  {
    {
      // (a copy that was enqueued before the failure must not outlive the caller's array)
      std::lock_guard<std::mutex> lock(ctx->mutex);
      if (f->uploaded && ctx->copy_in) (void)hipStreamSynchronize(ctx->copy_in);
    }
    delete static_cast<ScopedHostPin*>(f->pin);
    if (f->uploaded) (void)hipEventDestroy(f->uploaded);
    if (f->dev) (void)hipFree(f->dev);
    delete f;
    return FailHip("Failed to prepare filter grid", err);
  }
  AdoptChild(ctx);
  *out_filter = f;
  return VGT_HIP_OK;
}

void vgt_hip_filter_grid_destroy(vgt_hip_filter* filter)
{
  if (!filter) return;
  vgt_hip_ctx* const ctx = filter->ctx;
  (void)hipSetDevice(filter->device);
  if (filter->upload_pending) (void)hipEventSynchronize(filter->uploaded);
  UploadHasFinished(filter);
  if (filter->uploaded) (void)hipEventDestroy(filter->uploaded);
  if (ctx->destroyed.load())
    (void)hipDeviceSynchronize();
  else
    (void)hipStreamSynchronize(ctx->stream);
  PoolRelease(ctx, filter->dev, static_cast<size_t>(filter->num_cells) * sizeof(float));
  ReleaseChild(ctx, filter->device);
  delete filter;
}

int64_t vgt_hip_filter_grid_num_cells(const vgt_hip_filter* filter)
{
  return filter ? filter->num_cells : 0;
}
void* vgt_hip_filter_grid_dev_ptr(const vgt_hip_filter* filter)
{
  if (!filter) return nullptr;
  // A deferred upload is ordered behind nothing the caller's own streams know of: wait for it here, so that the
  // pointer can be used on any stream (the pin on the caller's array is released by retrieve / destroy as before).
  // (the flag is written under the context's mutex by filter / retrieve: read it the same way; a failed wait means the
  // grid's content is not known to be there -- no pointer then, the reason in vgt_hip_last_error(); the calling thread's
  // current device is left as it was)
  bool pending = false;
  {
    std::lock_guard<std::mutex> lock(filter->ctx->mutex);
    pending = filter->upload_pending;
  }
  if (pending)
  {
    int previous = -1;
    if (hipGetDevice(&previous) != hipSuccess) previous = -1;
    hipError_t err = hipSetDevice(filter->device);
    if (err == hipSuccess) err = hipEventSynchronize(filter->uploaded);
    if (previous >= 0 && previous != filter->device) (void)hipSetDevice(previous);
    if (err != hipSuccess)
    {
      (void)hipGetLastError();
      Fail(VGT_HIP_ERR_RUNTIME, std::string("waiting for the filter grid's upload: ") + hipGetErrorString(err));
      return nullptr;
    }
  }
  return filter->dev;
}

static int FilterImpl(vgt_hip_ctx* ctx, const vgt_hip_grids* grids, double percent_seen_free,
                      int32_t outlier_points_threshold, int32_t num_cameras_seen_free,
                      bool ratio_in_double, vgt_hip_filter* filter)
{
  if (!ctx || !grids || !filter) return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  if (grids->ctx != ctx || filter->ctx != ctx)
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "handles belong to another context");
  if (grids->num_cells != filter->num_cells)
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "tracking grids and filter grid differ in size");
  VGT_TRY_HIP(hipSetDevice(ctx->device), "set device");
  std::lock_guard<std::mutex> lock(ctx->mutex);
  VGT_TRY_HIP(OrderBehindUpload(ctx, filter), "order the filter behind the grid's upload");
  VGT_TRY_HIP(vgt::LaunchFilter(grids->dev, grids->num_cells, grids->num_grids, percent_seen_free,
                                outlier_points_threshold, num_cameras_seen_free, ratio_in_double,
                                filter->dev, ctx->threads_per_block, ctx->stream),
              "Failed to dispatch filter kernel");
  return VGT_HIP_OK;
}

int vgt_hip_filter_tracking_grids(vgt_hip_ctx* ctx, const vgt_hip_grids* grids,
                                  float percent_seen_free, int32_t outlier_points_threshold,
                                  int32_t num_cameras_seen_free, vgt_hip_filter* filter)
{
  return FilterImpl(ctx, grids, static_cast<double>(percent_seen_free), outlier_points_threshold,
                    num_cameras_seen_free, false, filter);
}

int vgt_hip_filter_tracking_grids_f64(vgt_hip_ctx* ctx, const vgt_hip_grids* grids,
                                      double percent_seen_free, int32_t outlier_points_threshold,
                                      int32_t num_cameras_seen_free, vgt_hip_filter* filter)
{
  return FilterImpl(ctx, grids, percent_seen_free, outlier_points_threshold,
                    num_cameras_seen_free, true, filter);
}

int vgt_hip_retrieve_tracking_grid(vgt_hip_ctx* ctx, const vgt_hip_grids* grids, size_t grid_index,
                                   void* host_out)
{
  if (!ctx || !grids || !host_out || grids->ctx != ctx)
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  if (grid_index >= static_cast<size_t>(grids->num_grids))
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "tracking grid index out of range");
  VGT_TRY_HIP(hipSetDevice(ctx->device), "set device");
  std::lock_guard<std::mutex> lock(ctx->mutex);
  const size_t bytes = static_cast<size_t>(grids->num_cells) * 2 * sizeof(int32_t);
  VGT_TRY_HIP(hipMemcpyAsync(host_out, vgt_hip_tracking_grids_dev_ptr(grids, grid_index), bytes,
                             hipMemcpyDeviceToHost, ctx->stream),
              "Failed to memcpy the tracking grid back to the host");
  VGT_TRY_HIP(hipStreamSynchronize(ctx->stream), "retrieve tracking grid");
  return VGT_HIP_OK;
}

int vgt_hip_retrieve_filtered_grid(vgt_hip_ctx* ctx, const vgt_hip_filter* filter, void* host_out)
{
  if (!ctx || !filter || !host_out || filter->ctx != ctx)
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  VGT_TRY_HIP(hipSetDevice(ctx->device), "set device");
  const size_t bytes = static_cast<size_t>(filter->num_cells) * sizeof(float);
  const ScopedHostPin pin(host_out, bytes);
  std::lock_guard<std::mutex> lock(ctx->mutex);
  VGT_TRY_HIP(OrderBehindUpload(ctx, filter), "order the download behind the grid's upload");
  VGT_TRY_HIP(hipMemcpyAsync(host_out, filter->dev, bytes, hipMemcpyDeviceToHost, ctx->stream),
              "Failed to memcpy the filter grid back to the host");
  VGT_TRY_HIP(hipStreamSynchronize(ctx->stream), "retrieve filtered grid");
  // (the handle is logically const for the caller; the finished upload's bookkeeping is not part of its value)
  if (filter->upload_pending) UploadHasFinished(const_cast<vgt_hip_filter*>(filter));
  return VGT_HIP_OK;
}

/* ----------------------------------- SDF ------------------------------------ */

size_t vgt_hip_sdf_workspace_bytes(int64_t nx, int64_t ny, int64_t nz)
{
  if (nx <= 0 || ny <= 0 || nz <= 0) return 0;
  return CarveWorkspace(nullptr, nx, ny, nz, vgt::EdtVariant::kDefault).bytes;
}

size_t vgt_hip_sdf_workspace_bytes_for_variant(int64_t nx, int64_t ny, int64_t nz, int variant)
{
  if (nx <= 0 || ny <= 0 || nz <= 0 || variant < 0 || variant > 1) return 0;
#ifndef VGT_HIP_TESTING
  if (variant != 0) return 0;  // the cross-check variants are not part of this build
#endif
  return CarveWorkspace(nullptr, nx, ny, nz, static_cast<vgt::EdtVariant>(variant)).bytes;
}

int vgt_hip_sdf_from_occupancy_f32(vgt_hip_ctx* ctx, const float* occupancy_host, int64_t nx,
                                   int64_t ny, int64_t nz, double resolution,
                                   int unknown_is_filled, int add_virtual_border, float* sdf_host,
                                   float* out_min, float* out_max)
{
  const vgt::SdfParams p{nx, ny, nz, resolution, unknown_is_filled ? 1 : 0,
                         add_virtual_border ? 1 : 0};
  return SdfFromHost<float>(ctx, occupancy_host, p, sdf_host, out_min, out_max);
}

int vgt_hip_sdf_from_mask_u8(vgt_hip_ctx* ctx, const uint8_t* filled_mask_host, int64_t nx,
                             int64_t ny, int64_t nz, double resolution, int add_virtual_border,
                             float* sdf_host, float* out_min, float* out_max)
{
  const vgt::SdfParams p{nx, ny, nz, resolution, 0, add_virtual_border ? 1 : 0};
  return SdfFromHost<uint8_t>(ctx, filled_mask_host, p, sdf_host, out_min, out_max);
}

int vgt_hip_sdf_dev(vgt_hip_ctx* ctx, const float* occupancy_dev, int64_t nx, int64_t ny,
                    int64_t nz, double resolution, int unknown_is_filled, int add_virtual_border,
                    float* sdf_dev, void* workspace_dev, size_t workspace_bytes, float* minmax_dev)
{
  if (!ctx || !occupancy_dev || !sdf_dev)
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  const int rc = CheckSdfShape(nx, ny, nz, resolution);
  if (rc != VGT_HIP_OK) return rc;
  VGT_TRY_HIP(hipSetDevice(ctx->device), "set device");
  const vgt::SdfParams p{nx, ny, nz, resolution, unknown_is_filled ? 1 : 0,
                         add_virtual_border ? 1 : 0};
  std::lock_guard<std::mutex> lock(ctx->mutex);
  hipEvent_t* slot = TimingSlot(ctx);
  const int result = RunSdfPipeline<float>(ctx, occupancy_dev, p, sdf_dev, workspace_dev, workspace_bytes,
                                           minmax_dev, slot);
  if (slot && result == VGT_HIP_OK) ctx->timing_kind[static_cast<size_t>(ctx->timing_used++)] = 1;
  return result;
}

size_t vgt_hip_sdf_batch_workspace_bytes(int64_t batch, int64_t nx, int64_t ny, int64_t nz)
{
  if (batch <= 0 || nx <= 0 || ny <= 0 || nz <= 0) return 0;
  return CarveWorkspace(nullptr, nx, ny, nz, vgt::EdtVariant::kDefault, batch).bytes;
}

int vgt_hip_sdf_batch_dev(vgt_hip_ctx* ctx, const float* occupancy_dev, int64_t batch, int64_t nx, int64_t ny,
                          int64_t nz, double resolution, int unknown_is_filled, int add_virtual_border,
                          float* sdf_dev, void* workspace_dev, size_t workspace_bytes, float* minmax_dev)
{
  if (!ctx || !occupancy_dev || !sdf_dev) return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  int rc = CheckSdfShape(nx, ny, nz, resolution);
  if (rc == VGT_HIP_OK) rc = CheckBatch(batch, nx, ny, nz);
  if (rc != VGT_HIP_OK) return rc;
  VGT_TRY_HIP(hipSetDevice(ctx->device), "set device");
  vgt::SdfParams p{nx, ny, nz, resolution, unknown_is_filled ? 1 : 0, add_virtual_border ? 1 : 0};
  p.batch = batch;
  std::lock_guard<std::mutex> lock(ctx->mutex);
  hipEvent_t* slot = TimingSlot(ctx);
  const int result = RunSdfPipeline<float>(ctx, occupancy_dev, p, sdf_dev, workspace_dev, workspace_bytes,
                                           minmax_dev, slot);
  if (slot && result == VGT_HIP_OK) ctx->timing_kind[static_cast<size_t>(ctx->timing_used++)] = 1;
  return result;
}

int vgt_hip_sdf_batch_from_occupancy_f32(vgt_hip_ctx* ctx, const float* const* occupancy_host, int64_t batch,
                                         int64_t nx, int64_t ny, int64_t nz, double resolution,
                                         int unknown_is_filled, int add_virtual_border, float* const* sdf_host,
                                         float* out_min, float* out_max)
{
  if (!ctx || !occupancy_host || !sdf_host) return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  int rc = CheckSdfShape(nx, ny, nz, resolution);
  if (rc != VGT_HIP_OK) return rc;
  if (batch <= 0) return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "batch must be positive");
  for (int64_t b = 0; b < batch; b++)
    if (!occupancy_host[b] || !sdf_host[b]) return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "null grid in the batch");
  VGT_TRY_HIP(hipSetDevice(ctx->device), "set device");
  const size_t n = static_cast<size_t>(nx * ny * nz);
  // Grids per launch: as many as the limits of a batch and a memory budget allow (the context keeps the buffers).
  const int64_t group = BatchGroup(batch, nx, ny, nz, size_t{2} << 30, n * 9);
  std::lock_guard<std::mutex> lock(ctx->mutex);
  const size_t ws_bytes = CarveWorkspace(nullptr, nx, ny, nz, ctx->variant, ctx->variant == vgt::EdtVariant::kDefault ? group : 1).bytes;
  VGT_TRY_HIP(Reserve(&ctx->sdf_in, &ctx->sdf_in_bytes, static_cast<size_t>(group) * n * sizeof(float)), "allocate SDF input");
  VGT_TRY_HIP(Reserve(&ctx->sdf_out, &ctx->sdf_out_bytes, static_cast<size_t>(group) * n * sizeof(float) + static_cast<size_t>(group) * 2 * sizeof(float)),
              "allocate SDF output");
  VGT_TRY_HIP(Reserve(&ctx->sdf_ws, &ctx->sdf_ws_bytes, ws_bytes), "allocate SDF workspace");
  float* const in_dev = static_cast<float*>(ctx->sdf_in);
  float* const out_dev = static_cast<float*>(ctx->sdf_out);
  float* const mm_dev = out_dev + static_cast<size_t>(group) * n;
  std::vector<float> mm(static_cast<size_t>(group) * 2);
  hipStream_t s = ctx->stream;
  for (int64_t first = 0; first < batch; first += group)
  {
    const int64_t count = batch - first < group ? batch - first : group;
    // Inputs of 32 MiB and more are page-locked for the group's uploads (they hold data, their pages exist); smaller
    // ones -- for which locking and unlocking costs several times the copy -- cross through the context's ring.
    std::vector<std::unique_ptr<ScopedHostPin>> pins;
    if (n * sizeof(float) >= (size_t{32} << 20))
    {
      for (int64_t b = 0; b < count; b++) pins.emplace_back(new ScopedHostPin(occupancy_host[first + b], n * sizeof(float)));
      for (int64_t b = 0; b < count; b++)
        VGT_TRY_HIP(hipMemcpyAsync(in_dev + static_cast<size_t>(b) * n, occupancy_host[first + b], n * sizeof(float),
                                   hipMemcpyHostToDevice, s),
                    "copy occupancy to device");
    }
    else
    {
      std::vector<HostArrayCopy> uploads;
      for (int64_t b = 0; b < count; b++)
        uploads.push_back(HostArrayCopy{in_dev + static_cast<size_t>(b) * n, const_cast<float*>(occupancy_host[first + b])});
      const hipError_t moved = UploadFromHostArrays(ctx, uploads, n * sizeof(float), s);
      if (moved != hipSuccess) (void)hipStreamSynchronize(s);
      VGT_TRY_HIP(moved, "copy occupancy to device");
    }
    vgt::SdfParams p{nx, ny, nz, resolution, unknown_is_filled ? 1 : 0, add_virtual_border ? 1 : 0};
    if (ctx->variant == vgt::EdtVariant::kDefault)
    {
      p.batch = count;
      rc = RunSdfPipeline<float>(ctx, in_dev, p, out_dev, ctx->sdf_ws, ctx->sdf_ws_bytes, mm_dev, nullptr);
      if (rc != VGT_HIP_OK) return rc;
    }
    else
    {
      // (testing builds with a cross-check variant selected: grid by grid)
      for (int64_t b = 0; b < count; b++)
      {
        rc = RunSdfPipeline<float>(ctx, in_dev + static_cast<size_t>(b) * n, p, out_dev + static_cast<size_t>(b) * n,
                                   ctx->sdf_ws, ctx->sdf_ws_bytes, mm_dev + 2 * b, nullptr);
        if (rc != VGT_HIP_OK) return rc;
      }
    }
    VGT_TRY_HIP(hipMemcpyAsync(mm.data(), mm_dev, static_cast<size_t>(count) * 2 * sizeof(float), hipMemcpyDeviceToHost, s),
                "copy extrema to host");
    {
      std::vector<HostArrayCopy> copies;
      for (int64_t b = 0; b < count; b++) copies.push_back(HostArrayCopy{out_dev + static_cast<size_t>(b) * n, sdf_host[first + b]});
      const hipError_t moved = DownloadToHostArrays(ctx, copies, n * sizeof(float), s);  // (synchronises the stream)
      if (moved != hipSuccess) (void)hipStreamSynchronize(s);
      VGT_TRY_HIP(moved, "copy SDF to host");
    }
    for (int64_t b = 0; b < count; b++)
    {
      if (out_min) out_min[first + b] = mm[static_cast<size_t>(2 * b)];
      if (out_max) out_max[first + b] = mm[static_cast<size_t>(2 * b + 1)];
    }
  }
  return VGT_HIP_OK;
}

int vgt_hip_sdf_dev_timed(vgt_hip_ctx* ctx, const float* occupancy_dev, int64_t nx, int64_t ny,
                          int64_t nz, double resolution, int unknown_is_filled,
                          int add_virtual_border, float* sdf_dev, void* workspace_dev,
                          size_t workspace_bytes, float* minmax_dev, float* kernel_ms)
{
  if (!ctx || !occupancy_dev || !sdf_dev || !kernel_ms)
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  const int rc = CheckSdfShape(nx, ny, nz, resolution);
  if (rc != VGT_HIP_OK) return rc;
  VGT_TRY_HIP(hipSetDevice(ctx->device), "set device");
  const vgt::SdfParams p{nx, ny, nz, resolution, unknown_is_filled ? 1 : 0,
                         add_virtual_border ? 1 : 0};
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  int result = VGT_HIP_OK;
  for (int i = 0; i < 4 && result == VGT_HIP_OK; i++)
  {
    const hipError_t err = hipEventCreate(&ev[i]);
    if (err != hipSuccess) result = FailHip("create event", err);
  }
  if (result == VGT_HIP_OK)
  {
    std::lock_guard<std::mutex> lock(ctx->mutex);
    result = RunSdfPipeline<float>(ctx, occupancy_dev, p, sdf_dev, workspace_dev, workspace_bytes,
                                   minmax_dev, ev);
    const hipError_t err = hipStreamSynchronize(ctx->stream);
    if (result == VGT_HIP_OK && err != hipSuccess) result = FailHip("synchronize", err);
    for (int i = 0; i < 3 && result == VGT_HIP_OK; i++)
    {
      const hipError_t terr = hipEventElapsedTime(&kernel_ms[i], ev[i], ev[i + 1]);
      if (terr != hipSuccess) result = FailHip("event elapsed", terr);
    }
  }
  for (int i = 0; i < 4; i++)
    if (ev[i]) (void)hipEventDestroy(ev[i]);
  return result;
}

/* ------------------------------ multi-GPU Z slabs ----------------------------- */

/* ------------------- SDFs of the map types with tagged cells ------------------- */

namespace
{
static void FreeCells(vgt_hip_cells* c)
{
  if (!c) return;
  if (c->records) (void)hipFree(c->records);
  if (c->mask) (void)hipFree(c->mask);
  if (c->sdf) (void)hipFree(c->sdf);
  if (c->sdf_named) (void)hipFree(c->sdf_named);
  if (c->workspace) (void)hipFree(c->workspace);
  if (c->objects) (void)hipFree(c->objects);
  if (c->scalar) (void)hipFree(c->scalar);
  if (c->batch_masks) (void)hipFree(c->batch_masks);
  if (c->batch_sdf) (void)hipFree(c->batch_sdf);
  if (c->batch_ws) (void)hipFree(c->batch_ws);
  delete c;
}

static int CheckCells(const vgt_hip_ctx* ctx, const vgt_hip_cells* cells)
{
  if (!ctx || !cells) return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  if (cells->ctx != ctx) return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "cells belong to another context");
  return VGT_HIP_OK;
}

// mask (mode, objects) -> signed distance field in `sdf_dev`, extrema in ctx->minmax_out.
// Caller holds the context mutex.
static int RunCellsSdf(vgt_hip_ctx* ctx, vgt_hip_cells* c, int mode, int num_objects, const vgt::SdfParams& p,
                float* sdf_dev, float* minmax_dev = nullptr)  // (extrema: the context's two floats unless told otherwise)
{
  const int64_t n = c->nx * c->ny * c->nz;
  VGT_TRY_HIP(vgt::LaunchCellMask(c->records, n, c->cell_bytes, c->object_id_offset, mode, c->objects,
                                  num_objects, p.unknown_is_filled, c->mask, ctx->stream),
              "cell predicate");
  vgt::SdfParams mask_params = p;
  mask_params.unknown_is_filled = 0;
  // (the workspace was sized for the default pipeline: a cross-check variant set later needs more)
  const size_t need = CarveWorkspace(nullptr, c->nx, c->ny, c->nz, ctx->variant).bytes;
  if (need > c->workspace_bytes)
  {
    VGT_TRY_HIP(hipStreamSynchronize(ctx->stream), "synchronize");
    VGT_TRY_HIP(Reserve(&c->workspace, &c->workspace_bytes, need), "allocate SDF workspace");
  }
  return RunSdfPipeline<uint8_t>(ctx, c->mask, mask_params, sdf_dev, c->workspace, c->workspace_bytes,
                                 minmax_dev ? minmax_dev : ctx->minmax_out, nullptr);
}

// A field of the context's stream into a host array (usually one the caller has just allocated) + the extrema.  The
// kernels that produce it are enqueued, not finished: preparing the array overlaps them.  Large arrays are faulted in
// by a few threads and page-locked for one DMA; medium ones cross through the context's page-locked ring
// (DownloadToHostArrays: locking fresh pages costs more than the copy); small ones take the runtime's pageable path.
static int CopySdfToHost(vgt_hip_ctx* ctx, const float* sdf_dev, int64_t n, float* sdf_host, float* out_min,
                  float* out_max)
{
  float mm[2] = {0.0f, 0.0f};
  const size_t bytes = static_cast<size_t>(n) * sizeof(float);
  hipError_t err = hipSuccess;
  if (bytes >= (size_t{32} << 20))
  {
    vgt::HostRangePopulator fresh(sdf_host, bytes);
    fresh.Wait();
    const ScopedHostPin pin(sdf_host, bytes);
    err = hipMemcpyAsync(sdf_host, sdf_dev, bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (err == hipSuccess) err = hipMemcpyAsync(mm, ctx->minmax_out, sizeof(mm), hipMemcpyDeviceToHost, ctx->stream);
    const hipError_t sync = hipStreamSynchronize(ctx->stream);  // (before the array is unlocked, whatever happened)
    if (err == hipSuccess) err = sync;
  }
  else if (bytes >= (size_t{1} << 20))
  {
    err = hipMemcpyAsync(mm, ctx->minmax_out, sizeof(mm), hipMemcpyDeviceToHost, ctx->stream);
    if (err == hipSuccess)
      err = DownloadToHostArrays(ctx, {HostArrayCopy{sdf_dev, sdf_host}}, bytes, ctx->stream);  // (synchronises)
    if (err != hipSuccess) (void)hipStreamSynchronize(ctx->stream);  // (`mm` is on this stack)
  }
  else
  {
    err = hipMemcpyAsync(sdf_host, sdf_dev, bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (err == hipSuccess) err = hipMemcpyAsync(mm, ctx->minmax_out, sizeof(mm), hipMemcpyDeviceToHost, ctx->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(ctx->stream);
  }
  if (err != hipSuccess) return FailHip("copy SDF to host", err);
  if (out_min) *out_min = mm[0];
  if (out_max) *out_max = mm[1];
  return VGT_HIP_OK;
}
}  // namespace

int vgt_hip_cells_create(vgt_hip_ctx* ctx, const void* cells_host, int64_t nx, int64_t ny, int64_t nz,
                         int32_t cell_bytes, int32_t object_id_offset, vgt_hip_cells** out_cells)
{
  if (!ctx || !cells_host || !out_cells) return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  *out_cells = nullptr;
  const int rc = CheckSdfShape(nx, ny, nz, 1.0);
  if (rc != VGT_HIP_OK) return rc;
  if (cell_bytes < 4 || cell_bytes % 4 != 0 || cell_bytes > 64)
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "cell records must be 4..64 bytes, a multiple of 4");
  if (object_id_offset != -1 &&
      (object_id_offset < 4 || object_id_offset % 4 != 0 || object_id_offset + 4 > cell_bytes))
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "object id offset outside the cell record");
  VGT_TRY_HIP(hipSetDevice(ctx->device), "set device");
  vgt_hip_cells* c = new (std::nothrow) vgt_hip_cells();
  if (!c) return Fail(VGT_HIP_ERR_RUNTIME, "out of host memory");
  c->ctx = ctx;
  c->device = ctx->device;
  c->nx = nx;
  c->ny = ny;
  c->nz = nz;
  c->cell_bytes = cell_bytes;
  c->object_id_offset = object_id_offset;
  const size_t n = static_cast<size_t>(nx * ny * nz);
  c->workspace_bytes = vgt_hip_sdf_workspace_bytes(nx, ny, nz);
  hipError_t err = hipMalloc(&c->records, n * static_cast<size_t>(cell_bytes));
  if (err == hipSuccess) err = hipMalloc(reinterpret_cast<void**>(&c->mask), n);
  if (err == hipSuccess) err = hipMalloc(reinterpret_cast<void**>(&c->sdf), n * sizeof(float));
  if (err == hipSuccess) err = hipMalloc(&c->workspace, c->workspace_bytes);
  if (err == hipSuccess) err = hipMalloc(reinterpret_cast<void**>(&c->scalar), 2 * sizeof(uint32_t));
  if (err == hipSuccess)
  {
    std::lock_guard<std::mutex> lock(ctx->mutex);
    err = hipMemcpyAsync(c->records, cells_host, n * static_cast<size_t>(cell_bytes), hipMemcpyHostToDevice,
                         ctx->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(ctx->stream);
  }
  if (err != hipSuccess)
  {
    FreeCells(c);
    return FailHip("upload cell records", err);
  }
  AdoptChild(ctx);
  *out_cells = c;
  return VGT_HIP_OK;
}

void vgt_hip_cells_destroy(vgt_hip_cells* cells)
{
  if (!cells) return;
  ReleaseChild(cells->ctx, cells->device);
  FreeCells(cells);
}

int vgt_hip_cells_object_ids(vgt_hip_ctx* ctx, vgt_hip_cells* cells, uint32_t* ids_out, int64_t capacity,
                             int64_t* count)
{
  const int rc = CheckCells(ctx, cells);
  if (rc != VGT_HIP_OK) return rc;
  if (!count || capacity < 0 || (capacity > 0 && !ids_out))
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "invalid id buffer");
  *count = 0;
  if (cells->object_id_offset < 0) return VGT_HIP_OK;  // this map type has no object ids
  VGT_TRY_HIP(hipSetDevice(ctx->device), "set device");
  std::lock_guard<std::mutex> lock(ctx->mutex);
  const int64_t n = cells->nx * cells->ny * cells->nz;
  // One pass: every id goes into a device hash set, the set is compacted and sorted on the host (the reference
  // collects a std::set, tagged_object_occupancy_map.hpp:268-289).  Only a grid with more distinct ids than the
  // table can hold falls back to the one-id-per-launch scan.
  // Tagged maps usually hold a handful of ids: a 2^16-slot table (0.5 MiB) serves them; only when it overflows is the
  // table sized for the worst case (two slots per cell, up to 2^27 slots = 1 GiB).
  int full_log2 = 10;
  while (full_log2 < 27 && (int64_t{1} << full_log2) < 2 * n) full_log2++;
  const int attempts[2] = {std::min(16, full_log2), full_log2};
  for (int attempt = 0; attempt < 2; attempt++)
  {
    if (attempt == 1 && attempts[1] == attempts[0]) break;
    const int kTableLog2 = attempts[attempt];
    const size_t slots = size_t{1} << kTableLog2;
    uint32_t* table = nullptr;
    hipError_t err = hipMalloc(reinterpret_cast<void**>(&table), (2 * slots + 64) * sizeof(uint32_t));
    if (err == hipSuccess)
    {
      uint32_t* ids_dev = table + slots;
      uint32_t* count_overflow = ids_dev + slots;
      uint32_t header[2] = {0u, 1u};
      std::vector<uint32_t> ids;
      err = hipMemsetAsync(table, 0, (2 * slots + 64) * sizeof(uint32_t), ctx->stream);
      if (err == hipSuccess)
        err = vgt::LaunchDistinctObjectIds(cells->records, n, cells->cell_bytes, cells->object_id_offset, table, kTableLog2,
                                           ids_dev, count_overflow, ctx->stream);
      if (err == hipSuccess)
        err = hipMemcpyAsync(header, count_overflow, sizeof(header), hipMemcpyDeviceToHost, ctx->stream);
      if (err == hipSuccess) err = hipStreamSynchronize(ctx->stream);
      if (err == hipSuccess && header[1] == 0u && header[0] > 0u)
      {
        ids.resize(header[0]);
        err = hipMemcpy(ids.data(), ids_dev, ids.size() * sizeof(uint32_t), hipMemcpyDeviceToHost);
      }
      (void)hipFree(table);
      VGT_TRY_HIP(err, "object id scan");
      if (header[1] == 0u)
      {
        std::sort(ids.begin(), ids.end());
        *count = static_cast<int64_t>(ids.size());
        for (int64_t k = 0; k < *count && k < capacity; k++) ids_out[k] = ids[static_cast<size_t>(k)];
        return VGT_HIP_OK;
      }
    }
    else
      (void)hipGetLastError();
  }
  uint32_t after = 0;  // ids > 0 only (tagged_object_occupancy_map.hpp:279-283)
  for (;;)
  {
    const uint32_t init[2] = {0xffffffffu, 0u};
    VGT_TRY_HIP(hipMemcpyAsync(cells->scalar, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream),
                "reset id scan");
    VGT_TRY_HIP(vgt::LaunchNextObjectId(cells->records, n, cells->cell_bytes, cells->object_id_offset, after,
                                        cells->scalar, ctx->stream),
                "object id scan");
    uint32_t next[2] = {0u, 0u};
    VGT_TRY_HIP(hipMemcpyAsync(next, cells->scalar, sizeof(next), hipMemcpyDeviceToHost, ctx->stream),
                "read id scan");
    VGT_TRY_HIP(hipStreamSynchronize(ctx->stream), "id scan");
    if (!next[1]) break;
    if (*count < capacity) ids_out[*count] = next[0];
    (*count)++;
    after = next[0];
    if (after == 0xffffffffu) break;
  }
  return VGT_HIP_OK;
}

int vgt_hip_cells_sdf(vgt_hip_ctx* ctx, vgt_hip_cells* cells, const uint32_t* objects_to_use,
                      int64_t num_objects, double resolution, int unknown_is_filled, int add_virtual_border,
                      float* sdf_host, float* out_min, float* out_max)
{
  int rc = CheckCells(ctx, cells);
  if (rc != VGT_HIP_OK) return rc;
  if (!sdf_host || num_objects < 0 || (num_objects > 0 && !objects_to_use))
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  if (num_objects > 0 && cells->object_id_offset < 0)
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "this cell type carries no object id");
  rc = CheckSdfShape(cells->nx, cells->ny, cells->nz, resolution);
  if (rc != VGT_HIP_OK) return rc;
  VGT_TRY_HIP(hipSetDevice(ctx->device), "set device");
  // the reference looks the ids up in a std::set (tagged_object_occupancy_map.hpp:206-211)
  std::vector<uint32_t> objects(objects_to_use, objects_to_use + num_objects);
  std::sort(objects.begin(), objects.end());
  objects.erase(std::unique(objects.begin(), objects.end()), objects.end());
  std::lock_guard<std::mutex> lock(ctx->mutex);
  if (objects.size() > cells->objects_capacity)
  {
    VGT_TRY_HIP(hipStreamSynchronize(ctx->stream), "drain before regrowing object list");
    if (cells->objects) (void)hipFree(cells->objects);
    cells->objects = nullptr;
    cells->objects_capacity = 0;
    VGT_TRY_HIP(hipMalloc(reinterpret_cast<void**>(&cells->objects), objects.size() * sizeof(uint32_t)),
                "allocate object list");
    cells->objects_capacity = objects.size();
  }
  if (!objects.empty())
  {
    VGT_TRY_HIP(hipMemcpyAsync(cells->objects, objects.data(), objects.size() * sizeof(uint32_t),
                               hipMemcpyHostToDevice, ctx->stream),
                "upload object list");
    // `objects` is pageable host memory: the copy has been staged when the call returns
  }
  const vgt::SdfParams p{cells->nx, cells->ny, cells->nz, resolution, unknown_is_filled ? 1 : 0,
                         add_virtual_border ? 1 : 0};
  const size_t n = static_cast<size_t>(cells->nx * cells->ny * cells->nz);
  if (n * sizeof(float) <= (size_t{1} << 19) && EnsureStaging(ctx) == hipSuccess)
  {
    // a small field: the X pass writes it, and the extrema behind it, straight into the context's page-locked ring
    // (as SdfFromHost does for small maps: no download, one wait, one memcpy)
    float* const field = reinterpret_cast<float*>(static_cast<char*>(ctx->host_staging) + kStagingSlotBytes);
    rc = RunCellsSdf(ctx, cells, objects.empty() ? 0 : 1, static_cast<int>(objects.size()), p, field, field + n);
    const hipError_t err = hipStreamSynchronize(ctx->stream);
    if (rc == VGT_HIP_OK && err != hipSuccess) rc = FailHip("small tagged-map SDF extraction", err);
    if (rc != VGT_HIP_OK) return rc;
    std::memcpy(sdf_host, field, n * sizeof(float));
    if (out_min) *out_min = field[n];
    if (out_max) *out_max = field[n + 1];
    return VGT_HIP_OK;
  }
  rc = RunCellsSdf(ctx, cells, objects.empty() ? 0 : 1, static_cast<int>(objects.size()), p, cells->sdf);
  if (rc != VGT_HIP_OK)
  {
    (void)hipStreamSynchronize(ctx->stream);
    return rc;
  }
  return CopySdfToHost(ctx, cells->sdf, cells->nx * cells->ny * cells->nz, sdf_host, out_min, out_max);
}

int vgt_hip_cells_object_sdfs(vgt_hip_ctx* ctx, vgt_hip_cells* cells, const uint32_t* object_ids,
                              int64_t num_objects, double resolution, int unknown_is_filled, int add_virtual_border,
                              float* const* sdf_host, float* out_min, float* out_max)
{
  int rc = CheckCells(ctx, cells);
  if (rc != VGT_HIP_OK) return rc;
  if (num_objects < 0 || (num_objects > 0 && (!object_ids || !sdf_host)))
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  if (num_objects > 0 && cells->object_id_offset < 0)
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "this cell type carries no object id");
  for (int64_t b = 0; b < num_objects; b++)
    if (!sdf_host[b]) return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "null field in the batch");
  rc = CheckSdfShape(cells->nx, cells->ny, cells->nz, resolution);
  if (rc != VGT_HIP_OK) return rc;
  if (num_objects == 0) return VGT_HIP_OK;
  VGT_TRY_HIP(hipSetDevice(ctx->device), "set device");
  const int64_t nx = cells->nx, ny = cells->ny, nz = cells->nz;
  const size_t n = static_cast<size_t>(nx * ny * nz);
  // Objects per launch: what the limits of a batch and a memory budget allow (9.25 bytes per voxel and object: mask,
  // intermediate field, field, records).
  const int64_t group = BatchGroup(num_objects, nx, ny, nz, size_t{4} << 30, n * 10);
  std::lock_guard<std::mutex> lock(ctx->mutex);
  if (ctx->variant != vgt::EdtVariant::kDefault)
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "batches run on the default EDT pipeline only");
  hipStream_t s = ctx->stream;
  const size_t ws_bytes = CarveWorkspace(nullptr, nx, ny, nz, ctx->variant, group).bytes;
  const size_t sdf_bytes = static_cast<size_t>(group) * (n + 2) * sizeof(float);
  if (cells->batch_masks_bytes < static_cast<size_t>(group) * n || cells->batch_sdf_bytes < sdf_bytes ||
      cells->batch_ws_bytes < ws_bytes || cells->objects_capacity < static_cast<size_t>(group))
    VGT_TRY_HIP(hipStreamSynchronize(s), "drain before regrowing the batch buffers");
  VGT_TRY_HIP(Reserve(&cells->batch_masks, &cells->batch_masks_bytes, static_cast<size_t>(group) * n), "allocate masks");
  VGT_TRY_HIP(Reserve(&cells->batch_sdf, &cells->batch_sdf_bytes, sdf_bytes), "allocate fields");
  VGT_TRY_HIP(Reserve(&cells->batch_ws, &cells->batch_ws_bytes, ws_bytes), "allocate SDF workspace");
  if (cells->objects_capacity < static_cast<size_t>(group))
  {
    if (cells->objects) (void)hipFree(cells->objects);
    cells->objects = nullptr;
    cells->objects_capacity = 0;
    VGT_TRY_HIP(hipMalloc(reinterpret_cast<void**>(&cells->objects), static_cast<size_t>(group) * sizeof(uint32_t)),
                "allocate object list");
    cells->objects_capacity = static_cast<size_t>(group);
  }
  float* const sdf_dev = static_cast<float*>(cells->batch_sdf);
  float* const mm_dev = sdf_dev + static_cast<size_t>(group) * n;
  std::vector<float> mm(static_cast<size_t>(group) * 2);
  for (int64_t first = 0; first < num_objects; first += group)
  {
    const int64_t count = num_objects - first < group ? num_objects - first : group;
    // (pageable source: staged when the call returns)
    VGT_TRY_HIP(hipMemcpyAsync(cells->objects, object_ids + first, static_cast<size_t>(count) * sizeof(uint32_t),
                               hipMemcpyHostToDevice, s),
                "upload object ids");
    VGT_TRY_HIP(vgt::LaunchCellObjectMasks(cells->records, static_cast<int64_t>(n), cells->cell_bytes,
                                           cells->object_id_offset, cells->objects, static_cast<int>(count),
                                           unknown_is_filled ? 1 : 0, static_cast<uint8_t*>(cells->batch_masks), s),
                "object masks");
    vgt::SdfParams p{nx, ny, nz, resolution, 0, add_virtual_border ? 1 : 0};
    p.batch = count;
    rc = RunSdfPipeline<uint8_t>(ctx, static_cast<const uint8_t*>(cells->batch_masks), p, sdf_dev, cells->batch_ws,
                                 cells->batch_ws_bytes, mm_dev, nullptr);
    if (rc != VGT_HIP_OK)
    {
      (void)hipStreamSynchronize(s);
      return rc;
    }
    {
      // the fields go through the context's page-locked ring and on into the callers' arrays on several threads
      std::vector<HostArrayCopy> copies;
      for (int64_t b = 0; b < count; b++) copies.push_back(HostArrayCopy{sdf_dev + static_cast<size_t>(b) * n, sdf_host[first + b]});
      hipError_t err = hipMemcpyAsync(mm.data(), mm_dev, static_cast<size_t>(count) * 2 * sizeof(float), hipMemcpyDeviceToHost, s);
      const hipError_t moved = DownloadToHostArrays(ctx, copies, n * sizeof(float), s);  // (synchronises the stream)
      if (err == hipSuccess) err = moved;
      if (err != hipSuccess) (void)hipStreamSynchronize(s);
      VGT_TRY_HIP(err, "copy the objects' fields to the host");
    }
    for (int64_t b = 0; b < count; b++)
    {
      if (out_min) out_min[first + b] = mm[static_cast<size_t>(2 * b)];
      if (out_max) out_max[first + b] = mm[static_cast<size_t>(2 * b + 1)];
    }
  }
  return VGT_HIP_OK;
}

int vgt_hip_cells_free_and_named_objects_sdf(vgt_hip_ctx* ctx, vgt_hip_cells* cells, double resolution,
                                             int unknown_is_filled, int add_virtual_border, float* sdf_host,
                                             float* out_min, float* out_max)
{
  int rc = CheckCells(ctx, cells);
  if (rc != VGT_HIP_OK) return rc;
  if (!sdf_host) return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  if (cells->object_id_offset < 0)
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "this cell type carries no object id");
  rc = CheckSdfShape(cells->nx, cells->ny, cells->nz, resolution);
  if (rc != VGT_HIP_OK) return rc;
  VGT_TRY_HIP(hipSetDevice(ctx->device), "set device");
  const int64_t n = cells->nx * cells->ny * cells->nz;
  std::lock_guard<std::mutex> lock(ctx->mutex);
  if (!cells->sdf_named)
    VGT_TRY_HIP(hipMalloc(reinterpret_cast<void**>(&cells->sdf_named), static_cast<size_t>(n) * sizeof(float)),
                "allocate second SDF");
  const vgt::SdfParams p{cells->nx, cells->ny, cells->nz, resolution, unknown_is_filled ? 1 : 0,
                         add_virtual_border ? 1 : 0};
  rc = RunCellsSdf(ctx, cells, 0, 0, p, cells->sdf);                      // free-space field: every filled cell
  if (rc == VGT_HIP_OK) rc = RunCellsSdf(ctx, cells, 2, 0, p, cells->sdf_named);  // cells of named objects only
  if (rc == VGT_HIP_OK)
  {
    uint32_t* enc = CarveWorkspace(cells->workspace, cells->nx, cells->ny, cells->nz, ctx->variant).minmax_enc;
    hipError_t err = vgt::LaunchInitMinMax(enc, ctx->stream);
    if (err == hipSuccess)
      err = vgt::LaunchCombineFreeAndNamed(cells->sdf, cells->sdf_named, n, cells->sdf, enc, ctx->stream);
    if (err == hipSuccess) err = vgt::LaunchDecodeMinMax(enc, ctx->minmax_out, ctx->stream);
    if (err != hipSuccess) rc = FailHip("combine fields", err);
  }
  if (rc != VGT_HIP_OK)
  {
    (void)hipStreamSynchronize(ctx->stream);
    return rc;
  }
  return CopySdfToHost(ctx, cells->sdf, n, sdf_host, out_min, out_max);
}

/* ------------------------------ deferred timing ------------------------------ */

namespace
{
static void ReleaseTiming(vgt_hip_ctx* ctx)
{
  for (hipEvent_t e : ctx->timing_events)
    if (e) (void)hipEventDestroy(e);
  ctx->timing_events.clear();
  ctx->timing_kind.clear();
  ctx->timing_slots = 0;
  ctx->timing_used = 0;
}
}  // namespace

int vgt_hip_timing_start(vgt_hip_ctx* ctx, int32_t max_calls)
{
  if (!ctx || max_calls <= 0 || max_calls > 4096)
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "invalid timing capacity");
  VGT_TRY_HIP(hipSetDevice(ctx->device), "set device");
  std::lock_guard<std::mutex> lock(ctx->mutex);
  ReleaseTiming(ctx);
  ctx->timing_events.assign(static_cast<size_t>(max_calls) * 8, nullptr);
  ctx->timing_kind.assign(static_cast<size_t>(max_calls), 0);
  for (hipEvent_t& e : ctx->timing_events)
  {
    const hipError_t err = hipEventCreate(&e);
    if (err != hipSuccess)
    {
      ReleaseTiming(ctx);
      return FailHip("create event", err);
    }
  }
  ctx->timing_slots = max_calls;
  return VGT_HIP_OK;
}

int vgt_hip_timing_stop(vgt_hip_ctx* ctx, float* kernel_ms, int32_t* num_calls)
{
  if (!ctx || !kernel_ms || !num_calls) return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  VGT_TRY_HIP(hipSetDevice(ctx->device), "set device");
  std::lock_guard<std::mutex> lock(ctx->mutex);
  hipError_t err = hipStreamSynchronize(ctx->stream);
  *num_calls = ctx->timing_used;
  for (int i = 0; i < ctx->timing_used && err == hipSuccess; i++)
  {
    hipEvent_t* e = &ctx->timing_events[static_cast<size_t>(i) * 8];
    float* out = kernel_ms + 3 * i;
    float fix = 0.0f;
    if (ctx->timing_kind[static_cast<size_t>(i)] == 1)
    {
      err = hipEventElapsedTime(&out[0], e[0], e[1]);
      if (err == hipSuccess) err = hipEventElapsedTime(&out[1], e[1], e[2]);
      if (err == hipSuccess) err = hipEventElapsedTime(&out[2], e[2], e[3]);
    }
    else
    {
      err = hipEventElapsedTime(&out[0], e[0], e[1]);
      if (err == hipSuccess) err = hipEventElapsedTime(&fix, e[4], e[5]);
      if (err == hipSuccess) err = hipEventElapsedTime(&out[1], e[5], e[6]);
      if (err == hipSuccess) err = hipEventElapsedTime(&out[2], e[6], e[7]);
      out[0] += fix;  // the slab fix-up belongs to the Z pass
    }
  }
  ReleaseTiming(ctx);
  VGT_TRY_HIP(err, "read kernel timing");
  return VGT_HIP_OK;
}

/* ------------------------------- SDF consumers ------------------------------- */

int vgt_hip_sdf_coarse_gradient_dev(vgt_hip_ctx* ctx, const float* sdf_dev, int64_t nx, int64_t ny, int64_t nz,
                                    double resolution, int enable_edge_gradients, const double* rotation,
                                    double* gradient_dev, uint8_t* has_value_dev)
{
  if (!ctx || !sdf_dev || !gradient_dev) return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  const int rc = CheckSdfShape(nx, ny, nz, resolution);
  if (rc != VGT_HIP_OK) return rc;
  VGT_TRY_HIP(hipSetDevice(ctx->device), "set device");
  std::lock_guard<std::mutex> lock(ctx->mutex);
  VGT_TRY_HIP(vgt::LaunchCoarseGradient(sdf_dev, nx, ny, nz, resolution, enable_edge_gradients ? 1 : 0, rotation,
                                        gradient_dev, has_value_dev, ctx->stream),
              "coarse gradient");
  return VGT_HIP_OK;
}

int vgt_hip_sdf_coarse_gradient(vgt_hip_ctx* ctx, const float* sdf_host, int64_t nx, int64_t ny, int64_t nz,
                                double resolution, int enable_edge_gradients, const double* rotation,
                                double* gradient_host, uint8_t* has_value_host)
{
  if (!ctx || !sdf_host || !gradient_host) return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  const int rc = CheckSdfShape(nx, ny, nz, resolution);
  if (rc != VGT_HIP_OK) return rc;
  VGT_TRY_HIP(hipSetDevice(ctx->device), "set device");
  const size_t n = static_cast<size_t>(nx * ny * nz);
  float* sdf_dev = nullptr;
  double* grad_dev = nullptr;
  uint8_t* has_dev = nullptr;
  hipError_t err = hipMalloc(reinterpret_cast<void**>(&sdf_dev), n * sizeof(float));
  if (err == hipSuccess) err = hipMalloc(reinterpret_cast<void**>(&grad_dev), n * 3 * sizeof(double));
  if (err == hipSuccess && has_value_host) err = hipMalloc(reinterpret_cast<void**>(&has_dev), n);
  if (err == hipSuccess)
  {
    std::lock_guard<std::mutex> lock(ctx->mutex);
    err = hipMemcpyAsync(sdf_dev, sdf_host, n * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (err == hipSuccess)
      err = vgt::LaunchCoarseGradient(sdf_dev, nx, ny, nz, resolution, enable_edge_gradients ? 1 : 0, rotation,
                                      grad_dev, has_dev, ctx->stream);
    if (err == hipSuccess)
      err = hipMemcpyAsync(gradient_host, grad_dev, n * 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (err == hipSuccess && has_value_host)
      err = hipMemcpyAsync(has_value_host, has_dev, n, hipMemcpyDeviceToHost, ctx->stream);
    const hipError_t sync = hipStreamSynchronize(ctx->stream);
    if (err == hipSuccess) err = sync;
  }
  if (sdf_dev) (void)hipFree(sdf_dev);
  if (grad_dev) (void)hipFree(grad_dev);
  if (has_dev) (void)hipFree(has_dev);
  VGT_TRY_HIP(err, "coarse gradient");
  return VGT_HIP_OK;
}

int vgt_hip_sdf_estimate_distance_dev(vgt_hip_ctx* ctx, const float* sdf_dev, int64_t nx, int64_t ny, int64_t nz,
                                      double resolution, const double* grid_from_world, const double* query_xyz_dev,
                                      int64_t num_queries, double* distance_dev, uint8_t* has_value_dev)
{
  if (!ctx || !sdf_dev || !distance_dev || num_queries < 0 || (num_queries > 0 && !query_xyz_dev))
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  const int rc = CheckSdfShape(nx, ny, nz, resolution);
  if (rc != VGT_HIP_OK) return rc;
  VGT_TRY_HIP(hipSetDevice(ctx->device), "set device");
  std::lock_guard<std::mutex> lock(ctx->mutex);
  VGT_TRY_HIP(vgt::LaunchEstimateDistance(sdf_dev, nx, ny, nz, resolution, grid_from_world, query_xyz_dev, num_queries,
                                          distance_dev, has_value_dev, ctx->stream),
              "estimate distance");
  return VGT_HIP_OK;
}

int vgt_hip_sdf_estimate_distance(vgt_hip_ctx* ctx, const float* sdf_host, int64_t nx, int64_t ny, int64_t nz,
                                  double resolution, const double* grid_from_world, const double* query_xyz_host,
                                  int64_t num_queries, double* distance_host, uint8_t* has_value_host)
{
  return SdfQueriesHost(ctx, sdf_host, nx, ny, nz, resolution, query_xyz_host, num_queries, 1, distance_host,
                        has_value_host, "estimate distance",
                        [&](const float* sdf_dev, const double* queries_dev, double* out_dev, uint8_t* has_dev,
                            uint32_t*, hipStream_t s) {
                          return vgt::LaunchEstimateDistance(sdf_dev, nx, ny, nz, resolution, grid_from_world,
                                                             queries_dev, num_queries, out_dev, has_dev, s);
                        });
}

int vgt_hip_sdf_fine_gradient(vgt_hip_ctx* ctx, const float* sdf_host, int64_t nx, int64_t ny, int64_t nz,
                              double resolution, const double* grid_from_world, const double* query_xyz_host,
                              int64_t num_queries, double nominal_window_size, double* gradient_host,
                              uint8_t* has_value_host)
{
  if (!std::isfinite(nominal_window_size) || nominal_window_size == 0.0)
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "the fine-gradient window must be finite and non-zero");
  return SdfQueriesHost(ctx, sdf_host, nx, ny, nz, resolution, query_xyz_host, num_queries, 3, gradient_host,
                        has_value_host, "fine gradient",
                        [&](const float* sdf_dev, const double* queries_dev, double* out_dev, uint8_t* has_dev,
                            uint32_t* flag_dev, hipStream_t s) {
                          return vgt::LaunchFineGradient(sdf_dev, nx, ny, nz, resolution, grid_from_world, queries_dev,
                                                         num_queries, nominal_window_size, out_dev, has_dev, flag_dev,
                                                         s);
                        });
}

int vgt_hip_sdf_local_extrema_map_dev(vgt_hip_ctx* ctx, const float* sdf_dev, int64_t nx, int64_t ny, int64_t nz,
                                      double resolution, const double* rotation, double* extrema_dev)
{
  if (!ctx || !sdf_dev || !extrema_dev) return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  const int rc = CheckSdfShape(nx, ny, nz, resolution);
  if (rc != VGT_HIP_OK) return rc;
  if (nx * ny * nz >= 0x7fffffffLL)
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "the local extrema map supports grids below 2^31 cells");
  VGT_TRY_HIP(hipSetDevice(ctx->device), "set device");
  void* scratch = nullptr;
  VGT_TRY_HIP(hipMalloc(&scratch, vgt::LocalExtremaScratchBytes(nx * ny * nz)), "allocate extrema scratch");
  hipError_t err;
  {
    std::lock_guard<std::mutex> lock(ctx->mutex);
    err = vgt::LaunchLocalExtremaMap(sdf_dev, nx, ny, nz, resolution, rotation, extrema_dev, scratch, ctx->stream);
    const hipError_t sync = hipStreamSynchronize(ctx->stream);  // the scratch is freed below
    if (err == hipSuccess) err = sync;
  }
  (void)hipFree(scratch);
  VGT_TRY_HIP(err, "local extrema map");
  return VGT_HIP_OK;
}

int vgt_hip_sdf_local_extrema_map(vgt_hip_ctx* ctx, const float* sdf_host, int64_t nx, int64_t ny, int64_t nz,
                                  double resolution, const double* rotation, double* extrema_host)
{
  if (!ctx || !sdf_host || !extrema_host) return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  const int rc = CheckSdfShape(nx, ny, nz, resolution);
  if (rc != VGT_HIP_OK) return rc;
  VGT_TRY_HIP(hipSetDevice(ctx->device), "set device");
  const size_t n = static_cast<size_t>(nx * ny * nz);
  float* sdf_dev = nullptr;
  double* out_dev = nullptr;
  hipError_t err = hipMalloc(reinterpret_cast<void**>(&sdf_dev), n * sizeof(float));
  if (err == hipSuccess) err = hipMalloc(reinterpret_cast<void**>(&out_dev), n * 3 * sizeof(double));
  int result = VGT_HIP_OK;
  if (err == hipSuccess)
  {
    {
      std::lock_guard<std::mutex> lock(ctx->mutex);
      err = hipMemcpyAsync(sdf_dev, sdf_host, n * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    }
    if (err == hipSuccess)
      result = vgt_hip_sdf_local_extrema_map_dev(ctx, sdf_dev, nx, ny, nz, resolution, rotation, out_dev);
    if (err == hipSuccess && result == VGT_HIP_OK)
    {
      std::lock_guard<std::mutex> lock(ctx->mutex);
      err = hipMemcpyAsync(extrema_host, out_dev, n * 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
      const hipError_t sync = hipStreamSynchronize(ctx->stream);
      if (err == hipSuccess) err = sync;
    }
  }
  if (sdf_dev) (void)hipFree(sdf_dev);
  if (out_dev) (void)hipFree(out_dev);
  if (result != VGT_HIP_OK) return result;
  VGT_TRY_HIP(err, "local extrema map");
  return VGT_HIP_OK;
}

/* --------------------------------- multi-GPU --------------------------------- */

size_t vgt_hip_sdf_slab_summary_bytes(int64_t nx, int64_t ny)
{
  if (nx <= 0 || ny <= 0) return 0;
  return static_cast<size_t>(nx) * static_cast<size_t>(ny) * sizeof(vgt::SlabLineSummary);
}

namespace
{
// Times `count` consecutive kernel groups with events when kernel_ms is requested.
struct StageTimer
{
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  int n = 0;
  bool enabled = false;
  int Init(bool on, int stages)
  {
    enabled = on;
    if (!on) return VGT_HIP_OK;
    for (int i = 0; i <= stages; i++)
    {
      const hipError_t err = hipEventCreate(&ev[i]);
      if (err != hipSuccess) return FailHip("create event", err);
      n = i + 1;
    }
    return VGT_HIP_OK;
  }
  hipError_t Mark(int i, hipStream_t s) { return enabled ? hipEventRecord(ev[i], s) : hipSuccess; }
  int Finish(hipStream_t s, float* out, int stages)
  {
    if (!enabled) return VGT_HIP_OK;
    hipError_t err = hipStreamSynchronize(s);
    for (int i = 0; i < stages && err == hipSuccess; i++)
      err = hipEventElapsedTime(&out[i], ev[i], ev[i + 1]);
    return err == hipSuccess ? VGT_HIP_OK : FailHip("stage timing", err);
  }
  ~StageTimer()
  {
    for (int i = 0; i < n; i++)
      if (ev[i]) (void)hipEventDestroy(ev[i]);
  }
};
}  // namespace

int vgt_hip_sdf_slab_begin_dev(vgt_hip_ctx* ctx, const float* occupancy_dev, int64_t nx, int64_t ny,
                               int64_t nz_local, int64_t z_offset, int unknown_is_filled,
                               void* workspace_dev, size_t workspace_bytes, void* summary_dev,
                               float* kernel_ms)
{
  if (!ctx || !occupancy_dev || !workspace_dev || !summary_dev)
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  const int rc = CheckSdfShape(nx, ny, nz_local, 1.0);
  if (rc != VGT_HIP_OK) return rc;
  if (z_offset < 0 || z_offset + nz_local > vgt::kMaxExtent)
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "slab lies outside the supported Z extent");
  VGT_TRY_HIP(hipSetDevice(ctx->device), "set device");
  vgt::SdfParams p{nx, ny, nz_local, 1.0, unknown_is_filled ? 1 : 0, 0};
  p.z_offset = z_offset;
  const SdfWorkspace ws = CarveWorkspace(workspace_dev, nx, ny, nz_local, ctx->variant);
  if (workspace_bytes < ws.bytes) return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "SDF workspace too small");
  StageTimer timer;
  const int trc = timer.Init(kernel_ms != nullptr, 1);
  if (trc != VGT_HIP_OK) return trc;
  std::lock_guard<std::mutex> lock(ctx->mutex);
  hipEvent_t* slot = kernel_ms ? nullptr : TimingSlot(ctx);
  VGT_TRY_HIP(timer.Mark(0, ctx->stream), "event record");
  if (slot) VGT_TRY_HIP(hipEventRecord(slot[0], ctx->stream), "event record");
  VGT_TRY_HIP(LaunchPassOne<float>(occupancy_dev, ws, p, 0, static_cast<vgt::SlabLineSummary*>(summary_dev), ctx->stream),
              "pass 1");
  if (slot) VGT_TRY_HIP(hipEventRecord(slot[1], ctx->stream), "event record");
  VGT_TRY_HIP(timer.Mark(1, ctx->stream), "event record");
  return timer.Finish(ctx->stream, kernel_ms, 1);
}

int vgt_hip_sdf_slab_range(int64_t nz_global, int32_t world, int32_t rank, int64_t* z_offset, int64_t* nz_local)
{
  if (!z_offset || !nz_local) return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  if (nz_global <= 0 || world <= 0 || world > nz_global || rank < 0 || rank >= world)
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "invalid slab partition");
  vgt::SlabRange(nz_global, world, rank, z_offset, nz_local);
  return VGT_HIP_OK;
}

size_t vgt_hip_sdf_slab_carries_bytes(int64_t nx, int64_t ny)
{
  if (nx <= 0 || ny <= 0) return 0;
  return static_cast<size_t>(nx) * static_cast<size_t>(ny) * sizeof(vgt::SlabLineCarry);
}

int vgt_hip_sdf_slab_carries_dev(vgt_hip_ctx* ctx, const void* gathered_summaries_dev, int32_t world,
                                 int32_t rank, int64_t nx, int64_t ny, int64_t nz_global, void* carries_dev)
{
  if (!ctx || !gathered_summaries_dev || !carries_dev)
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  if (world <= 0 || rank < 0 || rank >= world || nx <= 0 || ny <= 0 || nz_global < world ||
      nz_global > vgt::kMaxExtent)
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "invalid slab partition");
  VGT_TRY_HIP(hipSetDevice(ctx->device), "set device");
  std::lock_guard<std::mutex> lock(ctx->mutex);
  VGT_TRY_HIP(vgt::LaunchSlabCarries(static_cast<const vgt::SlabLineSummary*>(gathered_summaries_dev), world, rank,
                                     nx * ny, nz_global, static_cast<vgt::SlabLineCarry*>(carries_dev), ctx->stream),
              "slab carries");
  // the carries hold for the slab that vgt_hip_sdf_slab_range gives this rank, and for no other
  vgt_hip_ctx::SlabNote note{0, 0, nz_global};
  vgt::SlabRange(nz_global, world, rank, &note.z_offset, &note.nz_local);
  if (ctx->slab_notes.size() > 64) ctx->slab_notes.clear();
  ctx->slab_notes[carries_dev] = note;
  return VGT_HIP_OK;
}

int vgt_hip_sdf_slab_finish_dev(vgt_hip_ctx* ctx, int64_t nx, int64_t ny, int64_t nz_local,
                                int64_t z_offset, int64_t nz_global, double resolution,
                                int add_virtual_border, const void* carries_dev, float* sdf_dev,
                                void* workspace_dev, size_t workspace_bytes, float* minmax_dev,
                                float* kernel_ms)
{
  if (!ctx || !carries_dev || !sdf_dev || !workspace_dev)
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  const int rc = CheckSdfShape(nx, ny, nz_local, resolution);
  if (rc != VGT_HIP_OK) return rc;
  if (z_offset < 0 || nz_global < z_offset + nz_local || nz_global > vgt::kMaxExtent)
    return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "slab lies outside the global Z extent");
  VGT_TRY_HIP(hipSetDevice(ctx->device), "set device");
  vgt::SdfParams p{nx, ny, nz_local, resolution, 0, add_virtual_border ? 1 : 0};
  p.z_offset = z_offset;
  p.nz_global = nz_global;
  const SdfWorkspace ws = CarveWorkspace(workspace_dev, nx, ny, nz_local, ctx->variant);
  if (workspace_bytes < ws.bytes) return Fail(VGT_HIP_ERR_INVALID_ARGUMENT, "SDF workspace too small");
  StageTimer timer;
  const int trc = timer.Init(kernel_ms != nullptr, 3);
  if (trc != VGT_HIP_OK) return trc;
  std::lock_guard<std::mutex> lock(ctx->mutex);
  {
    const auto note = ctx->slab_notes.find(carries_dev);
    if (note != ctx->slab_notes.end())
    {
      const bool other = note->second.z_offset != z_offset || note->second.nz_local != nz_local ||
                         note->second.nz_global != nz_global;
      // (consumed either way: a later buffer at the same address -- a caching allocator's reuse -- starts without a note)
      ctx->slab_notes.erase(note);
      if (other)
        return Fail(VGT_HIP_ERR_INVALID_ARGUMENT,
                    "the carries were computed for another slab: slabs must be the ranges of vgt_hip_sdf_slab_range");
    }
  }
  hipStream_t s = ctx->stream;
  hipEvent_t* slot = kernel_ms ? nullptr : TimingSlot(ctx);
  VGT_TRY_HIP(vgt::LaunchInitMinMax(ws.minmax_enc, s), "init min/max");
  VGT_TRY_HIP(timer.Mark(0, s), "event record");
  if (slot) VGT_TRY_HIP(hipEventRecord(slot[4], s), "event record");
  if (ws.records)
    VGT_TRY_HIP(vgt::LaunchSlabRecordFixup(ws.records, static_cast<const vgt::SlabLineCarry*>(carries_dev), p, s),
                "slab fix-up");
#ifdef VGT_HIP_TESTING
  else
    VGT_TRY_HIP(vgt::LaunchSlabFixup(ws.t16, static_cast<const vgt::SlabLineCarry*>(carries_dev), p, s), "slab fix-up");
#endif
  VGT_TRY_HIP(timer.Mark(1, s), "event record");
  if (slot) VGT_TRY_HIP(hipEventRecord(slot[5], s), "event record");
  VGT_TRY_HIP(LaunchPassTwo(ws, p, 0, ctx->variant, s), "Y pass");
  VGT_TRY_HIP(timer.Mark(2, s), "event record");
  if (slot) VGT_TRY_HIP(hipEventRecord(slot[6], s), "event record");
  VGT_TRY_HIP(vgt::LaunchPassXFinalize(ws.t32, sdf_dev, ws.minmax_enc, ws.sweep_scratch, p, ctx->variant, s), "X pass");
  VGT_TRY_HIP(timer.Mark(3, s), "event record");
  if (slot)
  {
    VGT_TRY_HIP(hipEventRecord(slot[7], s), "event record");
    ctx->timing_kind[static_cast<size_t>(ctx->timing_used++)] = 2;
  }
  if (minmax_dev) VGT_TRY_HIP(vgt::LaunchDecodeMinMax(ws.minmax_enc, minmax_dev, s), "min/max");
  return timer.Finish(s, kernel_ms, 3);
}

}  // extern "C"
