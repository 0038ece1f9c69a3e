// vgt_hipx_sdf_multi: one process drives N devices through the Z-slab pipeline of vgt_hip.h
// (slab scan + per-line summaries -> ONE all-gather -> carries -> fix-up -> Y / X passes).
// This is the large-grid branch of OccupancyMap::ExtractSignedDistanceFieldFloat
// (S/occupancy_map.cpp:256-260): the caller of the reference is a single process with the grid
// in host memory, so the N > 1 path has to be reachable from C++ without torch.distributed.
//
// Data movement: the slab of device r is occupancy[:, :, z0_r : z1_r] -- nx*ny rows of nzl floats at
// a pitch of nz floats in the caller's array -- moved by hipMemcpy2DAsync on the device's own
// stream (the caller's arrays are page-locked with hipHostRegister for the duration of the call
// when the driver allows it), so the N uploads, the N pipelines and the N downloads overlap.
// Exchange: rccl's ncclAllGather (one call per device inside a group, 4 bytes per (x, y) line and
// slab) when the devices are distinct; when one device appears more than once in `devices`
// (several slabs on one GPU: the single-GPU test of this path, or a grid that does not fit one
// GPU's workspace in one piece) rccl cannot form a communicator and the summaries are copied
// slab to slab with hipMemcpyPeerAsync instead.  Communicators are formed once per device list and kept
// for the life of the process (CommSet below); so are the per-slab contexts, streams and device buffers of the
// last (device list, grid shape) served (SlabSet below).  The field's extrema are reduced on the host: this
// process already holds every device's (min, max) pair.
//
// vgt_hipx_raycast_points_split (end of the file): ONE point cloud over several devices -- contiguous shares of
// the points, a private tracking grid per device, then the grids are summed onto the caller's device (the
// counts are integers: the sum is the whole cloud's counts whatever the split).
#include "../../include/vgt_hip.h"

#include "vgt_internal.hpp"
#include "host_pages.hpp"

#include <rccl/rccl.h>  // types and prototypes only: the library itself is loaded on first use (below)

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace
{
// librccl is half a gigabyte of code objects: linking it would make every user of libvgt_hip.so -- also the
// single-GPU ones -- load and register it at start-up.  It is opened when the first multi-device extraction
// with distinct devices asks for a communicator.
struct Rccl
{
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclReduce) Reduce = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string error;  // why it is unavailable (empty = loaded)
};

const Rccl& GetRccl()
{
  static const Rccl api = [] {
    Rccl r;
    void* lib = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
    {
      lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (lib) break;
    }
    if (!lib)
    {
      r.error = std::string("librccl.so.1 could not be loaded: ") + dlerror();
      return r;
    }
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(lib, "ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(lib, "ncclAllGather"));
    r.Reduce = reinterpret_cast<decltype(r.Reduce)>(dlsym(lib, "ncclReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
    if (!r.CommInitAll || !r.CommDestroy || !r.GroupStart || !r.GroupEnd || !r.AllGather || !r.Reduce ||
        !r.GetErrorString)
      r.error = "librccl.so.1 lacks a required entry point";
    return r;
  }();
  return api;
}

// Communicators are expensive to form (hundreds of milliseconds for a clique), so one set per ordered
// device list is kept for the life of the process (never destroyed: rccl may already be unloading when
// static destructors run).  A communicator serves one collective at a time; `in_use` is held from the
// group that enqueues the all-gather until the streams have drained.
struct CommSet
{
  std::vector<ncclComm_t> comms;
  std::mutex in_use;
};

CommSet* GetCommSet(const Rccl& rccl, const std::vector<int>& devices, std::string* error)
{
  static std::mutex registry_lock;
  static auto* registry = new std::map<std::vector<int>, CommSet*>();
  std::lock_guard<std::mutex> guard(registry_lock);
  auto found = registry->find(devices);
  if (found != registry->end()) return found->second;
  auto* set = new CommSet();
  set->comms.assign(devices.size(), nullptr);
  const ncclResult_t res = rccl.CommInitAll(set->comms.data(), static_cast<int>(devices.size()), devices.data());
  if (res != ncclSuccess)
  {
    *error = std::string("[ncclCommInitAll] RCCL error [") + rccl.GetErrorString(res) + "]";
    delete set;
    return nullptr;
  }
  (*registry)[devices] = set;
  return set;
}

struct Slab
{
  int device = -1;
  int64_t z0 = 0, nzl = 0;
  vgt_hip_ctx* ctx = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t summary_ready = nullptr;
  hipEvent_t marks[4] = {nullptr, nullptr, nullptr, nullptr};  // start, uploaded, computed, downloaded
  float* occ = nullptr;
  float* sdf = nullptr;
  void* workspace = nullptr;
  size_t workspace_bytes = 0;
  void* summary = nullptr;
  void* gathered = nullptr;
  void* carries = nullptr;
  float* minmax = nullptr;
  float minmax_host[2] = {0.0f, 0.0f};
};

// Everything a (device list, grid shape) needs on the devices: contexts, streams, events and buffers of every
// slab.  Built on the first extraction with that key and kept for the following ones -- a context, a stream and
// seven hipMallocs per slab every call cost more than the pipeline itself on grids of a few hundred MiB -- until
// another key displaces it (one set at a time: the buffers are the size of the grid) or vgt_hipx_release is called.
struct SlabSet
{
  std::vector<int> devices;
  int64_t nx = 0, ny = 0, nz = 0;
  std::vector<Slab> slabs;

  ~SlabSet()
  {
    for (Slab& s : slabs)
    {
      if (s.device < 0) continue;
      (void)hipSetDevice(s.device);
      if (s.stream) (void)hipStreamSynchronize(s.stream);
    }
    for (Slab& s : slabs)
    {
      if (s.device < 0) continue;
      (void)hipSetDevice(s.device);
      if (s.ctx) vgt_hip_destroy(s.ctx);  // drains the stream it was given first
      if (s.summary_ready) (void)hipEventDestroy(s.summary_ready);
      for (hipEvent_t e : s.marks)
        if (e) (void)hipEventDestroy(e);
      if (s.stream) (void)hipStreamDestroy(s.stream);
      for (void* p : {static_cast<void*>(s.occ), static_cast<void*>(s.sdf), s.workspace, s.summary, s.gathered,
                      s.carries, static_cast<void*>(s.minmax)})
        if (p) (void)hipFree(p);
    }
  }
};

std::mutex g_set_lock;        // held for the whole of an extraction: one at a time per process
SlabSet* g_set = nullptr;     // the cached set (guarded by g_set_lock)
float g_last_timing[5] = {0, 0, 0, 0, 0};

// Per call: the page-locking of the caller's arrays and the communicator lock; drains the streams before either goes.
struct CallState
{
  SlabSet* set = nullptr;
  std::unique_lock<std::mutex> comm_lock;  // the communicator set, held until the streams have drained
  bool registered_in = false, registered_out = false;
  const void* host_in = nullptr;
  void* host_out = nullptr;

  ~CallState()
  {
    if (set)
      for (Slab& s : set->slabs)
      {
        if (s.device < 0) continue;
        (void)hipSetDevice(s.device);
        if (s.stream) (void)hipStreamSynchronize(s.stream);
      }
    if (comm_lock.owns_lock()) comm_lock.unlock();
    if (registered_in) (void)hipHostUnregister(const_cast<void*>(host_in));
    if (registered_out) (void)hipHostUnregister(host_out);
  }
};

int FailMulti(int code, const std::string& msg)
{
  vgt::SetLastError(msg);
  return code;
}

#define VGTX_HIP(expr, what)                                                                           \
  do                                                                                                   \
  {                                                                                                    \
    const hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess)                                                                              \
      return FailMulti(VGT_HIP_ERR_RUNTIME, std::string("[") + (what) + "] HIP error [" + hipGetErrorString(e_) + "]"); \
  } while (0)
#define VGTX_NCCL(expr, what)                                                                          \
  do                                                                                                   \
  {                                                                                                    \
    const ncclResult_t r_ = (expr);                                                                    \
    if (r_ != ncclSuccess)                                                                             \
      return FailMulti(VGT_HIP_ERR_RUNTIME, std::string("[") + (what) + "] RCCL error [" + GetRccl().GetErrorString(r_) + "]"); \
  } while (0)
#define VGTX_CALL(expr)                  \
  do                                     \
  {                                      \
    const int rc_ = (expr);              \
    if (rc_ != VGT_HIP_OK) return rc_;   \
  } while (0)

// Contexts, streams, events and buffers of every slab of `set` (whose key and slab ranges are filled in).
int BuildSlabSet(SlabSet* set)
{
  const size_t lines = static_cast<size_t>(set->nx * set->ny);
  const size_t record_bytes = vgt_hip_sdf_slab_summary_bytes(set->nx, set->ny);   // 4 bytes per line
  const size_t carries_bytes = vgt_hip_sdf_slab_carries_bytes(set->nx, set->ny);  // 8 bytes per line
  const int world = static_cast<int>(set->slabs.size());
  for (Slab& s : set->slabs)
  {
    VGTX_HIP(hipSetDevice(s.device), "set device");
    VGTX_CALL(vgt_hip_create(s.device, -1, &s.ctx));
    VGTX_HIP(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking), "create stream");
    VGTX_HIP(hipEventCreateWithFlags(&s.summary_ready, hipEventDisableTiming), "create event");
    for (hipEvent_t& e : s.marks) VGTX_HIP(hipEventCreate(&e), "create event");
    VGTX_CALL(vgt_hip_set_stream(s.ctx, s.stream));
    const size_t slab_voxels = lines * static_cast<size_t>(s.nzl);
    s.workspace_bytes = vgt_hip_sdf_workspace_bytes(set->nx, set->ny, s.nzl);
    VGTX_HIP(hipMalloc(reinterpret_cast<void**>(&s.occ), slab_voxels * sizeof(float)), "allocate slab occupancy");
    VGTX_HIP(hipMalloc(reinterpret_cast<void**>(&s.sdf), slab_voxels * sizeof(float)), "allocate slab SDF");
    VGTX_HIP(hipMalloc(&s.workspace, s.workspace_bytes), "allocate slab workspace");
    VGTX_HIP(hipMalloc(&s.summary, record_bytes), "allocate slab summary");
    VGTX_HIP(hipMalloc(&s.gathered, record_bytes * world), "allocate gathered summaries");
    VGTX_HIP(hipMalloc(&s.carries, carries_bytes), "allocate slab carries");
    VGTX_HIP(hipMalloc(reinterpret_cast<void**>(&s.minmax), 256), "allocate extrema");
  }
  return VGT_HIP_OK;
}

// The helper devices of vgt_hipx_raycast_points_split: per helper a context, a stream for the reduction and one
// private tracking grid; `landing` is a buffer on the caller's device that a helper's grid is copied to when rccl
// cannot do the sum (a device listed twice).  Kept for the next call with the same (caller device, helper list,
// cell count), like SlabSet.
struct RayHelper
{
  int device = -1;
  vgt_hip_ctx* ctx = nullptr;
  vgt_hip_grids* grids = nullptr;
  hipStream_t stream = nullptr;
};

struct RaySet
{
  int primary_device = -1;
  std::vector<int> helper_devices;
  int64_t num_cells = 0;
  std::vector<RayHelper> helpers;
  int32_t* landing = nullptr;

  ~RaySet()
  {
    for (RayHelper& h : helpers)
    {
      if (h.device < 0) continue;
      (void)hipSetDevice(h.device);
      if (h.stream) (void)hipStreamSynchronize(h.stream);
      if (h.grids) vgt_hip_tracking_grids_destroy(h.grids);
      if (h.ctx) vgt_hip_destroy(h.ctx);
      if (h.stream) (void)hipStreamDestroy(h.stream);
    }
    if (landing)
    {
      (void)hipSetDevice(primary_device);
      (void)hipFree(landing);
    }
  }
};

std::mutex g_ray_lock;       // held for the whole of a split raycast: one at a time per process
RaySet* g_ray_set = nullptr;  // guarded by g_ray_lock
}  // namespace

extern "C" void vgt_hipx_release(void)
{
  {
    std::lock_guard<std::mutex> guard(g_set_lock);
    delete g_set;
    g_set = nullptr;
  }
  std::lock_guard<std::mutex> guard(g_ray_lock);
  delete g_ray_set;
  g_ray_set = nullptr;
}

extern "C" int vgt_hipx_last_timing(float* ms5)
{
  if (!ms5) return FailMulti(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  std::lock_guard<std::mutex> guard(g_set_lock);
  for (int i = 0; i < 5; i++) ms5[i] = g_last_timing[i];
  return VGT_HIP_OK;
}

extern "C" int vgt_hipx_sdf_multi(const int* devices, int num_devices, const float* occupancy_host, int64_t nx,
                                  int64_t ny, int64_t nz, double resolution, int unknown_is_filled,
                                  int add_virtual_border, float* sdf_host, float* out_min, float* out_max)
{
  if (!devices || !occupancy_host || !sdf_host) return FailMulti(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  if (num_devices <= 0) return FailMulti(VGT_HIP_ERR_INVALID_ARGUMENT, "num_devices must be > 0");
  if (nx <= 0 || ny <= 0 || nz <= 0) return FailMulti(VGT_HIP_ERR_INVALID_ARGUMENT, "grid extents must be positive");
  if (nx > vgt::kMaxExtent || ny > vgt::kMaxExtent || nz > vgt::kMaxExtent)
    return FailMulti(VGT_HIP_ERR_INVALID_ARGUMENT, "grid extent exceeds 16384 voxels on an axis");
  if (!(resolution > 0.0) || !std::isfinite(resolution))
    return FailMulti(VGT_HIP_ERR_INVALID_ARGUMENT, "Grid must have uniform, positive resolution");
  int device_count = 0;
  if (hipGetDeviceCount(&device_count) != hipSuccess || device_count <= 0)
    return FailMulti(VGT_HIP_ERR_UNAVAILABLE, "no usable HIP device (libvgt_hip has no CPU fallback)");
  for (int i = 0; i < num_devices; i++)
    if (devices[i] < 0 || devices[i] >= device_count)
      return FailMulti(VGT_HIP_ERR_UNAVAILABLE, "device " + std::to_string(devices[i]) + " out of range for " +
                                                    std::to_string(device_count) + " devices");
  const auto t_begin = std::chrono::steady_clock::now();

  // slabs along Z: vgt_hip_sdf_slab_range (equal shares, earlier slabs take the remainder = multi_gpu.slab_bounds)
  const int world = static_cast<int>(std::min<int64_t>(num_devices, nz));
  const std::vector<int> devs(devices, devices + world);
  std::lock_guard<std::mutex> set_guard(g_set_lock);
  if (g_set && !(g_set->devices == devs && g_set->nx == nx && g_set->ny == ny && g_set->nz == nz))
  {
    delete g_set;  // another grid shape or device list: its buffers make room
    g_set = nullptr;
  }
  if (!g_set)
  {
    auto* fresh = new SlabSet();
    fresh->devices = devs;
    fresh->nx = nx;
    fresh->ny = ny;
    fresh->nz = nz;
    fresh->slabs.resize(static_cast<size_t>(world));
    for (int r = 0; r < world; r++)
    {
      fresh->slabs[r].device = devs[r];
      vgt::SlabRange(nz, world, r, &fresh->slabs[r].z0, &fresh->slabs[r].nzl);
    }
    const int rc = BuildSlabSet(fresh);
    if (rc != VGT_HIP_OK)
    {
      delete fresh;
      return rc;
    }
    g_set = fresh;
  }
  SlabSet& set = *g_set;
  CallState st;
  st.set = &set;
  bool distinct = true;
  for (int a = 0; a < world; a++)
    for (int b = a + 1; b < world; b++)
      if (set.slabs[a].device == set.slabs[b].device) distinct = false;

  const size_t lines = static_cast<size_t>(nx * ny);
  const size_t record_bytes = vgt_hip_sdf_slab_summary_bytes(nx, ny);
  const size_t total_bytes = static_cast<size_t>(nx * ny * nz) * sizeof(float);
  // page-lock the caller's arrays so that the strided slab copies are true asynchronous DMA (best effort; a few
  // milliseconds per GiB, measured in bench.py's host_path.register_ms)
  st.host_in = occupancy_host;
  st.host_out = sdf_host;
  st.registered_in = hipHostRegister(const_cast<float*>(occupancy_host), total_bytes, hipHostRegisterPortable) == hipSuccess;
  (void)hipGetLastError();
  // the output array is not needed before the first download: it is page-locked by a helper thread while the
  // uploads run (joined before the first download is enqueued)
  // (the thread writes the call state itself: whichever way this function returns, the joiner below joins the thread
  // before the state's destructor looks at the flag and unlocks the pages)
  // (an array the caller has just allocated has no pages yet: they are faulted in by several threads first -- page-locking
  // fresh memory does it on one, 0.3 s for 4 GiB; host_pages.hpp)
  std::thread pin_out([&st, sdf_host, total_bytes] {
    {
      vgt::HostRangePopulator fresh(sdf_host, total_bytes);
      fresh.Wait();
    }
    st.registered_out = hipHostRegister(sdf_host, total_bytes, hipHostRegisterPortable) == hipSuccess;
  });
  struct Joiner
  {
    std::thread& t;
    ~Joiner()
    {
      if (t.joinable()) t.join();
    }
  } joiner{pin_out};
  const auto t_setup = std::chrono::steady_clock::now();

  // per slab: upload, slab scan + summary
  for (Slab& s : set.slabs)
  {
    VGTX_HIP(hipSetDevice(s.device), "set device");
    VGTX_HIP(hipEventRecord(s.marks[0], s.stream), "record event");
    VGTX_HIP(hipMemcpy2DAsync(s.occ, static_cast<size_t>(s.nzl) * sizeof(float), occupancy_host + s.z0,
                              static_cast<size_t>(nz) * sizeof(float), static_cast<size_t>(s.nzl) * sizeof(float), lines,
                              hipMemcpyHostToDevice, s.stream),
             "copy slab occupancy to device");
    VGTX_HIP(hipEventRecord(s.marks[1], s.stream), "record event");
    VGTX_CALL(vgt_hip_sdf_slab_begin_dev(s.ctx, s.occ, nx, ny, s.nzl, s.z0, unknown_is_filled, s.workspace,
                                         s.workspace_bytes, s.summary, nullptr));
    VGTX_HIP(hipEventRecord(s.summary_ready, s.stream), "record event");
  }

  // the one exchange: every slab receives every slab's per-line summary
  if (distinct)
  {
    const Rccl& rccl = GetRccl();
    if (!rccl.error.empty()) return FailMulti(VGT_HIP_ERR_UNAVAILABLE, rccl.error);
    std::string comm_error;
    CommSet* const comms = GetCommSet(rccl, devs, &comm_error);
    if (!comms) return FailMulti(VGT_HIP_ERR_RUNTIME, comm_error);
    st.comm_lock = std::unique_lock<std::mutex>(comms->in_use);
    VGTX_NCCL(rccl.GroupStart(), "ncclGroupStart");
    for (int r = 0; r < world; r++)
    {
      Slab& s = set.slabs[r];
      // 4-byte records moved as int32 words; rank r's block lands at gathered + r * record_bytes
      const ncclResult_t res = rccl.AllGather(s.summary, s.gathered, record_bytes / sizeof(int32_t), ncclInt32,
                                              comms->comms[r], s.stream);
      if (res != ncclSuccess)
      {
        (void)rccl.GroupEnd();
        return FailMulti(VGT_HIP_ERR_RUNTIME,
                         std::string("[ncclAllGather] RCCL error [") + rccl.GetErrorString(res) + "]");
      }
    }
    VGTX_NCCL(rccl.GroupEnd(), "ncclGroupEnd");
  }
  else
  {
    for (int r = 0; r < world; r++)
    {
      Slab& dst = set.slabs[r];
      VGTX_HIP(hipSetDevice(dst.device), "set device");
      for (int q = 0; q < world; q++)
      {
        Slab& src = set.slabs[q];
        VGTX_HIP(hipStreamWaitEvent(dst.stream, src.summary_ready, 0), "wait for summary");
        char* to = static_cast<char*>(dst.gathered) + static_cast<size_t>(q) * record_bytes;
        if (src.device == dst.device)
          VGTX_HIP(hipMemcpyAsync(to, src.summary, record_bytes, hipMemcpyDeviceToDevice, dst.stream),
                   "copy summary between slabs");
        else
          VGTX_HIP(hipMemcpyPeerAsync(to, dst.device, src.summary, src.device, record_bytes, dst.stream),
                   "copy summary between slabs");
      }
    }
  }

  // per slab: carries, fix-up, Y and X passes, download
  pin_out.join();
  (void)hipGetLastError();
  for (int r = 0; r < world; r++)
  {
    Slab& s = set.slabs[r];
    VGTX_HIP(hipSetDevice(s.device), "set device");
    VGTX_CALL(vgt_hip_sdf_slab_carries_dev(s.ctx, s.gathered, world, r, nx, ny, nz, s.carries));
    VGTX_CALL(vgt_hip_sdf_slab_finish_dev(s.ctx, nx, ny, s.nzl, s.z0, nz, resolution, add_virtual_border, s.carries,
                                          s.sdf, s.workspace, s.workspace_bytes, s.minmax, nullptr));
    VGTX_HIP(hipEventRecord(s.marks[2], s.stream), "record event");
    VGTX_HIP(hipMemcpy2DAsync(sdf_host + s.z0, static_cast<size_t>(nz) * sizeof(float), s.sdf,
                              static_cast<size_t>(s.nzl) * sizeof(float), static_cast<size_t>(s.nzl) * sizeof(float), lines,
                              hipMemcpyDeviceToHost, s.stream),
             "copy slab SDF to host");
    VGTX_HIP(hipEventRecord(s.marks[3], s.stream), "record event");
    VGTX_HIP(hipMemcpyAsync(s.minmax_host, s.minmax, 2 * sizeof(float), hipMemcpyDeviceToHost, s.stream),
             "copy extrema to host");
  }
  float lo = INFINITY, hi = -INFINITY;
  float upload_ms = 0.0f, compute_ms = 0.0f, download_ms = 0.0f;
  for (Slab& s : set.slabs)
  {
    VGTX_HIP(hipSetDevice(s.device), "set device");
    VGTX_HIP(hipStreamSynchronize(s.stream), "wait for slab");
    lo = std::min(lo, s.minmax_host[0]);
    hi = std::max(hi, s.minmax_host[1]);
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, s.marks[0], s.marks[1]) == hipSuccess) upload_ms = std::max(upload_ms, ms);
    if (hipEventElapsedTime(&ms, s.marks[1], s.marks[2]) == hipSuccess) compute_ms = std::max(compute_ms, ms);
    if (hipEventElapsedTime(&ms, s.marks[2], s.marks[3]) == hipSuccess) download_ms = std::max(download_ms, ms);
  }
  if (out_min) *out_min = lo;
  if (out_max) *out_max = hi;
  const auto t_end = std::chrono::steady_clock::now();
  g_last_timing[0] = std::chrono::duration<float, std::milli>(t_setup - t_begin).count();
  g_last_timing[1] = upload_ms;
  g_last_timing[2] = compute_ms;
  g_last_timing[3] = download_ms;
  g_last_timing[4] = std::chrono::duration<float, std::milli>(t_end - t_begin).count();
  return VGT_HIP_OK;
}

extern "C" int vgt_hipx_point_share(int64_t num_points, int32_t shares, int32_t share, int64_t* first, int64_t* count)
{
  if (!first || !count) return FailMulti(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  *first = *count = 0;
  if (num_points < 0 || shares <= 0 || share < 0 || share >= shares)
    return FailMulti(VGT_HIP_ERR_INVALID_ARGUMENT, "invalid share of a point cloud");
  // equal shares, earlier shares take the remainder (the rule of vgt_hip_sdf_slab_range)
  vgt::SlabRange(num_points, shares, share, first, count);
  return VGT_HIP_OK;
}

extern "C" int vgt_hipx_raycast_points_split(vgt_hip_ctx* ctx, vgt_hip_grids* grids, size_t grid_index,
                                             const int* helper_devices, int num_helpers, const float* points_xyz_host,
                                             int64_t num_points, float max_range,
                                             const float* grid_pointcloud_transform, float voxel_size,
                                             float inverse_voxel_size, float grid_x_size, float grid_y_size,
                                             float grid_z_size, int32_t num_x_voxels, int32_t num_y_voxels,
                                             int32_t num_z_voxels)
{
  if (!ctx || !grids) return FailMulti(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  if (num_helpers < 0 || (num_helpers > 0 && !helper_devices))
    return FailMulti(VGT_HIP_ERR_INVALID_ARGUMENT, "invalid helper device list");
  if (num_points < 0 || (num_points > 0 && !points_xyz_host))
    return FailMulti(VGT_HIP_ERR_INVALID_ARGUMENT, "invalid point buffer");
  int device_count = 0;
  if (hipGetDeviceCount(&device_count) != hipSuccess || device_count <= 0)
    return FailMulti(VGT_HIP_ERR_UNAVAILABLE, "no usable HIP device (libvgt_hip has no CPU fallback)");
  for (int i = 0; i < num_helpers; i++)
    if (helper_devices[i] < 0 || helper_devices[i] >= device_count)
      return FailMulti(VGT_HIP_ERR_UNAVAILABLE, "device " + std::to_string(helper_devices[i]) + " out of range for " +
                                                    std::to_string(device_count) + " devices");
  const int primary_device = vgt_hip_device_of(ctx);
  const int64_t num_cells = vgt_hip_tracking_grids_num_cells(grids);
  // a share nobody would miss is not worth a device: at most one helper per point beyond the caller's own
  const int helpers = static_cast<int>(std::min<int64_t>(num_helpers, std::max<int64_t>(num_points - 1, 0)));
  const int shares = helpers + 1;
  int64_t first = 0, count = 0;
  vgt::SlabRange(num_points, shares, 0, &first, &count);
  if (helpers == 0)  // also the argument checks of the plain call
    return vgt_hip_raycast_points_f32(ctx, grids, grid_index, points_xyz_host, num_points, max_range,
                                      grid_pointcloud_transform, voxel_size, inverse_voxel_size, grid_x_size,
                                      grid_y_size, grid_z_size, num_x_voxels, num_y_voxels, num_z_voxels);
  if (!grid_pointcloud_transform) return FailMulti(VGT_HIP_ERR_INVALID_ARGUMENT, "null argument");
  if (grid_index >= static_cast<size_t>(vgt_hip_tracking_grids_num_grids(grids)))
    return FailMulti(VGT_HIP_ERR_INVALID_ARGUMENT, "tracking grid index out of range");
  if (num_x_voxels <= 0 || num_y_voxels <= 0 || num_z_voxels <= 0 ||
      static_cast<int64_t>(num_x_voxels) * num_y_voxels * num_z_voxels != num_cells)
    return FailMulti(VGT_HIP_ERR_INVALID_ARGUMENT, "voxel counts do not match the tracking grids");
  auto* const target = static_cast<int32_t*>(vgt_hip_tracking_grids_dev_ptr(grids, grid_index));
  if (!target) return FailMulti(VGT_HIP_ERR_INVALID_ARGUMENT, "tracking grids have been released");

  const std::vector<int> helper_list(helper_devices, helper_devices + helpers);
  std::lock_guard<std::mutex> ray_guard(g_ray_lock);
  if (g_ray_set && !(g_ray_set->primary_device == primary_device && g_ray_set->helper_devices == helper_list &&
                     g_ray_set->num_cells == num_cells))
  {
    delete g_ray_set;
    g_ray_set = nullptr;
  }
  bool distinct = true;
  for (int a = 0; a < helpers; a++)
  {
    if (helper_list[a] == primary_device) distinct = false;
    for (int b = a + 1; b < helpers; b++)
      if (helper_list[a] == helper_list[b]) distinct = false;
  }
  const size_t grid_bytes = static_cast<size_t>(num_cells) * 2 * sizeof(int32_t);
  if (!g_ray_set)
  {
    auto* fresh = new RaySet();
    fresh->primary_device = primary_device;
    fresh->helper_devices = helper_list;
    fresh->num_cells = num_cells;
    fresh->helpers.resize(static_cast<size_t>(helpers));
    int rc = VGT_HIP_OK;
    for (int k = 0; k < helpers && rc == VGT_HIP_OK; k++)
    {
      RayHelper& h = fresh->helpers[k];
      h.device = helper_list[k];
      if (hipSetDevice(h.device) != hipSuccess ||
          hipStreamCreateWithFlags(&h.stream, hipStreamNonBlocking) != hipSuccess)
        rc = FailMulti(VGT_HIP_ERR_RUNTIME, "[create helper stream] HIP error");
      if (rc == VGT_HIP_OK) rc = vgt_hip_create(h.device, -1, &h.ctx);
      if (rc == VGT_HIP_OK) rc = vgt_hip_tracking_grids_create(h.ctx, num_cells, 1, &h.grids);
    }
    if (rc == VGT_HIP_OK && !distinct)
    {
      bool need_landing = false;
      for (int k = 0; k < helpers; k++)
        if (helper_list[k] != primary_device) need_landing = true;
      if (need_landing && (hipSetDevice(primary_device) != hipSuccess ||
                           hipMalloc(reinterpret_cast<void**>(&fresh->landing), grid_bytes) != hipSuccess))
        rc = FailMulti(VGT_HIP_ERR_RUNTIME, "[allocate landing grid] HIP error");
    }
    if (rc != VGT_HIP_OK)
    {
      delete fresh;
      return rc;
    }
    g_ray_set = fresh;
  }
  RaySet& set = *g_ray_set;

  // every share on its own device, concurrently: the helpers on threads of their own (the call uploads its share
  // and returns when its kernel has finished), the caller's share here, straight into the caller's grid
  std::vector<int> codes(static_cast<size_t>(helpers), VGT_HIP_OK);
  std::vector<std::string> messages(static_cast<size_t>(helpers));
  std::vector<std::thread> workers;
  workers.reserve(static_cast<size_t>(helpers));
  for (int k = 0; k < helpers; k++)
    workers.emplace_back([&, k] {
      RayHelper& h = set.helpers[k];
      int64_t begin = 0, share = 0;
      vgt::SlabRange(num_points, shares, k + 1, &begin, &share);
      int rc = vgt_hip_tracking_grids_clear(h.ctx, h.grids);
      if (rc == VGT_HIP_OK)
        rc = vgt_hip_raycast_points_f32(h.ctx, h.grids, 0, points_xyz_host + 3 * begin, share, max_range,
                                        grid_pointcloud_transform, voxel_size, inverse_voxel_size, grid_x_size,
                                        grid_y_size, grid_z_size, num_x_voxels, num_y_voxels, num_z_voxels);
      codes[k] = rc;
      if (rc != VGT_HIP_OK) messages[k] = vgt_hip_last_error();  // the message is the worker thread's own
    });
  const int own_rc = vgt_hip_raycast_points_f32(ctx, grids, grid_index, points_xyz_host + 3 * first, count, max_range,
                                                grid_pointcloud_transform, voxel_size, inverse_voxel_size, grid_x_size,
                                                grid_y_size, grid_z_size, num_x_voxels, num_y_voxels, num_z_voxels);
  for (std::thread& t : workers) t.join();
  if (own_rc != VGT_HIP_OK) return own_rc;
  for (int k = 0; k < helpers; k++)
    if (codes[k] != VGT_HIP_OK) return FailMulti(codes[k], messages[k]);

  // the sum, onto the caller's grid, on the caller's stream (later calls on `ctx` -- the filter -- see it)
  const hipStream_t primary_stream = vgt::ContextStream(ctx);
  const size_t counts = static_cast<size_t>(num_cells) * 2;
  if (distinct)
  {
    const Rccl& rccl = GetRccl();
    if (!rccl.error.empty()) return FailMulti(VGT_HIP_ERR_UNAVAILABLE, rccl.error);
    std::vector<int> clique;
    clique.push_back(primary_device);
    clique.insert(clique.end(), helper_list.begin(), helper_list.end());
    std::string comm_error;
    CommSet* const comms = GetCommSet(rccl, clique, &comm_error);
    if (!comms) return FailMulti(VGT_HIP_ERR_RUNTIME, comm_error);
    std::unique_lock<std::mutex> comm_lock(comms->in_use);
    // held until the streams have drained, whatever happens below
    struct Drain
    {
      RaySet& set;
      int primary_device;
      hipStream_t primary_stream;
      ~Drain()
      {
        for (RayHelper& h : set.helpers)
        {
          (void)hipSetDevice(h.device);
          (void)hipStreamSynchronize(h.stream);
        }
        (void)hipSetDevice(primary_device);
        (void)hipStreamSynchronize(primary_stream);
      }
    } drain{set, primary_device, primary_stream};
    VGTX_NCCL(rccl.GroupStart(), "ncclGroupStart");
    ncclResult_t res = rccl.Reduce(target, target, counts, ncclInt32, ncclSum, 0, comms->comms[0], primary_stream);
    for (int k = 0; k < helpers && res == ncclSuccess; k++)
    {
      RayHelper& h = set.helpers[k];
      void* const mine = vgt_hip_tracking_grids_dev_ptr(h.grids, 0);
      res = rccl.Reduce(mine, nullptr, counts, ncclInt32, ncclSum, 0, comms->comms[k + 1], h.stream);
    }
    if (res != ncclSuccess)
    {
      (void)rccl.GroupEnd();
      return FailMulti(VGT_HIP_ERR_RUNTIME, std::string("[ncclReduce] RCCL error [") + rccl.GetErrorString(res) + "]");
    }
    VGTX_NCCL(rccl.GroupEnd(), "ncclGroupEnd");
    return VGT_HIP_OK;  // ~Drain waits for the reduction
  }
  VGTX_HIP(hipSetDevice(primary_device), "set device");
  for (int k = 0; k < helpers; k++)
  {
    RayHelper& h = set.helpers[k];
    const auto* mine = static_cast<const int32_t*>(vgt_hip_tracking_grids_dev_ptr(h.grids, 0));
    if (h.device != primary_device)
    {
      VGTX_HIP(hipMemcpyPeerAsync(set.landing, primary_device, mine, h.device, grid_bytes, primary_stream),
               "copy a share's tracking grid");
      mine = set.landing;
    }
    VGTX_HIP(vgt::LaunchAccumulateCounts(target, mine, static_cast<int64_t>(counts), primary_stream),
             "add a share's tracking grid");
  }
  VGTX_HIP(hipStreamSynchronize(primary_stream), "wait for the sum");
  return VGT_HIP_OK;
}
