// HipPointCloudVoxelizer / ExtractSignedDistanceField: host orchestration over the HIP helper
// and the C ABI.  Mirrors src/voxelized_geometry_tools/device_pointcloud_voxelization.cpp:65-181.
#include "../../../include/vgt_hip/hip_pointcloud_voxelizer.hpp"

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cstddef>
#include <exception>
#include <mutex>
#include <thread>

#include "../../../include/vgt_hip.h"

namespace vgt_hip
{
using voxelized_geometry_tools::pointcloud_voxelization::FilterGridHandle;
using voxelized_geometry_tools::pointcloud_voxelization::RetrieveOptionOrDefault;
using voxelized_geometry_tools::pointcloud_voxelization::TrackingGridsHandle;
namespace hip_helpers = voxelized_geometry_tools::pointcloud_voxelization::hip_helpers;

HipPointCloudVoxelizer::HipPointCloudVoxelizer(const std::map<std::string, int32_t>& options,
                                               const LoggingFunction& logging_fn)
{
  const int32_t parallelize = RetrieveOptionOrDefault(options, "DISPATCH_PARALLELIZE", 1, logging_fn);
  const int32_t num_threads = RetrieveOptionOrDefault(options, "DISPATCH_NUM_THREADS", -1, logging_fn);
  if (parallelize > 0 && num_threads >= 1)
    dispatch_threads_ = num_threads;
  else if (parallelize > 0)
    dispatch_threads_ = std::max(1u, std::thread::hardware_concurrency());
  else
    dispatch_threads_ = 1;
  helper_interface_ = hip_helpers::MakeHipVoxelizationHelper(options, logging_fn);
  EnforceAvailable();
}

void HipPointCloudVoxelizer::EnforceAvailable() const
{
  if (!helper_interface_)
    throw std::runtime_error(device_name_ + " is not available (feature was not built)");
  if (!helper_interface_->IsAvailable())
    throw std::runtime_error(device_name_ + " is not available (device cannot be used)");
}

OccupancyMap HipPointCloudVoxelizer::VoxelizePointClouds(
    const OccupancyMap& static_environment,
    const PointCloudVoxelizationFilterOptions& filter_options,
    const std::vector<PointCloudWrapperSharedPtr>& pointclouds,
    const std::function<void(const VoxelizerRuntime&)>& runtime_log_fn) const
{
  if (!static_environment.IsInitialized())
    throw std::invalid_argument("!static_environment.IsInitialized()");
  // The reference copies the static map (pointcloud_voxelization_interface.hpp:246-258) and RetrieveFilteredGrid then
  // overwrites every cell of the copy.  Here the returned map starts with the static map's frame and extents and
  // UNTOUCHED cells -- no 4 bytes per cell copied, no page faulted by this thread -- and helper threads fault its pages in
  // while the device raycasts, so that the download (which page-locks its destination) finds them resident.  (At 256^3 the
  // copy and its page faults were 10 of the call's 12.7 ms.)
  const auto allocate_time = std::chrono::steady_clock::now();
  OccupancyMap output_environment = OccupancyMap::UninitializedLike(static_environment);
  const double allocate_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - allocate_time).count();
  float* const cells = output_environment.GetMutableRawData().data();
  const size_t num_cells = output_environment.GetMutableRawData().size();
  constexpr size_t kPageFloats = 4096 / sizeof(float);
  const size_t num_pages = (num_cells + kPageFloats - 1) / kPageFloats;
  const int touchers = static_cast<int>(std::min<size_t>(static_cast<size_t>(std::max(1, std::min(dispatch_threads_, 8))),
                                                         std::max<size_t>(1, num_pages / 256)));
  std::vector<std::thread> pool;
  if (num_pages >= 256)
    for (int t = 0; t < touchers; t++)
      pool.emplace_back([=]() {
        // (one store per page; nothing reads these cells before the download has overwritten them)
        for (size_t page = num_pages * t / touchers; page < num_pages * (t + 1) / touchers; page++)
          reinterpret_cast<volatile float*>(cells)[page * kPageFloats] = 0.0f;
      });
  double pages_wait_s = 0.0;
  const auto join_touchers = [&pool, &pages_wait_s]() {
    const auto t0 = std::chrono::steady_clock::now();
    for (auto& th : pool) th.join();
    pool.clear();
    pages_wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  };
  VoxelizerRuntime runtime(0.0, 0.0);
  try
  {
    runtime = DoVoxelizePointClouds(static_environment, filter_options, pointclouds, output_environment, join_touchers);
  }
  catch (...)
  {
    join_touchers();
    throw;
  }
  join_touchers();
  {
    std::lock_guard<std::mutex> lock(phases_mutex_);
    last_phases_.output_allocate_s = allocate_s;
    last_phases_.output_pages_wait_s = pages_wait_s;
  }
  if (runtime_log_fn) runtime_log_fn(runtime);
  return output_environment;
}

VoxelizerRuntime HipPointCloudVoxelizer::VoxelizePointClouds(
    const OccupancyMap& static_environment,
    const PointCloudVoxelizationFilterOptions& filter_options,
    const std::vector<PointCloudWrapperSharedPtr>& pointclouds,
    OccupancyMap& output_environment) const
{
  return DoVoxelizePointClouds(static_environment, filter_options, pointclouds, output_environment, {});
}

VoxelizerRuntime HipPointCloudVoxelizer::DoVoxelizePointClouds(
    const OccupancyMap& static_environment,
    const PointCloudVoxelizationFilterOptions& filter_options,
    const std::vector<PointCloudWrapperSharedPtr>& pointclouds,
    OccupancyMap& output_environment,
    const std::function<void()>& before_download) const
{
  if (!static_environment.IsInitialized())
    throw std::invalid_argument("!static_environment.IsInitialized()");
  if (!output_environment.IsInitialized())
    throw std::invalid_argument("!output_environment.IsInitialized()");
  if (!static_environment.SameSizes(output_environment))
    throw std::invalid_argument(
        "static_environment.ControlSizes() != output_environment.ControlSizes()");
  for (size_t idx = 0; idx < pointclouds.size(); idx++)
    if (!pointclouds[idx])
      throw std::invalid_argument("pointclouds[" + std::to_string(idx) + "] is null");
  EnforceAvailable();

  const auto start_time = std::chrono::steady_clock::now();
  const auto seconds_since = [](const std::chrono::steady_clock::time_point& t0) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  };
  VoxelizePhases phases;

  // at least one tracking grid so that filtering is uniform when there are no clouds
  const size_t num_tracking_grids = std::max(pointclouds.size(), static_cast<size_t>(1));
  std::unique_ptr<TrackingGridsHandle> tracking_grids = helper_interface_->PrepareTrackingGrids(
      static_environment.NumTotalVoxels(), static_cast<int32_t>(num_tracking_grids));
  if (tracking_grids->GetNumTrackingGrids() != num_tracking_grids)
    throw std::runtime_error("Failed to allocate device tracking grid");
  phases.prepare_tracking_grids_s = seconds_since(start_time);

  // The static environment goes to the device FIRST, without waiting for the copy (the reference prepares the filter grid
  // after the raycasts, device_pointcloud_voxelization.cpp:161-165: with a blocking copy the order does not matter
  // there): its 4 bytes per cell then cross the link while the clouds are raycast.  The results are the same.
  std::unique_ptr<FilterGridHandle> filter_grid;
  const auto upload_time = std::chrono::steady_clock::now();
  hip_helpers::DeferredFilterGridInterface* deferred =
      dynamic_cast<hip_helpers::DeferredFilterGridInterface*>(helper_interface_.get());
  if (deferred != nullptr)
    filter_grid = deferred->PrepareFilterGridDeferred(static_environment.NumTotalVoxels(),
                                                      static_environment.GetImmutableRawData().data());
  phases.filter_grid_enqueue_s = seconds_since(upload_time);
  const auto raycast_start_time = std::chrono::steady_clock::now();

  const Isometry3& X_GW = static_environment.InverseOriginTransform();
  const float voxel_size = static_cast<float>(static_environment.VoxelXSize());
  const float inverse_voxel_size = static_cast<float>(1.0 / static_environment.VoxelXSize());
  const float grid_x_size = static_cast<float>(static_environment.GridXSize());
  const float grid_y_size = static_cast<float>(static_environment.GridYSize());
  const float grid_z_size = static_cast<float>(static_environment.GridZSize());
  const int32_t num_x_voxels = static_cast<int32_t>(static_environment.NumXVoxels());
  const int32_t num_y_voxels = static_cast<int32_t>(static_environment.NumYVoxels());
  const int32_t num_z_voxels = static_cast<int32_t>(static_environment.NumZVoxels());

  const auto raycast_cloud = [&](const size_t cloud_index)
  {
    const PointCloudWrapperSharedPtr& cloud = pointclouds[cloud_index];
    if (cloud->Size() <= 0) return;  // empty arrays never reach the device interface
    // HIP_EXACT_FP64: the reference's CPU voxelizer on the device -- double points (CopyPointLocationIntoDoublePtr),
    // double transform and sizes (cpu_pointcloud_voxelization.cpp:167-206), the float64 walk
    hip_helpers::ExactRaycastInterface* exact =
        dynamic_cast<hip_helpers::ExactRaycastInterface*>(helper_interface_.get());
    if (exact != nullptr && exact->ExactFp64())
    {
      const Isometry3 X_GC_exact = X_GW * cloud->PointCloudOriginTransform();
      std::vector<double> raw_points(static_cast<size_t>(cloud->Size()) * 3, 0.0);
      for (int64_t point = 0; point < cloud->Size(); point++)
        cloud->CopyPointLocationIntoDoublePtr(point, raw_points.data() + point * 3);
      exact->RaycastPointsExact(raw_points, cloud->MaxRange(), X_GC_exact.m.data(), static_environment.VoxelXSize(),
                                1.0 / static_environment.VoxelXSize(), static_environment.GridXSize(),
                                static_environment.GridYSize(), static_environment.GridZSize(), num_x_voxels, num_y_voxels,
                                num_z_voxels, *tracking_grids, cloud_index);
      return;
    }
    const std::array<float, 16> X_GC = (X_GW * cloud->PointCloudOriginTransform()).CastFloat();
    const float max_range = static_cast<float>(cloud->MaxRange());
    // a cloud that exposes a strided FLOAT32 layout (PointCloud2) is raycast in place
    const uint8_t* strided_data = nullptr;
    int64_t point_step = 0, xyz_offset = 0;
    hip_helpers::StridedRaycastInterface* strided =
        dynamic_cast<hip_helpers::StridedRaycastInterface*>(helper_interface_.get());
    if (strided != nullptr && cloud->StridedFloat32Layout(&strided_data, &point_step, &xyz_offset) &&
        point_step % 4 == 0 && xyz_offset % 4 == 0)
    {
      strided->RaycastStridedPoints(strided_data, cloud->Size(), point_step, xyz_offset, max_range, X_GC.data(),
                                    voxel_size, inverse_voxel_size, grid_x_size, grid_y_size, grid_z_size,
                                    num_x_voxels, num_y_voxels, num_z_voxels, *tracking_grids, cloud_index);
      return;
    }
    std::vector<float> raw_points(static_cast<size_t>(cloud->Size()) * 3, 0.0f);
    for (int64_t point = 0; point < cloud->Size(); point++)
      cloud->CopyPointLocationIntoFloatPtr(point, raw_points.data() + point * 3);
    helper_interface_->RaycastPoints(raw_points, max_range, X_GC.data(), voxel_size,
                                     inverse_voxel_size, grid_x_size, grid_y_size, grid_z_size,
                                     num_x_voxels, num_y_voxels, num_z_voxels, *tracking_grids,
                                     cloud_index);
  };

  // dynamic dispatch over clouds from several host threads, like DynamicParallelForIndexLoop
  const int workers =
      std::max(1, std::min(dispatch_threads_, static_cast<int>(pointclouds.size())));
  if (workers <= 1)
  {
    for (size_t idx = 0; idx < pointclouds.size(); idx++) raycast_cloud(idx);
  }
  else
  {
    std::atomic<size_t> next{0};
    std::exception_ptr failure;
    std::mutex failure_mutex;
    std::vector<std::thread> pool;
    for (int w = 0; w < workers; w++)
    {
      pool.emplace_back([&]()
      {
        for (;;)
        {
          const size_t idx = next.fetch_add(1);
          if (idx >= pointclouds.size()) return;
          try
          {
            raycast_cloud(idx);
          }
          catch (...)
          {
            std::lock_guard<std::mutex> lock(failure_mutex);
            if (!failure) failure = std::current_exception();
          }
        }
      });
    }
    for (auto& th : pool) th.join();
    if (failure) std::rethrow_exception(failure);
  }

  const auto raycasted_time = std::chrono::steady_clock::now();
  phases.raycast_s = std::chrono::duration<double>(raycasted_time - raycast_start_time).count();

  if (!filter_grid)
    filter_grid = helper_interface_->PrepareFilterGrid(static_environment.NumTotalVoxels(),
                                                       static_environment.GetImmutableRawData().data());
  phases.filter_grid_blocking_upload_s = seconds_since(raycasted_time);
  const auto filter_time = std::chrono::steady_clock::now();
  helper_interface_->FilterTrackingGrids(
      *tracking_grids, static_cast<float>(filter_options.PercentSeenFree()),
      filter_options.OutlierPointsThreshold(), filter_options.NumCamerasSeenFree(), *filter_grid);
  phases.filter_enqueue_s = seconds_since(filter_time);
  if (before_download) before_download();
  const auto download_time = std::chrono::steady_clock::now();
  helper_interface_->RetrieveFilteredGrid(*filter_grid,
                                          output_environment.GetMutableRawData().data());
  phases.filter_and_download_s = seconds_since(download_time);
  const auto release_time = std::chrono::steady_clock::now();
  filter_grid.reset();
  tracking_grids.reset();
  phases.release_s = seconds_since(release_time);
  phases.total_s = seconds_since(start_time);
  {
    std::lock_guard<std::mutex> lock(phases_mutex_);
    last_phases_ = phases;
  }

  const auto done_time = std::chrono::steady_clock::now();
  // The reference prepares the filter grid AFTER the raycasts, inside its filtering time
  // (S/device_pointcloud_voxelization.cpp:161-165); here the upload is enqueued first, so the host time that takes
  // (page-locking the static map, enqueueing the copy) is moved to where the reference accounts for it.
  return VoxelizerRuntime(
      std::chrono::duration<double>(raycasted_time - start_time).count() - phases.filter_grid_enqueue_s,
      std::chrono::duration<double>(done_time - raycasted_time).count() + phases.filter_grid_enqueue_s);
}

namespace
{
// One context per device for the free-standing SDF entry points, created on first use and kept for
// the life of the process (deliberately never destroyed: static destruction order against the HIP
// runtime is not defined).  The context caches its device buffers, so a caller that extracts
// fields repeatedly pays for context creation and hipMalloc once; the C ABI serialises concurrent
// calls on one context.
std::mutex& SharedContextGuard()
{
  static std::mutex guard;
  return guard;
}
std::map<int, vgt_hip_ctx*>& SharedContexts()
{
  static std::map<int, vgt_hip_ctx*>* contexts = new std::map<int, vgt_hip_ctx*>();
  return *contexts;
}

vgt_hip_ctx* SharedSdfContext(int device)
{
  std::lock_guard<std::mutex> lock(SharedContextGuard());
  auto& contexts = SharedContexts();
  auto found = contexts.find(device);
  if (found != contexts.end()) return found->second;
  vgt_hip_ctx* ctx = nullptr;
  if (vgt_hip_create(device, -1, &ctx) != VGT_HIP_OK)
    throw std::runtime_error(std::string("HIP SDF backend is not available: ") + vgt_hip_last_error());
  contexts[device] = ctx;
  return ctx;
}
}  // namespace

void ReleaseCachedDeviceMemory()
{
  // only contexts that exist: a device that was never used is not touched
  std::vector<vgt_hip_ctx*> existing;
  {
    std::lock_guard<std::mutex> lock(SharedContextGuard());
    for (const auto& kv : SharedContexts()) existing.push_back(kv.second);
  }
  for (vgt_hip_ctx* ctx : existing) (void)vgt_hip_trim(ctx);
  vgt_hipx_release();  // the multi-device entry point's slab set
  GridBlockCache::Release();  // host blocks of dropped grids (host_types.hpp)
}

SignedDistanceField ExtractSignedDistanceField(
    const OccupancyMap& map, const SignedDistanceFieldGenerationParameters& parameters)
{
  if (!map.IsInitialized()) throw std::invalid_argument("Grid must be initialized");
  SignedDistanceField sdf;
  sdf.oob_value = parameters.oob_value;
  sdf.grid = DenseGrid::Uninitialized(map.OriginTransform(), map.Frame(), map.Resolution(), map.NumXVoxels(),
                                      map.NumYVoxels(), map.NumZVoxels());  // every cell is written below
  if (!parameters.hip_devices.empty())
  {
    // the large-grid branch: one Z slab per listed device, one process, one RCCL exchange
    const int rc = vgt_hipx_sdf_multi(
        parameters.hip_devices.data(), static_cast<int>(parameters.hip_devices.size()),
        map.GetImmutableRawData().data(), map.NumXVoxels(), map.NumYVoxels(), map.NumZVoxels(), map.Resolution(),
        parameters.unknown_is_filled ? 1 : 0, parameters.add_virtual_border ? 1 : 0,
        sdf.grid.GetMutableRawData().data(), &sdf.minimum, &sdf.maximum);
    if (rc == VGT_HIP_ERR_INVALID_ARGUMENT) throw std::invalid_argument(vgt_hip_last_error());
    if (rc != VGT_HIP_OK) throw std::runtime_error(std::string("HIP SDF backend: ") + vgt_hip_last_error());
    sdf.locked = true;
    return sdf;
  }
  vgt_hip_ctx* ctx = SharedSdfContext(parameters.hip_device);
  const int rc = vgt_hip_sdf_from_occupancy_f32(
      ctx, map.GetImmutableRawData().data(), map.NumXVoxels(), map.NumYVoxels(), map.NumZVoxels(),
      map.Resolution(), parameters.unknown_is_filled ? 1 : 0,
      parameters.add_virtual_border ? 1 : 0, sdf.grid.GetMutableRawData().data(), &sdf.minimum,
      &sdf.maximum);
  const std::string msg = (rc == VGT_HIP_OK) ? std::string() : std::string(vgt_hip_last_error());
  if (rc == VGT_HIP_ERR_INVALID_ARGUMENT) throw std::invalid_argument(msg);
  if (rc != VGT_HIP_OK) throw std::runtime_error(msg);
  sdf.locked = true;  // min / max were computed on the device: Lock() has nothing left to scan
  return sdf;
}

std::vector<SignedDistanceField> ExtractSignedDistanceFields(
    const std::vector<const OccupancyMap*>& maps, const SignedDistanceFieldGenerationParameters& parameters)
{
  std::vector<SignedDistanceField> fields(maps.size());
  if (maps.empty()) return fields;
  std::vector<const float*> inputs;
  std::vector<float*> outputs;
  for (size_t i = 0; i < maps.size(); i++)
  {
    const OccupancyMap* map = maps[i];
    if (!map || !map->IsInitialized()) throw std::invalid_argument("Grid must be initialized");
    if (!map->SameSizes(*maps[0])) throw std::invalid_argument("the maps of a batch must have the same sizes");
    fields[i].oob_value = parameters.oob_value;
    fields[i].grid = DenseGrid::Uninitialized(map->OriginTransform(), map->Frame(), map->Resolution(),
                                              map->NumXVoxels(), map->NumYVoxels(), map->NumZVoxels());
    inputs.push_back(map->GetImmutableRawData().data());
    outputs.push_back(fields[i].grid.GetMutableRawData().data());
  }
  std::vector<float> minima(maps.size()), maxima(maps.size());
  vgt_hip_ctx* ctx = SharedSdfContext(parameters.hip_device);
  const OccupancyMap& first = *maps[0];
  const int rc = vgt_hip_sdf_batch_from_occupancy_f32(
      ctx, inputs.data(), static_cast<int64_t>(maps.size()), first.NumXVoxels(), first.NumYVoxels(),
      first.NumZVoxels(), first.Resolution(), parameters.unknown_is_filled ? 1 : 0,
      parameters.add_virtual_border ? 1 : 0, outputs.data(), minima.data(), maxima.data());
  const std::string msg = (rc == VGT_HIP_OK) ? std::string() : std::string(vgt_hip_last_error());
  if (rc == VGT_HIP_ERR_INVALID_ARGUMENT) throw std::invalid_argument(msg);
  if (rc != VGT_HIP_OK) throw std::runtime_error(msg);
  for (size_t i = 0; i < maps.size(); i++)
  {
    fields[i].minimum = minima[i];
    fields[i].maximum = maxima[i];
    fields[i].locked = true;
  }
  return fields;
}

namespace
{
[[noreturn]] void ThrowForCode(int rc, const std::string& msg)
{
  if (rc == VGT_HIP_ERR_INVALID_ARGUMENT) throw std::invalid_argument(msg);
  throw std::runtime_error(msg);
}

// rotation block of the field's origin transform, row-major (what the reference multiplies gradients with)
std::array<double, 9> RotationOf(const Isometry3& t)
{
  std::array<double, 9> r{};
  for (int row = 0; row < 3; row++)
    for (int col = 0; col < 3; col++) r[static_cast<size_t>(row * 3 + col)] = t(row, col);
  return r;
}
}  // namespace

// ---- SDF consumers ----

DistanceEstimates EstimateLocationDistances(const SignedDistanceField& sdf, const std::vector<double>& locations_xyz,
                                            int hip_device)
{
  if (locations_xyz.size() % 3 != 0) throw std::invalid_argument("locations_xyz must hold 3 doubles per point");
  const int64_t n = static_cast<int64_t>(locations_xyz.size() / 3);
  DistanceEstimates out;
  out.distance.resize(static_cast<size_t>(n));
  out.has_value.resize(static_cast<size_t>(n));
  const DenseGrid& g = sdf.grid;
  const int rc = vgt_hip_sdf_estimate_distance(SharedSdfContext(hip_device), g.GetImmutableRawData().data(), g.NumXVoxels(),
                                               g.NumYVoxels(), g.NumZVoxels(), g.Resolution(),
                                               g.InverseOriginTransform().m.data(), locations_xyz.data(), n,
                                               out.distance.data(), out.has_value.data());
  if (rc != VGT_HIP_OK) ThrowForCode(rc, vgt_hip_last_error());
  return out;
}

Gradients GetLocationFineGradients(const SignedDistanceField& sdf, const std::vector<double>& locations_xyz,
                                   double nominal_window_size, int hip_device)
{
  if (locations_xyz.size() % 3 != 0) throw std::invalid_argument("locations_xyz must hold 3 doubles per point");
  const int64_t n = static_cast<int64_t>(locations_xyz.size() / 3);
  Gradients out;
  out.gradient.resize(static_cast<size_t>(3 * n));
  out.has_value.resize(static_cast<size_t>(n));
  const DenseGrid& g = sdf.grid;
  const int rc = vgt_hip_sdf_fine_gradient(SharedSdfContext(hip_device), g.GetImmutableRawData().data(), g.NumXVoxels(),
                                           g.NumYVoxels(), g.NumZVoxels(), g.Resolution(),
                                           g.InverseOriginTransform().m.data(), locations_xyz.data(), n,
                                           nominal_window_size, out.gradient.data(), out.has_value.data());
  if (rc != VGT_HIP_OK)
  {
    const std::string msg = vgt_hip_last_error();
    // the reference throws std::runtime_error for a window that leaves the field on both sides
    if (msg.find("Window size") != std::string::npos) throw std::runtime_error(msg);
    ThrowForCode(rc, msg);
  }
  return out;
}

Gradients GetIndexCoarseGradients(const SignedDistanceField& sdf, bool enable_edge_gradients, int hip_device)
{
  const DenseGrid& g = sdf.grid;
  const size_t n = static_cast<size_t>(g.NumTotalVoxels());
  Gradients out;
  out.gradient.resize(3 * n);
  out.has_value.resize(n);
  const std::array<double, 9> rotation = RotationOf(g.OriginTransform());
  const int rc = vgt_hip_sdf_coarse_gradient(SharedSdfContext(hip_device), g.GetImmutableRawData().data(), g.NumXVoxels(),
                                             g.NumYVoxels(), g.NumZVoxels(), g.Resolution(), enable_edge_gradients ? 1 : 0,
                                             rotation.data(), out.gradient.data(), out.has_value.data());
  if (rc != VGT_HIP_OK) ThrowForCode(rc, vgt_hip_last_error());
  return out;
}

std::vector<double> ComputeLocalExtremaMap(const SignedDistanceField& sdf, int hip_device)
{
  const DenseGrid& g = sdf.grid;
  std::vector<double> extrema(3 * static_cast<size_t>(g.NumTotalVoxels()));
  const std::array<double, 9> rotation = RotationOf(g.OriginTransform());
  const int rc = vgt_hip_sdf_local_extrema_map(SharedSdfContext(hip_device), g.GetImmutableRawData().data(),
                                               g.NumXVoxels(), g.NumYVoxels(), g.NumZVoxels(), g.Resolution(),
                                               rotation.data(), extrema.data());
  if (rc != VGT_HIP_OK) ThrowForCode(rc, vgt_hip_last_error());
  return extrema;
}

// ---- the other three map types ----

SignedDistanceField ExtractSignedDistanceField(
    const OccupancyComponentMap& map, const SignedDistanceFieldGenerationParameters& parameters)
{
  if (!map.IsInitialized()) throw std::invalid_argument("Grid must be initialized");
  vgt_hip_ctx* ctx = SharedSdfContext(parameters.hip_device);  // (the process's context of that device: no set-up per call)
  vgt_hip_cells* cells = nullptr;
  // the component does not enter the SDF (occupancy_component_map.hpp:276-296): no object id
  int rc = vgt_hip_cells_create(ctx, map.GetImmutableRawData().data(), map.NumXVoxels(), map.NumYVoxels(),
                                map.NumZVoxels(), static_cast<int32_t>(sizeof(OccupancyComponentCell)), -1,
                                &cells);
  SignedDistanceField sdf;
  sdf.oob_value = parameters.oob_value;
  sdf.grid = DenseGrid::Uninitialized(map.OriginTransform(), map.Frame(), map.Resolution(), map.NumXVoxels(),
                                      map.NumYVoxels(), map.NumZVoxels());  // every cell is written below
  if (rc == VGT_HIP_OK)
    rc = vgt_hip_cells_sdf(ctx, cells, nullptr, 0, map.Resolution(), parameters.unknown_is_filled ? 1 : 0,
                           parameters.add_virtual_border ? 1 : 0, sdf.grid.GetMutableRawData().data(),
                           &sdf.minimum, &sdf.maximum);
  const std::string msg = (rc == VGT_HIP_OK) ? std::string() : std::string(vgt_hip_last_error());
  vgt_hip_cells_destroy(cells);
  if (rc != VGT_HIP_OK) ThrowForCode(rc, msg);
  sdf.locked = true;
  return sdf;
}

DeviceTaggedObjectMap::DeviceTaggedObjectMap(const TaggedObjectOccupancyMap& map, int hip_device)
{
  if (!map.IsInitialized()) throw std::invalid_argument("Grid must be initialized");
  shape_ = DenseGrid::ShapeOnly(map.OriginTransform(), map.Frame(), map.Resolution(), map.NumXVoxels(),
                                map.NumYVoxels(), map.NumZVoxels());
  Upload(map.GetImmutableRawData().data(), static_cast<int>(sizeof(TaggedObjectOccupancyCell)),
         static_cast<int>(offsetof(TaggedObjectOccupancyCell, object_id)), hip_device);
}

DeviceTaggedObjectMap::DeviceTaggedObjectMap(const TaggedObjectOccupancyComponentMap& map, int hip_device)
{
  if (!map.IsInitialized()) throw std::invalid_argument("Grid must be initialized");
  shape_ = DenseGrid::ShapeOnly(map.OriginTransform(), map.Frame(), map.Resolution(), map.NumXVoxels(),
                                map.NumYVoxels(), map.NumZVoxels());
  Upload(map.GetImmutableRawData().data(), static_cast<int>(sizeof(TaggedObjectOccupancyComponentCell)),
         static_cast<int>(offsetof(TaggedObjectOccupancyComponentCell, object_id)), hip_device);
}

void DeviceTaggedObjectMap::Upload(const void* cells, int cell_bytes, int object_id_offset, int hip_device)
{
  // The process's context of that device (SharedSdfContext), not one per map: creating a context -- streams, events, its
  // buffers -- costs 14 ms, ten times the extraction of a 128^3 map, and the reference's interface constructs a map's
  // device side per call.  Calls on the maps of one device queue behind each other on its stream.
  ctx_ = SharedSdfContext(hip_device);
  const int rc = vgt_hip_cells_create(ctx_, cells, shape_.NumXVoxels(), shape_.NumYVoxels(),
                                      shape_.NumZVoxels(), cell_bytes, object_id_offset, &cells_);
  if (rc != VGT_HIP_OK)
  {
    const std::string msg = vgt_hip_last_error();
    ctx_ = nullptr;
    ThrowForCode(rc, msg);
  }
}

DeviceTaggedObjectMap::~DeviceTaggedObjectMap()
{
  vgt_hip_cells_destroy(cells_);  // (the context is the process's, it stays)
}

SignedDistanceField DeviceTaggedObjectMap::EmptyField(
    const SignedDistanceFieldGenerationParameters& parameters) const
{
  SignedDistanceField sdf;
  sdf.oob_value = parameters.oob_value;
  sdf.grid = DenseGrid::Uninitialized(shape_.OriginTransform(), shape_.Frame(), shape_.Resolution(),
                                      shape_.NumXVoxels(), shape_.NumYVoxels(), shape_.NumZVoxels());
  return sdf;
}

SignedDistanceField DeviceTaggedObjectMap::ExtractSignedDistanceField(
    const std::vector<uint32_t>& objects_to_use,
    const SignedDistanceFieldGenerationParameters& parameters) const
{
  SignedDistanceField sdf = EmptyField(parameters);
  const int rc = vgt_hip_cells_sdf(
      ctx_, cells_, objects_to_use.empty() ? nullptr : objects_to_use.data(),
      static_cast<int64_t>(objects_to_use.size()), shape_.Resolution(), parameters.unknown_is_filled ? 1 : 0,
      parameters.add_virtual_border ? 1 : 0, sdf.grid.GetMutableRawData().data(), &sdf.minimum, &sdf.maximum);
  if (rc != VGT_HIP_OK) ThrowForCode(rc, vgt_hip_last_error());
  sdf.locked = true;
  return sdf;
}

std::map<uint32_t, SignedDistanceField> DeviceTaggedObjectMap::MakeSeparateObjectSDFs(
    const std::vector<uint32_t>& object_ids, const SignedDistanceFieldGenerationParameters& parameters) const
{
  // The reference runs one whole ExtractSignedDistanceField({id}) per object (tagged_object_occupancy_map.hpp:
  // 249-263); here the objects are one batch: one pass over the cells for all their masks, the EDT passes once over
  // all of them, the fields straight into the map's entries (vgt_hip_cells_object_sdfs).
  std::map<uint32_t, SignedDistanceField> per_object_sdfs;
  std::vector<uint32_t> ids;
  for (const uint32_t object_id : object_ids)
    if (per_object_sdfs.find(object_id) == per_object_sdfs.end())
    {
      per_object_sdfs.emplace(object_id, EmptyField(parameters));
      ids.push_back(object_id);
    }
  if (ids.empty()) return per_object_sdfs;
  std::vector<float*> fields;
  for (const uint32_t object_id : ids) fields.push_back(per_object_sdfs.at(object_id).grid.GetMutableRawData().data());
  std::vector<float> minima(ids.size()), maxima(ids.size());
  const int rc = vgt_hip_cells_object_sdfs(ctx_, cells_, ids.data(), static_cast<int64_t>(ids.size()),
                                           shape_.Resolution(), parameters.unknown_is_filled ? 1 : 0,
                                           parameters.add_virtual_border ? 1 : 0, fields.data(), minima.data(),
                                           maxima.data());
  if (rc != VGT_HIP_OK) ThrowForCode(rc, vgt_hip_last_error());
  for (size_t k = 0; k < ids.size(); k++)
  {
    SignedDistanceField& sdf = per_object_sdfs.at(ids[k]);
    sdf.minimum = minima[k];
    sdf.maximum = maxima[k];
    sdf.locked = true;
  }
  return per_object_sdfs;
}

std::vector<uint32_t> DeviceTaggedObjectMap::ObjectIds() const
{
  std::vector<uint32_t> ids(64);
  for (;;)
  {
    int64_t count = 0;
    const int rc = vgt_hip_cells_object_ids(ctx_, cells_, ids.data(), static_cast<int64_t>(ids.size()), &count);
    if (rc != VGT_HIP_OK) ThrowForCode(rc, vgt_hip_last_error());
    if (count <= static_cast<int64_t>(ids.size()))
    {
      ids.resize(static_cast<size_t>(count));
      return ids;
    }
    ids.assign(static_cast<size_t>(count), 0u);
  }
}

std::map<uint32_t, SignedDistanceField> DeviceTaggedObjectMap::MakeAllObjectSDFs(
    const SignedDistanceFieldGenerationParameters& parameters) const
{
  return MakeSeparateObjectSDFs(ObjectIds(), parameters);
}

SignedDistanceField DeviceTaggedObjectMap::ExtractFreeAndNamedObjectsSignedDistanceField(
    const SignedDistanceFieldGenerationParameters& parameters) const
{
  SignedDistanceField sdf = EmptyField(parameters);
  const int rc = vgt_hip_cells_free_and_named_objects_sdf(
      ctx_, cells_, shape_.Resolution(), parameters.unknown_is_filled ? 1 : 0,
      parameters.add_virtual_border ? 1 : 0, sdf.grid.GetMutableRawData().data(), &sdf.minimum, &sdf.maximum);
  if (rc != VGT_HIP_OK) ThrowForCode(rc, vgt_hip_last_error());
  sdf.locked = true;
  return sdf;
}
}  // namespace vgt_hip
