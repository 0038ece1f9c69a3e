// DeviceVoxelizationHelperInterface on top of the libvgt_hip C ABI.
//
// Behaviour follows the reference's CUDA helper (src/voxelized_geometry_tools/
// cuda_voxelization_helpers.cu:559-788): the constructor never throws (a bad device or a
// missing runtime leaves the helper unavailable), every failing call throws
// std::runtime_error("[...] HIP error [...]"), handles own their device memory, RaycastPoints
// may be called concurrently from several host threads, Retrieve* blocks.
// Plain C++ (no HIP headers): compiles with the host compiler and links against libvgt_hip.so.
#include "../../../include/vgt_hip/hip_voxelization_helpers.h"

#include <stdexcept>

#include "../../../include/vgt_hip.h"

namespace voxelized_geometry_tools
{
VGT_NAMESPACE_BEGIN
namespace pointcloud_voxelization
{
namespace hip_helpers
{
namespace
{
[[noreturn]] void ThrowLastError(const int code)
{
  const std::string msg(vgt_hip_last_error());
  if (code == VGT_HIP_ERR_INVALID_ARGUMENT) throw std::invalid_argument(msg);
  throw std::runtime_error(msg);
}

void Check(const int code)
{
  if (code != VGT_HIP_OK) ThrowLastError(code);
}

std::vector<int64_t> GridOffsets(const vgt_hip_grids* grids)
{
  std::vector<int64_t> offsets;
  const int32_t count = vgt_hip_tracking_grids_num_grids(grids);
  for (int32_t g = 0; g < count; g++)
    offsets.push_back(vgt_hip_tracking_grids_offset(grids, static_cast<size_t>(g)));
  return offsets;
}

class HipTrackingGridsHandle : public TrackingGridsHandle
{
public:
  explicit HipTrackingGridsHandle(vgt_hip_grids* grids)
      : TrackingGridsHandle(GridOffsets(grids), vgt_hip_tracking_grids_num_cells(grids)),
        grids_(grids) {}
  ~HipTrackingGridsHandle() override { vgt_hip_tracking_grids_destroy(grids_); }
  vgt_hip_grids* Get() const { return grids_; }

private:
  vgt_hip_grids* grids_;
};

class HipFilterGridHandle : public FilterGridHandle
{
public:
  explicit HipFilterGridHandle(vgt_hip_filter* filter)
      : FilterGridHandle(vgt_hip_filter_grid_num_cells(filter)), filter_(filter) {}
  ~HipFilterGridHandle() override { vgt_hip_filter_grid_destroy(filter_); }
  vgt_hip_filter* Get() const { return filter_; }

private:
  vgt_hip_filter* filter_;
};

class HipVoxelizationHelper : public DeviceVoxelizationHelperInterface,
                              public StridedRaycastInterface,
                              public DeferredFilterGridInterface,
                              public ExactRaycastInterface
{
public:
  HipVoxelizationHelper(const std::map<std::string, int32_t>& options,
                        const LoggingFunction& logging_fn)
  {
    const int32_t threads =
        RetrieveOptionOrDefault(options, "HIP_THREADS_PER_BLOCK", -1, logging_fn);
    const int32_t device = RetrieveOptionOrDefault(options, "HIP_DEVICE", 0, logging_fn);
    exact_fp64_ = RetrieveOptionOrDefault(options, "HIP_EXACT_FP64", 0, logging_fn) > 0;
    // One large cloud over several devices (vgt_hipx_raycast_points_split): HIP_SPLIT_HELPERS further devices,
    // counted on from HIP_DEVICE (wrapping around), share every cloud of at least HIP_SPLIT_MIN_POINTS points.
    const int32_t split_helpers = RetrieveOptionOrDefault(options, "HIP_SPLIT_HELPERS", 0, logging_fn);
    split_min_points_ = RetrieveOptionOrDefault(options, "HIP_SPLIT_MIN_POINTS", 1 << 20, logging_fn);
    int device_count = 0;
    if (split_helpers > 0 && vgt_hip_device_count(&device_count) == VGT_HIP_OK && device_count > 0)
      for (int32_t k = 0; k < split_helpers; k++) split_devices_.push_back((device + 1 + k) % device_count);
    if (logging_fn)
    {
      logging_fn(threads > 0 ? "Set HIP threads per block to specified " + std::to_string(threads)
                             : std::string("Set HIP threads per block to default 256"));
    }
    const int rc = vgt_hip_create(device, threads, &ctx_);
    if (rc != VGT_HIP_OK)
    {
      ctx_ = nullptr;
      if (logging_fn)
        logging_fn("Failed to load HIP runtime and set device: " +
                   std::string(vgt_hip_last_error()));
      return;
    }
    if (logging_fn)
    {
      char name[256] = {0};
      vgt_hip_device_name(device, name, sizeof(name));
      logging_fn("Using HIP device [" + std::to_string(device) + "] - Name: [" + name + "]");
    }
  }

  ~HipVoxelizationHelper() override { vgt_hip_destroy(ctx_); }

  bool IsAvailable() const override { return ctx_ != nullptr; }

  std::unique_ptr<TrackingGridsHandle> PrepareTrackingGrids(
      const int64_t num_cells, const int32_t num_grids) override
  {
    vgt_hip_grids* grids = nullptr;
    const int rc = vgt_hip_tracking_grids_create(ctx_, num_cells, num_grids, &grids);
    // zero-sized requests are runtime errors in the reference (cuda_voxelization_helpers.cu:457-460)
    if (rc != VGT_HIP_OK) throw std::runtime_error(vgt_hip_last_error());
    return std::unique_ptr<TrackingGridsHandle>(new HipTrackingGridsHandle(grids));
  }

  void RaycastPoints(
      const std::vector<float>& raw_points, const float max_range,
      const float* const grid_pointcloud_transform, const float voxel_size,
      const float inverse_voxel_size, const float grid_x_size, const float grid_y_size,
      const float grid_z_size, const int32_t num_x_voxels, const int32_t num_y_voxels,
      const int32_t num_z_voxels, TrackingGridsHandle& tracking_grids,
      const size_t tracking_grid_index) override
  {
    HipTrackingGridsHandle& real = dynamic_cast<HipTrackingGridsHandle&>(tracking_grids);
    const int64_t num_points = static_cast<int64_t>(raw_points.size() / 3);
    if (!split_devices_.empty() && num_points >= split_min_points_)
    {
      Check(vgt_hipx_raycast_points_split(
          ctx_, real.Get(), tracking_grid_index, split_devices_.data(), static_cast<int>(split_devices_.size()),
          raw_points.data(), num_points, max_range, grid_pointcloud_transform, voxel_size, inverse_voxel_size,
          grid_x_size, grid_y_size, grid_z_size, num_x_voxels, num_y_voxels, num_z_voxels));
      return;
    }
    Check(vgt_hip_raycast_points_f32(
        ctx_, real.Get(), tracking_grid_index, raw_points.data(), num_points, max_range, grid_pointcloud_transform,
        voxel_size, inverse_voxel_size, grid_x_size, grid_y_size, grid_z_size, num_x_voxels,
        num_y_voxels, num_z_voxels));
  }

  void RaycastStridedPoints(
      const uint8_t* data, const int64_t num_points, const int64_t point_step, const int64_t xyz_offset,
      const float max_range, const float* const grid_pointcloud_transform, const float voxel_size,
      const float inverse_voxel_size, const float grid_x_size, const float grid_y_size,
      const float grid_z_size, const int32_t num_x_voxels, const int32_t num_y_voxels,
      const int32_t num_z_voxels, TrackingGridsHandle& tracking_grids,
      const size_t tracking_grid_index) override
  {
    HipTrackingGridsHandle& real = dynamic_cast<HipTrackingGridsHandle&>(tracking_grids);
    Check(vgt_hip_raycast_pointcloud2_f32(
        ctx_, real.Get(), tracking_grid_index, data, num_points, point_step, xyz_offset, max_range,
        grid_pointcloud_transform, voxel_size, inverse_voxel_size, grid_x_size, grid_y_size, grid_z_size,
        num_x_voxels, num_y_voxels, num_z_voxels));
  }

  bool ExactFp64() const override { return exact_fp64_; }

  void RaycastPointsExact(
      const std::vector<double>& raw_points, const double max_range, const double* const grid_pointcloud_transform,
      const double voxel_size, const double inverse_voxel_size, const double grid_x_size, const double grid_y_size,
      const double grid_z_size, const int32_t num_x_voxels, const int32_t num_y_voxels, const int32_t num_z_voxels,
      TrackingGridsHandle& tracking_grids, const size_t tracking_grid_index) override
  {
    HipTrackingGridsHandle& real = dynamic_cast<HipTrackingGridsHandle&>(tracking_grids);
    Check(vgt_hip_raycast_points_f64(
        ctx_, real.Get(), tracking_grid_index, raw_points.data(), static_cast<int64_t>(raw_points.size() / 3), max_range,
        grid_pointcloud_transform, voxel_size, inverse_voxel_size, grid_x_size, grid_y_size, grid_z_size, num_x_voxels,
        num_y_voxels, num_z_voxels));
  }

  std::unique_ptr<FilterGridHandle> PrepareFilterGrid(
      const int64_t num_cells, const void* host_data_ptr) override
  {
    vgt_hip_filter* filter = nullptr;
    const int rc = vgt_hip_filter_grid_create(ctx_, num_cells,
                                              static_cast<const float*>(host_data_ptr), &filter);
    if (rc != VGT_HIP_OK) throw std::runtime_error(vgt_hip_last_error());
    return std::unique_ptr<FilterGridHandle>(new HipFilterGridHandle(filter));
  }

  std::unique_ptr<FilterGridHandle> PrepareFilterGridDeferred(
      const int64_t num_cells, const void* host_data_ptr) override
  {
    vgt_hip_filter* filter = nullptr;
    const int rc = vgt_hip_filter_grid_create_deferred(ctx_, num_cells,
                                                       static_cast<const float*>(host_data_ptr), &filter);
    if (rc != VGT_HIP_OK) throw std::runtime_error(vgt_hip_last_error());
    return std::unique_ptr<FilterGridHandle>(new HipFilterGridHandle(filter));
  }

  void FilterTrackingGrids(
      const TrackingGridsHandle& tracking_grids, const float percent_seen_free,
      const int32_t outlier_points_threshold, const int32_t num_cameras_seen_free,
      FilterGridHandle& filter_grid) override
  {
    const HipTrackingGridsHandle& real =
        dynamic_cast<const HipTrackingGridsHandle&>(tracking_grids);
    HipFilterGridHandle& real_filter = dynamic_cast<HipFilterGridHandle&>(filter_grid);
    if (exact_fp64_)
      Check(vgt_hip_filter_tracking_grids_f64(ctx_, real.Get(),
                                              static_cast<double>(percent_seen_free),
                                              outlier_points_threshold, num_cameras_seen_free,
                                              real_filter.Get()));
    else
      Check(vgt_hip_filter_tracking_grids(ctx_, real.Get(), percent_seen_free,
                                          outlier_points_threshold, num_cameras_seen_free,
                                          real_filter.Get()));
  }

  void RetrieveTrackingGrid(
      const TrackingGridsHandle& tracking_grids, const size_t tracking_grid_index,
      void* host_data_ptr) override
  {
    const HipTrackingGridsHandle& real =
        dynamic_cast<const HipTrackingGridsHandle&>(tracking_grids);
    Check(vgt_hip_retrieve_tracking_grid(ctx_, real.Get(), tracking_grid_index, host_data_ptr));
  }

  void RetrieveFilteredGrid(const FilterGridHandle& filter_grid, void* host_data_ptr) override
  {
    const HipFilterGridHandle& real = dynamic_cast<const HipFilterGridHandle&>(filter_grid);
    Check(vgt_hip_retrieve_filtered_grid(ctx_, real.Get(), host_data_ptr));
  }

private:
  vgt_hip_ctx* ctx_ = nullptr;
  bool exact_fp64_ = false;
  std::vector<int> split_devices_;   // helper devices of vgt_hipx_raycast_points_split (HIP_SPLIT_HELPERS)
  int64_t split_min_points_ = 1 << 20;
};
}  // namespace

std::vector<AvailableDevice> GetAvailableDevices()
{
  std::vector<AvailableDevice> devices;
  int count = 0;
  if (vgt_hip_device_count(&count) != VGT_HIP_OK) return devices;
  for (int idx = 0; idx < count; idx++)
  {
    char name[256] = {0};
    if (vgt_hip_device_name(idx, name, sizeof(name)) != VGT_HIP_OK) continue;
    std::map<std::string, int32_t> options;
    options["HIP_DEVICE"] = idx;
    devices.push_back(AvailableDevice("HIP - Device: [" + std::string(name) + "]", options));
  }
  return devices;
}

std::unique_ptr<DeviceVoxelizationHelperInterface> MakeHipVoxelizationHelper(
    const std::map<std::string, int32_t>& options, const LoggingFunction& logging_fn)
{
  return std::unique_ptr<DeviceVoxelizationHelperInterface>(
      new HipVoxelizationHelper(options, logging_fn));
}
}  // namespace hip_helpers
}  // namespace pointcloud_voxelization
VGT_NAMESPACE_END
}  // namespace voxelized_geometry_tools
