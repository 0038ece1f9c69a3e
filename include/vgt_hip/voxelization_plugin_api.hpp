// The device-helper plugin boundary of voxelized_geometry_tools, as seen by the HIP backend.
//
// When this repository's glue is compiled INSIDE the reference tree, the reference's own
// header is used (include/voxelized_geometry_tools/device_voxelization_interface.hpp, which
// only needs <std>).  Stand-alone (this repository, the GPU box) the declarations below take
// its place: same namespace, same names, same signatures -- the contract every helper
// (CUDA, OpenCL, and now HIP) implements:
//   LoggingFunction, AvailableDevice, RetrieveOptionOrDefault  device_voxelization_interface.hpp:20-70
//   TrackingGridsHandle / FilterGridHandle                      :73-127
//   DeviceVoxelizationHelperInterface (seven virtuals)          :129-175
#pragma once

#if __has_include(<voxelized_geometry_tools/device_voxelization_interface.hpp>)
#include <voxelized_geometry_tools/device_voxelization_interface.hpp>
#else

#include <cstddef>
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#ifndef VGT_NAMESPACE_BEGIN
#define VGT_NAMESPACE_BEGIN inline namespace v1 {
#endif
#ifndef VGT_NAMESPACE_END
#define VGT_NAMESPACE_END }
#endif

namespace voxelized_geometry_tools
{
VGT_NAMESPACE_BEGIN
namespace pointcloud_voxelization
{
using LoggingFunction = std::function<void(const std::string&)>;

class AvailableDevice
{
public:
  AvailableDevice(std::string name, std::map<std::string, int32_t> options)
      : name_(std::move(name)), options_(std::move(options)) {}
  const std::string& DeviceName() const { return name_; }
  const std::map<std::string, int32_t>& DeviceOptions() const { return options_; }

private:
  std::string name_;
  std::map<std::string, int32_t> options_;
};

// Looks `option` up in `options`, reports what it found through `logging_fn` (may be empty).
inline int32_t RetrieveOptionOrDefault(const std::map<std::string, int32_t>& options,
                                       const std::string& option, const int32_t default_value,
                                       const LoggingFunction& logging_fn)
{
  const auto it = options.find(option);
  const bool found = (it != options.end());
  const int32_t value = found ? it->second : default_value;
  if (logging_fn)
  {
    logging_fn(found ? ("Option [" + option + "] found, value [" + std::to_string(value) + "]")
                     : ("Option [" + option + "] not found, default [" +
                        std::to_string(value) + "]"));
  }
  return value;
}

// Opaque per-helper handles; helpers derive from them and down-cast what they are handed.
class TrackingGridsHandle
{
public:
  TrackingGridsHandle(const TrackingGridsHandle&) = delete;
  TrackingGridsHandle& operator=(const TrackingGridsHandle&) = delete;
  virtual ~TrackingGridsHandle() {}

  int64_t GetTrackingGridStartingOffset(const size_t index) const { return offsets_.at(index); }
  size_t GetNumTrackingGrids() const { return offsets_.size(); }
  int64_t NumCellsPerGrid() const { return num_cells_per_grid_; }

protected:
  TrackingGridsHandle(const std::vector<int64_t>& offsets, const int64_t num_cells_per_grid)
      : offsets_(offsets), num_cells_per_grid_(num_cells_per_grid) {}

private:
  std::vector<int64_t> offsets_;
  int64_t num_cells_per_grid_ = 0;
};

class FilterGridHandle
{
public:
  FilterGridHandle(const FilterGridHandle&) = delete;
  FilterGridHandle& operator=(const FilterGridHandle&) = delete;
  virtual ~FilterGridHandle() {}
  int64_t NumVoxels() const { return num_voxels_; }

protected:
  explicit FilterGridHandle(const int64_t num_voxels) : num_voxels_(num_voxels) {}

private:
  int64_t num_voxels_ = 0;
};

class DeviceVoxelizationHelperInterface
{
public:
  DeviceVoxelizationHelperInterface(const DeviceVoxelizationHelperInterface&) = delete;
  DeviceVoxelizationHelperInterface& operator=(const DeviceVoxelizationHelperInterface&) = delete;
  virtual ~DeviceVoxelizationHelperInterface() {}

  virtual bool IsAvailable() const = 0;

  virtual std::unique_ptr<TrackingGridsHandle> PrepareTrackingGrids(
      const int64_t num_cells, const int32_t num_grids) = 0;

  virtual void RaycastPoints(
      const std::vector<float>& raw_points, const float max_range,
      const float* const grid_pointcloud_transform, const float voxel_size,
      const float inverse_voxel_size, const float grid_x_size, const float grid_y_size,
      const float grid_z_size, const int32_t num_x_voxels, const int32_t num_y_voxels,
      const int32_t num_z_voxels, TrackingGridsHandle& tracking_grids,
      const size_t tracking_grid_index) = 0;

  virtual std::unique_ptr<FilterGridHandle> PrepareFilterGrid(
      const int64_t num_cells, const void* host_data_ptr) = 0;

  virtual void FilterTrackingGrids(
      const TrackingGridsHandle& tracking_grids, const float percent_seen_free,
      const int32_t outlier_points_threshold, const int32_t num_cameras_seen_free,
      FilterGridHandle& filter_grid) = 0;

  virtual void RetrieveTrackingGrid(
      const TrackingGridsHandle& tracking_grids, const size_t tracking_grid_index,
      void* host_data_ptr) = 0;

  virtual void RetrieveFilteredGrid(const FilterGridHandle& filter_grid, void* host_data_ptr) = 0;

protected:
  DeviceVoxelizationHelperInterface() = default;
};
}  // namespace pointcloud_voxelization
VGT_NAMESPACE_END
}  // namespace voxelized_geometry_tools
#endif  // reference header available
