// hip_helpers: the HIP sibling of cuda_helpers / opencl_helpers
// (reference: include/voxelized_geometry_tools/cuda_voxelization_helpers.h:19-24).
// Implemented in voxelized_geometry_tools_amd/csrc/host/hip_voxelization_helpers.cc on top of
// the C ABI (include/vgt_hip.h); host/dummy_hip_voxelization_helpers.cc is the link-time stub
// for builds without HIP (reference pattern: src/.../dummy_cuda_voxelization_helpers.cc:15-22).
//
// Options understood by MakeHipVoxelizationHelper:
//   HIP_DEVICE             device index, default 0
//   HIP_THREADS_PER_BLOCK  threads per workgroup of the raycast / filter kernels; not given: 256, and 512 for the
//                          raycast kernel of large (direction-sorted) clouds
//   HIP_EXACT_FP64         1: the arithmetic of the reference's CPU voxelizer -- the rays are walked in float64 from double
//                          points and transforms (cpu_pointcloud_voxelization.cpp:167-206,208-436; the helper's
//                          ExactRaycastInterface, which HipPointCloudVoxelizer feeds through
//                          CopyPointLocationIntoDoublePtr) and the filter's ratio is a double
//                          (pointcloud_voxelization_interface.hpp:55-86); default 0 (float32 walk and ratio, as the
//                          reference's device kernels)
#pragma once

#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "voxelization_plugin_api.hpp"

namespace voxelized_geometry_tools
{
VGT_NAMESPACE_BEGIN
namespace pointcloud_voxelization
{
namespace hip_helpers
{
// Extension of the plugin interface implemented by the HIP helper: raycast a
// buffer of fixed-size records holding x, y, z as consecutive FLOAT32 (a PointCloud2 data
// buffer) in place.  Callers discover it with dynamic_cast on the helper.
class StridedRaycastInterface
{
public:
  virtual ~StridedRaycastInterface() {}
  virtual void RaycastStridedPoints(
      const uint8_t* data, int64_t num_points, int64_t point_step, int64_t xyz_offset, float max_range,
      const float* grid_pointcloud_transform, float voxel_size, float inverse_voxel_size, float grid_x_size,
      float grid_y_size, float grid_z_size, int32_t num_x_voxels, int32_t num_y_voxels, int32_t num_z_voxels,
      TrackingGridsHandle& tracking_grids, size_t tracking_grid_index) = 0;
};

// Extension of the plugin interface implemented by the HIP helper: PrepareFilterGrid without waiting for the copy.
// The upload of the static environment (64 MiB at 256^3) then runs beside the raycasts instead of after them;
// FilterTrackingGrids / RetrieveFilteredGrid are ordered behind it.  `host_data_ptr` must stay valid and unchanged until
// RetrieveFilteredGrid has returned or the handle is destroyed.  Callers discover it with dynamic_cast on the helper.
class DeferredFilterGridInterface
{
public:
  virtual ~DeferredFilterGridInterface() {}
  virtual std::unique_ptr<FilterGridHandle> PrepareFilterGridDeferred(int64_t num_cells, const void* host_data_ptr) = 0;
};

// Extension of the plugin interface implemented by the HIP helper: the float64 walk of the reference's CPU voxelizer on the
// device (vgt_hip_raycast_points_f64).  ExactFp64(): whether the helper was made with HIP_EXACT_FP64 -- a caller that
// wants CPU-voxelizer-identical grids then hands its points over as doubles.  Discovered with dynamic_cast on the helper.
class ExactRaycastInterface
{
public:
  virtual ~ExactRaycastInterface() {}
  virtual bool ExactFp64() const = 0;
  virtual void RaycastPointsExact(
      const std::vector<double>& raw_points, double max_range, const double* grid_pointcloud_transform,
      double voxel_size, double inverse_voxel_size, double grid_x_size, double grid_y_size, double grid_z_size,
      int32_t num_x_voxels, int32_t num_y_voxels, int32_t num_z_voxels, TrackingGridsHandle& tracking_grids,
      size_t tracking_grid_index) = 0;
};

std::vector<AvailableDevice> GetAvailableDevices();

std::unique_ptr<DeviceVoxelizationHelperInterface> MakeHipVoxelizationHelper(
    const std::map<std::string, int32_t>& options, const LoggingFunction& logging_fn);
}  // namespace hip_helpers
}  // namespace pointcloud_voxelization
VGT_NAMESPACE_END
}  // namespace voxelized_geometry_tools
