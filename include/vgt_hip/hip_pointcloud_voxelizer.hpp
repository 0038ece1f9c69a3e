// Host drivers of the HIP backend:
//   HipPointCloudVoxelizer     the sibling of CudaPointCloudVoxelizer / OpenCLPointCloudVoxelizer
//                              (device_pointcloud_voxelization.hpp:73-91), running
//                              DevicePointCloudVoxelizer::DoVoxelizePointClouds
//                              (device_pointcloud_voxelization.cpp:65-181) over the HIP helper;
//   ExtractSignedDistanceField the device implementation of
//                              OccupancyMap::ExtractSignedDistanceField<float>
//                              (occupancy_map.hpp:174-210).
#pragma once

#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "hip_voxelization_helpers.h"
#include "host_types.hpp"

namespace vgt_hip
{
using voxelized_geometry_tools::pointcloud_voxelization::DeviceVoxelizationHelperInterface;
using voxelized_geometry_tools::pointcloud_voxelization::LoggingFunction;

class HipPointCloudVoxelizer
{
public:
  // Options: DISPATCH_PARALLELIZE (1), DISPATCH_NUM_THREADS (-1 = hardware concurrency), plus
  // the helper's HIP_* options.  Throws std::runtime_error when no device can be used, like
  // CudaPointCloudVoxelizer's constructor (device_pointcloud_voxelization.cpp:183-192).
  explicit HipPointCloudVoxelizer(const std::map<std::string, int32_t>& options,
                                  const LoggingFunction& logging_fn = {});

  // PointCloudVoxelizationInterface::VoxelizePointClouds
  // (pointcloud_voxelization_interface.hpp:246-292), both overloads.
  OccupancyMap VoxelizePointClouds(
      const OccupancyMap& static_environment,
      const PointCloudVoxelizationFilterOptions& filter_options,
      const std::vector<PointCloudWrapperSharedPtr>& pointclouds,
      const std::function<void(const VoxelizerRuntime&)>& runtime_log_fn = {}) const;

  VoxelizerRuntime VoxelizePointClouds(
      const OccupancyMap& static_environment,
      const PointCloudVoxelizationFilterOptions& filter_options,
      const std::vector<PointCloudWrapperSharedPtr>& pointclouds,
      OccupancyMap& output_environment) const;

private:
  void EnforceAvailable() const;
  std::unique_ptr<DeviceVoxelizationHelperInterface> helper_interface_;
  std::string device_name_ = "HipPointCloudVoxelizer";
  int dispatch_threads_ = 1;
};

// OccupancyMap::ExtractSignedDistanceField<float>.  Throws std::invalid_argument for grids the
// reference rejects, std::runtime_error when no HIP device can be used (no CPU fallback).
SignedDistanceField ExtractSignedDistanceField(
    const OccupancyMap& map, const SignedDistanceFieldGenerationParameters& parameters);
}  // namespace vgt_hip
