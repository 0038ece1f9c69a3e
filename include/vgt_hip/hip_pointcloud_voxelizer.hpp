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
#include <mutex>
#include <string>
#include <vector>

struct vgt_hip_ctx;
struct vgt_hip_cells;

#include "hip_voxelization_helpers.h"
#include "host_types.hpp"

namespace vgt_hip
{
using voxelized_geometry_tools::pointcloud_voxelization::DeviceVoxelizationHelperInterface;
using voxelized_geometry_tools::pointcloud_voxelization::LoggingFunction;

// Host-clock phases of one VoxelizePointClouds call (seconds), for benchmarks: what VoxelizerRuntime's two numbers are made of.
struct VoxelizePhases
{
  double prepare_tracking_grids_s = 0;       // device buffer (pooled) + zeroing enqueued
  double filter_grid_enqueue_s = 0;          // page-locking the static environment + enqueueing its upload (not waited for)
  double raycast_s = 0;                      // per-cloud upload + raycast kernels, all clouds, dispatch threads joined
  double filter_grid_blocking_upload_s = 0;  // only with helpers that cannot defer the upload
  double filter_enqueue_s = 0;               // filter kernel enqueued
  double filter_and_download_s = 0;          // rest of the upload, filter kernel, download of the filtered grid (blocking)
  double release_s = 0;                      // device buffers back to the pool
  double total_s = 0;
  // by-value overload only: making the returned map (no cell touched) and waiting, before the download, for the threads
  // that fault its pages in
  double output_allocate_s = 0;
  double output_pages_wait_s = 0;
};

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

  // Phases of the most recent call that finished on this object (any thread).
  VoxelizePhases LastPhases() const
  {
    std::lock_guard<std::mutex> lock(phases_mutex_);
    return last_phases_;
  }

private:
  void EnforceAvailable() const;
  // both overloads; before_download (optional) runs after the filter kernel is enqueued and before the download starts
  VoxelizerRuntime DoVoxelizePointClouds(const OccupancyMap& static_environment,
                                         const PointCloudVoxelizationFilterOptions& filter_options,
                                         const std::vector<PointCloudWrapperSharedPtr>& pointclouds,
                                         OccupancyMap& output_environment,
                                         const std::function<void()>& before_download) const;
  std::unique_ptr<DeviceVoxelizationHelperInterface> helper_interface_;
  std::string device_name_ = "HipPointCloudVoxelizer";
  int dispatch_threads_ = 1;
  mutable std::mutex phases_mutex_;
  mutable VoxelizePhases last_phases_;
};

// OccupancyMap::ExtractSignedDistanceField<float>.  Throws std::invalid_argument for grids the
// reference rejects, std::runtime_error when no HIP device can be used (no CPU fallback).
SignedDistanceField ExtractSignedDistanceField(
    const OccupancyMap& map, const SignedDistanceFieldGenerationParameters& parameters);
// The same for a batch of maps of one size (e.g. the per-object or per-frame maps of a planner): one batched extraction
// (vgt_hip_sdf_batch_from_occupancy_f32) instead of a loop of calls; fields[i] belongs to maps[i] and is what the single
// call returns for it, bit for bit.  Throws std::invalid_argument when the maps' sizes differ.
std::vector<SignedDistanceField> ExtractSignedDistanceFields(
    const std::vector<const OccupancyMap*>& maps, const SignedDistanceFieldGenerationParameters& parameters);
// The free-standing SDF entry points share one context per device for the life of the process; that
// context keeps its device buffers between calls (no hipMalloc / hipFree per extraction).  This
// returns the memory (e.g. after one very large grid).
void ReleaseCachedDeviceMemory();

// ---- SDF consumers on the device, batched: one call for many query points / all voxels ----
// Results carry the reference's query semantics: has_value[i] == 0 <=> the reference's query object is empty.
struct DistanceEstimates
{
  std::vector<double> distance;     // EstimateDistanceQuery::Value()
  std::vector<uint8_t> has_value;   // EstimateDistanceQuery::HasValue()
};
struct Gradients
{
  std::vector<double> gradient;     // 3 per query / voxel: GradientQuery::Value().head<3>()
  std::vector<uint8_t> has_value;
};
// SignedDistanceField::EstimateLocationDistance for every point of `locations_xyz` (3 doubles per point, in the
// frame the field's origin transform maps to), signed_distance_field.hpp:808-833.
DistanceEstimates EstimateLocationDistances(const SignedDistanceField& sdf, const std::vector<double>& locations_xyz,
                                            int hip_device = 0);
// SignedDistanceField::GetLocationFineGradient (:1050-1091); throws std::runtime_error("Window size for fine
// gradient is too large for SDF") exactly when the reference does for one of the points.
Gradients GetLocationFineGradients(const SignedDistanceField& sdf, const std::vector<double>& locations_xyz,
                                   double nominal_window_size, int hip_device = 0);
// SignedDistanceField::GetIndexCoarseGradient at every voxel (:906-1016), X-major / Z fastest.
Gradients GetIndexCoarseGradients(const SignedDistanceField& sdf, bool enable_edge_gradients = false,
                                  int hip_device = 0);
// SignedDistanceField::ComputeLocalExtremaMap (:1205-1231): 3 doubles per voxel.
std::vector<double> ComputeLocalExtremaMap(const SignedDistanceField& sdf, int hip_device = 0);

// ---- the other three map types ----
// OccupancyComponentMap::ExtractSignedDistanceField<float> (occupancy_component_map.hpp:270-306).
SignedDistanceField ExtractSignedDistanceField(
    const OccupancyComponentMap& map, const SignedDistanceFieldGenerationParameters& parameters);

// A tagged map's cells on the device, for any number of extractions.  The methods are the
// reference's (tagged_object_occupancy_map.hpp:199-378 and
// tagged_object_occupancy_component_map.hpp:361-540), float instantiation.
class DeviceTaggedObjectMap
{
public:
  DeviceTaggedObjectMap(const TaggedObjectOccupancyMap& map, int hip_device = 0);
  DeviceTaggedObjectMap(const TaggedObjectOccupancyComponentMap& map, int hip_device = 0);
  ~DeviceTaggedObjectMap();
  DeviceTaggedObjectMap(const DeviceTaggedObjectMap&) = delete;
  DeviceTaggedObjectMap& operator=(const DeviceTaggedObjectMap&) = delete;

  SignedDistanceField ExtractSignedDistanceField(
      const std::vector<uint32_t>& objects_to_use,
      const SignedDistanceFieldGenerationParameters& parameters) const;
  std::map<uint32_t, SignedDistanceField> MakeSeparateObjectSDFs(
      const std::vector<uint32_t>& object_ids,
      const SignedDistanceFieldGenerationParameters& parameters) const;
  std::map<uint32_t, SignedDistanceField> MakeAllObjectSDFs(
      const SignedDistanceFieldGenerationParameters& parameters) const;
  SignedDistanceField ExtractFreeAndNamedObjectsSignedDistanceField(
      const SignedDistanceFieldGenerationParameters& parameters) const;
  // distinct object ids > 0, ascending
  std::vector<uint32_t> ObjectIds() const;

private:
  void Upload(const void* cells, int cell_bytes, int object_id_offset, int hip_device);
  SignedDistanceField EmptyField(const SignedDistanceFieldGenerationParameters& parameters) const;
  ::vgt_hip_ctx* ctx_ = nullptr;  // the process's context of the device (not owned)
  ::vgt_hip_cells* cells_ = nullptr;
  DenseGrid shape_;  // origin / frame / sizes of the map, for the fields handed back
};
}  // namespace vgt_hip
