// hip_helpers: the HIP sibling of cuda_helpers / opencl_helpers
// (reference: include/voxelized_geometry_tools/cuda_voxelization_helpers.h:19-24).
// Implemented in voxelized_geometry_tools_amd/csrc/host/hip_voxelization_helpers.cc on top of
// the C ABI (include/vgt_hip.h); host/dummy_hip_voxelization_helpers.cc is the link-time stub
// for builds without HIP (reference pattern: src/.../dummy_cuda_voxelization_helpers.cc:15-22).
//
// Options understood by MakeHipVoxelizationHelper:
//   HIP_DEVICE             device index, default 0
//   HIP_THREADS_PER_BLOCK  threads per workgroup of the raycast / filter kernels, default 256
//   HIP_EXACT_FP64         1: filter ratio in double, as the reference's CPU voxelizer
//                          (pointcloud_voxelization_interface.hpp:55-86); default 0 (float,
//                          as the reference's device kernels)
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
std::vector<AvailableDevice> GetAvailableDevices();

std::unique_ptr<DeviceVoxelizationHelperInterface> MakeHipVoxelizationHelper(
    const std::map<std::string, int32_t>& options, const LoggingFunction& logging_fn);
}  // namespace hip_helpers
}  // namespace pointcloud_voxelization
VGT_NAMESPACE_END
}  // namespace voxelized_geometry_tools
