// Link-time stand-in for builds without HIP: no devices, no helper.  The factory then reports
// "feature was not built" exactly as for the reference's CUDA / OpenCL stubs
// (include/voxelized_geometry_tools/device_pointcloud_voxelization.hpp:34-46).
#include "../../../include/vgt_hip/hip_voxelization_helpers.h"

namespace voxelized_geometry_tools
{
VGT_NAMESPACE_BEGIN
namespace pointcloud_voxelization
{
namespace hip_helpers
{
std::vector<AvailableDevice> GetAvailableDevices() { return {}; }

std::unique_ptr<DeviceVoxelizationHelperInterface> MakeHipVoxelizationHelper(
    const std::map<std::string, int32_t>&, const LoggingFunction&)
{
  return nullptr;
}
}  // namespace hip_helpers
}  // namespace pointcloud_voxelization
VGT_NAMESPACE_END
}  // namespace voxelized_geometry_tools
