// SDF entry points of the map types whose cells carry more than an occupancy:
//   OccupancyComponentMap                  cell = { float occupancy; uint32 component }
//   TaggedObjectOccupancyMap               cell = { float occupancy; uint32 object_id }
//   TaggedObjectOccupancyComponentMap      cell = { float occupancy; uint32 object_id, component,
//                                                   spatial_segment }
// The reference evaluates an `is_filled_fn` per voxel on the host
// (occupancy_component_map.hpp:270-306, tagged_object_occupancy_map.hpp:199-247,292-378,
// tagged_object_occupancy_component_map.hpp:361-540) and then runs the same EDT.  Here the cell
// records are uploaded once; a kernel evaluates the predicate into one byte per voxel, which feeds
// the mask entry of the Z scan, so any number of per-object SDFs reuse the upload.
#include "edt_device.hpp"

namespace vgt
{
namespace
{
constexpr int kCellBlock = 256;

__device__ __forceinline__ bool OccupancyIsFilled(float occupancy, int unknown_is_filled)
{
  return (occupancy > 0.5f) || (unknown_is_filled && (occupancy == 0.5f));
}

// objects: sorted, distinct.  Short lists are scanned, long ones bisected.
__device__ __forceinline__ bool ContainsObject(const uint32_t* __restrict__ objects, int count, uint32_t id)
{
  if (count <= 8)
  {
    bool hit = false;
    for (int i = 0; i < count; i++) hit |= (objects[i] == id);
    return hit;
  }
  int lo = 0, hi = count - 1;
  while (lo <= hi)
  {
    const int mid = (lo + hi) >> 1;
    const uint32_t v = objects[mid];
    if (v == id) return true;
    if (v < id)
      lo = mid + 1;
    else
      hi = mid - 1;
  }
  return false;
}

// mode 0: every cell counts (no object list given, or a map type without object ids)
// mode 1: only cells whose object id is in `objects`  (tagged_object_occupancy_map.hpp:216-233)
// mode 2: only cells of a named object, id > 0        (tagged_object_occupancy_map.hpp:326-343)
__global__ __launch_bounds__(kCellBlock) void CellMaskKernel(const uint8_t* __restrict__ cells,
                                                            int64_t num_cells, int cell_bytes,
                                                            int object_id_offset, int mode,
                                                            const uint32_t* __restrict__ objects,
                                                            int num_objects, int unknown_is_filled,
                                                            uint8_t* __restrict__ mask)
{
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < num_cells;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
  {
    const uint8_t* rec = cells + i * cell_bytes;
    const float occupancy = *reinterpret_cast<const float*>(rec);
    bool filled = OccupancyIsFilled(occupancy, unknown_is_filled);
    if (mode != 0)
    {
      const uint32_t id = *reinterpret_cast<const uint32_t*>(rec + object_id_offset);
      filled = filled && ((mode == 1) ? ContainsObject(objects, num_objects, id) : (id > 0u));
    }
    mask[i] = filled ? 1 : 0;
  }
}

// Smallest object id strictly greater than `after`: MakeAllObjectSDFs' scan for the distinct
// ids (tagged_object_occupancy_map.hpp:268-289), one id per launch.  result[0] = the id,
// result[1] = 1 if one was found (so that id 0xffffffff is not mistaken for "none").
__global__ __launch_bounds__(kCellBlock) void NextObjectIdKernel(const uint8_t* __restrict__ cells,
                                                                int64_t num_cells, int cell_bytes,
                                                                int object_id_offset, uint32_t after,
                                                                uint32_t* __restrict__ result)
{
  uint32_t best = 0xffffffffu;
  int found = 0;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < num_cells;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
  {
    const uint32_t id = *reinterpret_cast<const uint32_t*>(cells + i * cell_bytes + object_id_offset);
    if (id > after)
    {
      best = min(best, id);
      found = 1;
    }
  }
  for (int off = kWaveSize / 2; off > 0; off >>= 1)
  {
    best = min(best, static_cast<uint32_t>(__shfl_xor(static_cast<int>(best), off)));
    found |= __shfl_xor(found, off);
  }
  if ((threadIdx.x & (kWaveSize - 1)) == 0 && found)
  {
    atomicMin(&result[0], best);
    atomicOr(&result[1], 1u);
  }
}

// ExtractFreeAndNamedObjectsSignedDistanceField's combination
// (tagged_object_occupancy_map.hpp:347-372): the free-space field where it is non-negative, the
// named-objects field where that one is non-positive, 0 elsewhere; extrema of the result.
__global__ __launch_bounds__(kCellBlock) void CombineFreeAndNamedKernel(
    const float* free_sdf, const float* __restrict__ named_sdf, int64_t num_cells,
    float* out /* may be free_sdf */, uint32_t* __restrict__ minmax_enc)
{
  uint32_t lo = 0xffffffffu, hi = 0u;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < num_cells;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
  {
    const float f = free_sdf[i];
    const float n = named_sdf[i];
    float v;
    if (f >= 0.0f)
      v = f;
    else if (n <= -0.0f)
      v = n;
    else
      v = 0.0f;
    out[i] = v;
    const uint32_t e = EncodeOrdered(v);
    lo = min(lo, e);
    hi = max(hi, e);
  }
  BlockMinMax(lo, hi, minmax_enc);
}

int CellGrid(int64_t num_cells)
{
  const int64_t blocks = (num_cells + kCellBlock - 1) / kCellBlock;
  return static_cast<int>(blocks < 1 ? 1 : (blocks > 65536 ? 65536 : blocks));
}
}  // namespace

hipError_t LaunchCellMask(const void* cells_dev, int64_t num_cells, int cell_bytes, int object_id_offset,
                          int mode, const uint32_t* objects_dev, int num_objects, int unknown_is_filled,
                          uint8_t* mask_dev, hipStream_t stream)
{
  hipLaunchKernelGGL(CellMaskKernel, dim3(CellGrid(num_cells)), dim3(kCellBlock), 0, stream,
                     static_cast<const uint8_t*>(cells_dev), num_cells, cell_bytes, object_id_offset, mode,
                     objects_dev, num_objects, unknown_is_filled, mask_dev);
  return hipGetLastError();
}

hipError_t LaunchNextObjectId(const void* cells_dev, int64_t num_cells, int cell_bytes, int object_id_offset,
                              uint32_t after, uint32_t* result_dev, hipStream_t stream)
{
  hipLaunchKernelGGL(NextObjectIdKernel, dim3(CellGrid(num_cells)), dim3(kCellBlock), 0, stream,
                     static_cast<const uint8_t*>(cells_dev), num_cells, cell_bytes, object_id_offset, after,
                     result_dev);
  return hipGetLastError();
}

hipError_t LaunchCombineFreeAndNamed(const float* free_sdf_dev, const float* named_sdf_dev, int64_t num_cells,
                                     float* out_dev, uint32_t* minmax_enc, hipStream_t stream)
{
  hipLaunchKernelGGL(CombineFreeAndNamedKernel, dim3(CellGrid(num_cells)), dim3(kCellBlock), 0, stream,
                     free_sdf_dev, named_sdf_dev, num_cells, out_dev, minmax_enc);
  return hipGetLastError();
}
}  // namespace vgt

// ---------------------------------------------------------------------------------------------
// SDF consumer: SignedDistanceField::GetGridAlignedIndexCoarseGradient
// (signed_distance_field.hpp:923-1016) for every voxel at once.  Interior voxels: central
// differences, the float difference taken in float and scaled in double exactly as the
// reference's expression evaluates; voxels on a face of the grid: one-sided differences in double
// when edge gradients are enabled (:951-1004), otherwise no value (NaN, has_value = 0).
// `rotation` (optional, 9 doubles row-major) turns the result into GetIndexCoarseGradient's
// (:906-921): OriginTransform() * gradient, translation not applied to a direction.
// ---------------------------------------------------------------------------------------------
namespace vgt
{
namespace
{
struct Rotation
{
  double m[9];
  int enabled;
};

__global__ __launch_bounds__(256) void CoarseGradientKernel(const float* __restrict__ sdf, int nx, int ny, int nz,
                                                           double resolution, int enable_edge_gradients,
                                                           const Rotation rot, double* __restrict__ gradient,
                                                           uint8_t* __restrict__ has_value)
{
  const int64_t total = static_cast<int64_t>(nx) * ny * nz;
  const int64_t sx = static_cast<int64_t>(ny) * nz, sy = nz;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
  {
    const int z = static_cast<int>(i % nz);
    const int y = static_cast<int>((i / nz) % ny);
    const int x = static_cast<int>(i / sx);
    double gx = 0.0, gy = 0.0, gz = 0.0;
    bool ok = true;
    if (x > 0 && y > 0 && z > 0 && x < nx - 1 && y < ny - 1 && z < nz - 1)
    {
      const double inv_twice_resolution = 1.0 / (2.0 * resolution);
      gx = static_cast<double>(sdf[i + sx] - sdf[i - sx]) * inv_twice_resolution;
      gy = static_cast<double>(sdf[i + sy] - sdf[i - sy]) * inv_twice_resolution;
      gz = static_cast<double>(sdf[i + 1] - sdf[i - 1]) * inv_twice_resolution;
    }
    else if (enable_edge_gradients)
    {
      const int lx = max(0, x - 1), hx = min(nx - 1, x + 1);
      const int ly = max(0, y - 1), hy = min(ny - 1, y + 1);
      const int lz = max(0, z - 1), hz = min(nz - 1, z + 1);
      const double x_increment = static_cast<double>(hx - lx) * resolution;
      const double y_increment = static_cast<double>(hy - ly) * resolution;
      const double z_increment = static_cast<double>(hz - lz) * resolution;
      if (x_increment > 0.0)
        gx = (static_cast<double>(sdf[i + (hx - x) * sx]) - static_cast<double>(sdf[i - (x - lx) * sx])) *
             (1.0 / x_increment);
      if (y_increment > 0.0)
        gy = (static_cast<double>(sdf[i + (hy - y) * sy]) - static_cast<double>(sdf[i - (y - ly) * sy])) *
             (1.0 / y_increment);
      if (z_increment > 0.0)
        gz = (static_cast<double>(sdf[i + (hz - z)]) - static_cast<double>(sdf[i - (z - lz)])) * (1.0 / z_increment);
    }
    else
    {
      ok = false;
      gx = gy = gz = __longlong_as_double(0x7ff8000000000000ll);
    }
    if (ok && rot.enabled)
    {
      const double wx = rot.m[0] * gx + rot.m[1] * gy + rot.m[2] * gz;
      const double wy = rot.m[3] * gx + rot.m[4] * gy + rot.m[5] * gz;
      const double wz = rot.m[6] * gx + rot.m[7] * gy + rot.m[8] * gz;
      gx = wx;
      gy = wy;
      gz = wz;
    }
    gradient[3 * i + 0] = gx;
    gradient[3 * i + 1] = gy;
    gradient[3 * i + 2] = gz;
    if (has_value) has_value[i] = ok ? 1 : 0;
  }
}
}  // namespace

hipError_t LaunchCoarseGradient(const float* sdf_dev, int64_t nx, int64_t ny, int64_t nz, double resolution,
                                int enable_edge_gradients, const double* rotation_host, double* gradient_dev,
                                uint8_t* has_value_dev, hipStream_t stream)
{
  Rotation rot;
  rot.enabled = rotation_host ? 1 : 0;
  for (int k = 0; k < 9; k++) rot.m[k] = rotation_host ? rotation_host[k] : 0.0;
  hipLaunchKernelGGL(CoarseGradientKernel, dim3(CellGrid(nx * ny * nz)), dim3(256), 0, stream, sdf_dev,
                     static_cast<int>(nx), static_cast<int>(ny), static_cast<int>(nz), resolution,
                     enable_edge_gradients, rot, gradient_dev, has_value_dev);
  return hipGetLastError();
}
}  // namespace vgt
