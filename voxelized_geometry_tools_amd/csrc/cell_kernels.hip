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
