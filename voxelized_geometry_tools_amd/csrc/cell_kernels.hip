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

// One byte mask PER LISTED OBJECT in one pass over the cells (MakeSeparateObjectSDFs / MakeAllObjectSDFs,
// tagged_object_occupancy_map.hpp:249-290: one ExtractSignedDistanceField({id}) per object): masks[b][i] = the cell is
// filled and belongs to object ids[b].  The masks lie one after the other, which is the layout of a batch
// (SdfParams::batch): the three EDT passes then run ONCE for all the objects.
__global__ __launch_bounds__(kCellBlock) void CellObjectMasksKernel(const uint8_t* __restrict__ cells,
                                                                   int64_t num_cells, int cell_bytes,
                                                                   int object_id_offset,
                                                                   const uint32_t* __restrict__ ids, int num_ids,
                                                                   int unknown_is_filled, uint8_t* __restrict__ masks)
{
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < num_cells;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
  {
    const uint8_t* rec = cells + i * cell_bytes;
    const bool filled = OccupancyIsFilled(*reinterpret_cast<const float*>(rec), unknown_is_filled);
    const uint32_t id = *reinterpret_cast<const uint32_t*>(rec + object_id_offset);
    for (int b = 0; b < num_ids; b++) masks[static_cast<int64_t>(b) * num_cells + i] = (filled && id == ids[b]) ? 1 : 0;
  }
}

// The distinct object ids > 0 of a grid in ONE pass (MakeAllObjectSDFs' std::set, tagged_object_occupancy_map.hpp:
// 268-289): a hash set in device memory (0 = empty slot: valid ids are > 0).  Objects are contiguous in space, so a
// lane whose id equals its left neighbour's leaves the insert to it.  count_overflow[1] is set when a key finds no
// slot within the probe limit (more distinct ids than the table can hold): the caller then falls back to the
// one-id-per-launch scan below.
constexpr int kIdProbes = 128;
__global__ __launch_bounds__(kCellBlock) void DistinctIdsKernel(const uint8_t* __restrict__ cells, int64_t num_cells,
                                                               int cell_bytes, int object_id_offset,
                                                               uint32_t* __restrict__ table, int table_log2,
                                                               uint32_t* __restrict__ count_overflow)
{
  const uint32_t mask = (1u << table_log2) - 1u;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const int64_t rounds = (num_cells + stride - 1) / stride;
  int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  for (int64_t round = 0; round < rounds; round++, i += stride)
  {
    const uint32_t id =
        (i < num_cells) ? *reinterpret_cast<const uint32_t*>(cells + i * cell_bytes + object_id_offset) : 0u;
    const uint32_t left = static_cast<uint32_t>(__shfl_up(static_cast<int>(id), 1));
    if (id == 0u || ((threadIdx.x & (kWaveSize - 1)) != 0 && left == id)) continue;
    uint32_t slot = (id * 2654435761u) >> (32 - table_log2);
    bool placed = false;
    for (int probe = 0; probe < kIdProbes && !placed; probe++)
    {
      uint32_t key = table[slot];
      if (key == 0u) key = atomicCAS(&table[slot], 0u, id);
      placed = (key == id) || (key == 0u);
      slot = (slot + 1u) & mask;
    }
    if (!placed) atomicOr(&count_overflow[1], 1u);
  }
}

__global__ __launch_bounds__(kCellBlock) void CompactIdsKernel(const uint32_t* __restrict__ table, int64_t slots,
                                                              uint32_t* __restrict__ ids,
                                                              uint32_t* __restrict__ count_overflow)
{
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < slots;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
  {
    const uint32_t key = table[i];
    if (key != 0u) ids[atomicAdd(&count_overflow[0], 1u)] = key;
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

hipError_t LaunchCellObjectMasks(const void* cells_dev, int64_t num_cells, int cell_bytes, int object_id_offset,
                                 const uint32_t* ids_dev, int num_ids, int unknown_is_filled, uint8_t* masks_dev,
                                 hipStream_t stream)
{
  hipLaunchKernelGGL(CellObjectMasksKernel, dim3(CellGrid(num_cells)), dim3(kCellBlock), 0, stream,
                     static_cast<const uint8_t*>(cells_dev), num_cells, cell_bytes, object_id_offset, ids_dev, num_ids,
                     unknown_is_filled, masks_dev);
  return hipGetLastError();
}

hipError_t LaunchDistinctObjectIds(const void* cells_dev, int64_t num_cells, int cell_bytes, int object_id_offset,
                                   uint32_t* table_dev, int table_log2, uint32_t* ids_dev, uint32_t* count_overflow_dev,
                                   hipStream_t stream)
{
  hipLaunchKernelGGL(DistinctIdsKernel, dim3(CellGrid(num_cells)), dim3(kCellBlock), 0, stream,
                     static_cast<const uint8_t*>(cells_dev), num_cells, cell_bytes, object_id_offset, table_dev,
                     table_log2, count_overflow_dev);
  hipLaunchKernelGGL(CompactIdsKernel, dim3(CellGrid(int64_t{1} << table_log2)), dim3(kCellBlock), 0, stream, table_dev,
                     int64_t{1} << table_log2, ids_dev, count_overflow_dev);
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

// ---------------------------------------------------------------------------------------------
// SDF consumers, batched queries: SignedDistanceField::EstimateLocationDistance (trilinear estimate,
// signed_distance_field.hpp:808-833 over :300-378) and ::GetLocationFineGradient (:1050-1091 over
// :214-254).  One thread per query point.
//
// Operation order (the part of the reference that lives in common_robotics_utilities, whose source is
// not available here, is TrilinearInterpolate; this is the order used, in double, no FMA contraction):
//   g        = M * (x, y, z, 1), M = grid_from_world (InverseOriginTransform, column-major), row by row,
//              left to right (identity when M is NULL)
//   index    = floor(g * (1 / resolution)) per axis; out of bounds -> no value
//   centre   = (index + 0.5) * resolution; offset = g - centre; (lower, upper) per axis by
//              GetAxisInterpolationIndices (:276-313)
//   corners  = double(sdf) -/+ resolution * 0.5 (GetCorrectedCenterDistance, :259-273)
//   low      = (lower + 0.5) * resolution; t = (g - low) / ((low + resolution) - low) per axis
//   along x first, then y, then z, each as a * (1 - t) + b * t.
// Agreement with the reference is therefore to rounding (tests: 1e-5 absolute, the tolerance of
// BASELINE.json), bit-exact against the oracle's restatement of the same order.
// ---------------------------------------------------------------------------------------------
namespace
{
struct GridFromWorld
{
  double m[16];
  int enabled;
};

struct EstimateResult
{
  double value;
  bool has_value;
};

__device__ __forceinline__ void AxisInterpolationIndices(int initial, int size, double offset, int& lower, int& upper)
{
  lower = initial;
  upper = initial;
  if (offset >= 0.0)
  {
    upper = initial + 1;
    if (upper >= size)
    {
      upper = initial;
      lower = initial - 1;
      if (lower < 0) lower = initial;
    }
  }
  else
  {
    lower = initial - 1;
    if (lower < 0)
    {
      upper = initial + 1;
      lower = initial;
      if (upper >= size) upper = initial;
    }
  }
}

__device__ __forceinline__ double CorrectedCenterDistance(const float* __restrict__ sdf, int64_t index, double resolution)
{
  const double nominal = static_cast<double>(sdf[index]);
  const double offset = resolution * 0.5;
  return (nominal >= 0.0) ? nominal - offset : nominal + offset;
}

__device__ __forceinline__ double Lerp(double a, double b, double t) { return a * (1.0 - t) + b * t; }

__device__ EstimateResult EstimateDistance(const float* __restrict__ sdf, int nx, int ny, int nz, double resolution,
                                           const GridFromWorld& xf, double x, double y, double z)
{
  EstimateResult r{0.0, false};
  double g[3] = {x, y, z};
  if (xf.enabled)
  {
    const double* M = xf.m;
    g[0] = M[0] * x + M[4] * y + M[8] * z + M[12];
    g[1] = M[1] * x + M[5] * y + M[9] * z + M[13];
    g[2] = M[2] * x + M[6] * y + M[10] * z + M[14];
  }
  const double inv = 1.0 / resolution;
  const double fx = floor(g[0] * inv), fy = floor(g[1] * inv), fz = floor(g[2] * inv);
  if (!(fx >= 0.0 && fx < nx && fy >= 0.0 && fy < ny && fz >= 0.0 && fz < nz)) return r;  // also rejects NaN
  const int ix = static_cast<int>(fx), iy = static_cast<int>(fy), iz = static_cast<int>(fz);
  const double cx = (static_cast<double>(ix) + 0.5) * resolution;
  const double cy = (static_cast<double>(iy) + 0.5) * resolution;
  const double cz = (static_cast<double>(iz) + 0.5) * resolution;
  int lx, ux, ly, uy, lz, uz;
  AxisInterpolationIndices(ix, nx, g[0] - cx, lx, ux);
  AxisInterpolationIndices(iy, ny, g[1] - cy, ly, uy);
  AxisInterpolationIndices(iz, nz, g[2] - cz, lz, uz);
  const int64_t sx = static_cast<int64_t>(ny) * nz, sy = nz;
  auto at = [&](int a, int b, int c) { return CorrectedCenterDistance(sdf, a * sx + b * sy + c, resolution); };
  const double mmm = at(lx, ly, lz), mmp = at(lx, ly, uz), mpm = at(lx, uy, lz), mpp = at(lx, uy, uz);
  const double pmm = at(ux, ly, lz), pmp = at(ux, ly, uz), ppm = at(ux, uy, lz), ppp = at(ux, uy, uz);
  const double low_x = (static_cast<double>(lx) + 0.5) * resolution;
  const double low_y = (static_cast<double>(ly) + 0.5) * resolution;
  const double low_z = (static_cast<double>(lz) + 0.5) * resolution;
  const double tx = (g[0] - low_x) / ((low_x + resolution) - low_x);
  const double ty = (g[1] - low_y) / ((low_y + resolution) - low_y);
  const double tz = (g[2] - low_z) / ((low_z + resolution) - low_z);
  const double mm = Lerp(mmm, pmm, tx), mp = Lerp(mmp, pmp, tx), pm = Lerp(mpm, ppm, tx), pp = Lerp(mpp, ppp, tx);
  const double lo = Lerp(mm, pm, ty), hi = Lerp(mp, pp, ty);
  r.value = Lerp(lo, hi, tz);
  r.has_value = true;
  return r;
}

__global__ __launch_bounds__(256) void EstimateDistanceKernel(const float* __restrict__ sdf, int nx, int ny, int nz,
                                                             double resolution, const GridFromWorld xf,
                                                             const double* __restrict__ queries, int64_t num_queries,
                                                             double* __restrict__ distance,
                                                             uint8_t* __restrict__ has_value)
{
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < num_queries;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
  {
    const EstimateResult r = EstimateDistance(sdf, nx, ny, nz, resolution, xf, queries[3 * i], queries[3 * i + 1],
                                              queries[3 * i + 2]);
    distance[i] = r.has_value ? r.value : __longlong_as_double(0x7ff8000000000000ll);
    if (has_value) has_value[i] = r.has_value ? 1 : 0;
  }
}

// ComputeAxisFineGradient (:214-254); returns false for the reference's "window too large" throw
__device__ __forceinline__ bool AxisFineGradient(const EstimateResult& point, const EstimateResult& minus,
                                                 const EstimateResult& plus, double q, double lo, double hi,
                                                 double& gradient)
{
  if (point.has_value && minus.has_value && plus.has_value)
    gradient = (plus.value - minus.value) / (hi - lo);
  else if (point.has_value && minus.has_value)
    gradient = (point.value - minus.value) / (q - lo);
  else if (point.has_value && plus.has_value)
    gradient = (plus.value - point.value) / (hi - q);
  else
    return false;
  return true;
}

__global__ __launch_bounds__(256) void FineGradientKernel(const float* __restrict__ sdf, int nx, int ny, int nz,
                                                         double resolution, const GridFromWorld xf,
                                                         const double* __restrict__ queries, int64_t num_queries,
                                                         double window, double* __restrict__ gradient,
                                                         uint8_t* __restrict__ has_value,
                                                         uint32_t* __restrict__ window_too_large)
{
  const double nan = __longlong_as_double(0x7ff8000000000000ll);
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < num_queries;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
  {
    const double x = queries[3 * i], y = queries[3 * i + 1], z = queries[3 * i + 2];
    const EstimateResult point = EstimateDistance(sdf, nx, ny, nz, resolution, xf, x, y, z);
    double gx = nan, gy = nan, gz = nan;
    bool ok = point.has_value;  // CheckLocationInBounds == the estimate has a value
    if (ok)
    {
      const EstimateResult mx = EstimateDistance(sdf, nx, ny, nz, resolution, xf, x - window, y, z);
      const EstimateResult px = EstimateDistance(sdf, nx, ny, nz, resolution, xf, x + window, y, z);
      const EstimateResult my = EstimateDistance(sdf, nx, ny, nz, resolution, xf, x, y - window, z);
      const EstimateResult py = EstimateDistance(sdf, nx, ny, nz, resolution, xf, x, y + window, z);
      const EstimateResult mz = EstimateDistance(sdf, nx, ny, nz, resolution, xf, x, y, z - window);
      const EstimateResult pz = EstimateDistance(sdf, nx, ny, nz, resolution, xf, x, y, z + window);
      const bool fine = AxisFineGradient(point, mx, px, x, x - window, x + window, gx) &&
                        AxisFineGradient(point, my, py, y, y - window, y + window, gy) &&
                        AxisFineGradient(point, mz, pz, z, z - window, z + window, gz);
      if (!fine)
      {
        atomicOr(window_too_large, 1u);
        ok = false;
        gx = gy = gz = nan;
      }
    }
    gradient[3 * i] = gx;
    gradient[3 * i + 1] = gy;
    gradient[3 * i + 2] = gz;
    if (has_value) has_value[i] = ok ? 1 : 0;
  }
}

GridFromWorld MakeGridFromWorld(const double* grid_from_world_host)
{
  GridFromWorld xf;
  xf.enabled = grid_from_world_host ? 1 : 0;
  for (int k = 0; k < 16; k++) xf.m[k] = grid_from_world_host ? grid_from_world_host[k] : 0.0;
  return xf;
}
}  // namespace

hipError_t LaunchEstimateDistance(const float* sdf_dev, int64_t nx, int64_t ny, int64_t nz, double resolution,
                                  const double* grid_from_world_host, const double* queries_dev, int64_t num_queries,
                                  double* distance_dev, uint8_t* has_value_dev, hipStream_t stream)
{
  if (num_queries <= 0) return hipSuccess;
  hipLaunchKernelGGL(EstimateDistanceKernel, dim3(CellGrid(num_queries)), dim3(256), 0, stream, sdf_dev,
                     static_cast<int>(nx), static_cast<int>(ny), static_cast<int>(nz), resolution,
                     MakeGridFromWorld(grid_from_world_host), queries_dev, num_queries, distance_dev, has_value_dev);
  return hipGetLastError();
}

hipError_t LaunchFineGradient(const float* sdf_dev, int64_t nx, int64_t ny, int64_t nz, double resolution,
                              const double* grid_from_world_host, const double* queries_dev, int64_t num_queries,
                              double nominal_window_size, double* gradient_dev, uint8_t* has_value_dev,
                              uint32_t* window_too_large_dev, hipStream_t stream)
{
  if (num_queries <= 0) return hipSuccess;
  hipLaunchKernelGGL(FineGradientKernel, dim3(CellGrid(num_queries)), dim3(256), 0, stream, sdf_dev,
                     static_cast<int>(nx), static_cast<int>(ny), static_cast<int>(nz), resolution,
                     MakeGridFromWorld(grid_from_world_host), queries_dev, num_queries, fabs(nominal_window_size),
                     gradient_dev, has_value_dev, window_too_large_dev);
  return hipGetLastError();
}

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

// ---------------------------------------------------------------------------------------------
// SDF consumer: SignedDistanceField::ComputeLocalExtremaMap (signed_distance_field.hpp:1205-1231 over
// FollowGradientsToLocalExtremaUnsafe :385-480, GradientIsEffectiveFlat :482-497, GetNextFromGradient
// :499-541).  The reference follows, from every cell in X-major order, the coarse gradient (edge gradients
// on, rotated by the origin transform) one cell at a time -- up the field outside obstacles, down inside --
// until it meets a flat cell, leaves the grid, meets a cell whose answer is known, or revisits a cell of its
// own path, and stores the grid-frame location of the cell it ended at in every cell of the path.
//
// Every cell has exactly one successor, so the cells form a functional graph and the stored value of a cell
// is a property of where its chain ends:
//   * a cell that is its own successor (flat gradient, or no component beyond the step threshold): its location;
//   * a successor outside the grid: +infinity in all three components;
//   * a cycle of two or more cells: the reference stores the location of the cycle cell at which the FIRST walk
//     that reaches the cycle entered it, and walks start in X-major order, so that is the entry cell of the
//     chain from the smallest cell index of the cycle's basin.
// Device formulation: successor table -> pointer doubling to a representative of every chain's end ->
// cycles are walked once for their smallest member and membership marks -> per cycle the smallest basin
// index (atomicMin) -> its chain is walked to the entry cell -> labels.
// ---------------------------------------------------------------------------------------------
namespace
{
constexpr int32_t kOffGrid = -2;

__device__ __forceinline__ bool EffectiveFlat(double gx, double gy, double gz, double step)
{
  return fabs(gx) <= step && fabs(gy) <= step && fabs(gz) <= step;
}

__global__ __launch_bounds__(256) void ExtremaSuccessorKernel(const float* __restrict__ sdf, int nx, int ny, int nz,
                                                             double resolution, const Rotation rot,
                                                             int32_t* __restrict__ next)
{
  const int64_t total = static_cast<int64_t>(nx) * ny * nz;
  const int64_t sx = static_cast<int64_t>(ny) * nz, sy = nz;
  const double step = resolution * 0.06125;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
  {
    const int z = static_cast<int>(i % nz);
    const int y = static_cast<int>((i / nz) % ny);
    const int x = static_cast<int>(i / sx);
    // GetGridAlignedIndexCoarseGradient with edge gradients (:923-1016), then the rotation (:906-921)
    double gx = 0.0, gy = 0.0, gz = 0.0;
    if (x > 0 && y > 0 && z > 0 && x < nx - 1 && y < ny - 1 && z < nz - 1)
    {
      const double inv_twice_resolution = 1.0 / (2.0 * resolution);
      gx = static_cast<double>(sdf[i + sx] - sdf[i - sx]) * inv_twice_resolution;
      gy = static_cast<double>(sdf[i + sy] - sdf[i - sy]) * inv_twice_resolution;
      gz = static_cast<double>(sdf[i + 1] - sdf[i - 1]) * inv_twice_resolution;
    }
    else
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
    if (rot.enabled)
    {
      const double wx = rot.m[0] * gx + rot.m[1] * gy + rot.m[2] * gz;
      const double wy = rot.m[3] * gx + rot.m[4] * gy + rot.m[5] * gz;
      const double wz = rot.m[6] * gx + rot.m[7] * gy + rot.m[8] * gz;
      gx = wx;
      gy = wy;
      gz = wz;
    }
    int32_t successor = static_cast<int32_t>(i);
    if (!EffectiveFlat(gx, gy, gz, step))
    {
      // GetNextFromGradient: downhill inside an obstacle, uphill outside
      if (sdf[i] < 0.0f)
      {
        gx = gx * -1.0;
        gy = gy * -1.0;
        gz = gz * -1.0;
      }
      int tx = x, ty = y, tz = z;
      if (gx > step) tx += 1; else if (gx < -step) tx -= 1;
      if (gy > step) ty += 1; else if (gy < -step) ty -= 1;
      if (gz > step) tz += 1; else if (gz < -step) tz -= 1;
      if (tx < 0 || tx >= nx || ty < 0 || ty >= ny || tz < 0 || tz >= nz)
        successor = kOffGrid;
      else
        successor = static_cast<int32_t>(tx * sx + ty * sy + tz);
    }
    next[i] = successor;
  }
}

// one round of pointer doubling, in place (an entry read mid-update is still a valid, farther jump)
__global__ __launch_bounds__(256) void ExtremaJumpKernel(int32_t* __restrict__ jump, int64_t total)
{
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
  {
    const int32_t j = jump[i];
    if (j >= 0)
    {
      const int32_t jj = jump[j];
      if (jj != j) jump[i] = jj;
    }
  }
}

// representatives that sit on a cycle of two or more cells: walk the cycle once for its smallest member, once
// more to mark every member with it (idempotent: representatives of the same cycle write the same values)
__global__ __launch_bounds__(256) void ExtremaCycleKernel(const int32_t* __restrict__ next,
                                                         const int32_t* __restrict__ jump, int64_t total,
                                                         int32_t* __restrict__ cycle_id)
{
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
  {
    const int32_t r = jump[i];
    if (r < 0 || next[r] == r) continue;  // chain leaves the grid or ends at a fixed cell
    if (cycle_id[r] >= 0) continue;       // another thread has taken this cycle (it finishes within this launch)
    int32_t smallest = r;
    for (int32_t w = next[r]; w != r; w = next[w]) smallest = min(smallest, w);
    cycle_id[r] = smallest;
    for (int32_t w = next[r]; w != r; w = next[w]) cycle_id[w] = smallest;
  }
}

__global__ __launch_bounds__(256) void ExtremaBasinMinKernel(const int32_t* __restrict__ jump,
                                                            const int32_t* __restrict__ cycle_id, int64_t total,
                                                            int32_t* __restrict__ basin_min)
{
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
  {
    const int32_t r = jump[i];
    if (r < 0) continue;
    const int32_t c = cycle_id[r];
    if (c >= 0) atomicMin(&basin_min[c], static_cast<int32_t>(i));
  }
}

// per cycle (thread of its smallest member): the cycle cell at which the chain from the basin's smallest cell enters
__global__ __launch_bounds__(256) void ExtremaEntryKernel(const int32_t* __restrict__ next,
                                                         const int32_t* __restrict__ cycle_id,
                                                         const int32_t* __restrict__ basin_min, int64_t total,
                                                         int32_t* __restrict__ entry)
{
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
  {
    if (cycle_id[i] != static_cast<int32_t>(i)) continue;
    int32_t w = basin_min[i];
    while (cycle_id[w] != static_cast<int32_t>(i)) w = next[w];
    entry[i] = w;
  }
}

__global__ __launch_bounds__(256) void ExtremaLabelKernel(const int32_t* __restrict__ next,
                                                         const int32_t* __restrict__ jump,
                                                         const int32_t* __restrict__ cycle_id,
                                                         const int32_t* __restrict__ entry, int nx, int ny, int nz,
                                                         double resolution, double* __restrict__ extrema)
{
  const int64_t total = static_cast<int64_t>(nx) * ny * nz;
  const int64_t sx = static_cast<int64_t>(ny) * nz;
  const double inf = __longlong_as_double(0x7ff0000000000000ll);
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
  {
    const int32_t r = jump[i];
    double ex = inf, ey = inf, ez = inf;
    if (r >= 0)
    {
      const int32_t c = cycle_id[r];
      const int32_t cell = (c >= 0) ? entry[c] : r;
      // GridIndexToLocationInGridFrame: the cell centre
      ex = (static_cast<double>(cell / sx) + 0.5) * resolution;
      ey = (static_cast<double>((cell / nz) % ny) + 0.5) * resolution;
      ez = (static_cast<double>(cell % nz) + 0.5) * resolution;
    }
    extrema[3 * i] = ex;
    extrema[3 * i + 1] = ey;
    extrema[3 * i + 2] = ez;
  }
}

__global__ __launch_bounds__(256) void FillInt32Kernel(int32_t* __restrict__ p, int64_t total, int32_t value)
{
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
    p[i] = value;
}
}  // namespace

size_t LocalExtremaScratchBytes(int64_t num_cells) { return static_cast<size_t>(num_cells) * 5 * sizeof(int32_t); }

// scratch_dev: LocalExtremaScratchBytes(nx * ny * nz) bytes.  num_cells must be below 2^31.
hipError_t LaunchLocalExtremaMap(const float* sdf_dev, int64_t nx, int64_t ny, int64_t nz, double resolution,
                                 const double* rotation_host, double* extrema_dev, void* scratch_dev,
                                 hipStream_t stream)
{
  const int64_t total = nx * ny * nz;
  int32_t* next = static_cast<int32_t*>(scratch_dev);
  int32_t* jump = next + total;
  int32_t* cycle_id = jump + total;
  int32_t* basin_min = cycle_id + total;
  int32_t* entry = basin_min + total;
  Rotation rot;
  rot.enabled = rotation_host ? 1 : 0;
  for (int k = 0; k < 9; k++) rot.m[k] = rotation_host ? rotation_host[k] : 0.0;
  const dim3 grid(CellGrid(total)), block(256);
  hipLaunchKernelGGL(ExtremaSuccessorKernel, grid, block, 0, stream, sdf_dev, static_cast<int>(nx),
                     static_cast<int>(ny), static_cast<int>(nz), resolution, rot, next);
  hipError_t err = hipMemcpyAsync(jump, next, static_cast<size_t>(total) * sizeof(int32_t), hipMemcpyDeviceToDevice, stream);
  if (err != hipSuccess) return err;
  // after k rounds every entry has advanced at least 2^k steps along its chain (or reached its end)
  int rounds = 1;
  while ((int64_t{1} << rounds) < total) rounds++;
  for (int k = 0; k < rounds; k++) hipLaunchKernelGGL(ExtremaJumpKernel, grid, block, 0, stream, jump, total);
  hipLaunchKernelGGL(FillInt32Kernel, grid, block, 0, stream, cycle_id, total, int32_t{-1});
  hipLaunchKernelGGL(FillInt32Kernel, grid, block, 0, stream, basin_min, total, int32_t{0x7fffffff});
  hipLaunchKernelGGL(ExtremaCycleKernel, grid, block, 0, stream, next, jump, total, cycle_id);
  hipLaunchKernelGGL(ExtremaBasinMinKernel, grid, block, 0, stream, jump, cycle_id, total, basin_min);
  hipLaunchKernelGGL(ExtremaEntryKernel, grid, block, 0, stream, next, cycle_id, basin_min, total, entry);
  hipLaunchKernelGGL(ExtremaLabelKernel, grid, block, 0, stream, next, jump, cycle_id, entry, static_cast<int>(nx),
                     static_cast<int>(ny), static_cast<int>(nz), resolution, extrema_dev);
  return hipGetLastError();
}

}  // namespace vgt
