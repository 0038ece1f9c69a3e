/*
 * vgt_hip.h -- C ABI of libvgt_hip.so, the MI355X (gfx950) backend for the SDF/EDT and
 * pointcloud-raycast-voxelization hot path of calderpg/voxelized_geometry_tools.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch types.
 * Each entry point names the reference interface it stands in for (paths relative to
 * the reference checkout; I/ = include/voxelized_geometry_tools/,
 * S/ = src/voxelized_geometry_tools/).  The reference-side binding a maintainer would
 * add is shown in INTEGRATION.md; include/vgt_hip/ holds the C++ glue that implements
 * the reference's DeviceVoxelizationHelperInterface on top of these calls.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; the message is then
 *     available from vgt_hip_last_error() (thread-local).  No exception crosses the ABI.
 *   - dense grids are X-major / Z fastest: index = x*(ny*nz) + y*nz + z
 *     (S/cuda_voxelization_helpers.cu:683-684).
 *   - "host" pointers are ordinary process memory; "dev" pointers are HIP device memory
 *     on the context's device.  The caller owns every buffer it passes in; the library
 *     owns what it hands out behind the opaque handles.
 *   - there is no CPU fallback: without a usable HIP device vgt_hip_create() fails.
 */
#ifndef VGT_HIP_H_
#define VGT_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: the functions declared between this push and the pop at the end
 * of the header are its whole dynamic symbol table (tests/test_capi_symbols.py checks `nm -D`). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define VGT_HIP_ABI_VERSION 2

typedef struct vgt_hip_ctx vgt_hip_ctx;       /* one context <-> one device + one stream */
typedef struct vgt_hip_grids vgt_hip_grids;   /* tracking grids (TrackingGridsHandle)    */
typedef struct vgt_hip_filter vgt_hip_filter; /* filter grid (FilterGridHandle)          */

/* Error codes (also the return values). */
enum {
  VGT_HIP_OK = 0,
  VGT_HIP_ERR_INVALID_ARGUMENT = 1, /* maps to std::invalid_argument in the C++ glue */
  VGT_HIP_ERR_RUNTIME = 2,          /* HIP error; maps to std::runtime_error          */
  VGT_HIP_ERR_UNAVAILABLE = 3       /* no device / device index out of range           */
};

int vgt_hip_abi_version(void);
const char* vgt_hip_last_error(void);

/* ---- device enumeration: hip_helpers::GetAvailableDevices()
 *      (sibling of cuda_helpers::GetAvailableDevices, S/cuda_voxelization_helpers.cu:791-821) */
int vgt_hip_device_count(int* count);
int vgt_hip_device_name(int device, char* buffer, size_t buffer_size);

/* ---- context: the state behind hip_helpers::MakeHipVoxelizationHelper(options, log)
 *      (I/cuda_voxelization_helpers.h:19-24; ctor S/cuda_voxelization_helpers.cu:562-639).
 *      threads_per_block <= 0 selects the defaults: 256 for the filter and the small-cloud raycast
 *      kernel, and the size the direction-sorted raycast kernel is tuned for (512: its LDS table, the
 *      re-deal of rays by walk length and the flush scale with the workgroup); a positive value (a
 *      multiple of 64, at most 1024) is used for all of them.  Options HIP_DEVICE /
 *      HIP_THREADS_PER_BLOCK of the C++ glue land here. */
int vgt_hip_create(int device, int threads_per_block, vgt_hip_ctx** out_ctx);
void vgt_hip_destroy(vgt_hip_ctx* ctx);
/* A context keeps the device buffers of its host-pointer entry points (SDF input / field /
 * workspace, point-cloud staging) across calls, growing them on demand, so that repeated calls do
 * not pay for hipMalloc / hipFree (the reference allocates per call, S/cuda_voxelization_helpers.cu:
 * 676-680).  vgt_hip_trim gives that memory back (waits for the context's stream first). */
int vgt_hip_trim(vgt_hip_ctx* ctx);
/* Run all work of this context on an externally owned hipStream_t (e.g. the caller's
 * framework stream); NULL is HIP's legacy default stream.  vgt_hip_reset_stream goes back to the
 * context's own (non-blocking) stream.  Both drain the stream in use first. */
int vgt_hip_set_stream(vgt_hip_ctx* ctx, void* hip_stream);
int vgt_hip_reset_stream(vgt_hip_ctx* ctx);
int vgt_hip_synchronize(vgt_hip_ctx* ctx);
int vgt_hip_device_of(const vgt_hip_ctx* ctx);

/* =====================  pointcloud raycast voxelization  ===================== */

/* DeviceVoxelizationHelperInterface::PrepareTrackingGrids
 * (I/device_voxelization_interface.hpp:148-149; S/cuda_voxelization_helpers.cu:641-658):
 * num_grids zeroed grids of int32[2*num_cells] = (seen_free, seen_filled) per cell,
 * grid g starting at element offset g*num_cells*2. */
int vgt_hip_tracking_grids_create(vgt_hip_ctx* ctx, int64_t num_cells, int32_t num_grids,
                                  vgt_hip_grids** out_grids);
void vgt_hip_tracking_grids_destroy(vgt_hip_grids* grids);
int64_t vgt_hip_tracking_grids_num_cells(const vgt_hip_grids* grids);
int32_t vgt_hip_tracking_grids_num_grids(const vgt_hip_grids* grids);
int64_t vgt_hip_tracking_grids_offset(const vgt_hip_grids* grids, size_t grid_index);
void* vgt_hip_tracking_grids_dev_ptr(const vgt_hip_grids* grids, size_t grid_index);
int vgt_hip_tracking_grids_clear(vgt_hip_ctx* ctx, vgt_hip_grids* grids);

/* DeviceVoxelizationHelperInterface::RaycastPoints
 * (I/device_voxelization_interface.hpp:151-158; kernel S/cuda_voxelization_helpers.cu:73-356):
 * float32 DDA of num_points xyz points (AoS, cloud frame) through the grid, transform =
 * 16 floats column-major (grid <- cloud).  Safe to call concurrently from several host
 * threads on one context with distinct grid_index
 * (S/device_pointcloud_voxelization.cpp:147-149).  Returns after the kernel has been
 * enqueued AND the host point buffer has been consumed. */
int vgt_hip_raycast_points_f32(vgt_hip_ctx* ctx, vgt_hip_grids* grids, size_t grid_index,
                               const float* points_xyz_host, int64_t num_points,
                               float max_range, const float* grid_pointcloud_transform,
                               float voxel_size, float inverse_voxel_size,
                               float grid_x_size, float grid_y_size, float grid_z_size,
                               int32_t num_x_voxels, int32_t num_y_voxels,
                               int32_t num_z_voxels);
/* PointCloud2 ingestion: the message's data buffer is uploaded as it is and the
 * kernel reads x, y, z in place -- instead of the per-point virtual
 * CopyPointLocationIntoFloatPtr gather of S/device_pointcloud_voxelization.cpp:130-136 over
 * PointCloud2Wrapper (I/pointcloud_voxelization_ros_interface.hpp:68-91).
 *   cloud_data_host  sensor_msgs/PointCloud2::data, num_points = width * height records
 *   point_step       bytes per record;  xyz_offset = offset of field "x" (y and z follow, FLOAT32;
 *                    S/pointcloud_voxelization_ros_interface.cpp:49-78).  Both multiples of 4. */
int vgt_hip_raycast_pointcloud2_f32(vgt_hip_ctx* ctx, vgt_hip_grids* grids, size_t grid_index,
                                    const uint8_t* cloud_data_host, int64_t num_points,
                                    int64_t point_step, int64_t xyz_offset, float max_range,
                                    const float* grid_pointcloud_transform, float voxel_size,
                                    float inverse_voxel_size, float grid_x_size,
                                    float grid_y_size, float grid_z_size, int32_t num_x_voxels,
                                    int32_t num_y_voxels, int32_t num_z_voxels);
/* Same, points already resident on the device (bench / device-resident pipelines). */
int vgt_hip_raycast_points_f32_dev(vgt_hip_ctx* ctx, vgt_hip_grids* grids, size_t grid_index,
                                   const float* points_xyz_dev, int64_t num_points,
                                   float max_range, const float* grid_pointcloud_transform,
                                   float voxel_size, float inverse_voxel_size,
                                   float grid_x_size, float grid_y_size, float grid_z_size,
                                   int32_t num_x_voxels, int32_t num_y_voxels,
                                   int32_t num_z_voxels);
/* HIP_EXACT_FP64 mode: float64 DDA with the arithmetic of the reference's CPU voxelizer
 * (CpuPointCloudVoxelizer::DoRaycastSinglePoint, S/cpu_pointcloud_voxelization.cpp:208-436);
 * points and transform are doubles. */
int vgt_hip_raycast_points_f64(vgt_hip_ctx* ctx, vgt_hip_grids* grids, size_t grid_index,
                               const double* points_xyz_host, int64_t num_points,
                               double max_range, const double* grid_pointcloud_transform,
                               double voxel_size, double inverse_voxel_size,
                               double grid_x_size, double grid_y_size, double grid_z_size,
                               int32_t num_x_voxels, int32_t num_y_voxels,
                               int32_t num_z_voxels);

/* DeviceVoxelizationHelperInterface::PrepareFilterGrid
 * (I/device_voxelization_interface.hpp:160-161; S/cuda_voxelization_helpers.cu:701-708):
 * device copy of the static environment's float occupancy. */
int vgt_hip_filter_grid_create(vgt_hip_ctx* ctx, int64_t num_cells,
                               const float* occupancy_host, vgt_hip_filter** out_filter);
/* The same without waiting for the copy: the upload runs on a copy stream beside whatever the context does next (the
 * raycasts of HipPointCloudVoxelizer, which prepares the filter grid first), and the calls that use the grid --
 * filter, retrieve, destroy -- are ordered behind it.  `occupancy_host` must stay valid and unchanged until one of
 * vgt_hip_retrieve_filtered_grid / vgt_hip_filter_grid_destroy has returned for this grid (it is page-locked by the
 * library for that time). */
int vgt_hip_filter_grid_create_deferred(vgt_hip_ctx* ctx, int64_t num_cells,
                                        const float* occupancy_host, vgt_hip_filter** out_filter);
void vgt_hip_filter_grid_destroy(vgt_hip_filter* filter);
int64_t vgt_hip_filter_grid_num_cells(const vgt_hip_filter* filter);
/* The grid's device buffer.  Waits for a deferred upload first (the pointer is then usable on any stream); NULL, with the
 * reason in vgt_hip_last_error(), when that wait fails. */
void* vgt_hip_filter_grid_dev_ptr(const vgt_hip_filter* filter);

/* DeviceVoxelizationHelperInterface::FilterTrackingGrids
 * (I/device_voxelization_interface.hpp:163-166; kernel S/cuda_voxelization_helpers.cu:358-426).
 * ratio_in_double = 0: float ratio as the reference device kernels; 1: double ratio as
 * PointCloudVoxelizationFilterOptions::CountsSeenAs (I/pointcloud_voxelization_interface.hpp:55-86). */
int vgt_hip_filter_tracking_grids(vgt_hip_ctx* ctx, const vgt_hip_grids* grids,
                                  float percent_seen_free, int32_t outlier_points_threshold,
                                  int32_t num_cameras_seen_free, vgt_hip_filter* filter);
int vgt_hip_filter_tracking_grids_f64(vgt_hip_ctx* ctx, const vgt_hip_grids* grids,
                                      double percent_seen_free,
                                      int32_t outlier_points_threshold,
                                      int32_t num_cameras_seen_free, vgt_hip_filter* filter);

/* DeviceVoxelizationHelperInterface::RetrieveTrackingGrid / RetrieveFilteredGrid
 * (I/device_voxelization_interface.hpp:168-173; S/cuda_voxelization_helpers.cu:734-767):
 * blocking copies of num_cells*8 / num_cells*4 bytes; all earlier work of the context
 * has finished when they return. */
int vgt_hip_retrieve_tracking_grid(vgt_hip_ctx* ctx, const vgt_hip_grids* grids,
                                   size_t grid_index, void* host_out);
int vgt_hip_retrieve_filtered_grid(vgt_hip_ctx* ctx, const vgt_hip_filter* filter,
                                   void* host_out);

/* ==========================  signed distance field  ========================== */

/* OccupancyMap::ExtractSignedDistanceField<float>(params)
 * (I/occupancy_map.hpp:174-210 -> I/signed_distance_field_generation.hpp:39-285 ->
 *  S/signed_distance_field_generation.cpp:258-391).  occupancy_host / sdf_host are
 * float[nx*ny*nz]; out_min / out_max receive what SignedDistanceField::Lock() caches
 * (I/signed_distance_field.hpp:765-787) and may be NULL.  Blocking.  The two host arrays are
 * page-locked for the call, the context keeps its device buffers between calls (vgt_hip_trim
 * returns them), and grids of 2^27 voxels and more overlap upload, kernels and download chunk
 * by chunk on three streams (the threshold is fixed in the product library; testing builds can move
 * it with vgt_hip_testing_set_host_pipeline_min_voxels); the result does not depend on it. */
int vgt_hip_sdf_from_occupancy_f32(vgt_hip_ctx* ctx, const float* occupancy_host,
                                   int64_t nx, int64_t ny, int64_t nz, double resolution,
                                   int unknown_is_filled, int add_virtual_border,
                                   float* sdf_host, float* out_min, float* out_max);
/* Same for the map types whose predicate is not a pure occupancy threshold
 * (I/occupancy_component_map.hpp:270-306, I/tagged_object_occupancy_map.hpp:199-247):
 * the caller evaluates is_filled on the host into one byte per voxel. */
int vgt_hip_sdf_from_mask_u8(vgt_hip_ctx* ctx, const uint8_t* filled_mask_host, int64_t nx,
                             int64_t ny, int64_t nz, double resolution,
                             int add_virtual_border, float* sdf_host, float* out_min,
                             float* out_max);

/* Device-resident form (bench, device pipelines): input and output stay
 * in HBM, the caller provides the scratch workspace.  minmax_dev, if non-NULL, receives
 * {min, max} as two floats on the device after the call (stream-ordered). */
size_t vgt_hip_sdf_workspace_bytes(int64_t nx, int64_t ny, int64_t nz);
/* As above for a context set to EDT variant `variant` (0 = the default pipeline = vgt_hip_sdf_workspace_bytes; the
 * cross-check variant 1 exists in testing builds only, the product library returns 0 for it).  The default
 * workspace holds the class records of pass 1 (0.25 bytes per voxel), the int32 intermediate field (4 bytes per voxel)
 * and the line passes' scratch, which grows with the axis lengths, not with the volume (the spilled stack entries and
 * sign words of the at most 4096 waves in flight: 1.1 GB for a 1024^3 grid, 4.4 GB at 2048 x 2048 x 1024; the launches
 * use as many workgroups as the scratch they are given holds).  5.8 GB in all at 1024^3. */
size_t vgt_hip_sdf_workspace_bytes_for_variant(int64_t nx, int64_t ny, int64_t nz, int variant);
int vgt_hip_sdf_dev(vgt_hip_ctx* ctx, const float* occupancy_dev, int64_t nx, int64_t ny,
                    int64_t nz, double resolution, int unknown_is_filled,
                    int add_virtual_border, float* sdf_dev, void* workspace_dev,
                    size_t workspace_bytes, float* minmax_dev);
/* As vgt_hip_sdf_dev, bracketing each kernel with HIP events on the context's stream.
 * kernel_ms[0..2] = pass 1 (class records), Y-pass, X-pass(+finalize) durations of this call. Blocking. */
int vgt_hip_sdf_dev_timed(vgt_hip_ctx* ctx, const float* occupancy_dev, int64_t nx,
                          int64_t ny, int64_t nz, double resolution, int unknown_is_filled,
                          int add_virtual_border, float* sdf_dev, void* workspace_dev,
                          size_t workspace_bytes, float* minmax_dev, float* kernel_ms);
/* ---- Batches: `batch` grids of the same extents in one call (many small maps, or many masks of one map).
 * The reference extracts one field per call and loops -- TaggedObjectOccupancyMap::MakeSeparateObjectSDFs /
 * MakeAllObjectSDFs run one whole ExtractSignedDistanceField per object id
 * (I/tagged_object_occupancy_map.hpp:249-290) -- and on the grid sizes of its own examples and tests (8^3 - 40^3)
 * one extraction is a few hundred work items for a GPU that holds 16 384 waves.  Here the three passes run ONCE
 * over the whole batch: the grids lie one after the other, [batch][nx][ny][nz], in the input, in the output and in
 * every intermediate buffer; pass 1 and the Y pass see one grid of batch * nx slices (their lines never leave a
 * slice), the X pass deals (grid, y, z segment) items to the same persistent workgroups, every grid has its own
 * extrema.  Results are bit-identical to `batch` single calls.
 *   limits      the per-axis limit of every SDF entry point for nx, ny, nz; batch * nx * ny < 2^28
 *   minmax_dev  NULL or 2 * batch floats: {min, max} of grid 0, of grid 1, ...
 * vgt_hip_sdf_batch_from_occupancy_f32 takes `batch` host arrays (any addresses) and hands back `batch` fields and
 * their extrema (out_min / out_max: NULL or `batch` floats each); it cuts batches that exceed the limits or 2 GiB of
 * device buffers into several launches by itself.  Blocking. */
size_t vgt_hip_sdf_batch_workspace_bytes(int64_t batch, int64_t nx, int64_t ny, int64_t nz);
int vgt_hip_sdf_batch_dev(vgt_hip_ctx* ctx, const float* occupancy_dev, int64_t batch, int64_t nx, int64_t ny,
                          int64_t nz, double resolution, int unknown_is_filled, int add_virtual_border,
                          float* sdf_dev, void* workspace_dev, size_t workspace_bytes, float* minmax_dev);
int vgt_hip_sdf_batch_from_occupancy_f32(vgt_hip_ctx* ctx, const float* const* occupancy_host, int64_t batch,
                                         int64_t nx, int64_t ny, int64_t nz, double resolution,
                                         int unknown_is_filled, int add_virtual_border, float* const* sdf_host,
                                         float* out_min, float* out_max);
#ifdef VGT_HIP_TESTING
/* ---- Testing builds only: libvgt_hip_testing.so (make -C voxelized_geometry_tools_amd/csrc testing), which the parity
 * tests load next to the product library.  The product library exports none of these and contains none of the
 * cross-check implementations. ----
 * Selects the EDT pipeline (both exact): 0 = default (pass 1 writes class records, lane-per-line sweeps: one lane runs
 * the Felzenszwalb-Huttenlocher stack of one line, stack tops in LDS, any extent); 1 = the independent cross-check: an
 * int16 distance field along Z as pass 1, then a pruned outward search per voxel straight from HBM (any size). */
int vgt_hip_set_edt_variant(vgt_hip_ctx* ctx, int variant);
/* The final conversion float(sqrt(double(d2)) * resolution) has a fast evaluation with an exact fallback
 * (csrc/edt_device.hpp); this runs both over d2 in [first_d2, first_d2 + count) on the device and reports how many
 * results differ (must be 0) and the first differing d2 (UINT64_MAX if none). */
int vgt_hip_debug_finalize_check(vgt_hip_ctx* ctx, int64_t first_d2, int64_t count,
                                 double resolution, uint64_t* mismatches,
                                 uint64_t* first_mismatch);
/* Smallest grid (voxels) that the host-pointer SDF entry points pipeline (upload / kernels / download overlapped);
 * default 2^27, negative = never.  Lets the tests run that path on small grids. */
int vgt_hip_testing_set_host_pipeline_min_voxels(int64_t min_voxels);
/* Lines of at most `rows` rows (0 - 128) take the short-line kernels (csrc/edt_short_kernels.hip) instead of the sweeps,
 * whatever the number of items: lets the tests and benches run either formulation on any length.  Negative: back to the
 * product's rule (64 rows; the Y pass of launches of at most 1024 items: 128). */
int vgt_hip_testing_set_short_line_rows(int rows);
/* Pass 1 alone, for a test of the record format itself (csrc/vgt_internal.hpp, ClassRecord): the class records of a
 * device-resident occupancy grid, [x][64-voxel word][y] x 4 uint32 (mask_lo, mask_hi, below2, above2), into records_dev
 * (vgt_hip_testing_class_record_bytes bytes); summary_dev (optional): the 4-byte slab summaries per line, in which case the
 * records carry no one-class marks (a slab cannot know).  Blocking. */
size_t vgt_hip_testing_class_record_bytes(int64_t nx, int64_t ny, int64_t nz);
int vgt_hip_testing_class_records_dev(vgt_hip_ctx* ctx, const float* occupancy_dev, int64_t nx, int64_t ny, int64_t nz,
                                      int unknown_is_filled, int64_t z_offset, void* records_dev, void* summary_dev);
#endif /* VGT_HIP_TESTING */

/* ---- SDFs of the map types whose cells carry more than an occupancy ----
 * Replaces the per-voxel `is_filled_fn` + EDT of
 *   OccupancyComponentMap::ExtractSignedDistanceField            occupancy_component_map.hpp:270-306
 *   TaggedObjectOccupancyMap::ExtractSignedDistanceField         tagged_object_occupancy_map.hpp:199-247
 *     ::MakeSeparateObjectSDFs / ::MakeAllObjectSDFs             :249-290
 *     ::ExtractFreeAndNamedObjectsSignedDistanceField            :292-378
 *   TaggedObjectOccupancyComponentMap (same four)                tagged_object_occupancy_component_map.hpp:361-540
 * The raw cell store (`GetImmutableRawData().data()`) is uploaded once; every extraction after that
 * evaluates its predicate on the device and runs the same exact EDT.
 *   cell_bytes        sizeof the cell: 8 (OccupancyComponentCell, TaggedObjectOccupancyCell) or
 *                     16 (TaggedObjectOccupancyComponentCell); 4 (plain OccupancyCell) also works.
 *                     The float occupancy is the first member of all of them.
 *   object_id_offset  byte offset of the uint32 object id inside a cell (4 for both tagged types),
 *                     or -1 for a type without one (OccupancyComponentCell: the component is not
 *                     used by its SDF). */
typedef struct vgt_hip_cells vgt_hip_cells;
int vgt_hip_cells_create(vgt_hip_ctx* ctx, const void* cells_host, int64_t nx, int64_t ny,
                         int64_t nz, int32_t cell_bytes, int32_t object_id_offset,
                         vgt_hip_cells** out_cells);
void vgt_hip_cells_destroy(vgt_hip_cells* cells);
/* Distinct object ids > 0 in ascending order (what MakeAllObjectSDFs collects in a std::set).
 * Writes at most `capacity` ids; *count receives the number found. */
int vgt_hip_cells_object_ids(vgt_hip_ctx* ctx, vgt_hip_cells* cells, uint32_t* ids_out,
                             int64_t capacity, int64_t* count);
/* ExtractSignedDistanceField(objects_to_use, parameters): a cell is filled when its occupancy is
 * (> 0.5, or == 0.5 with unknown_is_filled) AND (num_objects == 0 or its object id is listed).
 * One call per object id = MakeSeparateObjectSDFs. */
int vgt_hip_cells_sdf(vgt_hip_ctx* ctx, vgt_hip_cells* cells, const uint32_t* objects_to_use,
                      int64_t num_objects, double resolution, int unknown_is_filled,
                      int add_virtual_border, float* sdf_host, float* out_min, float* out_max);
/* MakeSeparateObjectSDFs(object_ids) / MakeAllObjectSDFs (:249-290) as ONE batch: sdf_host[k] receives
 * ExtractSignedDistanceField({object_ids[k]}) -- the field of vgt_hip_cells_sdf with that one id, bit for bit --
 * and out_min[k] / out_max[k] (NULL or num_objects floats each) its extrema.  One pass over the cells writes every
 * object's mask, the EDT passes run once over all of them (see "Batches" above), the fields come back through
 * page-locked copies.  Object lists that exceed the limits of a batch or 4 GiB of device buffers are cut into several
 * launches.  Blocking. */
int vgt_hip_cells_object_sdfs(vgt_hip_ctx* ctx, vgt_hip_cells* cells, const uint32_t* object_ids,
                              int64_t num_objects, double resolution, int unknown_is_filled, int add_virtual_border,
                              float* const* sdf_host, float* out_min, float* out_max);
/* ExtractFreeAndNamedObjectsSignedDistanceField: the field of all filled cells where it is >= 0,
 * the field of the filled cells of named objects (id > 0) where that is <= 0, else 0. */
int vgt_hip_cells_free_and_named_objects_sdf(vgt_hip_ctx* ctx, vgt_hip_cells* cells,
                                             double resolution, int unknown_is_filled,
                                             int add_virtual_border, float* sdf_host,
                                             float* out_min, float* out_max);

/* ---- deferred per-kernel timing (benchmarks) ----
 * Between start and stop every vgt_hip_sdf_dev call (or vgt_hip_sdf_slab_begin_dev /
 * _finish_dev pair called with kernel_ms == NULL) on this context records HIP events around its
 * kernels on the stream they run on, WITHOUT synchronising; stop waits for the stream once and
 * returns, per call, the milliseconds of {pass 1 (+ slab record fix-up), Y pass, X pass}.  Calls beyond
 * max_calls are not recorded. */
int vgt_hip_timing_start(vgt_hip_ctx* ctx, int32_t max_calls);
int vgt_hip_timing_stop(vgt_hip_ctx* ctx, float* kernel_ms /* [max_calls][3] */,
                        int32_t* num_calls);

/* ---- SDF consumers ----
 * SignedDistanceField<float>::GetGridAlignedIndexCoarseGradient
 * (I/signed_distance_field.hpp:923-1016) for every voxel of a field at once: gradient[3 * i + a]
 * (double), i = x*ny*nz + y*nz + z.  Voxels on a face of the grid get one-sided differences when
 * enable_edge_gradients is set, otherwise NaN and has_value[i] = 0 (has_value may be NULL).
 * rotation (NULL or 9 doubles, row-major) = the rotation of OriginTransform(): with it the
 * result is GetIndexCoarseGradient's (:906-921). */
int vgt_hip_sdf_coarse_gradient(vgt_hip_ctx* ctx, const float* sdf_host, int64_t nx, int64_t ny,
                                int64_t nz, double resolution, int enable_edge_gradients,
                                const double* rotation, double* gradient_host,
                                uint8_t* has_value_host);
/* Same on device buffers (e.g. straight after vgt_hip_sdf_dev, without leaving the device). */
int vgt_hip_sdf_coarse_gradient_dev(vgt_hip_ctx* ctx, const float* sdf_dev, int64_t nx, int64_t ny,
                                    int64_t nz, double resolution, int enable_edge_gradients,
                                    const double* rotation, double* gradient_dev,
                                    uint8_t* has_value_dev);

/* Batched point queries against a float SDF (x, y, z as 3 doubles per query, in the frame
 * `grid_from_world` maps from: 16 doubles column-major = InverseOriginTransform, NULL = the grid
 * frame itself).
 *   vgt_hip_sdf_estimate_distance   SignedDistanceField::EstimateLocationDistance (trilinear estimate
 *                                   over the eight surrounding cell centres, I/signed_distance_field.hpp:
 *                                   808-833 over :259-378); distance[i] = NaN and has_value[i] = 0 for a
 *                                   query outside the grid.
 *   vgt_hip_sdf_fine_gradient       ::GetLocationFineGradient (:1050-1091 over :214-254): differences of
 *                                   seven estimates at +-|nominal_window_size| along the query frame's
 *                                   axes; 3 doubles per query.  A query in the grid whose window leaves it
 *                                   on both sides of an axis is the reference's std::runtime_error
 *                                   "Window size for fine gradient is too large for SDF": the call then
 *                                   returns VGT_HIP_ERR_INVALID_ARGUMENT with that message.
 * The one operation whose order the reference takes from common_robotics_utilities
 * (TrilinearInterpolate) is evaluated along x, then y, then z, each as a*(1-t) + b*t in double
 * (csrc/cell_kernels.hip); results therefore agree with the reference to rounding (tests: 1e-5). */
int vgt_hip_sdf_estimate_distance(vgt_hip_ctx* ctx, const float* sdf_host, int64_t nx, int64_t ny, int64_t nz,
                                  double resolution, const double* grid_from_world, const double* query_xyz_host,
                                  int64_t num_queries, double* distance_host, uint8_t* has_value_host);
int vgt_hip_sdf_estimate_distance_dev(vgt_hip_ctx* ctx, const float* sdf_dev, int64_t nx, int64_t ny, int64_t nz,
                                      double resolution, const double* grid_from_world, const double* query_xyz_dev,
                                      int64_t num_queries, double* distance_dev, uint8_t* has_value_dev);
int vgt_hip_sdf_fine_gradient(vgt_hip_ctx* ctx, const float* sdf_host, int64_t nx, int64_t ny, int64_t nz,
                              double resolution, const double* grid_from_world, const double* query_xyz_host,
                              int64_t num_queries, double nominal_window_size, double* gradient_host,
                              uint8_t* has_value_host);

/* SignedDistanceField::ComputeLocalExtremaMap (I/signed_distance_field.hpp:1205-1231 over :385-541; consumed by
 * TaggedObjectOccupancyComponentMap::UpdateSpatialSegments, S/tagged_object_occupancy_component_map.cpp:775-868):
 * for every voxel the grid-frame location (3 doubles) of the cell its gradient chain ends at -- the chain follows
 * the coarse gradient with edge gradients (rotated by `rotation`, 9 doubles row-major, NULL = none; the
 * reference applies the origin transform's rotation) uphill outside obstacles and downhill inside, one of
 * the 26 neighbours at a time -- or +infinity x3 when the chain leaves the grid.  Where chains run into a
 * cycle the reference's answer depends on its X-major visiting order (the first walk that reaches the cycle fixes
 * the cell every later walk inherits); the device formulation reproduces it: bit-identical results
 * (csrc/cell_kernels.hip).  Grids below 2^31 cells. */
int vgt_hip_sdf_local_extrema_map(vgt_hip_ctx* ctx, const float* sdf_host, int64_t nx, int64_t ny, int64_t nz,
                                  double resolution, const double* rotation, double* extrema_host);
int vgt_hip_sdf_local_extrema_map_dev(vgt_hip_ctx* ctx, const float* sdf_dev, int64_t nx, int64_t ny, int64_t nz,
                                      double resolution, const double* rotation, double* extrema_dev);

/* ---- multi-GPU: the grid is cut into Z slabs, one device per slab (BASELINE.json config 5).
 * Lines along Y and X are local to a slab; only the first pass (nearest voxel of the other class
 * along Z) crosses slabs, and all it needs from the other slabs is, per (x, y) line, the nearest
 * filled / free voxel below and above.  So:
 *   1. vgt_hip_sdf_slab_begin_dev   local pass 1 (class records) + per-line summary of this slab: 4 bytes per line.  A slab's
 *                                    first voxel is filled or free, so the record holds, for the slab's first and
 *                                    for its last voxel, the class (bit 15: filled) and the global z of the first /
 *                                    last voxel of the OTHER class inside the slab (bits 0-14, 0x7fff when absent)
 *   2. the caller all-gathers the summaries (one RCCL all-gather; torch.distributed in
 *      voxelized_geometry_tools_amd/multi_gpu.py) into [world][lines] records and
 *      vgt_hip_sdf_slab_carries_dev reduces them to this slab's per-line carries
 *      (4 x int16: prev_filled, next_filled, prev_free, next_free as global z, -1 when absent)
 *   3. vgt_hip_sdf_slab_finish_dev  folds the carries in, then Y pass and X pass + finalize.
 * The slabs must be the ranges of vgt_hip_sdf_slab_range (equal shares of nz_global, earlier slabs take the
 * remainder): the carries are decoded with them.  As a guard against the likeliest mistake,
 * vgt_hip_sdf_slab_finish_dev rejects carries that vgt_hip_sdf_slab_carries_dev computed ON THE SAME CONTEXT for
 * another (z_offset, nz_local, nz_global) than the one it is given (VGT_HIP_ERR_INVALID_ARGUMENT; ABI version 2.
 * Version 1 exchanged 8-byte records with absolute positions and accepted any partition).  The guard is
 * best-effort: it goes by the address of the carries buffer, one finish consumes it, and carries that were written
 * by another context, a copy or a collective are not checked -- the partition rule above is the contract.
 * The workspace is the one of vgt_hip_sdf_dev for the slab's extents and must be the same buffer
 * in both calls.  kernel_ms (optional): begin -> [scan]; finish -> [fix-up, Y pass, X pass];
 * when given, the call blocks until the work has finished.
 * Limits: summaries and carries hold GLOBAL z in 15 / 16 bits, so the whole grid's Z extent (nz_global,
 * and z_offset + nz_local of every slab) must not exceed 16384 -- the per-axis limit of every SDF
 * entry point; larger values are rejected with VGT_HIP_ERR_INVALID_ARGUMENT. */
int vgt_hip_sdf_slab_range(int64_t nz_global, int32_t world, int32_t rank, int64_t* z_offset, int64_t* nz_local);
size_t vgt_hip_sdf_slab_summary_bytes(int64_t nx, int64_t ny);
int vgt_hip_sdf_slab_begin_dev(vgt_hip_ctx* ctx, const float* occupancy_dev, int64_t nx, int64_t ny,
                               int64_t nz_local, int64_t z_offset, int unknown_is_filled,
                               void* workspace_dev, size_t workspace_bytes, void* summary_dev,
                               float* kernel_ms);
size_t vgt_hip_sdf_slab_carries_bytes(int64_t nx, int64_t ny);
int vgt_hip_sdf_slab_carries_dev(vgt_hip_ctx* ctx, const void* gathered_summaries_dev, int32_t world,
                                 int32_t rank, int64_t nx, int64_t ny, int64_t nz_global, void* carries_dev);
int vgt_hip_sdf_slab_finish_dev(vgt_hip_ctx* ctx, int64_t nx, int64_t ny, int64_t nz_local,
                                int64_t z_offset, int64_t nz_global, double resolution,
                                int add_virtual_border, const void* carries_dev, float* sdf_dev,
                                void* workspace_dev, size_t workspace_bytes, float* minmax_dev,
                                float* kernel_ms);

/* ---- multi-GPU from ONE process, host buffers in and out: the large-grid branch of
 *      OccupancyMap::ExtractSignedDistanceFieldFloat (S/occupancy_map.cpp:256-260;
 *      I/occupancy_map.hpp:174-210).  The grid is cut into min(num_devices, nz) Z slabs, slab r runs
 *      on devices[r]: strided upload of occupancy[:, :, z0:z1], the slab pipeline above, ONE exchange
 *      of the per-line summaries (rccl ncclAllGather, one call per device in a group), download.
 *      The N uploads / pipelines / downloads run concurrently on per-device streams; the caller's
 *      arrays are page-locked for the duration of the call when possible.  A device may be listed
 *      more than once (several slabs on one GPU); rccl cannot form a communicator then, and the
 *      summaries are copied slab to slab instead.  Result and extrema are bit-identical to
 *      vgt_hip_sdf_from_occupancy_f32 on one device.  Blocking. */
int vgt_hipx_sdf_multi(const int* devices, int num_devices, const float* occupancy_host, int64_t nx,
                       int64_t ny, int64_t nz, double resolution, int unknown_is_filled,
                       int add_virtual_border, float* sdf_host, float* out_min, float* out_max);
/* vgt_hipx_sdf_multi keeps the per-slab contexts, streams and device buffers of the last (device list, grid shape)
 * it served for the next call with the same key (one extraction at a time per process); this frees them. */
void vgt_hipx_release(void);
/* Phases of the last vgt_hipx_sdf_multi call, milliseconds: [0] set-up (slab set on a miss + page-locking),
 * [1] slowest slab's upload, [2] its scan + exchange + passes, [3] its download (events on the slab streams; the
 * phases of different slabs overlap), [4] the whole call. */
int vgt_hipx_last_timing(float* ms5);


/* ---- ONE point cloud over several devices (the reference dispatches whole clouds,
 *      S/device_pointcloud_voxelization.cpp:147-149, so a single large cloud uses one device there).
 *      DeviceVoxelizationHelperInterface::RaycastPoints (I/device_voxelization_interface.hpp:151-158) with the
 *      points cut into 1 + num_helpers contiguous shares (vgt_hipx_point_share: equal shares, earlier shares take
 *      the remainder): share 0 is cast on `ctx`'s device straight into grid `grid_index`, share k on
 *      helper_devices[k-1] into a private tracking grid, and the private grids are then ADDED into grid
 *      `grid_index` -- rccl ncclReduce(sum, int32, root = ctx's device) when all the devices are distinct,
 *      copy + add otherwise (a device listed twice, or the caller's own: the one-GPU test of this path).
 *      Tracking counts are integers, so the result equals vgt_hip_raycast_points_f32 on the whole cloud bit for
 *      bit, whatever the split; counts already in the grid are kept.  Blocking; the sum is ordered on ctx's
 *      stream, so the filter that follows sees it.  Helper contexts and grids are kept for the next call with
 *      the same (device, helper list, cell count); vgt_hipx_release frees them.  num_helpers = 0 is the plain call. */
int vgt_hipx_point_share(int64_t num_points, int32_t shares, int32_t share, int64_t* first, int64_t* count);
int vgt_hipx_raycast_points_split(vgt_hip_ctx* ctx, vgt_hip_grids* grids, size_t grid_index,
                                  const int* helper_devices, int num_helpers, const float* points_xyz_host,
                                  int64_t num_points, float max_range, const float* grid_pointcloud_transform,
                                  float voxel_size, float inverse_voxel_size, float grid_x_size,
                                  float grid_y_size, float grid_z_size, int32_t num_x_voxels,
                                  int32_t num_y_voxels, int32_t num_z_voxels);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* VGT_HIP_H_ */
