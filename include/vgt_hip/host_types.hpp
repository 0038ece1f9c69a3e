// Stand-alone host-side value types for the HIP backend's C++ layer.
//
// Inside the reference tree these roles are played by Eigen::Isometry3d, OccupancyMap,
// SignedDistanceField<float>, PointCloudWrapper, ... (none of which can be compiled here:
// Eigen and common_robotics_utilities are not available).  The types below carry exactly what
// the hot path needs, with the reference's names for the operations the hot path uses, so that
// the C++ tests read like the reference's and the reference-side wiring in INTEGRATION.md is a
// field-by-field substitution.
#pragma once

#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>
#ifdef __linux__
#include <sys/mman.h>
#endif

namespace vgt_hip
{
// Rigid transform, 4x4 column-major doubles (the layout of Eigen::Isometry3d::data()).
struct Isometry3
{
  std::array<double, 16> m{{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}};

  static Isometry3 Identity() { return Isometry3(); }
  static Isometry3 Translation(double x, double y, double z)
  {
    Isometry3 t;
    t.m[12] = x;
    t.m[13] = y;
    t.m[14] = z;
    return t;
  }
  // Rotation from a unit quaternion (w, x, y, z) + translation.
  static Isometry3 FromQuaternion(double w, double x, double y, double z, double tx, double ty,
                                  double tz)
  {
    Isometry3 t;
    const double x2 = 2 * x, y2 = 2 * y, z2 = 2 * z;
    t.m[0] = 1 - (y2 * y + z2 * z);
    t.m[4] = y2 * x - z2 * w;
    t.m[8] = z2 * x + y2 * w;
    t.m[1] = y2 * x + z2 * w;
    t.m[5] = 1 - (x2 * x + z2 * z);
    t.m[9] = z2 * y - x2 * w;
    t.m[2] = z2 * x - y2 * w;
    t.m[6] = z2 * y + x2 * w;
    t.m[10] = 1 - (x2 * x + y2 * y);
    t.m[12] = tx;
    t.m[13] = ty;
    t.m[14] = tz;
    return t;
  }
  double operator()(int r, int c) const { return m[static_cast<size_t>(c * 4 + r)]; }
  double& operator()(int r, int c) { return m[static_cast<size_t>(c * 4 + r)]; }

  Isometry3 operator*(const Isometry3& o) const  // affine product
  {
    Isometry3 out;
    for (int r = 0; r < 3; r++)
    {
      for (int c = 0; c < 3; c++)
        out(r, c) = (*this)(r, 0) * o(0, c) + (*this)(r, 1) * o(1, c) + (*this)(r, 2) * o(2, c);
      out(r, 3) = (*this)(r, 0) * o(0, 3) + (*this)(r, 1) * o(1, 3) + (*this)(r, 2) * o(2, 3) +
                  (*this)(r, 3);
    }
    return out;
  }
  Isometry3 Inverse() const  // rigid: R^T, -R^T t
  {
    Isometry3 out;
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) out(r, c) = (*this)(c, r);
    for (int r = 0; r < 3; r++)
      out(r, 3) = -(out(r, 0) * (*this)(0, 3) + out(r, 1) * (*this)(1, 3) +
                    out(r, 2) * (*this)(2, 3));
    return out;
  }
  std::array<float, 16> CastFloat() const
  {
    std::array<float, 16> f{};
    for (size_t i = 0; i < 16; i++) f[i] = static_cast<float>(m[i]);
    return f;
  }
};

// Large blocks (a grid's cells) that a destroyed grid hands back and the next grid of the same size takes over, pages and
// all: a map of 256^3 cells is 64 MiB, which malloc maps afresh every time -- 16 384 page faults for the first thread that
// writes to it, more than a voxelization takes on the device.  A caller that voxelizes frame after frame (the by-value
// VoxelizePointClouds returns a new map per call) gets the block of the map it dropped a frame ago.  Bounded: at most
// Limit() bytes are kept, larger or surplus blocks go back to the system.  Process-wide, thread-safe.
class GridBlockCache
{
public:
  static constexpr size_t kMinBytes = size_t{1} << 20, kDefaultMaxBytes = size_t{512} << 20, kHugePage = size_t{2} << 20;
  // How much the cache may hold (default 512 MiB).  A process that extracts fields of several GiB in a loop can raise it
  // so that the block of the field it dropped is the block of the next one: no unmapping (4 GiB: 0.2 - 0.4 s on the
  // caller's thread) and no page faults.
  static void SetLimit(size_t bytes) noexcept
  {
    std::lock_guard<std::mutex> lock(Guard());
    Limit() = bytes;
  }
  static void* Take(size_t bytes)
  {
    if (bytes >= kMinBytes)
    {
      std::lock_guard<std::mutex> lock(Guard());
      auto& blocks = Blocks();
      for (size_t i = 0; i < blocks.size(); i++)
        if (blocks[i].second == bytes)
        {
          void* const p = blocks[i].first;
          Held() -= bytes;
          blocks.erase(blocks.begin() + static_cast<std::ptrdiff_t>(i));
          return p;
        }
    }
    // Fresh blocks of 2 MiB and more start on a huge-page boundary and ask for huge pages (a hint; where the kernel gives
    // them on request a 4 GiB grid is 2 048 page faults instead of a million, and page-locking it for a copy takes 8 ms
    // instead of 70 - 140).
    void* p = nullptr;
    const size_t alignment = bytes >= kHugePage ? kHugePage : 4096;
    if (posix_memalign(&p, alignment, bytes < 4096 ? 4096 : bytes) != 0) throw std::bad_alloc();
#ifdef __linux__
    if (bytes >= kHugePage) (void)madvise(p, bytes & ~(kHugePage - 1), MADV_HUGEPAGE);
#endif
    return p;
  }
  static void Give(void* p, size_t bytes) noexcept
  {
    if (bytes >= kMinBytes)
    {
      std::lock_guard<std::mutex> lock(Guard());
      if (Held() + bytes <= Limit())
      {
        Blocks().emplace_back(p, bytes);
        Held() += bytes;
        return;
      }
    }
    std::free(p);
  }
  // gives every cached block back to the system
  static void Release() noexcept
  {
    std::lock_guard<std::mutex> lock(Guard());
    for (const auto& b : Blocks()) std::free(b.first);
    Blocks().clear();
    Held() = 0;
  }

private:
  // (never destroyed: grids with static storage duration may give their blocks back after any static of this class)
  static std::mutex& Guard()
  {
    static std::mutex* const guard = new std::mutex();
    return *guard;
  }
  static std::vector<std::pair<void*, size_t>>& Blocks()
  {
    static std::vector<std::pair<void*, size_t>>* const blocks = new std::vector<std::pair<void*, size_t>>();
    return *blocks;
  }
  static size_t& Held()
  {
    static size_t held = 0;
    return held;
  }
  static size_t& Limit()
  {
    static size_t limit = kDefaultMaxBytes;
    return limit;
  }
};

// Allocator of a grid's cells: blocks come from / go back to GridBlockCache, and its value-less construct() leaves the
// memory as it is -- a vector sized through it has its elements' storage but has not touched a page of it
// (DenseGrid::UninitializedLike: a map whose every cell a download overwrites).
template <typename T>
struct DefaultInitAllocator
{
  using value_type = T;
  template <typename U>
  struct rebind
  {
    using other = DefaultInitAllocator<U>;
  };
  DefaultInitAllocator() = default;
  template <typename U>
  DefaultInitAllocator(const DefaultInitAllocator<U>&) noexcept {}
  T* allocate(size_t count) { return static_cast<T*>(GridBlockCache::Take(count * sizeof(T))); }
  void deallocate(T* p, size_t count) noexcept { GridBlockCache::Give(p, count * sizeof(T)); }
  template <typename U>
  bool operator==(const DefaultInitAllocator<U>&) const noexcept { return true; }
  template <typename U>
  bool operator!=(const DefaultInitAllocator<U>&) const noexcept { return false; }
  template <typename U>
  void construct(U* p) noexcept(std::is_nothrow_default_constructible<U>::value)
  {
    ::new (static_cast<void*>(p)) U;  // default-, not value-initialised: a no-op for float
  }
  template <typename U, typename... Args>
  void construct(U* p, Args&&... args)
  {
    ::new (static_cast<void*>(p)) U(std::forward<Args>(args)...);
  }
};

// Dense X-major / Z-fastest grid of floats with a uniform voxel size: the part of
// OccupancyMap (occupancy_map.hpp:65-217) and SignedDistanceField<float>
// (signed_distance_field.hpp:193-789) that the hot path touches.
class DenseGrid
{
public:
  DenseGrid() = default;
  DenseGrid(const Isometry3& origin_transform, const std::string& frame, double resolution,
            int64_t num_x, int64_t num_y, int64_t num_z, float default_value)
      : origin_(origin_transform), inverse_origin_(origin_transform.Inverse()), frame_(frame),
        resolution_(resolution), nx_(num_x), ny_(num_y), nz_(num_z)
  {
    if (!(resolution > 0.0) || num_x <= 0 || num_y <= 0 || num_z <= 0)
      throw std::invalid_argument("Grid must have positive resolution and voxel counts");
    data_.assign(static_cast<size_t>(num_x * num_y * num_z), default_value);
  }
  // A grid with the frame, transform and extents of `other` whose cells are NOT initialised -- not even touched: for a
  // caller that overwrites every cell (the by-value VoxelizePointClouds, whose result the download fills) and does not
  // want to pay for a copy and its page faults first.
  static DenseGrid UninitializedLike(const DenseGrid& other)
  {
    DenseGrid grid;
    grid.origin_ = other.origin_;
    grid.inverse_origin_ = other.inverse_origin_;
    grid.frame_ = other.frame_;
    grid.resolution_ = other.resolution_;
    grid.nx_ = other.nx_;
    grid.ny_ = other.ny_;
    grid.nz_ = other.nz_;
    grid.data_.resize(other.data_.size());
    return grid;
  }
  // A grid of the given frame and extents whose cells are allocated but NOT initialised (not even touched): for the
  // fields an extraction hands back, every cell of which the download writes.  A fill first would cost more than the
  // whole device path (1024^3: 4 GiB of stores and a million page faults before a 153 ms extraction).
  static DenseGrid Uninitialized(const Isometry3& origin_transform, const std::string& frame, double resolution,
                                 int64_t num_x, int64_t num_y, int64_t num_z)
  {
    DenseGrid grid = ShapeOnly(origin_transform, frame, resolution, num_x, num_y, num_z);
    grid.data_.resize(static_cast<size_t>(num_x * num_y * num_z));
    return grid;
  }
  // Frame, transform and extents without cells (IsInitialized() is false): the geometry a device-resident map keeps
  // for the fields it hands back.
  static DenseGrid ShapeOnly(const Isometry3& origin_transform, const std::string& frame, double resolution,
                             int64_t num_x, int64_t num_y, int64_t num_z)
  {
    if (!(resolution > 0.0) || num_x <= 0 || num_y <= 0 || num_z <= 0)
      throw std::invalid_argument("Grid must have positive resolution and voxel counts");
    DenseGrid grid;
    grid.origin_ = origin_transform;
    grid.inverse_origin_ = origin_transform.Inverse();
    grid.frame_ = frame;
    grid.resolution_ = resolution;
    grid.nx_ = num_x;
    grid.ny_ = num_y;
    grid.nz_ = num_z;
    return grid;
  }
  // VoxelGridSizes::FromGridSizes: counts = size / resolution
  static DenseGrid FromGridSizes(const Isometry3& origin_transform, const std::string& frame,
                                 double resolution, double x_size, double y_size, double z_size,
                                 float default_value)
  {
    return DenseGrid(origin_transform, frame, resolution,
                     static_cast<int64_t>(std::ceil(x_size / resolution)),
                     static_cast<int64_t>(std::ceil(y_size / resolution)),
                     static_cast<int64_t>(std::ceil(z_size / resolution)), default_value);
  }

  bool IsInitialized() const { return !data_.empty(); }
  int64_t NumXVoxels() const { return nx_; }
  int64_t NumYVoxels() const { return ny_; }
  int64_t NumZVoxels() const { return nz_; }
  int64_t NumTotalVoxels() const { return nx_ * ny_ * nz_; }
  double Resolution() const { return resolution_; }
  double VoxelXSize() const { return resolution_; }
  double GridXSize() const { return static_cast<double>(nx_) * resolution_; }
  double GridYSize() const { return static_cast<double>(ny_) * resolution_; }
  double GridZSize() const { return static_cast<double>(nz_) * resolution_; }
  const Isometry3& OriginTransform() const { return origin_; }
  const Isometry3& InverseOriginTransform() const { return inverse_origin_; }
  const std::string& Frame() const { return frame_; }
  bool SameSizes(const DenseGrid& o) const
  {
    return nx_ == o.nx_ && ny_ == o.ny_ && nz_ == o.nz_ && resolution_ == o.resolution_;
  }

  bool IndexInBounds(int64_t x, int64_t y, int64_t z) const
  {
    return x >= 0 && x < nx_ && y >= 0 && y < ny_ && z >= 0 && z < nz_;
  }
  float GetIndexImmutable(int64_t x, int64_t y, int64_t z) const
  {
    if (!IndexInBounds(x, y, z)) throw std::runtime_error("index out of grid bounds");
    return data_[static_cast<size_t>((x * ny_ + y) * nz_ + z)];
  }
  void SetIndex(int64_t x, int64_t y, int64_t z, float value)
  {
    if (!IndexInBounds(x, y, z)) throw std::runtime_error("index out of grid bounds");
    data_[static_cast<size_t>((x * ny_ + y) * nz_ + z)] = value;
  }
  using Storage = std::vector<float, DefaultInitAllocator<float>>;
  const Storage& GetImmutableRawData() const { return data_; }
  Storage& GetMutableRawData() { return data_; }

private:
  Isometry3 origin_, inverse_origin_;
  std::string frame_;
  double resolution_ = 0.0;
  int64_t nx_ = 0, ny_ = 0, nz_ = 0;
  Storage data_;
};

using OccupancyMap = DenseGrid;

// Cell types of the other three map types, laid out exactly as the reference's
// (occupancy_component_map.hpp:29-72, tagged_object_occupancy_map.hpp:29-69,
// tagged_object_occupancy_component_map.hpp:30-99): the float occupancy first, then uint32s.
struct OccupancyComponentCell
{
  float occupancy = 0.0f;
  uint32_t component = 0u;
};
struct TaggedObjectOccupancyCell
{
  float occupancy = 0.0f;
  uint32_t object_id = 0u;
};
struct TaggedObjectOccupancyComponentCell
{
  float occupancy = 0.0f;
  uint32_t object_id = 0u;
  uint32_t component = 0u;
  uint32_t spatial_segment = 0u;
};
static_assert(sizeof(OccupancyComponentCell) == 8 && sizeof(TaggedObjectOccupancyCell) == 8 &&
                  sizeof(TaggedObjectOccupancyComponentCell) == 16,
              "cell records must match the reference's raw store");

// The part of OccupancyComponentMap / TaggedObjectOccupancyMap /
// TaggedObjectOccupancyComponentMap that their SDF entry points touch: a dense X-major /
// Z-fastest grid of cells with a uniform voxel size.
template <typename Cell>
class CellGrid
{
public:
  CellGrid() = default;
  CellGrid(const Isometry3& origin_transform, const std::string& frame, double resolution,
           int64_t num_x, int64_t num_y, int64_t num_z, const Cell& default_value)
      : origin_(origin_transform), frame_(frame), resolution_(resolution), nx_(num_x), ny_(num_y),
        nz_(num_z)
  {
    if (!(resolution > 0.0) || num_x <= 0 || num_y <= 0 || num_z <= 0)
      throw std::invalid_argument("Grid must have positive resolution and voxel counts");
    data_.assign(static_cast<size_t>(num_x * num_y * num_z), default_value);
  }
  bool IsInitialized() const { return !data_.empty(); }
  int64_t NumXVoxels() const { return nx_; }
  int64_t NumYVoxels() const { return ny_; }
  int64_t NumZVoxels() const { return nz_; }
  double Resolution() const { return resolution_; }
  const Isometry3& OriginTransform() const { return origin_; }
  const std::string& Frame() const { return frame_; }
  const Cell& GetIndexImmutable(int64_t x, int64_t y, int64_t z) const
  {
    if (x < 0 || x >= nx_ || y < 0 || y >= ny_ || z < 0 || z >= nz_)
      throw std::runtime_error("index out of grid bounds");
    return data_[static_cast<size_t>((x * ny_ + y) * nz_ + z)];
  }
  void SetIndex(int64_t x, int64_t y, int64_t z, const Cell& value)
  {
    if (x < 0 || x >= nx_ || y < 0 || y >= ny_ || z < 0 || z >= nz_)
      throw std::runtime_error("index out of grid bounds");
    data_[static_cast<size_t>((x * ny_ + y) * nz_ + z)] = value;
  }
  const std::vector<Cell>& GetImmutableRawData() const { return data_; }

private:
  Isometry3 origin_;
  std::string frame_;
  double resolution_ = 0.0;
  int64_t nx_ = 0, ny_ = 0, nz_ = 0;
  std::vector<Cell> data_;
};

using OccupancyComponentMap = CellGrid<OccupancyComponentCell>;
using TaggedObjectOccupancyMap = CellGrid<TaggedObjectOccupancyCell>;
using TaggedObjectOccupancyComponentMap = CellGrid<TaggedObjectOccupancyComponentCell>;

// SignedDistanceFieldGenerationParameters<float> (signed_distance_field.hpp:1234-1264), minus
// the CPU parallelism knob, plus the device to run on.
struct SignedDistanceFieldGenerationParameters
{
  float oob_value = std::numeric_limits<float>::infinity();
  bool unknown_is_filled = true;
  bool add_virtual_border = false;
  int hip_device = 0;
  // Large grids: when not empty, the field is extracted on these devices, one Z slab per entry
  // (vgt_hipx_sdf_multi); a device may be listed more than once to cut the grid into more slabs
  // than there are GPUs.
  std::vector<int> hip_devices;
};

// SignedDistanceField<float>: values + the minimum / maximum cached by Lock()
// (signed_distance_field.hpp:765-789).
struct SignedDistanceField
{
  DenseGrid grid;
  float oob_value = std::numeric_limits<float>::infinity();
  float minimum = 0.0f, maximum = 0.0f;
  bool locked = false;
  bool IsLocked() const { return locked; }
  float GetIndexImmutable(int64_t x, int64_t y, int64_t z) const
  {
    return grid.GetIndexImmutable(x, y, z);
  }
};

// PointCloudVoxelizationFilterOptions (pointcloud_voxelization_interface.hpp:20-92).
class PointCloudVoxelizationFilterOptions
{
public:
  PointCloudVoxelizationFilterOptions() = default;
  PointCloudVoxelizationFilterOptions(double percent_seen_free, int32_t outlier_points_threshold,
                                      int32_t num_cameras_seen_free)
      : percent_seen_free_(percent_seen_free), outlier_points_threshold_(outlier_points_threshold),
        num_cameras_seen_free_(num_cameras_seen_free)
  {
    if (percent_seen_free_ <= 0.0 || percent_seen_free_ > 1.0)
      throw std::invalid_argument("0 < percent_seen_free_ <= 1 must be true");
    if (outlier_points_threshold_ <= 0) throw std::invalid_argument("outlier_points_threshold_ <= 0");
    if (num_cameras_seen_free_ <= 0) throw std::invalid_argument("num_cameras_seen_free_ <= 0");
  }
  double PercentSeenFree() const { return percent_seen_free_; }
  int32_t OutlierPointsThreshold() const { return outlier_points_threshold_; }
  int32_t NumCamerasSeenFree() const { return num_cameras_seen_free_; }

private:
  double percent_seen_free_ = 1.0;
  int32_t outlier_points_threshold_ = 1;
  int32_t num_cameras_seen_free_ = 1;
};

// PointCloudWrapper (pointcloud_voxelization_interface.hpp:94-202): abstract point source.
class PointCloudWrapper
{
public:
  virtual ~PointCloudWrapper() {}
  virtual double MaxRange() const = 0;
  virtual int64_t Size() const = 0;
  virtual const Isometry3& PointCloudOriginTransform() const = 0;
  void CopyPointLocationIntoFloatPtr(int64_t point_index, float* destination) const
  {
    if (point_index < 0 || point_index >= Size()) throw std::out_of_range("point_index out of range");
    CopyPointLocationIntoFloatPtrImpl(point_index, destination);
  }
  // (pointcloud_voxelization_interface.hpp:117-133: what the reference's CPU voxelizer reads its points through)
  void CopyPointLocationIntoDoublePtr(int64_t point_index, double* destination) const
  {
    if (point_index < 0 || point_index >= Size()) throw std::out_of_range("point_index out of range");
    CopyPointLocationIntoDoublePtrImpl(point_index, destination);
  }

  // Extension: a wrapper whose points sit in one buffer of fixed-size records
  // with x, y, z as consecutive FLOAT32 reports the layout, and the HIP voxelizer raycasts the
  // buffer in place instead of gathering point by point.  Default: not available.
  virtual bool StridedFloat32Layout(const uint8_t** data, int64_t* point_step, int64_t* xyz_offset) const
  {
    (void)data;
    (void)point_step;
    (void)xyz_offset;
    return false;
  }

protected:
  virtual void CopyPointLocationIntoFloatPtrImpl(int64_t point_index, float* destination) const = 0;
  // Default: the float location widened -- exact for clouds whose storage is FLOAT32 (PointCloud2); a wrapper that
  // stores doubles overrides it (the reference declares it pure, pointcloud_voxelization_interface.hpp:193-199).
  virtual void CopyPointLocationIntoDoublePtrImpl(int64_t point_index, double* destination) const
  {
    float xyz[3];
    CopyPointLocationIntoFloatPtrImpl(point_index, xyz);
    for (int a = 0; a < 3; a++) destination[a] = static_cast<double>(xyz[a]);
  }
};
using PointCloudWrapperSharedPtr = std::shared_ptr<PointCloudWrapper>;

// sensor_msgs/PointCloud2, the members the voxelizer reads (ROS is not a dependency here).
struct PointField
{
  enum : uint8_t { INT8 = 1, UINT8 = 2, INT16 = 3, UINT16 = 4, INT32 = 5, UINT32 = 6, FLOAT32 = 7, FLOAT64 = 8 };
  std::string name;
  uint32_t offset = 0;
  uint8_t datatype = 0;
  uint32_t count = 1;
};
struct PointCloud2
{
  uint32_t height = 0, width = 0;
  std::vector<PointField> fields;
  uint32_t point_step = 0;
  std::vector<uint8_t> data;
};

// NonOwningPointCloud2Wrapper (pointcloud_voxelization_ros_interface.hpp:36-119; constructor
// checks: pointcloud_voxelization_ros_interface.cpp:28-78).
class NonOwningPointCloud2Wrapper : public PointCloudWrapper
{
public:
  NonOwningPointCloud2Wrapper(const PointCloud2* cloud_ptr, const Isometry3& origin_transform,
                              double max_range = std::numeric_limits<double>::infinity())
      : cloud_ptr_(cloud_ptr), origin_transform_(origin_transform), max_range_(max_range)
  {
    if (cloud_ptr_ == nullptr) throw std::invalid_argument("cloud_ptr_ == nullptr");
    if (max_range_ <= 0.0) throw std::runtime_error("max_range_ <= 0.0");
    const PointField* xyz[3] = {nullptr, nullptr, nullptr};
    for (const PointField& field : cloud_ptr_->fields)
    {
      if (field.name == "x") xyz[0] = &field;
      if (field.name == "y") xyz[1] = &field;
      if (field.name == "z") xyz[2] = &field;
    }
    const char* names[3] = {"x", "y", "z"};
    for (int a = 0; a < 3; a++)
    {
      if (xyz[a] == nullptr) throw std::out_of_range("map::at");  // field_type_map.at(...) of the reference
      if (xyz[a]->datatype != PointField::FLOAT32)
        throw std::invalid_argument(std::string("PointCloud ") + names[a] + " field is not FLOAT32");
    }
    if ((xyz[2]->offset - xyz[1]->offset) == sizeof(float) && (xyz[1]->offset - xyz[0]->offset) == sizeof(float))
      xyz_offset_from_point_start_ = xyz[0]->offset;
    else
      throw std::invalid_argument("PointCloud does not have sequential xyz fields");
  }
  double MaxRange() const override { return max_range_; }
  int64_t Size() const override { return static_cast<int64_t>(cloud_ptr_->width) * cloud_ptr_->height; }
  const Isometry3& PointCloudOriginTransform() const override { return origin_transform_; }
  bool StridedFloat32Layout(const uint8_t** data, int64_t* point_step, int64_t* xyz_offset) const override
  {
    *data = cloud_ptr_->data.data();
    *point_step = static_cast<int64_t>(cloud_ptr_->point_step);
    *xyz_offset = static_cast<int64_t>(xyz_offset_from_point_start_);
    return true;
  }

private:
  void CopyPointLocationIntoFloatPtrImpl(int64_t point_index, float* destination) const override
  {
    const size_t starting_offset =
        static_cast<size_t>(point_index) * static_cast<size_t>(cloud_ptr_->point_step) + xyz_offset_from_point_start_;
    std::memcpy(destination, &(cloud_ptr_->data.at(starting_offset)), sizeof(float) * 3);
  }
  const PointCloud2* const cloud_ptr_;
  size_t xyz_offset_from_point_start_ = 0;
  Isometry3 origin_transform_;
  double max_range_;
};

// VoxelizerRuntime (pointcloud_voxelization_interface.hpp:206-229).
class VoxelizerRuntime
{
public:
  VoxelizerRuntime(double raycasting_time, double filtering_time)
      : raycasting_time_(raycasting_time), filtering_time_(filtering_time)
  {
    if (raycasting_time_ < 0.0) throw std::invalid_argument("raycasting_time < 0.0");
    if (filtering_time_ < 0.0) throw std::invalid_argument("filtering_time < 0.0");
  }
  double RaycastingTime() const { return raycasting_time_; }
  double FilteringTime() const { return filtering_time_; }

private:
  double raycasting_time_ = 0.0, filtering_time_ = 0.0;
};
}  // namespace vgt_hip
