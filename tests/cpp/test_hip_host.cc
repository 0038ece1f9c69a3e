// C++ parity tests through the host layer (include/vgt_hip/), written after the reference's
// own gtest suites: test/sdf_generation_test.cpp and test/pointcloud_voxelization_test.cpp.
// No gtest in this image: a tiny EXPECT macro set; exit code = number of failures.
//   test_hip_host            all tests (needs a HIP device)
//   test_hip_host --no-device  only the "backend unavailable" behaviour (CPU box)
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <limits>
#include <thread>
#include <vector>

#include "../../include/vgt_hip/hip_pointcloud_voxelizer.hpp"
#include "../../oracle/vgt_oracle.h"  // the checker of the CPU-exact mode (test infrastructure: oracle/ never ships)

using namespace vgt_hip;
namespace hip_helpers = voxelized_geometry_tools::pointcloud_voxelization::hip_helpers;

static int g_failures = 0;
#define EXPECT_TRUE(cond)                                                              \
  do {                                                                                 \
    if (!(cond)) { g_failures++; std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); } \
  } while (0)
#define EXPECT_EQ(a, b) EXPECT_TRUE((a) == (b))
#define EXPECT_FLOAT_EQ(a, b) EXPECT_TRUE(AlmostEqualUlps((a), (b)))

static bool AlmostEqualUlps(float a, float b)  // gtest's EXPECT_FLOAT_EQ: within 4 ULPs
{
  if (a == b) return true;
  int32_t ia, ib;
  std::memcpy(&ia, &a, 4);
  std::memcpy(&ib, &b, 4);
  if ((ia < 0) != (ib < 0)) return false;
  return std::abs(ia - ib) <= 4;
}

static bool CloseEnough(float a, float b) { return a == b || std::abs(a - b) <= 0.0001f; }

static void FillBox(OccupancyMap& m, int x0, int x1, int y0, int y1, int z0, int z1)
{
  for (int x = x0; x < x1; x++)
    for (int y = y0; y < y1; y++)
      for (int z = z0; z < z1; z++) m.SetIndex(x, y, z, 1.0f);
}

// The four map types of one test scene, built cell by cell as test/sdf_generation_test.cpp does (:279-294, :387-418):
// filled cells are OccupancyCell(1), OccupancyComponentCell(1), TaggedObjectOccupancyCell(1, 1u) and
// TaggedObjectOccupancyComponentCell(1, 1u), the background cells (v) / (v) / (v, v != 0 ? 1u : 0u).
struct FourMaps
{
  OccupancyMap occupancy;
  OccupancyComponentMap component;
  TaggedObjectOccupancyMap tagged;
  TaggedObjectOccupancyComponentMap tagged_component;
  FourMaps(const Isometry3& origin, double res, double xs, double ys, double zs, float fill)
      : occupancy(OccupancyMap::FromGridSizes(origin, "test_frame", res, xs, ys, zs, fill)),
        component(origin, "test_frame", res, occupancy.NumXVoxels(), occupancy.NumYVoxels(), occupancy.NumZVoxels(),
                  OccupancyComponentCell{fill, 0u}),
        tagged(origin, "test_frame", res, occupancy.NumXVoxels(), occupancy.NumYVoxels(), occupancy.NumZVoxels(),
               TaggedObjectOccupancyCell{fill, fill != 0.0f ? 1u : 0u}),
        tagged_component(origin, "test_frame", res, occupancy.NumXVoxels(), occupancy.NumYVoxels(),
                         occupancy.NumZVoxels(), TaggedObjectOccupancyComponentCell{fill, fill != 0.0f ? 1u : 0u, 0u, 0u})
  {
  }
  void FillBox(int x0, int x1, int y0, int y1, int z0, int z1)
  {
    for (int x = x0; x < x1; x++)
      for (int y = y0; y < y1; y++)
        for (int z = z0; z < z1; z++)
        {
          occupancy.SetIndex(x, y, z, 1.0f);
          component.SetIndex(x, y, z, OccupancyComponentCell{1.0f, 0u});
          tagged.SetIndex(x, y, z, TaggedObjectOccupancyCell{1.0f, 1u});
          tagged_component.SetIndex(x, y, z, TaggedObjectOccupancyComponentCell{1.0f, 1u, 0u, 0u});
        }
  }
  // GenerateSignedDistanceFields (test/sdf_generation_test.cpp:42-110): each type through its own entry point, the
  // tagged ones with an empty object list
  std::vector<SignedDistanceField> Fields() const
  {
    const SignedDistanceFieldGenerationParameters params;
    std::vector<SignedDistanceField> fields;
    fields.push_back(ExtractSignedDistanceField(occupancy, params));
    fields.push_back(ExtractSignedDistanceField(component, params));
    fields.push_back(DeviceTaggedObjectMap(tagged).ExtractSignedDistanceField({}, params));
    fields.push_back(DeviceTaggedObjectMap(tagged_component).ExtractSignedDistanceField({}, params));
    return fields;
  }
};

// TestSDFGeneration (test/sdf_generation_test.cpp:140-260): extrema + sign of every voxel, on all four map types.
static void CheckSdf(const FourMaps& maps, float expected_min, float expected_max)
{
  const OccupancyMap& map = maps.occupancy;
  for (const SignedDistanceField& sdf : maps.Fields())
  {
    EXPECT_TRUE(sdf.IsLocked());
    EXPECT_TRUE(CloseEnough(sdf.minimum, expected_min));
    EXPECT_TRUE(CloseEnough(sdf.maximum, expected_max));
    EXPECT_EQ(sdf.grid.NumXVoxels(), map.NumXVoxels());
    EXPECT_EQ(sdf.grid.NumYVoxels(), map.NumYVoxels());
    EXPECT_EQ(sdf.grid.NumZVoxels(), map.NumZVoxels());
    for (int64_t x = 0; x < map.NumXVoxels(); x++)
      for (int64_t y = 0; y < map.NumYVoxels(); y++)
        for (int64_t z = 0; z < map.NumZVoxels(); z++)
        {
          if (map.GetIndexImmutable(x, y, z) >= 0.5f)
            EXPECT_TRUE(sdf.GetIndexImmutable(x, y, z) < 0.0f);
          else
            EXPECT_TRUE(sdf.GetIndexImmutable(x, y, z) > 0.0f);
        }
  }
}

static void SdfGenerationTests()
{
  const float inf = std::numeric_limits<float>::infinity();
  const Isometry3 origin = Isometry3::Translation(-5.0, -5.0, -5.0);
  CheckSdf(FourMaps(origin, 0.25, 1, 2, 3, 1.0f), -inf, -inf);  // FullyFilledTest
  CheckSdf(FourMaps(origin, 0.25, 1, 2, 3, 0.0f), inf, inf);    // FullyEmptyTest
  {                                                              // CenterObstacleTest
    FourMaps m(origin, 0.25, 1, 2, 3, 0.0f);
    EXPECT_EQ(m.occupancy.NumXVoxels(), 4);
    EXPECT_EQ(m.occupancy.NumYVoxels(), 8);
    EXPECT_EQ(m.occupancy.NumZVoxels(), 12);
    m.FillBox(1, 3, 2, 6, 3, 9);
    CheckSdf(m, -0.25f, static_cast<float>(std::sqrt(0.25 * 0.25 + 0.5 * 0.5 + 0.75 * 0.75)));
  }
  {  // CornerObstacleTest
    FourMaps m(origin, 0.25, 1, 2, 3, 0.0f);
    m.FillBox(0, 2, 0, 4, 0, 6);
    CheckSdf(m, -0.5f, 1.8708f);
  }
  {  // FaceObstacleTest
    FourMaps m(origin, 0.25, 1, 2, 3, 0.0f);
    m.FillBox(0, 4, 0, 8, 0, 1);
    CheckSdf(m, -0.25f, 2.75f);
  }
  {  // LinearExactTest (:586-701)
    FourMaps m(Isometry3::Identity(), 1.0, 1, 1, 4, 0.0f);
    m.FillBox(0, 1, 0, 1, 0, 2);
    const float want[4] = {-2.0f, -1.0f, 1.0f, 2.0f};
    for (const SignedDistanceField& sdf : m.Fields())
      for (int z = 0; z < 4; z++) EXPECT_FLOAT_EQ(sdf.GetIndexImmutable(0, 0, z), want[z]);
  }
  {  // PlanarExactTest (:703-902)
    FourMaps m(Isometry3::Identity(), 1.0, 1, 4, 4, 0.0f);
    m.FillBox(0, 1, 0, 2, 0, 2);
    const float s2 = std::sqrt(2.0f), s5 = std::sqrt(5.0f), s8 = std::sqrt(8.0f);
    const float want[4][4] = {{-2, -1, 1, 2}, {-1, -1, 1, 2}, {1, 1, s2, s5}, {2, 2, s5, s8}};
    for (const SignedDistanceField& sdf : m.Fields())
      for (int y = 0; y < 4; y++)
        for (int z = 0; z < 4; z++) EXPECT_FLOAT_EQ(sdf.GetIndexImmutable(0, y, z), want[y][z]);
  }
  {  // CubeExactTest (:904-1055)
    FourMaps m(Isometry3::Identity(), 1.0, 2, 2, 2, 0.0f);
    m.FillBox(0, 1, 0, 1, 0, 1);
    const float s2 = std::sqrt(2.0f), s3 = std::sqrt(3.0f);
    const float want[8] = {-1, 1, 1, s2, 1, s2, s2, s3};
    for (const SignedDistanceField& sdf : m.Fields())
      for (int i = 0; i < 8; i++) EXPECT_FLOAT_EQ(sdf.GetIndexImmutable(i >> 2, (i >> 1) & 1, i & 1), want[i]);
  }
}

// Batches: many maps of one size in one extraction, and MakeAllObjectSDFs as one batch -- each field bit-equal to the
// single call's.
static void BatchedSdfTests()
{
  std::vector<OccupancyMap> maps;
  for (int i = 0; i < 9; i++)
  {
    OccupancyMap m = OccupancyMap::FromGridSizes(Isometry3::Translation(0.1 * i, 0, 0), "test_frame", 0.05, 1.0, 1.2, 0.9, 0.0f);
    FillBox(m, i, i + 4, 2 + i, 9 + i, 1, 3 + i);
    if (i == 4) FillBox(m, 0, 20, 0, 24, 0, 18);  // a full grid inside the batch
    if (i == 7) m.SetIndex(3, 3, 3, 0.5f);
    maps.push_back(m);
  }
  maps.push_back(OccupancyMap::FromGridSizes(Isometry3::Identity(), "test_frame", 0.05, 1.0, 1.2, 0.9, 0.0f));  // empty
  std::vector<const OccupancyMap*> pointers;
  for (const OccupancyMap& m : maps) pointers.push_back(&m);
  SignedDistanceFieldGenerationParameters params;
  for (int vb = 0; vb < 2; vb++)
  {
    params.add_virtual_border = vb != 0;
    const std::vector<SignedDistanceField> fields = ExtractSignedDistanceFields(pointers, params);
    EXPECT_EQ(fields.size(), maps.size());
    for (size_t i = 0; i < maps.size(); i++)
    {
      const SignedDistanceField single = ExtractSignedDistanceField(maps[i], params);
      EXPECT_TRUE(fields[i].IsLocked());
      EXPECT_TRUE(std::memcmp(fields[i].grid.GetImmutableRawData().data(), single.grid.GetImmutableRawData().data(),
                              single.grid.GetImmutableRawData().size() * sizeof(float)) == 0);
      EXPECT_TRUE(fields[i].minimum == single.minimum && fields[i].maximum == single.maximum);
    }
  }
  bool threw = false;
  try
  {
    const OccupancyMap other = OccupancyMap::FromGridSizes(Isometry3::Identity(), "test_frame", 0.05, 1.0, 1.0, 0.9, 0.0f);
    ExtractSignedDistanceFields({&maps[0], &other}, params);
  }
  catch (const std::invalid_argument&)
  {
    threw = true;
  }
  EXPECT_TRUE(threw);
  // tagged map: 12 box-shaped objects; the batched MakeAllObjectSDFs against one ExtractSignedDistanceField({id}) each
  TaggedObjectOccupancyMap tagged(Isometry3::Identity(), "test_frame", 0.1, 30, 26, 22, TaggedObjectOccupancyCell());
  for (uint32_t id = 1; id <= 12; id++)
    for (int x = 0; x < 4; x++)
      for (int y = 0; y < 3; y++)
        for (int z = 0; z < 5; z++)
          tagged.SetIndex((id * 7) % 26 + x, (id * 5) % 23 + y, (id * 3) % 17 + z,
                          TaggedObjectOccupancyCell{(id == 5 && z == 0) ? 0.5f : 1.0f, id});
  const DeviceTaggedObjectMap device_map(tagged);
  SignedDistanceFieldGenerationParameters tagged_params;
  const std::map<uint32_t, SignedDistanceField> all = device_map.MakeAllObjectSDFs(tagged_params);
  const std::vector<uint32_t> ids = device_map.ObjectIds();
  EXPECT_EQ(all.size(), ids.size());
  for (const uint32_t id : ids)
  {
    const SignedDistanceField single = device_map.ExtractSignedDistanceField({id}, tagged_params);
    const SignedDistanceField& batched = all.at(id);
    EXPECT_TRUE(batched.IsLocked());
    EXPECT_TRUE(std::memcmp(batched.grid.GetImmutableRawData().data(), single.grid.GetImmutableRawData().data(),
                            single.grid.GetImmutableRawData().size() * sizeof(float)) == 0);
    EXPECT_TRUE(batched.minimum == single.minimum && batched.maximum == single.maximum);
  }
  // duplicated ids in the list: one entry per distinct id, as a std::map gives
  EXPECT_EQ(device_map.MakeSeparateObjectSDFs({3u, 3u, 9u}, tagged_params).size(), 2u);
}

// The large-grid branch (vgt_hipx_sdf_multi: one process, one Z slab per listed device).  With one GPU the
// slabs share device 0 ({0, 0}, {0, 0, 0}: summaries copied slab to slab) or there is one slab ({0}: the RCCL
// communicator path with one rank); every variant must reproduce the single-device field bit for bit.
static void MultiDeviceSdfTests()
{
  OccupancyMap m = OccupancyMap::FromGridSizes(Isometry3::Identity(), "test_frame", 0.05, 2.0, 1.5, 3.0, 0.0f);
  uint64_t state = 12345;
  auto next = [&state]() {
    state = state * 6364136223846793005ull + 1442695040888963407ull;
    return static_cast<uint32_t>(state >> 33);
  };
  for (int64_t x = 0; x < m.NumXVoxels(); x++)
    for (int64_t y = 0; y < m.NumYVoxels(); y++)
      for (int64_t z = 0; z < m.NumZVoxels(); z++)
      {
        const uint32_t r = next() % 1000;
        if (r < 15) m.SetIndex(x, y, z, 1.0f);
        else if (r < 20) m.SetIndex(x, y, z, 0.5f);
      }
  FillBox(m, 5, 20, 3, 11, 20, 47);  // a solid that spans several slabs
  for (const bool border : {false, true})
  {
    SignedDistanceFieldGenerationParameters single;
    single.add_virtual_border = border;
    const SignedDistanceField want = ExtractSignedDistanceField(m, single);
    for (const std::vector<int>& devices : {std::vector<int>{0}, std::vector<int>{0, 0}, std::vector<int>{0, 0, 0}})
    {
      SignedDistanceFieldGenerationParameters multi = single;
      multi.hip_devices = devices;
      const SignedDistanceField got = ExtractSignedDistanceField(m, multi);
      EXPECT_TRUE(got.locked);
      EXPECT_TRUE(std::memcmp(got.grid.GetImmutableRawData().data(), want.grid.GetImmutableRawData().data(),
                              want.grid.GetImmutableRawData().size() * sizeof(float)) == 0);
      EXPECT_EQ(got.minimum, want.minimum);
      EXPECT_EQ(got.maximum, want.maximum);
    }
  }
  {
    SignedDistanceFieldGenerationParameters bad;
    bad.hip_devices = {9999};
    bool threw = false;
    try { (void)ExtractSignedDistanceField(m, bad); } catch (const std::runtime_error&) { threw = true; }
    EXPECT_TRUE(threw);
  }
}

// SDF consumers through the C++ layer: on the field of one filled voxel the estimate at a cell centre is the stored
// distance moved half a cell towards the surface, the fine and coarse gradients point away from the voxel, every
// gradient chain outside ends off the grid or at a flat cell, and an oversized window throws like the reference.
static void SdfConsumerTests()
{
  OccupancyMap m = OccupancyMap::FromGridSizes(Isometry3::Translation(1.0, -2.0, 0.5), "test_frame", 0.5, 4.0, 4.0, 4.0, 0.0f);
  m.SetIndex(3, 4, 2, 1.0f);
  const SignedDistanceField sdf = ExtractSignedDistanceField(m, {});
  // world location of the centre of cell (6, 4, 2): origin translation + (idx + 0.5) * res
  const std::vector<double> q = {1.0 + 6.5 * 0.5, -2.0 + 4.5 * 0.5, 0.5 + 2.5 * 0.5, 100.0, 0.0, 0.0};
  const DistanceEstimates est = EstimateLocationDistances(sdf, q);
  EXPECT_EQ(est.has_value[0], 1);
  EXPECT_EQ(est.has_value[1], 0);
  EXPECT_TRUE(std::abs(est.distance[0] - (3.0 * 0.5 - 0.25)) < 1e-6);
  const Gradients fine = GetLocationFineGradients(sdf, {q[0], q[1], q[2]}, 0.1);
  EXPECT_EQ(fine.has_value[0], 1);
  EXPECT_TRUE(fine.gradient[0] > 0.9 && std::abs(fine.gradient[1]) < 1e-6 && std::abs(fine.gradient[2]) < 1e-6);
  const Gradients coarse = GetIndexCoarseGradients(sdf, true);
  const size_t cell = static_cast<size_t>((6 * 8 + 4) * 8 + 2);
  EXPECT_EQ(coarse.has_value[cell], 1);
  EXPECT_TRUE(std::abs(coarse.gradient[3 * cell] - 1.0) < 1e-6);
  bool threw = false;
  try { (void)GetLocationFineGradients(sdf, {q[0], q[1], q[2]}, 50.0); } catch (const std::runtime_error&) { threw = true; }
  EXPECT_TRUE(threw);
  const std::vector<double> extrema = ComputeLocalExtremaMap(sdf);
  EXPECT_EQ(extrema.size(), static_cast<size_t>(3 * 8 * 8 * 8));
  bool all_set = true;
  for (const double v : extrema) all_set = all_set && !(v == -std::numeric_limits<double>::infinity()) && !std::isnan(v);
  EXPECT_TRUE(all_set);
}

// ---- test/pointcloud_voxelization_test.cpp ----
class VectorPointCloudWrapper : public PointCloudWrapper
{
public:
  void PushBack(double x, double y, double z) { points_.push_back({x, y, z}); }
  double MaxRange() const override { return std::numeric_limits<double>::infinity(); }
  int64_t Size() const override { return static_cast<int64_t>(points_.size()); }
  const Isometry3& PointCloudOriginTransform() const override { return origin_; }
  void SetPointCloudOriginTransform(const Isometry3& t) { origin_ = t; }

private:
  void CopyPointLocationIntoFloatPtrImpl(int64_t i, float* dst) const override
  {
    for (int a = 0; a < 3; a++) dst[a] = static_cast<float>(points_[static_cast<size_t>(i)][a]);
  }
  void CopyPointLocationIntoDoublePtrImpl(int64_t i, double* dst) const override
  {
    for (int a = 0; a < 3; a++) dst[a] = points_[static_cast<size_t>(i)][a];
  }
  std::vector<std::array<double, 3>> points_;
  Isometry3 origin_;
};

// The reference's CPU voxelizer, as the oracle restates it (oracle/vgt_oracle.c: the float64 walk of
// cpu_pointcloud_voxelization.cpp:167-206,208-436 and the double-ratio filter of :438-497), on the clouds of a scene.
static OccupancyMap OracleCpuVoxelization(const OccupancyMap& static_environment,
                                          const PointCloudVoxelizationFilterOptions& filter_options,
                                          const std::vector<PointCloudWrapperSharedPtr>& clouds)
{
  const int64_t cells = static_environment.NumTotalVoxels();
  const size_t num_grids = std::max<size_t>(clouds.size(), 1);
  std::vector<int32_t> tracking(static_cast<size_t>(2 * cells) * num_grids, 0);
  for (size_t c = 0; c < clouds.size(); c++)
  {
    const PointCloudWrapper& cloud = *clouds[c];
    if (cloud.Size() <= 0) continue;
    const Isometry3 X_GC = static_environment.InverseOriginTransform() * cloud.PointCloudOriginTransform();
    std::vector<double> points(static_cast<size_t>(cloud.Size()) * 3);
    for (int64_t i = 0; i < cloud.Size(); i++) cloud.CopyPointLocationIntoDoublePtr(i, points.data() + 3 * i);
    vgt_oracle_raycast_f64(points.data(), cloud.Size(), cloud.MaxRange(), X_GC.m.data(), static_environment.VoxelXSize(),
                           1.0 / static_environment.VoxelXSize(), static_environment.GridXSize(),
                           static_environment.GridYSize(), static_environment.GridZSize(), static_environment.NumXVoxels(),
                           static_environment.NumYVoxels(), static_environment.NumZVoxels(),
                           tracking.data() + c * static_cast<size_t>(2 * cells), 1);
  }
  OccupancyMap filtered = static_environment;
  vgt_oracle_filter(tracking.data(), cells, static_cast<int32_t>(num_grids), filter_options.PercentSeenFree(),
                    filter_options.OutlierPointsThreshold(), filter_options.NumCamerasSeenFree(), 1,
                    filtered.GetMutableRawData().data(), 1);
  return filtered;
}

static void check_empty_voxelization(const OccupancyMap& occupancy)
{
  for (int64_t x = 0; x < occupancy.NumXVoxels(); x++)
    for (int64_t y = 0; y < occupancy.NumYVoxels(); y++)
      for (int64_t z = 0; z < occupancy.NumZVoxels(); z++)
        EXPECT_EQ(occupancy.GetIndexImmutable(x, y, z), (z == 0) ? 1.0f : 0.5f);
}

static void check_voxelization(const OccupancyMap& occupancy)
{
  for (int64_t x = 0; x < occupancy.NumXVoxels(); x++)
    for (int64_t y = 0; y < occupancy.NumYVoxels(); y++)
      for (int64_t z = 0; z < occupancy.NumZVoxels(); z++)
      {
        const float occ = occupancy.GetIndexImmutable(x, y, z);
        if (z == 0) EXPECT_EQ(occ, 1.0f);
        if ((x == 3) && (y >= 3) && (z >= 1)) EXPECT_EQ(occ, 0.0f);
        if ((x >= 3) && (y == 3) && (z >= 1)) EXPECT_EQ(occ, 0.0f);
        if ((x == 4) && (y >= 4) && (z >= 1)) EXPECT_EQ(occ, 1.0f);
        if ((x >= 4) && (y == 4) && (z >= 1)) EXPECT_EQ(occ, 1.0f);
        if ((x > 4) && (y > 4) && (z >= 1)) EXPECT_EQ(occ, 0.5f);
      }
}

static Isometry3 QuatZ_then_X(double az, double ax, double tx, double ty, double tz)
{
  // Quaternion(AngleAxis(az, Z) * AngleAxis(ax, X))
  const double wz = std::cos(az / 2), sz = std::sin(az / 2);
  const double wx = std::cos(ax / 2), sx = std::sin(ax / 2);
  const double w = wz * wx, x = wz * sx, y = sz * sx, z = sz * wx;
  return Isometry3::FromQuaternion(w, x, y, z, tx, ty, tz);
}

static void PointCloudVoxelizationTests(int dispatch_threads)
{
  OccupancyMap static_environment = OccupancyMap::FromGridSizes(
      Isometry3::Translation(-1.0, -1.0, -1.0), "world", 0.25, 2.0, 2.0, 2.0, 0.0f);
  for (int64_t x = 0; x < static_environment.NumXVoxels(); x++)
    for (int64_t y = 0; y < static_environment.NumYVoxels(); y++)
      static_environment.SetIndex(x, y, 0, 1.0f);

  const Isometry3 X_CO = QuatZ_then_X(-M_PI_2, -M_PI_2, 0, 0, 0);
  auto cam1 = std::make_shared<VectorPointCloudWrapper>();
  cam1->SetPointCloudOriginTransform(Isometry3::Translation(-2.0, 0.0, 0.0) * X_CO);
  auto cam2 = std::make_shared<VectorPointCloudWrapper>();
  cam2->SetPointCloudOriginTransform(QuatZ_then_X(M_PI_2, 0.0, 0.0, -2.0, 0.0) * X_CO);
  for (double x = -2.0; x <= 2.0; x += 0.03125)
    for (double y = -2.0; y <= 2.0; y += 0.03125)
    {
      cam1->PushBack(x, y, (x <= 0.0) ? 2.125 : 4.0);
      cam2->PushBack(x, y, (x >= 0.0) ? 2.125 : 4.0);
    }
  auto cam3 = std::make_shared<VectorPointCloudWrapper>();  // empty cloud
  cam3->SetPointCloudOriginTransform(X_CO);

  const PointCloudVoxelizationFilterOptions filter_options(1.0, 1, 1);
  std::map<std::string, int32_t> options;
  options["DISPATCH_PARALLELIZE"] = dispatch_threads > 1 ? 1 : 0;
  options["DISPATCH_NUM_THREADS"] = dispatch_threads;
  const auto logging_fn = [](const std::string& msg) { std::cout << msg << std::endl; };

  for (const auto& device : hip_helpers::GetAvailableDevices())
  {
    std::map<std::string, int32_t> merged = device.DeviceOptions();
    for (const auto& kv : options) merged[kv.first] = kv.second;
    std::cout << "Trying voxelizer " << device.DeviceName() << std::endl;
    const HipPointCloudVoxelizer voxelizer(merged, logging_fn);
    check_empty_voxelization(voxelizer.VoxelizePointClouds(static_environment, filter_options, {}));
    double raycast_s = -1.0;
    const OccupancyMap voxelized = voxelizer.VoxelizePointClouds(
        static_environment, filter_options, {cam1, cam2, cam3},
        [&](const VoxelizerRuntime& rt) { raycast_s = rt.RaycastingTime(); });
    check_voxelization(voxelized);
    EXPECT_TRUE(raycast_s >= 0.0);
    {
      // the same clouds as PointCloud2 messages (32-byte records: intensity, x, y, z, ring, pad)
      // are raycast in place and must give the identical grid.
      const auto to_cloud2 = [](const VectorPointCloudWrapper& src, PointCloud2& msg) {
        msg.height = 1;
        msg.width = static_cast<uint32_t>(src.Size());
        msg.point_step = 32;
        msg.fields = {PointField{"intensity", 0, PointField::FLOAT32, 1}, PointField{"x", 4, PointField::FLOAT32, 1},
                      PointField{"y", 8, PointField::FLOAT32, 1}, PointField{"z", 12, PointField::FLOAT32, 1},
                      PointField{"ring", 16, PointField::UINT16, 1}};
        msg.data.assign(static_cast<size_t>(msg.width) * msg.point_step, 0xAB);
        for (int64_t i = 0; i < src.Size(); i++)
          src.CopyPointLocationIntoFloatPtr(i, reinterpret_cast<float*>(msg.data.data() + i * 32 + 4));
      };
      PointCloud2 msg1, msg2, msg3;
      to_cloud2(*cam1, msg1);
      to_cloud2(*cam2, msg2);
      to_cloud2(*cam3, msg3);
      auto w1 = std::make_shared<NonOwningPointCloud2Wrapper>(&msg1, cam1->PointCloudOriginTransform());
      auto w2 = std::make_shared<NonOwningPointCloud2Wrapper>(&msg2, cam2->PointCloudOriginTransform());
      auto w3 = std::make_shared<NonOwningPointCloud2Wrapper>(&msg3, cam3->PointCloudOriginTransform());
      const OccupancyMap from_msgs = voxelizer.VoxelizePointClouds(static_environment, filter_options, {w1, w2, w3});
      check_voxelization(from_msgs);
      EXPECT_TRUE(from_msgs.GetImmutableRawData() == voxelized.GetImmutableRawData());
      // constructor checks of pointcloud_voxelization_ros_interface.cpp:28-78
      bool rejected = false;
      PointCloud2 bad = msg1;
      bad.fields[2].offset = 20;  // y no longer follows x
      try { NonOwningPointCloud2Wrapper w(&bad, Isometry3::Identity()); }
      catch (const std::invalid_argument&) { rejected = true; }
      EXPECT_TRUE(rejected);
      rejected = false;
      bad = msg1;
      bad.fields[3].datatype = PointField::FLOAT64;
      try { NonOwningPointCloud2Wrapper w(&bad, Isometry3::Identity()); }
      catch (const std::invalid_argument&) { rejected = true; }
      EXPECT_TRUE(rejected);
      rejected = false;
      try { NonOwningPointCloud2Wrapper w(nullptr, Isometry3::Identity()); }
      catch (const std::invalid_argument&) { rejected = true; }
      EXPECT_TRUE(rejected);
    }
    {
      // SURVEY 8e, single cloud over several devices: HIP_SPLIT_HELPERS shares every cloud of at least
      // HIP_SPLIT_MIN_POINTS points (the helpers wrap around to this device on a one-GPU box); same grid.
      std::map<std::string, int32_t> split = merged;
      split["HIP_SPLIT_HELPERS"] = 3;
      split["HIP_SPLIT_MIN_POINTS"] = 100;
      const HipPointCloudVoxelizer split_voxelizer(split, logging_fn);
      const OccupancyMap from_shares =
          split_voxelizer.VoxelizePointClouds(static_environment, filter_options, {cam1, cam2, cam3});
      check_voxelization(from_shares);
      EXPECT_TRUE(from_shares.GetImmutableRawData() == voxelized.GetImmutableRawData());
    }
    {
      // SURVEY 8c, the CPU-exact mode: HIP_EXACT_FP64 walks the rays in float64 from the wrappers' DOUBLE points and
      // transforms and filters with a double ratio -- the grid of the reference's CPU voxelizer, which the oracle
      // restates.  Equal cell for cell, on the reference scene and on one where float and double walks differ.
      std::map<std::string, int32_t> exact = merged;
      exact["HIP_EXACT_FP64"] = 1;
      const HipPointCloudVoxelizer exact_voxelizer(exact, logging_fn);
      const OccupancyMap got = exact_voxelizer.VoxelizePointClouds(static_environment, filter_options, {cam1, cam2, cam3});
      check_voxelization(got);
      const OccupancyMap want = OracleCpuVoxelization(static_environment, filter_options, {cam1, cam2, cam3});
      EXPECT_TRUE(got.GetImmutableRawData() == want.GetImmutableRawData());
      // (points that are no floats, a max range that clips, a filter ratio below one)
      auto cam4 = std::make_shared<VectorPointCloudWrapper>();
      cam4->SetPointCloudOriginTransform(QuatZ_then_X(0.3, -1.1, -0.7, 0.2, 0.4));
      for (int i = 0; i < 20000; i++)
      {
        const double a = 0.001 * i, b = 0.37 * i;
        cam4->PushBack(1.7 * std::cos(a) * std::sin(b) + 1e-9 * i, 1.9 * std::sin(a) * std::sin(b), 2.3 * std::cos(b) + 0.1);
      }
      const PointCloudVoxelizationFilterOptions loose(0.6, 2, 1);
      const OccupancyMap got4 = exact_voxelizer.VoxelizePointClouds(static_environment, loose, {cam4, cam1});
      const OccupancyMap want4 = OracleCpuVoxelization(static_environment, loose, {cam4, cam1});
      EXPECT_TRUE(got4.GetImmutableRawData() == want4.GetImmutableRawData());
    }
    // argument validation of the public entry point (pointcloud_voxelization_interface.hpp:267-289)
    bool threw = false;
    try { voxelizer.VoxelizePointClouds(static_environment, filter_options, {nullptr}); }
    catch (const std::invalid_argument&) { threw = true; }
    EXPECT_TRUE(threw);
    // null logging function must be accepted
    const HipPointCloudVoxelizer quiet(merged, {});
  }
  EXPECT_TRUE(!hip_helpers::GetAvailableDevices().empty());
}

static void UnavailableBackendTests()
{
  // a device index that cannot exist: the helper constructs but is unavailable, the voxelizer
  // constructor throws std::runtime_error (device_pointcloud_voxelization.hpp:34-46)
  std::map<std::string, int32_t> options;
  options["HIP_DEVICE"] = 4096;
  const auto helper = hip_helpers::MakeHipVoxelizationHelper(options, {});
  EXPECT_TRUE(helper != nullptr);
  EXPECT_TRUE(!helper->IsAvailable());
  bool threw = false;
  try { HipPointCloudVoxelizer v(options); }
  catch (const std::runtime_error& ex) { threw = std::string(ex.what()).find("not available") != std::string::npos; }
  EXPECT_TRUE(threw);
  threw = false;
  try { PointCloudVoxelizationFilterOptions bad(0.0, 1, 1); }
  catch (const std::invalid_argument&) { threw = true; }
  EXPECT_TRUE(threw);
}

// ---- tagged map types: values derivable by hand on a 1 x 1 x 8 line ----
static void TaggedObjectSdfTests()
{
  // z:          0    1    2    3    4    5    6    7
  // occupancy:  1    1    0    0   .5    0    1    0
  // object id:  3    3    0    0    0    0    9    0
  TaggedObjectOccupancyMap map(Isometry3::Identity(), "test_frame", 1.0, 1, 1, 8, TaggedObjectOccupancyCell());
  const float occ[8] = {1, 1, 0, 0, 0.5f, 0, 1, 0};
  const uint32_t ids[8] = {3, 3, 0, 0, 0, 0, 9, 0};
  for (int z = 0; z < 8; z++) map.SetIndex(0, 0, z, TaggedObjectOccupancyCell{occ[z], ids[z]});
  const DeviceTaggedObjectMap device_map(map);
  SignedDistanceFieldGenerationParameters params;
  {  // no object list: every filled cell, the unknown one included
    const SignedDistanceField sdf = device_map.ExtractSignedDistanceField({}, params);
    const float want[8] = {-2, -1, 1, 1, -1, 1, -1, 1};
    for (int z = 0; z < 8; z++) EXPECT_FLOAT_EQ(sdf.GetIndexImmutable(0, 0, z), want[z]);
    EXPECT_FLOAT_EQ(sdf.minimum, -2.0f);
    EXPECT_FLOAT_EQ(sdf.maximum, 1.0f);
    EXPECT_TRUE(sdf.IsLocked());
  }
  {  // object 9 only
    const SignedDistanceField sdf = device_map.ExtractSignedDistanceField({9u}, params);
    const float want[8] = {6, 5, 4, 3, 2, 1, -1, 1};
    for (int z = 0; z < 8; z++) EXPECT_FLOAT_EQ(sdf.GetIndexImmutable(0, 0, z), want[z]);
  }
  {  // MakeAllObjectSDFs finds ids 3 and 9
    const std::vector<uint32_t> found = device_map.ObjectIds();
    EXPECT_EQ(found.size(), 2u);
    const std::map<uint32_t, SignedDistanceField> all = device_map.MakeAllObjectSDFs(params);
    EXPECT_EQ(all.size(), 2u);
    EXPECT_TRUE(all.count(3u) == 1 && all.count(9u) == 1);
    const float want3[8] = {-2, -1, 1, 2, 3, 4, 5, 6};
    for (int z = 0; z < 8; z++) EXPECT_FLOAT_EQ(all.at(3u).GetIndexImmutable(0, 0, z), want3[z]);
  }
  {  // free-and-named: free field where >= 0, named field where <= 0, else 0 (the unknown cell of object 0)
    const SignedDistanceField sdf = device_map.ExtractFreeAndNamedObjectsSignedDistanceField(params);
    const float want[8] = {-2, -1, 1, 1, 0, 1, -1, 1};
    for (int z = 0; z < 8; z++) EXPECT_FLOAT_EQ(sdf.GetIndexImmutable(0, 0, z), want[z]);
    EXPECT_FLOAT_EQ(sdf.minimum, -2.0f);
    EXPECT_FLOAT_EQ(sdf.maximum, 1.0f);
  }
  {  // the 16-byte cell type gives the same fields; the component map ignores its component
    TaggedObjectOccupancyComponentMap map16(Isometry3::Identity(), "test_frame", 1.0, 1, 1, 8,
                                            TaggedObjectOccupancyComponentCell());
    OccupancyComponentMap map_c(Isometry3::Identity(), "test_frame", 1.0, 1, 1, 8, OccupancyComponentCell());
    for (int z = 0; z < 8; z++)
    {
      map16.SetIndex(0, 0, z, TaggedObjectOccupancyComponentCell{occ[z], ids[z], 77u + z, 5u});
      map_c.SetIndex(0, 0, z, OccupancyComponentCell{occ[z], 1000u + z});
    }
    const DeviceTaggedObjectMap device_map16(map16);
    const SignedDistanceField a = device_map16.ExtractSignedDistanceField({9u}, params);
    const float want9[8] = {6, 5, 4, 3, 2, 1, -1, 1};
    for (int z = 0; z < 8; z++) EXPECT_FLOAT_EQ(a.GetIndexImmutable(0, 0, z), want9[z]);
    const SignedDistanceField c = ExtractSignedDistanceField(map_c, params);
    const float want_all[8] = {-2, -1, 1, 1, -1, 1, -1, 1};
    for (int z = 0; z < 8; z++) EXPECT_FLOAT_EQ(c.GetIndexImmutable(0, 0, z), want_all[z]);
  }
}

// host_types.hpp: the grids' cells -- values of ordinary maps, the untouched map the by-value voxelizer returns, the block
// cache that hands a dropped map's block to the next map of its size (and to no other).
static void HostGridStorageTests()
{
  GridBlockCache::Release();
  const float* first_block = nullptr;
  {
    OccupancyMap a = OccupancyMap::FromGridSizes(Isometry3::Identity(), "world", 0.01, 0.64, 0.64, 0.64, 0.25f);  // 1 MiB
    EXPECT_EQ(a.NumTotalVoxels(), 64 * 64 * 64);
    bool filled = true;
    for (const float v : a.GetImmutableRawData()) filled = filled && v == 0.25f;
    EXPECT_TRUE(filled);
    first_block = a.GetImmutableRawData().data();
    const OccupancyMap copy = a;  // (a copy is a copy: its own block, the same cells)
    EXPECT_TRUE(copy.GetImmutableRawData().data() != first_block);
    EXPECT_TRUE(copy.GetImmutableRawData() == a.GetImmutableRawData());
    OccupancyMap shaped = OccupancyMap::UninitializedLike(a);
    EXPECT_TRUE(shaped.IsInitialized() && shaped.SameSizes(a) && shaped.Frame() == a.Frame());
    EXPECT_EQ(shaped.GetImmutableRawData().size(), a.GetImmutableRawData().size());
    shaped.SetIndex(63, 63, 63, 2.0f);
    EXPECT_EQ(shaped.GetIndexImmutable(63, 63, 63), 2.0f);
  }
  // `a` was dropped last: a map of the same size takes its block over, one of another size does not
  {
    OccupancyMap other_size = OccupancyMap::FromGridSizes(Isometry3::Identity(), "world", 0.01, 0.64, 0.64, 1.28, 0.0f);
    EXPECT_TRUE(other_size.GetImmutableRawData().data() != first_block);
    OccupancyMap again = OccupancyMap::FromGridSizes(Isometry3::Identity(), "world", 0.01, 0.64, 0.64, 0.64, 0.5f);
    bool refilled = true;
    for (const float v : again.GetImmutableRawData()) refilled = refilled && v == 0.5f;
    EXPECT_TRUE(refilled);  // (recycled storage is still initialised storage)
  }
  GridBlockCache::Release();
  // the grids an extraction hands back: cells allocated but not filled; the geometry of a device-resident map: no cells
  {
    const DenseGrid field = DenseGrid::Uninitialized(Isometry3::Translation(1.0, 2.0, 3.0), "world", 0.5, 160, 128, 32);  // 2.5 MiB
    EXPECT_TRUE(field.IsInitialized());
    EXPECT_EQ(field.GetImmutableRawData().size(), static_cast<size_t>(160 * 128 * 32));
    EXPECT_EQ(reinterpret_cast<uintptr_t>(field.GetImmutableRawData().data()) % (size_t{2} << 20), 0u);  // huge-page aligned
    const DenseGrid shape = DenseGrid::ShapeOnly(Isometry3::Translation(1.0, 2.0, 3.0), "world", 0.5, 160, 128, 32);
    EXPECT_TRUE(!shape.IsInitialized() && shape.GetImmutableRawData().empty());
    EXPECT_TRUE(shape.SameSizes(field) && shape.Frame() == "world" && shape.Resolution() == 0.5);
    bool threw = false;
    try { (void)DenseGrid::Uninitialized(Isometry3::Identity(), "world", 0.0, 4, 4, 4); } catch (const std::invalid_argument&) { threw = true; }
    EXPECT_TRUE(threw);
  }
  // the cache's limit: nothing is kept at 0, the default keeps the block
  {
    GridBlockCache::SetLimit(0);
    const float* block = nullptr;
    { OccupancyMap a = OccupancyMap::FromGridSizes(Isometry3::Identity(), "world", 0.01, 0.64, 0.64, 0.64, 0.0f); block = a.GetImmutableRawData().data(); }
    GridBlockCache::SetLimit(GridBlockCache::kDefaultMaxBytes);
    { OccupancyMap b = OccupancyMap::FromGridSizes(Isometry3::Identity(), "world", 0.01, 0.64, 0.64, 0.64, 0.0f); block = b.GetImmutableRawData().data(); }
    { OccupancyMap c = OccupancyMap::FromGridSizes(Isometry3::Identity(), "world", 0.01, 0.64, 0.64, 0.64, 0.0f); EXPECT_TRUE(c.GetImmutableRawData().data() == block); }
    GridBlockCache::Release();
  }
  // small grids never enter the cache
  { OccupancyMap tiny = OccupancyMap::FromGridSizes(Isometry3::Identity(), "world", 0.25, 1.0, 2.0, 3.0, 0.0f); (void)tiny; }
}

// Several host threads on the process's ONE context per device (every map type shares it since round 6): extractions of
// all four map types, small (kernels on the page-locked ring), medium (ring transfers) and batched, at the same time.
// Every field must equal the one the same call returned on a quiet context.
static void ConcurrentCallersTests()
{
  const SignedDistanceFieldGenerationParameters params;
  std::vector<FourMaps> scenes;
  for (const double res : {0.25, 0.125, 0.0625})  // 16^3 ... 64^3 cells for 4 x 4 x 4 m
  {
    scenes.emplace_back(Isometry3::Identity(), res, 4.0, 4.0, 4.0, 0.0f);
    const int n = static_cast<int>(scenes.back().occupancy.NumXVoxels());
    scenes.back().FillBox(n / 4, n / 2, n / 8, n / 3, n / 5, n - 2);
    scenes.back().FillBox(0, 2, n - 3, n, 1, 3);
  }
  OccupancyMap medium = OccupancyMap::FromGridSizes(Isometry3::Identity(), "test_frame", 0.03125, 4.0, 4.0, 4.0, 0.0f);  // 128^3
  FillBox(medium, 20, 50, 60, 100, 5, 120);
  std::vector<std::vector<SignedDistanceField>> quiet;
  for (const FourMaps& scene : scenes) quiet.push_back(scene.Fields());
  const SignedDistanceField quiet_medium = ExtractSignedDistanceField(medium, params);
  std::vector<const OccupancyMap*> batch;
  for (int i = 0; i < 6; i++) batch.push_back(&scenes[2].occupancy);
  std::atomic<int> mismatches{0};
  const auto same = [](const SignedDistanceField& a, const SignedDistanceField& b) {
    return a.grid.GetImmutableRawData().size() == b.grid.GetImmutableRawData().size() && a.minimum == b.minimum &&
           a.maximum == b.maximum &&
           std::memcmp(a.grid.GetImmutableRawData().data(), b.grid.GetImmutableRawData().data(),
                       a.grid.GetImmutableRawData().size() * sizeof(float)) == 0;
  };
  std::vector<std::thread> pool;
  for (int t = 0; t < 6; t++)
    pool.emplace_back([&, t]() {
      try
      {
        for (int round = 0; round < 12; round++)
        {
          const size_t k = static_cast<size_t>((t + round) % 3);
          const std::vector<SignedDistanceField> fields = scenes[k].Fields();
          for (size_t f = 0; f < fields.size(); f++)
            if (!same(fields[f], quiet[k][f])) mismatches++;
          if ((t + round) % 4 == 0 && !same(ExtractSignedDistanceField(medium, params), quiet_medium)) mismatches++;
          if ((t + round) % 5 == 0)
            for (const SignedDistanceField& field : ExtractSignedDistanceFields(batch, params))
              if (!same(field, quiet[2][0])) mismatches++;
        }
      }
      catch (const std::exception& ex)
      {
        std::printf("concurrent caller threw: %s\n", ex.what());
        mismatches++;
      }
    });
  for (auto& th : pool) th.join();
  EXPECT_EQ(mismatches.load(), 0);
}

int main(int argc, char** argv)
{
  const bool no_device = (argc > 1 && std::strcmp(argv[1], "--no-device") == 0);
  UnavailableBackendTests();
  HostGridStorageTests();
  if (!no_device)
  {
    SdfGenerationTests();
    MultiDeviceSdfTests();
    SdfConsumerTests();
    TaggedObjectSdfTests();
    BatchedSdfTests();
    ConcurrentCallersTests();
    PointCloudVoxelizationTests(1);
    PointCloudVoxelizationTests(4);
  }
  std::printf("%s: %d failure(s)\n", g_failures ? "FAILED" : "PASSED", g_failures);
  return g_failures ? 1 : 0;
}
