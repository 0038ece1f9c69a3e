// Clock of the SDF entry points of the C++ host layer (csrc/host/hip_pointcloud_voxelizer.cc) as a caller of the
// reference sees them: a map in host memory in, a SignedDistanceField (a fresh grid in host memory) out --
// ExtractSignedDistanceField (occupancy_map.hpp:189-199 in the reference), the batch over many maps, and
// MakeAllObjectSDFs of a tagged map (tagged_object_occupancy_map.hpp:249-290).  Prints one JSON line.
//   bench_sdf_host [edge of the large map = 512] [repetitions = 5] [anything: also the two-slab entry point]
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <vector>

#include "../../include/vgt_hip/hip_pointcloud_voxelizer.hpp"

using namespace vgt_hip;

namespace
{
double Now()
{
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// boxes of filled cells, written as runs along z
void FillBoxes(OccupancyMap& map, int boxes, uint64_t seed)
{
  std::mt19937_64 rng(seed);
  const int64_t nx = map.NumXVoxels(), ny = map.NumYVoxels(), nz = map.NumZVoxels();
  float* cells = map.GetMutableRawData().data();
  for (int b = 0; b < boxes; b++)
  {
    const int64_t ex = 1 + static_cast<int64_t>(rng() % std::max<int64_t>(1, nx / 8));
    const int64_t ey = 1 + static_cast<int64_t>(rng() % std::max<int64_t>(1, ny / 8));
    const int64_t ez = 1 + static_cast<int64_t>(rng() % std::max<int64_t>(1, nz / 8));
    const int64_t x0 = static_cast<int64_t>(rng() % (nx - ex + 1)), y0 = static_cast<int64_t>(rng() % (ny - ey + 1));
    const int64_t z0 = static_cast<int64_t>(rng() % (nz - ez + 1));
    for (int64_t x = x0; x < x0 + ex; x++)
      for (int64_t y = y0; y < y0 + ey; y++) std::fill_n(cells + (x * ny + y) * nz + z0, ez, 1.0f);
  }
}

struct Clock
{
  double first = 0.0, best = 1e30, sum = 0.0;
  int n = 0;
  void Add(double s)
  {
    if (n == 0) first = s;
    best = std::min(best, s);
    sum += s;
    n++;
  }
  void Print(const char* name, bool last) const
  {
    std::printf("\"%s\": {\"first_ms\": %.3f, \"best_ms\": %.3f, \"mean_ms\": %.3f, \"calls\": %d}%s", name, first * 1e3,
                best * 1e3, sum / n * 1e3, n, last ? "" : ", ");
  }
};
}  // namespace

int main(int argc, char** argv)
{
  const int64_t edge = argc > 1 ? std::atoll(argv[1]) : 512;
  const int reps = argc > 2 ? std::atoi(argv[2]) : 5;
  SignedDistanceFieldGenerationParameters params;
  double checksum = 0.0;

  // one large map
  Clock large;
  {
    OccupancyMap map(Isometry3::Identity(), "bench", 0.01, edge, edge, edge, 0.0f);
    FillBoxes(map, 40, 7);
    for (int r = 0; r < reps; r++)
    {
      const double t0 = Now();
      const SignedDistanceField sdf = ExtractSignedDistanceField(map, params);
      large.Add(Now() - t0);
      checksum += sdf.minimum + sdf.maximum + sdf.grid.GetImmutableRawData()[static_cast<size_t>(r) * 977];
    }
  }
  // the sizes of the reference's own examples
  Clock small;
  {
    OccupancyMap map(Isometry3::Identity(), "bench", 0.25, 40, 40, 40, 0.0f);
    FillBoxes(map, 6, 11);
    for (int r = 0; r < 200; r++)
    {
      const double t0 = Now();
      const SignedDistanceField sdf = ExtractSignedDistanceField(map, params);
      small.Add(Now() - t0);
      checksum += sdf.maximum;
    }
  }
  // 64 maps of 64^3 as one batch
  Clock batch;
  {
    std::vector<OccupancyMap> maps;
    for (int i = 0; i < 64; i++)
    {
      maps.emplace_back(Isometry3::Identity(), "bench", 0.05, 64, 64, 64, 0.0f);
      FillBoxes(maps.back(), 8, 100 + static_cast<uint64_t>(i));
    }
    std::vector<const OccupancyMap*> pointers;
    for (const OccupancyMap& m : maps) pointers.push_back(&m);
    for (int r = 0; r < reps * 4; r++)
    {
      const double t0 = Now();
      const std::vector<SignedDistanceField> fields = ExtractSignedDistanceFields(pointers, params);
      batch.Add(Now() - t0);
      checksum += fields.back().maximum;
    }
  }
  // 32 objects in a 128^3 tagged map
  Clock objects;
  {
    TaggedObjectOccupancyMap tagged(Isometry3::Identity(), "bench", 0.02, 128, 128, 128, TaggedObjectOccupancyCell());
    for (uint32_t id = 1; id <= 32; id++)
    {
      const int64_t x0 = (id * 37) % 112, y0 = (id * 53) % 112, z0 = (id * 71) % 112;
      for (int64_t x = x0; x < x0 + 12; x++)
        for (int64_t y = y0; y < y0 + 12; y++)
          for (int64_t z = z0; z < z0 + 12; z++) tagged.SetIndex(x, y, z, TaggedObjectOccupancyCell{1.0f, id});
    }
    const DeviceTaggedObjectMap device_map(tagged);
    for (int r = 0; r < reps * 4; r++)
    {
      const double t0 = Now();
      const std::map<uint32_t, SignedDistanceField> all = device_map.MakeAllObjectSDFs(params);
      objects.Add(Now() - t0);
      checksum += all.begin()->second.maximum + static_cast<double>(all.size());
    }
  }
  // one field of a tagged map: 256^3 (64 MiB out) and 128^3 (8 MiB out)
  Clock tagged_large, tagged_medium, tagged_one_shot;
  for (const int64_t n : {int64_t{256}, int64_t{128}})
  {
    TaggedObjectOccupancyMap tagged(Isometry3::Identity(), "bench", 0.02, n, n, n, TaggedObjectOccupancyCell());
    for (uint32_t id = 1; id <= 8; id++)
      for (int64_t x = 0; x < n / 10; x++)
        for (int64_t y = 0; y < n / 10; y++)
          for (int64_t z = 0; z < n / 10; z++)
            tagged.SetIndex((id * 37) % (n - n / 10) + x, (id * 53) % (n - n / 10) + y, (id * 71) % (n - n / 10) + z,
                            TaggedObjectOccupancyCell{1.0f, id});
    if (n == 128)
      for (int r = 0; r < reps * 4; r++)
      {
        // the reference's one-shot interface: upload, extract, drop
        const double t0 = Now();
        const SignedDistanceField sdf = DeviceTaggedObjectMap(tagged).ExtractSignedDistanceField({}, params);
        tagged_one_shot.Add(Now() - t0);
        checksum += sdf.maximum;
      }
    const DeviceTaggedObjectMap device_map(tagged);
    Clock& clock = n == 256 ? tagged_large : tagged_medium;
    for (int r = 0; r < reps * 4; r++)
    {
      const double t0 = Now();
      const SignedDistanceField sdf = device_map.ExtractSignedDistanceField({}, params);
      clock.Add(Now() - t0);
      checksum += sdf.maximum;
    }
  }
  // a tagged map of the reference's own test sizes: one field from the resident map, and the one-shot form
  Clock tagged_small, tagged_small_one_shot;
  {
    TaggedObjectOccupancyMap tagged(Isometry3::Identity(), "bench", 0.25, 40, 40, 40, TaggedObjectOccupancyCell());
    for (uint32_t id = 1; id <= 4; id++)
      for (int64_t x = 0; x < 6; x++)
        for (int64_t y = 0; y < 6; y++)
          for (int64_t z = 0; z < 6; z++)
            tagged.SetIndex((id * 7) % 34 + x, (id * 11) % 34 + y, (id * 13) % 34 + z, TaggedObjectOccupancyCell{1.0f, id});
    for (int r = 0; r < 200; r++)
    {
      const double t0 = Now();
      const SignedDistanceField sdf = DeviceTaggedObjectMap(tagged).ExtractSignedDistanceField({}, params);
      tagged_small_one_shot.Add(Now() - t0);
      checksum += sdf.maximum;
    }
    const DeviceTaggedObjectMap device_map(tagged);
    for (int r = 0; r < 200; r++)
    {
      const double t0 = Now();
      const SignedDistanceField sdf = device_map.ExtractSignedDistanceField({}, params);
      tagged_small.Add(Now() - t0);
      checksum += sdf.maximum;
    }
  }
  // a component map (uploaded, extracted and dropped per call, as the reference's interface has it)
  Clock component;
  {
    OccupancyComponentMap map(Isometry3::Identity(), "bench", 0.02, 128, 128, 128, OccupancyComponentCell());
    for (int64_t x = 40; x < 60; x++)
      for (int64_t y = 30; y < 90; y++)
        for (int64_t z = 50; z < 70; z++) map.SetIndex(x, y, z, OccupancyComponentCell{1.0f, 1u});
    for (int r = 0; r < reps * 4; r++)
    {
      const double t0 = Now();
      const SignedDistanceField sdf = ExtractSignedDistanceField(map, params);
      component.Add(Now() - t0);
      checksum += sdf.maximum;
    }
  }
  // the large map over two Z slabs of device 0 (the one-process multi-device entry point)
  Clock slabs;
  if (argc > 3)
  {
    OccupancyMap map(Isometry3::Identity(), "bench", 0.01, edge, edge, edge, 0.0f);
    FillBoxes(map, 40, 7);
    SignedDistanceFieldGenerationParameters two = params;
    two.hip_devices = {0, 0};
    for (int r = 0; r < reps; r++)
    {
      const double t0 = Now();
      const SignedDistanceField sdf = ExtractSignedDistanceField(map, two);
      slabs.Add(Now() - t0);
      checksum += sdf.maximum;
    }
  }
  std::printf("{\"edge\": %lld, ", static_cast<long long>(edge));
  large.Print("ExtractSignedDistanceField, one large map", false);
  small.Print("ExtractSignedDistanceField, 40^3", false);
  batch.Print("ExtractSignedDistanceFields, 64 maps of 64^3", false);
  objects.Print("MakeAllObjectSDFs, 32 objects in 128^3", false);
  tagged_large.Print("tagged map 256^3, one field", false);
  tagged_medium.Print("tagged map 128^3, one field", false);
  tagged_one_shot.Print("tagged map 128^3, uploaded per call", false);
  tagged_small.Print("tagged map 40^3, one field", false);
  tagged_small_one_shot.Print("tagged map 40^3, uploaded per call", false);
  component.Print("component map 128^3", false);
  if (slabs.n > 0) slabs.Print("one large map over two slabs of device 0", false);
  std::printf("\"checksum\": %.6g}\n", checksum);
  return 0;
}
