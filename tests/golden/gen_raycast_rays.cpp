// Regenerates the 1000 (origin, point) pairs drawn by the reference's
// test/voxel_raycasting_test.cpp:26-55,88-100:
//   std::mt19937_64 prng(42); u = std::generate_canonical<double, 53>(prng);
//   coordinate = Interpolate(-2.0, 7.0, u)  ==  -2.0 + (7.0 - (-2.0)) * u
// six draws per iteration in the order origin.x, origin.y, origin.z, point.x,
// point.y, point.z.  Writes 6000 little-endian doubles to argv[1].
// (Fixture generator; make_golden.py compiles and runs it.)
#include <cstdio>
#include <limits>
#include <random>

int main(int argc, char** argv)
{
  if (argc < 2) return 2;
  std::mt19937_64 prng(42);
  std::FILE* out = std::fopen(argv[1], "wb");
  if (!out) return 3;
  for (int iter = 0; iter < 1000; iter++)
  {
    double v[6];
    for (int i = 0; i < 6; i++)
    {
      const double u = std::generate_canonical<
          double, std::numeric_limits<double>::digits>(prng);
      v[i] = -2.0 + ((7.0 - (-2.0)) * u);
    }
    std::fwrite(v, sizeof(double), 6, out);
  }
  std::fclose(out);
  return 0;
}
