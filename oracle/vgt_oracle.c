/*
 * vgt_oracle.c -- CPU oracle (see vgt_oracle.h: TEST INFRASTRUCTURE ONLY).
 *
 * Every function cites the reference file:line it restates.  Paths are relative
 * to the reference checkout: S/ = src/voxelized_geometry_tools/,
 * I/ = include/voxelized_geometry_tools/.
 *
 * Build: see oracle/Makefile (gcc -O3 -march=native -fopenmp -ffp-contract=off).
 * -ffp-contract=off keeps the float DDA free of FMA contraction so that it is
 * the canonical scalar left-to-right evaluation the HIP kernel is compared to.
 */
#include "vgt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

int vgt_oracle_max_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

static int resolve_threads(int threads)
{
  if (threads > 0) return threads;
  return vgt_oracle_max_threads();
}

/* ------------------------------------------------------------------------- */
/* 1-D squared distance transform                                            */
/* ------------------------------------------------------------------------- */

#define EDT_AT(line, stride, i) ((line)[(i) * (stride)])

static double sq_i64(int64_t v) { return (double)(v * v); }

/* S/signed_distance_field_generation.cpp:85-122 -- O(n^2) for short lines. */
static void edt1d_bruteforce(double* line, int64_t n, int64_t stride,
                             double* d)
{
  for (int64_t q = 0; q < n; q++) d[q] = INFINITY;
  for (int64_t q = 0; q < n; q++)
  {
    for (int64_t o = 0; o < n; o++)
    {
      const double cand = sq_i64(q - o) + EDT_AT(line, stride, o);
      if (cand < d[q]) d[q] = cand;
    }
  }
  for (int64_t q = 0; q < n; q++) EDT_AT(line, stride, q) = d[q];
}

/* S/signed_distance_field_generation.cpp:153-172 -- inf-safe subtraction. */
static double inf_safe_sub(double a, double b)
{
  if (a == INFINITY && b == INFINITY) return 0.0;
  if (a == INFINITY) return INFINITY;
  if (b == INFINITY) return -INFINITY;
  return a - b;
}

/* S/signed_distance_field_generation.cpp:175-184 -- parabola intersection. */
static double edt_intersection(const double* line, int64_t stride, int64_t q,
                               int64_t vk)
{
  const double fq = EDT_AT(line, stride, q);
  const double fv = EDT_AT(line, stride, vk);
  const double top = inf_safe_sub(fq + sq_i64(q), fv + sq_i64(vk));
  const double bottom = (double)((2 * q) - (2 * vk));
  return top / bottom;
}

/* S/signed_distance_field_generation.cpp:124-226 -- Felzenszwalb-Huttenlocher
 * lower envelope; z has n+1 entries, v and d have n. */
static void edt1d_linear(double* line, int64_t n, int64_t stride, double* z,
                         int64_t* v, double* d)
{
  memset(z, 0, sizeof(double) * (size_t)(n + 1));
  memset(v, 0, sizeof(int64_t) * (size_t)n);
  memset(d, 0, sizeof(double) * (size_t)n);
  z[0] = -INFINITY;
  z[1] = INFINITY;

  /* phase 1 (:187-204) */
  int64_t k = 0;
  for (int64_t q = 1; q < n; q++)
  {
    double s = edt_intersection(line, stride, q, v[k]);
    while (k > 0 && s <= z[k])
    {
      k--;
      s = edt_intersection(line, stride, q, v[k]);
    }
    k++;
    v[k] = q;
    z[k] = s;
    z[k + 1] = INFINITY;
  }

  /* phase 2 (:207-225) */
  k = 0;
  for (int64_t q = 0; q < n; q++)
  {
    while (z[k + 1] < (double)q) k++;
    const int64_t vk = v[k];
    d[q] = sq_i64(q - vk) + EDT_AT(line, stride, vk);
  }
  for (int64_t q = 0; q < n; q++) EDT_AT(line, stride, q) = d[q];
}

/* S/signed_distance_field_generation.cpp:229-248 -- strategy threshold 8. */
static void edt1d_dispatch(double* line, int64_t n, int64_t stride, double* z,
                           int64_t* v, double* d)
{
  if (n > 8)
    edt1d_linear(line, n, stride, z, v, d);
  else
    edt1d_bruteforce(line, n, stride, d);
}

void vgt_oracle_edt1d_inplace(double* line, int64_t n, int64_t stride)
{
  double* z = (double*)malloc(sizeof(double) * (size_t)(n + 1));
  int64_t* v = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
  double* d = (double*)malloc(sizeof(double) * (size_t)n);
  edt1d_dispatch(line, n, stride, z, v, d);
  free(z);
  free(v);
  free(d);
}

/* One axis sweep over `count` independent lines; line i starts at
 * field + (i / step) * outer + (i % step) * inner  (:265-271 index split).
 * Static contiguous ranges per thread with per-thread scratch, like
 * StaticParallelForRangeLoop at :286-311. */
static void edt_axis_sweep(double* field, int64_t n, int64_t stride,
                           int64_t count, int64_t step, int64_t outer,
                           int64_t inner, int threads)
{
#pragma omp parallel num_threads(threads)
  {
    double* z = (double*)malloc(sizeof(double) * (size_t)(n + 1));
    int64_t* v = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
    double* d = (double*)malloc(sizeof(double) * (size_t)n);
#pragma omp for schedule(static)
    for (int64_t it = 0; it < count; it++)
    {
      const int64_t first = it / step;
      const int64_t second = it % step;
      edt1d_dispatch(field + first * outer + second * inner, n, stride, z, v,
                     d);
    }
    free(z);
    free(v);
    free(d);
  }
}

/* S/signed_distance_field_generation.cpp:258-391. */
void vgt_oracle_edt3d_inplace(double* field, int64_t nx, int64_t ny,
                              int64_t nz, int threads)
{
  threads = resolve_threads(threads);
  const int64_t sx = ny * nz, sy = nz, sz = 1;
  if (nx > 1) /* X lines, iteration -> (y, z) by / and % nz (:275-312) */
    edt_axis_sweep(field, nx, sx, ny * nz, nz, sy, sz, threads);
  if (ny > 1) /* Y lines, iteration -> (x, z) (:314-351) */
    edt_axis_sweep(field, ny, sy, nx * nz, nz, sx, sz, threads);
  if (nz > 1) /* Z lines, iteration -> (x, y) by / and % ny (:353-390) */
    edt_axis_sweep(field, nz, sz, nx * ny, ny, sx, sy, threads);
}

/* ------------------------------------------------------------------------- */
/* SDF assembly                                                              */
/* ------------------------------------------------------------------------- */

/* I/signed_distance_field_generation.hpp:39-113. */
int vgt_oracle_sdf_from_mask(const uint8_t* filled, int64_t nx, int64_t ny,
                             int64_t nz, double resolution, float* sdf_out,
                             int threads)
{
  const int64_t n = nx * ny * nz;
  if (n <= 0) return 1;
  double* to_filled = (double*)malloc(sizeof(double) * (size_t)n);
  double* to_free = (double*)malloc(sizeof(double) * (size_t)n);
  if (!to_filled || !to_free)
  {
    free(to_filled);
    free(to_free);
    return 2;
  }
  /* :47-74 -- both fields start at +inf, marked cells get 0 (serial loop) */
  for (int64_t i = 0; i < n; i++)
  {
    to_filled[i] = INFINITY;
    to_free[i] = INFINITY;
  }
  for (int64_t i = 0; i < n; i++)
  {
    if (filled[i])
      to_filled[i] = 0.0;
    else
      to_free[i] = 0.0;
  }
  /* :77-80 */
  vgt_oracle_edt3d_inplace(to_filled, nx, ny, nz, threads);
  vgt_oracle_edt3d_inplace(to_free, nx, ny, nz, threads);
  /* :85-108 (serial loop) */
  for (int64_t i = 0; i < n; i++)
  {
    const double distance1 = sqrt(to_filled[i]) * resolution;
    const double distance2 = sqrt(to_free[i]) * resolution;
    const double distance = distance1 - distance2;
    sdf_out[i] = (float)distance;
  }
  free(to_filled);
  free(to_free);
  return 0;
}

/* I/occupancy_map.hpp:181-205 -- the is_filled predicate. */
static int occupancy_is_filled(float occupancy, int unknown_is_filled)
{
  if (occupancy > 0.5) return 1;
  if (unknown_is_filled && (occupancy == 0.5)) return 1;
  return 0;
}

int vgt_oracle_sdf_from_occupancy(const float* occupancy, int64_t nx,
                                  int64_t ny, int64_t nz, double resolution,
                                  int unknown_is_filled, int add_virtual_border,
                                  float* sdf_out, float* out_min,
                                  float* out_max, int threads)
{
  const int64_t n = nx * ny * nz;
  if (n <= 0 || nx <= 0 || ny <= 0 || nz <= 0) return 1;
  int rc = 0;
  if (!add_virtual_border)
  {
    /* I/signed_distance_field_generation.hpp:127-133 */
    uint8_t* mask = (uint8_t*)malloc((size_t)n);
    if (!mask) return 2;
    for (int64_t i = 0; i < n; i++)
      mask[i] = (uint8_t)occupancy_is_filled(occupancy[i], unknown_is_filled);
    rc = vgt_oracle_sdf_from_mask(mask, nx, ny, nz, resolution, sdf_out,
                                  threads);
    free(mask);
  }
  else
  {
    /* I/signed_distance_field_generation.hpp:134-284 -- pad every axis that
     * has more than one voxel by one cell on each side, build one SDF with
     * the border filled and one with the border empty, combine. */
    const int64_t ox = (nx > 1) ? 1 : 0, oy = (ny > 1) ? 1 : 0,
                  oz = (nz > 1) ? 1 : 0;
    const int64_t px = nx + 2 * ox, py = ny + 2 * oy, pz = nz + 2 * oz;
    const int64_t pn = px * py * pz;
    uint8_t* border_filled = (uint8_t*)malloc((size_t)pn);
    uint8_t* border_empty = (uint8_t*)malloc((size_t)pn);
    float* sdf_free = (float*)malloc(sizeof(float) * (size_t)pn);
    float* sdf_filled = (float*)malloc(sizeof(float) * (size_t)pn);
    if (!border_filled || !border_empty || !sdf_free || !sdf_filled)
    {
      free(border_filled);
      free(border_empty);
      free(sdf_free);
      free(sdf_filled);
      return 2;
    }
    for (int64_t x = 0; x < px; x++)
      for (int64_t y = 0; y < py; y++)
        for (int64_t z = 0; z < pz; z++)
        {
          const int on_border = (ox && (x == 0 || x == px - 1)) ||
                                (oy && (y == 0 || y == py - 1)) ||
                                (oz && (z == 0 || z == pz - 1));
          const int64_t pi = (x * py + y) * pz + z;
          if (on_border)
          {
            border_filled[pi] = 1; /* :156-194 */
            border_empty[pi] = 0;  /* :197-235 */
          }
          else
          {
            const int64_t ri = ((x - ox) * ny + (y - oy)) * nz + (z - oz);
            const uint8_t f = (uint8_t)occupancy_is_filled(occupancy[ri],
                                                           unknown_is_filled);
            border_filled[pi] = f;
            border_empty[pi] = f;
          }
        }
    rc = vgt_oracle_sdf_from_mask(border_filled, px, py, pz, resolution,
                                  sdf_free, threads);
    if (rc == 0)
      rc = vgt_oracle_sdf_from_mask(border_empty, px, py, pz, resolution,
                                    sdf_filled, threads);
    if (rc == 0)
    {
      /* :247-279 */
      for (int64_t x = 0; x < nx; x++)
        for (int64_t y = 0; y < ny; y++)
          for (int64_t z = 0; z < nz; z++)
          {
            const int64_t pi = ((x + ox) * py + (y + oy)) * pz + (z + oz);
            const float free_value = sdf_free[pi];
            const float filled_value = sdf_filled[pi];
            float out;
            if (free_value >= 0.0)
              out = free_value;
            else if (filled_value <= -0.0)
              out = filled_value;
            else
              out = 0.0f;
            sdf_out[(x * ny + y) * nz + z] = out;
          }
    }
    free(border_filled);
    free(border_empty);
    free(sdf_free);
    free(sdf_filled);
  }
  if (rc != 0) return rc;
  /* I/signed_distance_field.hpp:765-787 -- Lock(): serial minmax scan */
  float lo = sdf_out[0], hi = sdf_out[0];
  for (int64_t i = 1; i < n; i++)
  {
    if (sdf_out[i] < lo) lo = sdf_out[i];
    if (hi < sdf_out[i]) hi = sdf_out[i];
  }
  if (out_min) *out_min = lo;
  if (out_max) *out_max = hi;
  return 0;
}

/* ------------------------------------------------------------------------- */
/* Raycast DDA -- float32 (device kernel restatement)                        */
/* ------------------------------------------------------------------------- */
/* Map types whose cells carry an object id.                    */
/* ------------------------------------------------------------------------- */

void vgt_oracle_cells_filled_mask(const void* cells, int64_t num_cells, int cell_bytes,
                                  int object_id_offset, int mode, const uint32_t* objects,
                                  int64_t num_objects, int unknown_is_filled,
                                  uint8_t* mask_out)
{
  const uint8_t* base = (const uint8_t*)cells;
  for (int64_t i = 0; i < num_cells; i++)
  {
    float occupancy;
    uint32_t object_id = 0;
    memcpy(&occupancy, base + i * cell_bytes, sizeof(float));
    if (object_id_offset >= 0)
      memcpy(&object_id, base + i * cell_bytes + object_id_offset, sizeof(uint32_t));
    int considered = 1;
    if (mode == 1)
    {
      /* tagged_object_occupancy_map.hpp:219-221: in the set, or no objects supplied */
      considered = (num_objects == 0);
      for (int64_t k = 0; k < num_objects && !considered; k++)
        if (objects[k] == object_id) considered = 1;
    }
    else if (mode == 2)
    {
      considered = (object_id > 0u); /* :330 */
    }
    mask_out[i] = (uint8_t)(considered && occupancy_is_filled(occupancy, unknown_is_filled));
  }
}

void vgt_oracle_combine_free_and_named(const float* free_sdf, const float* named_sdf,
                                       int64_t num_cells, float* out, float* out_min,
                                       float* out_max)
{
  float lo = INFINITY, hi = -INFINITY;
  for (int64_t i = 0; i < num_cells; i++)
  {
    const float free_sdf_value = free_sdf[i];
    const float named_objects_sdf_value = named_sdf[i];
    float v;
    if (free_sdf_value >= 0.0)
      v = free_sdf_value;
    else if (named_objects_sdf_value <= -0.0)
      v = named_objects_sdf_value;
    else
      v = 0.0f;
    out[i] = v;
    if (v < lo) lo = v;
    if (v > hi) hi = v;
  }
  if (out_min) *out_min = lo;
  if (out_max) *out_max = hi;
}

/* ------------------------------------------------------------------------- */
/* SDF consumers: trilinear distance estimate, fine gradient.   */
/* ------------------------------------------------------------------------- */

/* GetAxisInterpolationIndices, signed_distance_field.hpp:276-313 */
static void axis_interpolation_indices(int64_t initial_index, int64_t axis_size, double axis_offset,
                                       int64_t* lower_out, int64_t* upper_out)
{
  int64_t lower = initial_index;
  int64_t upper = initial_index;
  if (axis_offset >= 0.0)
  {
    upper = initial_index + 1;
    if (upper >= axis_size)
    {
      upper = initial_index;
      lower = initial_index - 1;
      if (lower < 0) lower = initial_index;
    }
  }
  else
  {
    lower = initial_index - 1;
    if (lower < 0)
    {
      upper = initial_index + 1;
      lower = initial_index;
      if (upper >= axis_size) upper = initial_index;
    }
  }
  *lower_out = lower;
  *upper_out = upper;
}

/* GetCorrectedCenterDistance, :259-273 */
static double corrected_center_distance(const float* sdf, int64_t ny, int64_t nz, int64_t x, int64_t y, int64_t z,
                                        double resolution)
{
  const double nominal_sdf_distance = (double)sdf[(x * ny + y) * nz + z];
  const double cell_center_distance_offset = resolution * 0.5;
  if (nominal_sdf_distance >= 0.0) return nominal_sdf_distance - cell_center_distance_offset;
  return nominal_sdf_distance + cell_center_distance_offset;
}

static double lerp(double a, double b, double t) { return a * (1.0 - t) + b * t; }

/* EstimateLocationDistance4d (:822-833) -> EstimateDistanceInterpolateFromNeighbors (:316-378).  The trilinear
 * interpolation itself is common_robotics_utilities::math::TrilinearInterpolate, whose source is not in the
 * container: evaluated here along x, then y, then z, each a*(1-t) + b*t (the order csrc/cell_kernels.hip
 * documents).  Returns 0 when the location is outside the grid. */
static int estimate_location_distance(const float* sdf, int64_t nx, int64_t ny, int64_t nz, double resolution,
                                      const double* grid_from_world, double x, double y, double z, double* out)
{
  double g[3] = {x, y, z};
  if (grid_from_world)
  {
    const double* M = grid_from_world;
    g[0] = M[0] * x + M[4] * y + M[8] * z + M[12];
    g[1] = M[1] * x + M[5] * y + M[9] * z + M[13];
    g[2] = M[2] * x + M[6] * y + M[10] * z + M[14];
  }
  const double inv = 1.0 / resolution;
  const double fx = floor(g[0] * inv), fy = floor(g[1] * inv), fz = floor(g[2] * inv);
  if (!(fx >= 0.0 && fx < (double)nx && fy >= 0.0 && fy < (double)ny && fz >= 0.0 && fz < (double)nz)) return 0;
  const int64_t x_idx = (int64_t)fx, y_idx = (int64_t)fy, z_idx = (int64_t)fz;
  const double cx = ((double)x_idx + 0.5) * resolution;
  const double cy = ((double)y_idx + 0.5) * resolution;
  const double cz = ((double)z_idx + 0.5) * resolution;
  int64_t lx, ux, ly, uy, lz, uz;
  axis_interpolation_indices(x_idx, nx, g[0] - cx, &lx, &ux);
  axis_interpolation_indices(y_idx, ny, g[1] - cy, &ly, &uy);
  axis_interpolation_indices(z_idx, nz, g[2] - cz, &lz, &uz);
#define CCD(a, b, c) corrected_center_distance(sdf, ny, nz, (a), (b), (c), resolution)
  const double mxmymz = CCD(lx, ly, lz), mxmypz = CCD(lx, ly, uz), mxpymz = CCD(lx, uy, lz), mxpypz = CCD(lx, uy, uz);
  const double pxmymz = CCD(ux, ly, lz), pxmypz = CCD(ux, ly, uz), pxpymz = CCD(ux, uy, lz), pxpypz = CCD(ux, uy, uz);
#undef CCD
  const double low_x = ((double)lx + 0.5) * resolution;
  const double low_y = ((double)ly + 0.5) * resolution;
  const double low_z = ((double)lz + 0.5) * resolution;
  const double tx = (g[0] - low_x) / ((low_x + resolution) - low_x);
  const double ty = (g[1] - low_y) / ((low_y + resolution) - low_y);
  const double tz = (g[2] - low_z) / ((low_z + resolution) - low_z);
  const double mm = lerp(mxmymz, pxmymz, tx), mp = lerp(mxmypz, pxmypz, tx);
  const double pm = lerp(mxpymz, pxpymz, tx), pp = lerp(mxpypz, pxpypz, tx);
  const double lo = lerp(mm, pm, ty), hi = lerp(mp, pp, ty);
  *out = lerp(lo, hi, tz);
  return 1;
}

void vgt_oracle_estimate_distance(const float* sdf, int64_t nx, int64_t ny, int64_t nz, double resolution,
                                  const double* grid_from_world, const double* queries, int64_t num_queries,
                                  double* distance, uint8_t* has_value)
{
  for (int64_t i = 0; i < num_queries; i++)
  {
    double v = NAN;
    const int ok = estimate_location_distance(sdf, nx, ny, nz, resolution, grid_from_world, queries[3 * i],
                                              queries[3 * i + 1], queries[3 * i + 2], &v);
    distance[i] = ok ? v : NAN;
    if (has_value) has_value[i] = (uint8_t)ok;
  }
}

/* ComputeAxisFineGradient, :214-254; returns 0 where the reference throws */
static int axis_fine_gradient(int point_ok, double point, int minus_ok, double minus, int plus_ok, double plus,
                              double query_point_axis_value, double minus_point_axis_value,
                              double plus_point_axis_value, double* out)
{
  if (point_ok && minus_ok && plus_ok)
  {
    const double window_size = plus_point_axis_value - minus_point_axis_value;
    const double distance_delta = plus - minus;
    *out = distance_delta / window_size;
  }
  else if (point_ok && minus_ok)
  {
    const double window_size = query_point_axis_value - minus_point_axis_value;
    const double distance_delta = point - minus;
    *out = distance_delta / window_size;
  }
  else if (point_ok && plus_ok)
  {
    const double window_size = plus_point_axis_value - query_point_axis_value;
    const double distance_delta = plus - point;
    *out = distance_delta / window_size;
  }
  else
    return 0;
  return 1;
}

/* GetLocationFineGradient, :1050-1091.  Return value: 1 if some query made the reference throw
 * "Window size for fine gradient is too large for SDF" (its entries are NaN / has_value 0). */
int vgt_oracle_fine_gradient(const float* sdf, int64_t nx, int64_t ny, int64_t nz, double resolution,
                             const double* grid_from_world, const double* queries, int64_t num_queries,
                             double nominal_window_size, double* gradient, uint8_t* has_value)
{
  const double ideal_window_size = fabs(nominal_window_size);
  int too_large = 0;
  for (int64_t i = 0; i < num_queries; i++)
  {
    const double x = queries[3 * i], y = queries[3 * i + 1], z = queries[3 * i + 2];
    double g[3] = {NAN, NAN, NAN};
    double point = 0.0;
    int ok = estimate_location_distance(sdf, nx, ny, nz, resolution, grid_from_world, x, y, z, &point);
    if (ok)
    {
      const double min_x = x - ideal_window_size, max_x = x + ideal_window_size;
      const double min_y = y - ideal_window_size, max_y = y + ideal_window_size;
      const double min_z = z - ideal_window_size, max_z = z + ideal_window_size;
      double mx = 0, px = 0, my = 0, py = 0, mz = 0, pz = 0;
      const int mx_ok = estimate_location_distance(sdf, nx, ny, nz, resolution, grid_from_world, min_x, y, z, &mx);
      const int px_ok = estimate_location_distance(sdf, nx, ny, nz, resolution, grid_from_world, max_x, y, z, &px);
      const int my_ok = estimate_location_distance(sdf, nx, ny, nz, resolution, grid_from_world, x, min_y, z, &my);
      const int py_ok = estimate_location_distance(sdf, nx, ny, nz, resolution, grid_from_world, x, max_y, z, &py);
      const int mz_ok = estimate_location_distance(sdf, nx, ny, nz, resolution, grid_from_world, x, y, min_z, &mz);
      const int pz_ok = estimate_location_distance(sdf, nx, ny, nz, resolution, grid_from_world, x, y, max_z, &pz);
      const int fine = axis_fine_gradient(1, point, mx_ok, mx, px_ok, px, x, min_x, max_x, &g[0]) &&
                       axis_fine_gradient(1, point, my_ok, my, py_ok, py, y, min_y, max_y, &g[1]) &&
                       axis_fine_gradient(1, point, mz_ok, mz, pz_ok, pz, z, min_z, max_z, &g[2]);
      if (!fine)
      {
        too_large = 1;
        ok = 0;
        g[0] = g[1] = g[2] = NAN;
      }
    }
    gradient[3 * i] = g[0];
    gradient[3 * i + 1] = g[1];
    gradient[3 * i + 2] = g[2];
    if (has_value) has_value[i] = (uint8_t)ok;
  }
  return too_large;
}

/* ------------------------------------------------------------------------- */
/* SDF consumer: ComputeLocalExtremaMap.                        */
/* ------------------------------------------------------------------------- */

/* GetIndexCoarseGradient(x, y, z, true): the grid-aligned gradient with edge gradients
 * (signed_distance_field.hpp:923-1004) rotated by the origin transform (:906-921; rotation = 9 doubles
 * row-major or NULL). */
static void coarse_gradient_with_edges(const float* sdf, int64_t nx, int64_t ny, int64_t nz, double resolution,
                                       const double* rotation, int64_t x_index, int64_t y_index, int64_t z_index,
                                       double g[3])
{
#define SDF_AT(xi, yi, zi) sdf[((xi) * ny + (yi)) * nz + (zi)]
  double gx = 0.0, gy = 0.0, gz = 0.0;
  if ((x_index > 0) && (y_index > 0) && (z_index > 0) && (x_index < (nx - 1)) && (y_index < (ny - 1)) &&
      (z_index < (nz - 1)))
  {
    const double inv_twice_resolution = 1.0 / (2.0 * resolution);
    const float dx = SDF_AT(x_index + 1, y_index, z_index) - SDF_AT(x_index - 1, y_index, z_index);
    const float dy = SDF_AT(x_index, y_index + 1, z_index) - SDF_AT(x_index, y_index - 1, z_index);
    const float dz = SDF_AT(x_index, y_index, z_index + 1) - SDF_AT(x_index, y_index, z_index - 1);
    gx = dx * inv_twice_resolution;
    gy = dy * inv_twice_resolution;
    gz = dz * inv_twice_resolution;
  }
  else
  {
    const int64_t low_x_index = (x_index - 1 > 0) ? x_index - 1 : 0;
    const int64_t high_x_index = (x_index + 1 < nx - 1) ? x_index + 1 : nx - 1;
    const int64_t low_y_index = (y_index - 1 > 0) ? y_index - 1 : 0;
    const int64_t high_y_index = (y_index + 1 < ny - 1) ? y_index + 1 : ny - 1;
    const int64_t low_z_index = (z_index - 1 > 0) ? z_index - 1 : 0;
    const int64_t high_z_index = (z_index + 1 < nz - 1) ? z_index + 1 : nz - 1;
    const double x_increment = (double)(high_x_index - low_x_index) * resolution;
    const double y_increment = (double)(high_y_index - low_y_index) * resolution;
    const double z_increment = (double)(high_z_index - low_z_index) * resolution;
    if (x_increment > 0.0)
      gx = ((double)SDF_AT(high_x_index, y_index, z_index) - (double)SDF_AT(low_x_index, y_index, z_index)) *
           (1.0 / x_increment);
    if (y_increment > 0.0)
      gy = ((double)SDF_AT(x_index, high_y_index, z_index) - (double)SDF_AT(x_index, low_y_index, z_index)) *
           (1.0 / y_increment);
    if (z_increment > 0.0)
      gz = ((double)SDF_AT(x_index, y_index, high_z_index) - (double)SDF_AT(x_index, y_index, low_z_index)) *
           (1.0 / z_increment);
  }
#undef SDF_AT
  if (rotation)
  {
    const double wx = rotation[0] * gx + rotation[1] * gy + rotation[2] * gz;
    const double wy = rotation[3] * gx + rotation[4] * gy + rotation[5] * gz;
    const double wz = rotation[6] * gx + rotation[7] * gy + rotation[8] * gz;
    gx = wx;
    gy = wy;
    gz = wz;
  }
  g[0] = gx;
  g[1] = gy;
  g[2] = gz;
}

/* GradientIsEffectiveFlat, :482-497 */
static int gradient_is_effective_flat(const double g[3], double resolution)
{
  const double step_resolution = resolution * 0.06125;
  return (fabs(g[0]) <= step_resolution && fabs(g[1]) <= step_resolution && fabs(g[2]) <= step_resolution);
}

/* GetNextFromGradient, :499-541 */
static void next_from_gradient(const float* sdf, int64_t ny, int64_t nz, double resolution, const int64_t index[3],
                               const double gradient[3], int64_t next_index[3])
{
  const float stored_distance = sdf[(index[0] * ny + index[1]) * nz + index[2]];
  double working_gradient[3] = {gradient[0], gradient[1], gradient[2]};
  if (stored_distance < 0.0)
  {
    for (int a = 0; a < 3; a++) working_gradient[a] = gradient[a] * -1.0;
  }
  const double step_resolution = resolution * 0.06125;
  for (int a = 0; a < 3; a++)
  {
    next_index[a] = index[a];
    if (working_gradient[a] > step_resolution)
      next_index[a] += 1;
    else if (working_gradient[a] < -step_resolution)
      next_index[a] -= 1;
  }
}

/* ComputeLocalExtremaMap (:1205-1231) with FollowGradientsToLocalExtremaUnsafe (:385-480), literally: cells are
 * visited in X-major order, a walk stops at a flat cell, off the grid, at a cell already stored, or at a cell of
 * its own path, and every cell of the path receives the result.  extrema: 3 doubles per cell, the default value
 * (-inf x 3) never survives. */
void vgt_oracle_local_extrema_map(const float* sdf, int64_t nx, int64_t ny, int64_t nz, double resolution,
                                  const double* rotation, double* extrema)
{
  const int64_t total = nx * ny * nz;
  for (int64_t i = 0; i < 3 * total; i++) extrema[i] = -INFINITY;
  int64_t* path = (int64_t*)malloc((size_t)(total + 1) * sizeof(int64_t));
  int64_t* stamp = (int64_t*)calloc((size_t)total, sizeof(int64_t)); /* walk id that last put the cell on its path */
  int64_t walk = 0;
  for (int64_t x_idx = 0; x_idx < nx; x_idx++)
    for (int64_t y_idx = 0; y_idx < ny; y_idx++)
      for (int64_t z_idx = 0; z_idx < nz; z_idx++)
      {
        const int64_t start = (x_idx * ny + y_idx) * nz + z_idx;
        if (extrema[3 * start] != -INFINITY && extrema[3 * start + 1] != -INFINITY && extrema[3 * start + 2] != -INFINITY)
          continue; /* already found for this cell */
        double gradient_vector[3];
        coarse_gradient_with_edges(sdf, nx, ny, nz, resolution, rotation, x_idx, y_idx, z_idx, gradient_vector);
        if (gradient_is_effective_flat(gradient_vector, resolution))
        {
          extrema[3 * start] = ((double)x_idx + 0.5) * resolution;
          extrema[3 * start + 1] = ((double)y_idx + 0.5) * resolution;
          extrema[3 * start + 2] = ((double)z_idx + 0.5) * resolution;
          continue;
        }
        walk++;
        int64_t path_length = 0;
        int64_t current_index[3] = {x_idx, y_idx, z_idx};
        path[path_length++] = start;
        stamp[start] = walk;
        double local_extrema[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (;;)
        {
          int64_t next_index[3];
          next_from_gradient(sdf, ny, nz, resolution, current_index, gradient_vector, next_index);
          for (int a = 0; a < 3; a++) current_index[a] = next_index[a];
          const int in_bounds = current_index[0] >= 0 && current_index[0] < nx && current_index[1] >= 0 &&
                                current_index[1] < ny && current_index[2] >= 0 && current_index[2] < nz;
          const int64_t current = in_bounds ? (current_index[0] * ny + current_index[1]) * nz + current_index[2] : -1;
          if (in_bounds && stamp[current] == walk)
          {
            /* we have been here on this walk: done */
            for (int a = 0; a < 3; a++) local_extrema[a] = ((double)current_index[a] + 0.5) * resolution;
            break;
          }
          if (!in_bounds)
          {
            for (int a = 0; a < 3; a++) local_extrema[a] = INFINITY;
            break;
          }
          path[path_length++] = current;
          stamp[current] = walk;
          if (extrema[3 * current] != -INFINITY && extrema[3 * current + 1] != -INFINITY &&
              extrema[3 * current + 2] != -INFINITY)
          {
            for (int a = 0; a < 3; a++) local_extrema[a] = extrema[3 * current + a];
            break;
          }
          coarse_gradient_with_edges(sdf, nx, ny, nz, resolution, rotation, current_index[0], current_index[1],
                                     current_index[2], gradient_vector);
          if (gradient_is_effective_flat(gradient_vector, resolution))
          {
            for (int a = 0; a < 3; a++) local_extrema[a] = ((double)current_index[a] + 0.5) * resolution;
            break;
          }
        }
        for (int64_t k = 0; k < path_length; k++)
          for (int a = 0; a < 3; a++) extrema[3 * path[k] + a] = local_extrema[a];
      }
  free(path);
  free(stamp);
}

/* ------------------------------------------------------------------------- */
/* SDF consumer: coarse gradient.                               */
/* ------------------------------------------------------------------------- */

void vgt_oracle_coarse_gradient(const float* sdf, int64_t nx, int64_t ny, int64_t nz,
                                double resolution, int enable_edge_gradients, double* gradient,
                                uint8_t* has_value)
{
#define SDF_AT(xi, yi, zi) sdf[((xi) * ny + (yi)) * nz + (zi)]
  for (int64_t x_index = 0; x_index < nx; x_index++)
    for (int64_t y_index = 0; y_index < ny; y_index++)
      for (int64_t z_index = 0; z_index < nz; z_index++)
      {
        const int64_t i = (x_index * ny + y_index) * nz + z_index;
        double gx = NAN, gy = NAN, gz = NAN;
        int ok = 1;
        /* signed_distance_field.hpp:933-950 */
        if ((x_index > 0) && (y_index > 0) && (z_index > 0) && (x_index < (nx - 1)) &&
            (y_index < (ny - 1)) && (z_index < (nz - 1)))
        {
          const double inv_twice_resolution = 1.0 / (2.0 * resolution);
          /* the operands are floats: the difference is a float, the product a double */
          const float dx = SDF_AT(x_index + 1, y_index, z_index) - SDF_AT(x_index - 1, y_index, z_index);
          const float dy = SDF_AT(x_index, y_index + 1, z_index) - SDF_AT(x_index, y_index - 1, z_index);
          const float dz = SDF_AT(x_index, y_index, z_index + 1) - SDF_AT(x_index, y_index, z_index - 1);
          gx = dx * inv_twice_resolution;
          gy = dy * inv_twice_resolution;
          gz = dz * inv_twice_resolution;
        }
        else if (enable_edge_gradients) /* :955-1004 */
        {
          const int64_t low_x_index = (x_index - 1 > 0) ? x_index - 1 : 0;
          const int64_t high_x_index = (x_index + 1 < nx - 1) ? x_index + 1 : nx - 1;
          const int64_t low_y_index = (y_index - 1 > 0) ? y_index - 1 : 0;
          const int64_t high_y_index = (y_index + 1 < ny - 1) ? y_index + 1 : ny - 1;
          const int64_t low_z_index = (z_index - 1 > 0) ? z_index - 1 : 0;
          const int64_t high_z_index = (z_index + 1 < nz - 1) ? z_index + 1 : nz - 1;
          const double x_increment = (double)(high_x_index - low_x_index) * resolution;
          const double y_increment = (double)(high_y_index - low_y_index) * resolution;
          const double z_increment = (double)(high_z_index - low_z_index) * resolution;
          gx = 0.0;
          gy = 0.0;
          gz = 0.0;
          if (x_increment > 0.0)
          {
            const double inv_x_increment = 1.0 / x_increment;
            const double high_x_value = SDF_AT(high_x_index, y_index, z_index);
            const double low_x_value = SDF_AT(low_x_index, y_index, z_index);
            gx = (high_x_value - low_x_value) * inv_x_increment;
          }
          if (y_increment > 0.0)
          {
            const double inv_y_increment = 1.0 / y_increment;
            const double high_y_value = SDF_AT(x_index, high_y_index, z_index);
            const double low_y_value = SDF_AT(x_index, low_y_index, z_index);
            gy = (high_y_value - low_y_value) * inv_y_increment;
          }
          if (z_increment > 0.0)
          {
            const double inv_z_increment = 1.0 / z_increment;
            const double high_z_value = SDF_AT(x_index, y_index, high_z_index);
            const double low_z_value = SDF_AT(x_index, y_index, low_z_index);
            gz = (high_z_value - low_z_value) * inv_z_increment;
          }
        }
        else
        {
          ok = 0; /* :1006-1010 empty GradientQuery */
        }
        gradient[3 * i + 0] = gx;
        gradient[3 * i + 1] = gy;
        gradient[3 * i + 2] = gz;
        if (has_value) has_value[i] = (uint8_t)ok;
      }
#undef SDF_AT
}

/* ------------------------------------------------------------------------- */

/* float -> int32 as the DEVICE kernels' cast behaves (cuda_voxelization_helpers.cu:140-144, :229-240 run as CUDA
 * cvt.rzi.s32.f32, and as v_cvt_i32_f32 on AMD): NaN -> 0, out of range -> saturated.  x86's cvttss2si would answer
 * INT32_MIN to all of those; the difference shows on a ray of length zero seen from outside the grid (direction 0 / 0,
 * entry point NaN): the device kernels start it in voxel (0, 0, 0). */
static inline int32_t device_index_f32(float x)
{
  if (isnan(x)) return 0;
  if (x >= 2147483648.0f) return INT32_MAX;
  if (x <= -2147483648.0f) return INT32_MIN;
  return (int32_t)x;
}
/* double -> int64 as the CPU voxelizer's cast behaves on x86-64 (cvttsd2si; cpu_pointcloud_voxelization.cpp:107, :181,
 * :294-297 through LocationInGridFrameToGridIndex4d): NaN and out of range -> INT64_MIN, made explicit so that the
 * oracle says the same on any host. */
static inline int64_t host_index_f64(double x)
{
  if (!(x > -9223372036854775808.0 && x < 9223372036854775808.0)) return INT64_MIN;
  return (int64_t)x;
}

/* (the step of an axis, S/cuda...cu:35-50: the sign of final - start index, taken inline below) */

/* S/cuda_voxelization_helpers.cu:52-71 */
static float axis_t_f32(float point_axis, float ray_axis, float vmin,
                        float vmax)
{
  if (ray_axis > 0.0f) return fabsf((vmax - point_axis) / ray_axis);
  if (ray_axis < -0.0f) return fabsf((point_axis - vmin) / ray_axis);
  return INFINITY;
}

static void atomic_inc_i32(int32_t* p)
{
#pragma omp atomic
  (*p)++;
}

/* S/cuda_voxelization_helpers.cu:73-356, one point. */
static void raycast_one_f32(const float* pt, float max_range, const float* T,
                            float vs, float ivs, const float gs[3], int32_t nx,
                            int32_t ny, int32_t nz, int32_t* grid)
{
  const float px = pt[0], py = pt[1], pz = pt[2];
  if (!isfinite(px) || !isfinite(py) || !isfinite(pz)) return; /* :97-100 */

  /* :103-114 */
  const float gx = T[0] * px + T[4] * py + T[8] * pz + T[12];
  const float gy = T[1] * px + T[5] * py + T[9] * pz + T[13];
  const float gz = T[2] * px + T[6] * py + T[10] * pz + T[14];
  /* :117-119 */
  const float o[3] = {T[12], T[13], T[14]};
  /* :122-136 */
  const float ray[3] = {gx - o[0], gy - o[1], gz - o[2]};
  const float len = sqrtf(ray[0] * ray[0] + ray[1] * ray[1] + ray[2] * ray[2]);
  const int clipped = len > max_range;
  float fin[3] = {gx, gy, gz};
  if (clipped)
  {
    fin[0] = o[0] + (ray[0] * (max_range / len));
    fin[1] = o[1] + (ray[1] * (max_range / len));
    fin[2] = o[2] + (ray[2] * (max_range / len));
  }
  /* :139-149 */
  const int32_t oi[3] = {device_index_f32(floorf(o[0] * ivs)), device_index_f32(floorf(o[1] * ivs)),
                         device_index_f32(floorf(o[2] * ivs))};
  const int32_t dims[3] = {nx, ny, nz};
  const int origin_in_grid = oi[0] >= 0 && oi[0] < nx && oi[1] >= 0 &&
                             oi[1] < ny && oi[2] >= 0 && oi[2] < nz;
  float start[3] = {o[0], o[1], o[2]};
  if (!origin_in_grid) /* :154-225 */
  {
    float tmin = 0.0f, tmax = max_range;
    const float dir[3] = {ray[0] / len, ray[1] / len, ray[2] / len};
    const float flat_threshold = 1e-10f;
    for (int a = 0; a < 3; a++)
    {
      if (fabsf(dir[a]) < flat_threshold)
      {
        const int in_slab = o[a] >= 0.0f && o[a] < gs[a];
        if (!in_slab) return;
      }
      else
      {
        const float ood = 1.0f / dir[a];
        const float tlow = (0.0f - o[a]) * ood;
        const float thigh = (gs[a] - o[a]) * ood;
        const float t1 = (tlow <= thigh) ? tlow : thigh;
        const float t2 = (tlow <= thigh) ? thigh : tlow;
        if (t1 > tmin) tmin = t1;
        if (t2 > tmax) tmax = t2; /* sic: reference quirk, :206-209 */
        if (tmin > tmax) return;
      }
    }
    const float nudge = 1e-10f;
    start[0] = o[0] + (dir[0] * (tmin + nudge));
    start[1] = o[1] + (dir[1] * (tmin + nudge));
    start[2] = o[2] + (dir[2] * (tmin + nudge));
  }
  /* :228-245 */
  int32_t si[3], fi[3], step[3];
  for (int a = 0; a < 3; a++)
  {
    si[a] = device_index_f32(floorf(start[a] * ivs));
    fi[a] = device_index_f32(floorf(fin[a] * ivs));
    /* (the difference in 64 bits: saturated indices must not overflow it) */
    const int64_t diff = (int64_t)fi[a] - (int64_t)si[a];
    step[a] = (diff > 0) ? 1 : ((diff < 0) ? -1 : 0);
  }
  /* :248-274 */
  const float half = vs * 0.5f;
  float t[3], dt[3];
  for (int a = 0; a < 3; a++)
  {
    const float centre = ((float)si[a] + 0.5f) * vs;
    const float lo = centre - half, hi = centre + half;
    t[a] = axis_t_f32(start[a], ray[a], lo, hi);
    dt[a] = fabsf(vs / ray[a]);
  }
  const int32_t stride1 = ny * nz, stride2 = nz; /* :683-684 */
  /* :277-293 -- the final voxel is marked first */
  if (fi[0] >= 0 && fi[0] < nx && fi[1] >= 0 && fi[1] < ny && fi[2] >= 0 &&
      fi[2] < nz)
  {
    const int32_t di = (fi[0] * stride1) + (fi[1] * stride2) + fi[2];
    atomic_inc_i32(&grid[(di * 2) + (clipped ? 0 : 1)]);
  }
  /* :295-355 */
  int32_t c[3] = {si[0], si[1], si[2]};
  while (c[0] != fi[0] || c[1] != fi[1] || c[2] != fi[2])
  {
    if (c[0] >= 0 && c[0] < dims[0] && c[1] >= 0 && c[1] < dims[1] &&
        c[2] >= 0 && c[2] < dims[2])
    {
      const int32_t di = (c[0] * stride1) + (c[1] * stride2) + c[2];
      atomic_inc_i32(&grid[(di * 2) + 0]);
    }
    else
      break;
    int a;
    if (t[0] <= t[1] && t[0] <= t[2])
      a = 0;
    else if (t[1] <= t[0] && t[1] <= t[2])
      a = 1;
    else
      a = 2;
    if (c[a] == fi[a]) break;
    c[a] += step[a];
    t[a] += dt[a];
  }
}

void vgt_oracle_raycast_f32(const float* points, int64_t num_points,
                            float max_range, const float* xform,
                            float voxel_size, float inverse_voxel_size,
                            float grid_x_size, float grid_y_size,
                            float grid_z_size, int32_t nx, int32_t ny,
                            int32_t nz, int32_t* tracking, int threads)
{
  threads = resolve_threads(threads);
  const float gs[3] = {grid_x_size, grid_y_size, grid_z_size};
#pragma omp parallel for schedule(static) num_threads(threads)
  for (int64_t i = 0; i < num_points; i++)
    raycast_one_f32(points + 3 * i, max_range, xform, voxel_size,
                    inverse_voxel_size, gs, nx, ny, nz, tracking);
}

/* ------------------------------------------------------------------------- */
/* Raycast DDA -- float64 (CPU path restatement)                             */
/* ------------------------------------------------------------------------- */

static double axis_t_f64(double point_axis, double ray_axis, double vmin,
                         double vmax) /* S/cpu...cpp:336-353 */
{
  if (ray_axis > 0.0) return fabs((vmax - point_axis) / ray_axis);
  if (ray_axis < -0.0) return fabs((point_axis - vmin) / ray_axis);
  return INFINITY;
}

/* S/cpu_pointcloud_voxelization.cpp:208-436, one point already in grid frame
 * (o = p_GCo, g = p_GP). */
static void raycast_one_f64(const double o[3], const int64_t oi[3],
                            const double g[3], double max_range, double vs,
                            double ivs, const double gs[3], int64_t nx,
                            int64_t ny, int64_t nz, int32_t* grid)
{
  const int64_t dims[3] = {nx, ny, nz};
  /* :217-228 (Vector4d with w = 0: norm == sqrt(x^2+y^2+z^2+0)) */
  const double ray[3] = {g[0] - o[0], g[1] - o[1], g[2] - o[2]};
  const double len =
      sqrt(ray[0] * ray[0] + ray[1] * ray[1] + ray[2] * ray[2] + 0.0);
  const int clipped = len > max_range;
  double fin[3] = {g[0], g[1], g[2]};
  if (clipped)
  {
    const double scale = max_range / len;
    fin[0] = o[0] + (ray[0] * scale);
    fin[1] = o[1] + (ray[1] * scale);
    fin[2] = o[2] + (ray[2] * scale);
  }
  /* :231-290 */
  const int origin_in_grid = oi[0] >= 0 && oi[0] < nx && oi[1] >= 0 &&
                             oi[1] < ny && oi[2] >= 0 && oi[2] < nz;
  double start[3] = {o[0], o[1], o[2]};
  if (!origin_in_grid)
  {
    double tmin = 0.0, tmax = max_range;
    const double dir[3] = {ray[0] / len, ray[1] / len, ray[2] / len};
    const double flat_threshold = 1e-10;
    for (int a = 0; a < 3; a++)
    {
      if (fabs(dir[a]) < flat_threshold)
      {
        const int in_slab = o[a] >= 0.0 && o[a] < gs[a];
        if (!in_slab) return;
      }
      else
      {
        const double ood = 1.0 / dir[a];
        const double tlow = (0.0 - o[a]) * ood;
        const double thigh = (gs[a] - o[a]) * ood;
        const double t1 = (tlow <= thigh) ? tlow : thigh;
        const double t2 = (tlow <= thigh) ? thigh : tlow;
        if (t1 > tmin) tmin = t1;
        if (t2 > tmax) tmax = t2; /* sic, :274-277 */
        if (tmin > tmax) return;
      }
    }
    const double nudge = 1e-10;
    start[0] = o[0] + (dir[0] * (tmin + nudge));
    start[1] = o[1] + (dir[1] * (tmin + nudge));
    start[2] = o[2] + (dir[2] * (tmin + nudge));
  }
  /* :293-321 */
  int64_t si[3], fi[3], step[3];
  for (int a = 0; a < 3; a++)
  {
    si[a] = host_index_f64(floor(start[a] * ivs));
    fi[a] = host_index_f64(floor(fin[a] * ivs));
    /* (the sign of final - start by comparison: with an "indefinite" index the subtraction overflows, and a ray that
     * has one never walks) */
    step[a] = (fi[a] > si[a]) ? 1 : ((fi[a] < si[a]) ? -1 : 0);
  }
  /* :324-364 */
  const double half = vs * 0.5;
  double t[3], dt[3];
  for (int a = 0; a < 3; a++)
  {
    const double centre = ((double)si[a] + 0.5) * vs;
    const double lo = centre - half, hi = centre + half;
    t[a] = axis_t_f64(start[a], ray[a], lo, hi);
    dt[a] = fabs(vs / ray[a]);
  }
  const int64_t stride1 = ny * nz, stride2 = nz;
  /* :367-381 */
  if (fi[0] >= 0 && fi[0] < nx && fi[1] >= 0 && fi[1] < ny && fi[2] >= 0 &&
      fi[2] < nz)
  {
    const int64_t di = (fi[0] * stride1) + (fi[1] * stride2) + fi[2];
    atomic_inc_i32(&grid[(di * 2) + (clipped ? 0 : 1)]);
  }
  /* :384-435 */
  int64_t c[3] = {si[0], si[1], si[2]};
  while (c[0] != fi[0] || c[1] != fi[1] || c[2] != fi[2])
  {
    if (c[0] >= 0 && c[0] < dims[0] && c[1] >= 0 && c[1] < dims[1] &&
        c[2] >= 0 && c[2] < dims[2])
    {
      const int64_t di = (c[0] * stride1) + (c[1] * stride2) + c[2];
      atomic_inc_i32(&grid[(di * 2) + 0]);
    }
    else
      break;
    int a;
    if (t[0] <= t[1] && t[0] <= t[2])
      a = 0;
    else if (t[1] <= t[0] && t[1] <= t[2])
      a = 1;
    else
      a = 2;
    if (c[a] == fi[a]) break;
    c[a] += step[a];
    t[a] += dt[a];
  }
}

/* S/cpu_pointcloud_voxelization.cpp:167-206 -- per-cloud driver.  xform is
 * X_GC (16 doubles, column-major); p_GP = X_GC * p_CP evaluated per row as
 * ((m0*x + m4*y) + m8*z) + m12, the canonical scalar order. */
void vgt_oracle_raycast_f64(const double* points, int64_t num_points,
                            double max_range, const double* T,
                            double voxel_size, double inverse_voxel_size,
                            double grid_x_size, double grid_y_size,
                            double grid_z_size, int64_t nx, int64_t ny,
                            int64_t nz, int32_t* tracking, int threads)
{
  threads = resolve_threads(threads);
  const double gs[3] = {grid_x_size, grid_y_size, grid_z_size};
  const double o[3] = {T[12], T[13], T[14]}; /* :178 */
  const int64_t oi[3] = {host_index_f64(floor(o[0] * inverse_voxel_size)),
                         host_index_f64(floor(o[1] * inverse_voxel_size)),
                         host_index_f64(floor(o[2] * inverse_voxel_size))}; /* :180 */
#pragma omp parallel for schedule(static) num_threads(threads)
  for (int64_t i = 0; i < num_points; i++)
  {
    const double px = points[3 * i + 0], py = points[3 * i + 1],
                 pz = points[3 * i + 2];
    if (!(isfinite(px) && isfinite(py) && isfinite(pz))) continue; /* :191 */
    const double g[3] = {T[0] * px + T[4] * py + T[8] * pz + T[12],
                         T[1] * px + T[5] * py + T[9] * pz + T[13],
                         T[2] * px + T[6] * py + T[10] * pz + T[14]};
    raycast_one_f64(o, oi, g, max_range, voxel_size, inverse_voxel_size, gs,
                    nx, ny, nz, tracking);
  }
}

/* ------------------------------------------------------------------------- */
/* Combine + filter                                                          */
/* ------------------------------------------------------------------------- */

void vgt_oracle_filter(const int32_t* tracking, int64_t num_cells,
                       int32_t num_grids, double percent_seen_free,
                       int32_t outlier_points_threshold,
                       int32_t num_cameras_seen_free, int ratio_in_double,
                       float* occupancy, int threads)
{
  threads = resolve_threads(threads);
  const float pct_f32 = (float)percent_seen_free; /* S/device...cpp:155-156 */
#pragma omp parallel for schedule(static) num_threads(threads)
  for (int64_t cell = 0; cell < num_cells; cell++)
  {
    /* S/cuda...cu:368-371 / S/cpu...cpp:451-453: filled cells stay filled */
    if (!(occupancy[cell] <= 0.5f)) continue;
    int32_t seen_filled = 0, seen_free = 0;
    for (int32_t g = 0; g < num_grids; g++)
    {
      const int32_t* tc = tracking + ((int64_t)g * num_cells + cell) * 2;
      const int32_t free_count = tc[0];
      const int32_t filled_count = tc[1];
      const int32_t filtered_filled =
          (filled_count >= outlier_points_threshold) ? filled_count : 0;
      if (free_count > 0 && filtered_filled > 0)
      {
        int is_free;
        if (ratio_in_double) /* I/pointcloud_voxelization_interface.hpp:60-73 */
          is_free = ((double)free_count /
                     (double)(free_count + filtered_filled)) >=
                    percent_seen_free;
        else /* S/cuda...cu:386-399 */
          is_free = ((float)free_count /
                     (float)(free_count + filtered_filled)) >= pct_f32;
        if (is_free)
          seen_free += 1;
        else
          seen_filled += 1;
      }
      else if (free_count > 0)
        seen_free += 1;
      else if (filtered_filled > 0)
        seen_filled += 1;
    }
    if (seen_filled > 0)
      occupancy[cell] = 1.0f;
    else if (seen_free >= num_cameras_seen_free)
      occupancy[cell] = 0.0f;
    else
      occupancy[cell] = 0.5f;
  }
}
