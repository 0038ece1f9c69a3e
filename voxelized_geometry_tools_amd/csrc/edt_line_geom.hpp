// Geometry of a line pass (Y or X) of the EDT, shared by the lane-per-line sweep kernels (edt_sweep_kernels.hip) and the
// short-line kernels (edt_short_kernels.hip): one wave owns 64 neighbouring Z positions of one outer index, a lane one
// line of the pass axis.
#pragma once

#include "vgt_internal.hpp"

namespace vgt
{
struct SweepGeom
{
  int n;                 // rows along the pass axis
  int nz;                // extent of the contiguous axis
  int zsegs;             // waves per outer index
  int items;             // outer indices x zsegs: units of work, dealt to the workgroups through counters
  int outers;            // outer indices
  int groups;            // workgroup b draws from counter b % groups, which deals the outer indices = b (mod groups)
  int nwords;            // ceil(n / 32)
  int chunks;            // spill chunks per lane
  int64_t row_stride;    // elements between consecutive rows
  int64_t outer_stride;  // elements between consecutive outer indices
  int nx, ny;
  int pass_axis;         // 0 = X pass (outer = y), 1 = Y pass (outer = x)
  double resolution;
  int add_virtual_border;
  int z_offset, nz_global;
  int outer_begin;
  // Batches of equal grids (X pass; the Y pass sees a batch as one grid of batch x nx slices): outer index o belongs to
  // grid o / batch_outers, whose lines start batch_skip elements further on per grid than outer_stride alone says, and
  // whose extrema go to minmax_enc[2 * grid].  One grid: batch_outers = outers (grid 0 for every item), batch_skip = 0.
  int batch_outers;
  int64_t batch_skip;
};

inline SweepGeom SweepGeometry(const SdfParams& p, int axis, int64_t* outer_count)
{
  SweepGeom g{};
  g.nz = static_cast<int>(p.nz);
  g.nx = static_cast<int>(p.nx);
  g.ny = static_cast<int>(p.ny);
  g.pass_axis = axis;
  if (axis == 0)
  {
    g.n = static_cast<int>(p.nx);
    g.row_stride = p.ny * p.nz;
    g.outer_stride = p.nz;
    *outer_count = p.ny;
  }
  else
  {
    g.n = static_cast<int>(p.ny);
    g.row_stride = p.nz;
    g.outer_stride = p.ny * p.nz;
    *outer_count = p.nx;
  }
  g.resolution = p.resolution;
  g.add_virtual_border = p.add_virtual_border;
  g.z_offset = static_cast<int>(p.z_offset);
  g.nz_global = static_cast<int>(p.nz_global > 0 ? p.nz_global : p.nz);
  return g;
}
}  // namespace vgt
