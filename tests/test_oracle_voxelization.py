"""The CPU oracle's DDA + filter against the predicates of the reference's own tests.

(not gpu).  Reference-pinned: the cell predicates of
test/pointcloud_voxelization_test.cpp:84-158 and the count invariants of
test/voxel_raycasting_test.cpp:57-82.  Tracking counts themselves are not pinned by any
reference test -- they are compared per precision against this oracle.
"""
import numpy as np
import pytest

from conftest import check_empty_voxelization, check_voxelization, scene_clouds
from oracle import oracle as O


def _voxelize(scene, precision, ratio_in_double):
    occ = scene["static_occupancy"]
    vs = float(scene["voxel_size"])
    counts = occ.shape
    grids = []
    if precision == "f32":
        vs32 = np.float32(vs)
        ivs32 = np.float32(1.0 / vs)
        sizes = [np.float32(c * vs) for c in counts]
        for pts, xf in scene_clouds(scene, np.float32):
            grids.append(O.raycast_f32(pts, np.float32(np.inf), xf, vs32, ivs32, sizes, counts))
    else:
        sizes = [c * vs for c in counts]
        for pts, xf in scene_clouds(scene, np.float64):
            grids.append(O.raycast_f64(pts, np.inf, xf, vs, 1.0 / vs, sizes, counts))
    grids.append(np.zeros_like(grids[0]))  # third, empty cloud (:230-235)
    pct, outlier, ncam = scene["filter"]
    out = O.filter_grids(np.stack(grids), occ, pct, int(outlier), int(ncam), ratio_in_double)
    return out, grids


@pytest.mark.parametrize("precision,ratio_in_double", [("f32", False), ("f64", True)])
def test_reference_scene_predicates(voxelization_scene, precision, ratio_in_double):
    out, grids = _voxelize(voxelization_scene, precision, ratio_in_double)
    check_voxelization(out)
    for g in grids[:2]:
        assert g[..., 1].sum() == 2048  # every unclipped ray ends in exactly one in-grid voxel


def test_reference_scene_empty(voxelization_scene):
    """No clouds: one zeroed tracking grid is still filtered (device_pointcloud_voxelization.cpp:79-80)."""
    occ = voxelization_scene["static_occupancy"]
    grids = np.zeros((1,) + occ.shape + (2,), dtype=np.int32)
    check_empty_voxelization(O.filter_grids(grids, occ, 1.0, 1, 1, False))
    check_empty_voxelization(O.filter_grids(grids, occ, 1.0, 1, 1, True))


@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_raycasting_invariants(raycast_rays, precision):
    """test/voxel_raycasting_test.cpp: 40^3 @0.125, max_range 10, one ray at a time:
    every cell's counts are in {0,1} and never both set."""
    counts = (40, 40, 40)
    vs = 0.125
    for ray in raycast_rays:
        origin, point = ray[:3], ray[3:]
        xf = np.eye(4)
        xf[:3, 3] = origin
        xf = xf.T.reshape(16)
        local = (point - origin).reshape(1, 3)  # cloud frame = grid frame shifted to the origin
        if precision == "f64":
            g = O.raycast_f64(local, 10.0, xf, vs, 1.0 / vs, [5.0, 5.0, 5.0], counts)
        else:
            g = O.raycast_f32(local, 10.0, xf, vs, 1.0 / vs, [5.0, 5.0, 5.0], counts)
        assert g.min() >= 0 and g.max() <= 1
        assert not np.any((g[..., 0] > 0) & (g[..., 1] > 0))


def test_filter_rules():
    """CountsSeenAs table (pointcloud_voxelization_interface.hpp:55-86) on hand-made counts."""
    # cells: 0 free only, 1 filled only, 2 mixed 3:1 (75 % free), 3 nothing, 4 already filled
    tracking = np.zeros((1, 5, 1, 1, 2), dtype=np.int32)
    tracking[0, 0, 0, 0] = (5, 0)
    tracking[0, 1, 0, 0] = (0, 2)
    tracking[0, 2, 0, 0] = (3, 1)
    tracking[0, 4, 0, 0] = (9, 0)
    occ = np.array([0.0, 0.0, 0.5, 0.0, 1.0], dtype=np.float32).reshape(5, 1, 1)
    for dbl in (False, True):
        out = O.filter_grids(tracking, occ, 1.0, 1, 1, dbl).ravel()
        assert list(out) == [0.0, 1.0, 1.0, 0.5, 1.0]
        out = O.filter_grids(tracking, occ, 0.75, 1, 1, dbl).ravel()
        assert list(out) == [0.0, 1.0, 0.0, 0.5, 1.0]
        out = O.filter_grids(tracking, occ, 1.0, 3, 1, dbl).ravel()  # outlier threshold hides the 1-2 hits
        assert list(out) == [0.0, 0.5, 0.0, 0.5, 1.0]
        out = O.filter_grids(tracking, occ, 1.0, 1, 2, dbl).ravel()  # needs two cameras to call it free
        assert list(out) == [0.5, 1.0, 1.0, 0.5, 1.0]


def _zero_length_ray_scene():
    from voxelized_geometry_tools_amd import synthetic
    counts = (20, 6, 5)
    vs = np.float32(0.1)
    pts = np.zeros((3, 3), dtype=np.float32)       # three points AT the sensor
    xf = synthetic.translation_xform(-0.55, 0.25, 0.15)   # the sensor beside the grid
    return counts, vs, pts, xf


def test_zero_length_ray_outside_the_grid_follows_the_cast_of_where_it_runs():
    """A point at the sensor, the sensor outside the grid: direction 0 / 0, entry point NaN, and the voxel index is what
    the cast of NaN gives WHERE THE REFERENCE RUNS.  The float walk is the device kernels' (cuda_voxelization_helpers.cu:
    229-240): the device cast gives 0, the ray starts in voxel (0, 0, 0) and -- all its boundary times infinite -- walks
    along +x / -x towards the sensor's (off-grid) voxel until it leaves the grid.  The double walk is the CPU
    voxelizer's (cpu_pointcloud_voxelization.cpp:294-297): x86's cast gives the most negative integer, never in the
    grid, and the ray is dropped.  The oracle states both explicitly (device_index_f32, host_index_f64)."""
    counts, vs, pts, xf = _zero_length_ray_scene()
    sizes = [np.float32(c) * vs for c in counts]
    got = O.raycast_f32(pts, 3.0, xf.astype(np.float32), vs, np.float32(1.0) / vs, sizes, counts)
    # sensor voxel x = floor(-0.55 / 0.1) = -6 < 0: step -1 from voxel (0, 0, 0): one visit, then out of the grid
    assert got[0, 0, 0, 0] == 3 and got.sum() == 3
    sizes64 = [float(c) * float(vs) for c in counts]
    got64 = O.raycast_f64(pts.astype(np.float64), 3.0, xf, float(vs), 1.0 / float(vs), sizes64, counts)
    assert got64.sum() == 0
