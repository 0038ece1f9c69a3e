"""ctypes front-end for the CPU oracle (oracle/vgt_oracle.c).

TEST INFRASTRUCTURE ONLY: the checker for tests/, __graft_entry__.smoke() and
bench.py's ``cpu_baseline`` leg.  The product package never imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_i64 = ctypes.c_int64
_i32 = ctypes.c_int32
_f32 = ctypes.c_float
_f64 = ctypes.c_double
_p = ctypes.c_void_p


def build(march="x86-64-v3", out=None):
    """Compile the oracle with the committed Makefile; returns the .so path."""
    out = out or os.path.join(_HERE, "libvgt_oracle.so")
    subprocess.check_call(
        ["make", "-s", "-C", _HERE, "MARCH=" + march, "OUT=" + out],
        stdout=subprocess.DEVNULL)
    return out


def load(path=None):
    global _LIB
    if path is None and _LIB is not None:
        return _LIB
    path = path or os.environ.get("VGT_ORACLE_LIB")  # (a sanitizer build: tests/test_sanitizers.py)
    so = path or os.path.join(_HERE, "libvgt_oracle.so")
    src = os.path.join(_HERE, "vgt_oracle.c")
    if path is None and (not os.path.exists(so)
                         or os.path.getmtime(so) < os.path.getmtime(src)):
        build()
    lib = ctypes.CDLL(so)
    lib.vgt_oracle_max_threads.restype = ctypes.c_int
    lib.vgt_oracle_edt3d_inplace.argtypes = [_p, _i64, _i64, _i64, ctypes.c_int]
    lib.vgt_oracle_edt1d_inplace.argtypes = [_p, _i64, _i64]
    lib.vgt_oracle_sdf_from_mask.argtypes = [_p, _i64, _i64, _i64, _f64, _p, ctypes.c_int]
    lib.vgt_oracle_sdf_from_mask.restype = ctypes.c_int
    lib.vgt_oracle_sdf_from_occupancy.argtypes = [
        _p, _i64, _i64, _i64, _f64, ctypes.c_int, ctypes.c_int, _p, _p, _p, ctypes.c_int]
    lib.vgt_oracle_sdf_from_occupancy.restype = ctypes.c_int
    lib.vgt_oracle_cells_filled_mask.argtypes = [_p, _i64, ctypes.c_int, ctypes.c_int, ctypes.c_int, _p, _i64,
                                                 ctypes.c_int, _p]
    lib.vgt_oracle_combine_free_and_named.argtypes = [_p, _p, _i64, _p, _p, _p]
    lib.vgt_oracle_coarse_gradient.argtypes = [_p, _i64, _i64, _i64, _f64, ctypes.c_int, _p, _p]
    lib.vgt_oracle_local_extrema_map.argtypes = [_p, _i64, _i64, _i64, _f64, _p, _p]
    lib.vgt_oracle_estimate_distance.argtypes = [_p, _i64, _i64, _i64, _f64, _p, _p, _i64, _p, _p]
    lib.vgt_oracle_fine_gradient.argtypes = [_p, _i64, _i64, _i64, _f64, _p, _p, _i64, _f64, _p, _p]
    lib.vgt_oracle_fine_gradient.restype = ctypes.c_int
    lib.vgt_oracle_raycast_f32.argtypes = [
        _p, _i64, _f32, _p, _f32, _f32, _f32, _f32, _f32, _i32, _i32, _i32, _p, ctypes.c_int]
    lib.vgt_oracle_raycast_f64.argtypes = [
        _p, _i64, _f64, _p, _f64, _f64, _f64, _f64, _f64, _i64, _i64, _i64, _p, ctypes.c_int]
    lib.vgt_oracle_filter.argtypes = [
        _p, _i64, _i32, _f64, _i32, _i32, ctypes.c_int, _p, ctypes.c_int]
    if path is None:
        _LIB = lib
    return lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def max_threads():
    return int(load().vgt_oracle_max_threads())


def edt3d(field, threads=0):
    """Squared EDT of a (nx,ny,nz) float64 array holding 0 at sites, +inf elsewhere."""
    f = np.ascontiguousarray(field, dtype=np.float64).copy()
    nx, ny, nz = f.shape
    load().vgt_oracle_edt3d_inplace(_ptr(f), nx, ny, nz, threads)
    return f


def edt1d(line):
    f = np.ascontiguousarray(line, dtype=np.float64).copy()
    load().vgt_oracle_edt1d_inplace(_ptr(f), f.shape[0], 1)
    return f


def sdf_from_mask(filled, resolution, threads=0):
    m = np.ascontiguousarray(filled, dtype=np.uint8)
    nx, ny, nz = m.shape
    out = np.empty((nx, ny, nz), dtype=np.float32)
    rc = load().vgt_oracle_sdf_from_mask(_ptr(m), nx, ny, nz, float(resolution), _ptr(out), threads)
    if rc != 0:
        raise RuntimeError("vgt_oracle_sdf_from_mask failed rc=%d" % rc)
    return out


def sdf_from_occupancy(occupancy, resolution, unknown_is_filled=True,
                       add_virtual_border=False, threads=0, lib=None):
    """ExtractSignedDistanceField<float>; returns (sdf, min, max)."""
    occ = np.ascontiguousarray(occupancy, dtype=np.float32)
    nx, ny, nz = occ.shape
    out = np.empty((nx, ny, nz), dtype=np.float32)
    lo = ctypes.c_float()
    hi = ctypes.c_float()
    rc = (lib or load()).vgt_oracle_sdf_from_occupancy(
        _ptr(occ), nx, ny, nz, float(resolution), int(bool(unknown_is_filled)),
        int(bool(add_virtual_border)), _ptr(out), ctypes.byref(lo), ctypes.byref(hi), threads)
    if rc != 0:
        raise RuntimeError("vgt_oracle_sdf_from_occupancy failed rc=%d" % rc)
    return out, float(lo.value), float(hi.value)


def cells_filled_mask(records, shape, mode, objects=(), unknown_is_filled=True, object_id_offset=4):
    """is_filled_fn of the tagged map types over a numpy record array (mode 0 / 1 / 2, see vgt_oracle.h)."""
    rec = np.ascontiguousarray(records)
    objs = np.ascontiguousarray(np.asarray(list(objects), dtype=np.uint32))
    mask = np.empty(rec.size, dtype=np.uint8)
    load().vgt_oracle_cells_filled_mask(_ptr(rec), rec.size, rec.dtype.itemsize, int(object_id_offset),
                                        int(mode), _ptr(objs) if objs.size else None, objs.size,
                                        int(bool(unknown_is_filled)), _ptr(mask))
    return mask.reshape(shape)


def sdf_from_cells(records, shape, resolution, objects_to_use=(), unknown_is_filled=True,
                   add_virtual_border=False, mode=None, object_id_offset=4):
    """ExtractSignedDistanceField of the tagged map types; returns (sdf, min, max)."""
    if mode is None:
        mode = 1 if object_id_offset >= 0 else 0
    mask = cells_filled_mask(records, shape, mode, objects_to_use, unknown_is_filled, object_id_offset)
    # a 0/1 occupancy through the plain predicate is the same is_filled_fn
    return sdf_from_occupancy(mask.astype(np.float32), resolution, False, add_virtual_border)


def free_and_named_objects_sdf(records, shape, resolution, unknown_is_filled=True, add_virtual_border=False,
                               object_id_offset=4):
    """ExtractFreeAndNamedObjectsSignedDistanceField; returns (sdf, min, max)."""
    free_sdf, _, _ = sdf_from_cells(records, shape, resolution, (), unknown_is_filled, add_virtual_border, 0,
                                    object_id_offset)
    named_sdf, _, _ = sdf_from_cells(records, shape, resolution, (), unknown_is_filled, add_virtual_border, 2,
                                     object_id_offset)
    out = np.empty_like(free_sdf)
    lo, hi = ctypes.c_float(), ctypes.c_float()
    load().vgt_oracle_combine_free_and_named(_ptr(free_sdf), _ptr(named_sdf), free_sdf.size, _ptr(out),
                                             ctypes.byref(lo), ctypes.byref(hi))
    return out, float(lo.value), float(hi.value)


def coarse_gradient(sdf, resolution, enable_edge_gradients=False):
    """GetGridAlignedIndexCoarseGradient at every voxel: (gradient [nx, ny, nz, 3] float64, has_value bool)."""
    field = np.ascontiguousarray(sdf, dtype=np.float32)
    nx, ny, nz = field.shape
    grad = np.empty((nx, ny, nz, 3), dtype=np.float64)
    has = np.empty((nx, ny, nz), dtype=np.uint8)
    load().vgt_oracle_coarse_gradient(_ptr(field), nx, ny, nz, float(resolution), int(bool(enable_edge_gradients)),
                                      _ptr(grad), _ptr(has))
    return grad, has.astype(bool)


def local_extrema_map(sdf, resolution, rotation=None):
    """ComputeLocalExtremaMap: [nx, ny, nz, 3] float64."""
    field = np.ascontiguousarray(sdf, dtype=np.float32)
    rot = None if rotation is None else np.ascontiguousarray(rotation, dtype=np.float64).reshape(9)
    out = np.empty(field.shape + (3,), dtype=np.float64)
    load().vgt_oracle_local_extrema_map(_ptr(field), *field.shape, float(resolution),
                                        _ptr(rot) if rot is not None else None, _ptr(out))
    return out


def estimate_distance(sdf, resolution, queries, grid_from_world=None):
    """EstimateLocationDistance for query points [N, 3]: (distance [N] float64, has_value [N] bool)."""
    field = np.ascontiguousarray(sdf, dtype=np.float32)
    q = np.ascontiguousarray(queries, dtype=np.float64).reshape(-1, 3)
    xf = None if grid_from_world is None else np.ascontiguousarray(grid_from_world, dtype=np.float64).reshape(16)
    out = np.empty(len(q), dtype=np.float64)
    has = np.empty(len(q), dtype=np.uint8)
    load().vgt_oracle_estimate_distance(_ptr(field), *field.shape, float(resolution), _ptr(xf) if xf is not None else None,
                                        _ptr(q), len(q), _ptr(out), _ptr(has))
    return out, has.astype(bool)


def fine_gradient(sdf, resolution, queries, window, grid_from_world=None):
    """GetLocationFineGradient for query points [N, 3]: (gradient [N, 3], has_value [N], window_too_large)."""
    field = np.ascontiguousarray(sdf, dtype=np.float32)
    q = np.ascontiguousarray(queries, dtype=np.float64).reshape(-1, 3)
    xf = None if grid_from_world is None else np.ascontiguousarray(grid_from_world, dtype=np.float64).reshape(16)
    out = np.empty((len(q), 3), dtype=np.float64)
    has = np.empty(len(q), dtype=np.uint8)
    too_large = load().vgt_oracle_fine_gradient(_ptr(field), *field.shape, float(resolution),
                                                _ptr(xf) if xf is not None else None, _ptr(q), len(q), float(window),
                                                _ptr(out), _ptr(has))
    return out, has.astype(bool), bool(too_large)


def raycast_f32(points, max_range, xform, voxel_size, inverse_voxel_size,
                grid_sizes, counts, tracking=None, threads=0, lib=None):
    """Float32 DDA; returns int32 tracking grid of shape (nx,ny,nz,2) = (free, filled)."""
    pts = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
    T = np.ascontiguousarray(xform, dtype=np.float32).reshape(16)
    nx, ny, nz = (int(c) for c in counts)
    if tracking is None:
        tracking = np.zeros((nx, ny, nz, 2), dtype=np.int32)
    (lib or load()).vgt_oracle_raycast_f32(
        _ptr(pts), pts.shape[0], float(max_range), _ptr(T), float(voxel_size),
        float(inverse_voxel_size), float(grid_sizes[0]), float(grid_sizes[1]),
        float(grid_sizes[2]), nx, ny, nz, _ptr(tracking), threads)
    return tracking


def raycast_f64(points, max_range, xform, voxel_size, inverse_voxel_size,
                grid_sizes, counts, tracking=None, threads=0):
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    T = np.ascontiguousarray(xform, dtype=np.float64).reshape(16)
    nx, ny, nz = (int(c) for c in counts)
    if tracking is None:
        tracking = np.zeros((nx, ny, nz, 2), dtype=np.int32)
    load().vgt_oracle_raycast_f64(
        _ptr(pts), pts.shape[0], float(max_range), _ptr(T), float(voxel_size),
        float(inverse_voxel_size), float(grid_sizes[0]), float(grid_sizes[1]),
        float(grid_sizes[2]), nx, ny, nz, _ptr(tracking), threads)
    return tracking


def filter_grids(tracking_grids, occupancy, percent_seen_free=1.0,
                 outlier_points_threshold=1, num_cameras_seen_free=1,
                 ratio_in_double=False, threads=0):
    """tracking_grids: (G,nx,ny,nz,2) int32; occupancy (nx,ny,nz) float32 -> filtered copy."""
    tg = np.ascontiguousarray(tracking_grids, dtype=np.int32)
    occ = np.ascontiguousarray(occupancy, dtype=np.float32).copy()
    g = tg.shape[0]
    cells = occ.size
    assert tg.size == g * cells * 2
    load().vgt_oracle_filter(_ptr(tg), cells, g, float(percent_seen_free),
                             int(outlier_points_threshold), int(num_cameras_seen_free),
                             int(bool(ratio_in_double)), _ptr(occ), threads)
    return occ
