"""ctypes binding of libvgt_hip.so (the C ABI declared in include/vgt_hip.h).

This is test / bench plumbing: the product is the shared library and the C++ glue in
include/vgt_hip/.  There is no CPU fallback here -- if the library is missing, or no HIP
device is usable, the calls raise.
"""
import ctypes
import threading
import weakref
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VGT_HIP_LIB") or os.path.join(_HERE, "libvgt_hip.so")  # override: diagnostic builds
# The product library plus the cross-check EDT implementations and the testing hooks (-DVGT_HIP_TESTING): what the
# parity tests load NEXT TO the product library (Context(testing=True)) to check the default pipeline against
# independent implementations.  Nothing in the product path uses it.
TESTING_LIB_PATH = os.environ.get("VGT_HIP_TESTING_LIB") or os.path.join(_HERE, "libvgt_hip_testing.so")
_LIB = None

_i64 = ctypes.c_int64
_i32 = ctypes.c_int32
_f32 = ctypes.c_float
_f64 = ctypes.c_double
_p = ctypes.c_void_p
_sz = ctypes.c_size_t
_int = ctypes.c_int

# name -> (restype, argtypes); mirrors include/vgt_hip.h one to one.
SIGNATURES = {
    "vgt_hip_abi_version": (_int, []),
    "vgt_hip_last_error": (ctypes.c_char_p, []),
    "vgt_hip_device_count": (_int, [ctypes.POINTER(_int)]),
    "vgt_hip_device_name": (_int, [_int, ctypes.c_char_p, _sz]),
    "vgt_hip_create": (_int, [_int, _int, ctypes.POINTER(_p)]),
    "vgt_hip_destroy": (None, [_p]),
    "vgt_hip_trim": (_int, [_p]),
    "vgt_hip_set_stream": (_int, [_p, _p]),
    "vgt_hip_reset_stream": (_int, [_p]),
    "vgt_hip_synchronize": (_int, [_p]),
    "vgt_hip_device_of": (_int, [_p]),
    "vgt_hip_tracking_grids_create": (_int, [_p, _i64, _i32, ctypes.POINTER(_p)]),
    "vgt_hip_tracking_grids_destroy": (None, [_p]),
    "vgt_hip_tracking_grids_num_cells": (_i64, [_p]),
    "vgt_hip_tracking_grids_num_grids": (_i32, [_p]),
    "vgt_hip_tracking_grids_offset": (_i64, [_p, _sz]),
    "vgt_hip_tracking_grids_dev_ptr": (_p, [_p, _sz]),
    "vgt_hip_tracking_grids_clear": (_int, [_p, _p]),
    "vgt_hip_raycast_points_f32": (_int, [_p, _p, _sz, _p, _i64, _f32, _p, _f32, _f32, _f32, _f32,
                                          _f32, _i32, _i32, _i32]),
    "vgt_hip_raycast_pointcloud2_f32": (_int, [_p, _p, _sz, _p, _i64, _i64, _i64, _f32, _p, _f32, _f32, _f32,
                                               _f32, _f32, _i32, _i32, _i32]),
    "vgt_hip_raycast_points_f32_dev": (_int, [_p, _p, _sz, _p, _i64, _f32, _p, _f32, _f32, _f32,
                                              _f32, _f32, _i32, _i32, _i32]),
    "vgt_hip_raycast_points_f64": (_int, [_p, _p, _sz, _p, _i64, _f64, _p, _f64, _f64, _f64, _f64,
                                          _f64, _i32, _i32, _i32]),
    "vgt_hip_filter_grid_create": (_int, [_p, _i64, _p, ctypes.POINTER(_p)]),
    "vgt_hip_filter_grid_create_deferred": (_int, [_p, _i64, _p, ctypes.POINTER(_p)]),
    "vgt_hip_filter_grid_destroy": (None, [_p]),
    "vgt_hip_filter_grid_num_cells": (_i64, [_p]),
    "vgt_hip_filter_grid_dev_ptr": (_p, [_p]),
    "vgt_hip_filter_tracking_grids": (_int, [_p, _p, _f32, _i32, _i32, _p]),
    "vgt_hip_filter_tracking_grids_f64": (_int, [_p, _p, _f64, _i32, _i32, _p]),
    "vgt_hip_retrieve_tracking_grid": (_int, [_p, _p, _sz, _p]),
    "vgt_hip_retrieve_filtered_grid": (_int, [_p, _p, _p]),
    "vgt_hip_sdf_from_occupancy_f32": (_int, [_p, _p, _i64, _i64, _i64, _f64, _int, _int, _p, _p, _p]),
    "vgt_hip_sdf_from_mask_u8": (_int, [_p, _p, _i64, _i64, _i64, _f64, _int, _p, _p, _p]),
    "vgt_hip_sdf_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "vgt_hip_sdf_workspace_bytes_for_variant": (_sz, [_i64, _i64, _i64, _int]),
    "vgt_hip_sdf_estimate_distance": (_int, [_p, _p, _i64, _i64, _i64, _f64, _p, _p, _i64, _p, _p]),
    "vgt_hip_sdf_estimate_distance_dev": (_int, [_p, _p, _i64, _i64, _i64, _f64, _p, _p, _i64, _p, _p]),
    "vgt_hip_sdf_fine_gradient": (_int, [_p, _p, _i64, _i64, _i64, _f64, _p, _p, _i64, _f64, _p, _p]),
    "vgt_hip_sdf_local_extrema_map": (_int, [_p, _p, _i64, _i64, _i64, _f64, _p, _p]),
    "vgt_hip_sdf_local_extrema_map_dev": (_int, [_p, _p, _i64, _i64, _i64, _f64, _p, _p]),
    "vgt_hipx_sdf_multi": (_int, [_p, _int, _p, _i64, _i64, _i64, _f64, _int, _int, _p, _p, _p]),
    "vgt_hipx_release": (None, []),
    "vgt_hipx_last_timing": (_int, [_p]),
    "vgt_hipx_point_share": (_int, [_i64, _i32, _i32, _p, _p]),
    "vgt_hipx_raycast_points_split": (_int, [_p, _p, _sz, _p, _int, _p, _i64, _f32, _p, _f32, _f32, _f32, _f32,
                                             _f32, _i32, _i32, _i32]),
    "vgt_hip_sdf_dev": (_int, [_p, _p, _i64, _i64, _i64, _f64, _int, _int, _p, _p, _sz, _p]),
    "vgt_hip_sdf_dev_timed": (_int, [_p, _p, _i64, _i64, _i64, _f64, _int, _int, _p, _p, _sz, _p, _p]),
    "vgt_hip_sdf_batch_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64]),
    "vgt_hip_sdf_batch_dev": (_int, [_p, _p, _i64, _i64, _i64, _i64, _f64, _int, _int, _p, _p, _sz, _p]),
    "vgt_hip_sdf_batch_from_occupancy_f32": (_int, [_p, _p, _i64, _i64, _i64, _i64, _f64, _int, _int, _p, _p, _p]),
    "vgt_hip_cells_object_sdfs": (_int, [_p, _p, _p, _i64, _f64, _int, _int, _p, _p, _p]),
    "vgt_hip_cells_create": (_int, [_p, _p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32,
                                   ctypes.c_int32, _p]),
    "vgt_hip_cells_destroy": (None, [_p]),
    "vgt_hip_cells_object_ids": (_int, [_p, _p, _p, ctypes.c_int64, _p]),
    "vgt_hip_cells_sdf": (_int, [_p, _p, _p, ctypes.c_int64, ctypes.c_double, _int, _int, _p, _p, _p]),
    "vgt_hip_cells_free_and_named_objects_sdf": (_int, [_p, _p, ctypes.c_double, _int, _int, _p, _p, _p]),
    "vgt_hip_sdf_coarse_gradient": (_int, [_p, _p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_double,
                                          _int, _p, _p, _p]),
    "vgt_hip_sdf_coarse_gradient_dev": (_int, [_p, _p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                              ctypes.c_double, _int, _p, _p, _p]),
    "vgt_hip_sdf_slab_carries_dev": (_int, [_p, _p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.c_int64,
                                            ctypes.c_int64, _p]),
    "vgt_hip_timing_start": (_int, [_p, ctypes.c_int32]),
    "vgt_hip_timing_stop": (_int, [_p, _p, _p]),
    "vgt_hip_sdf_slab_summary_bytes": (_sz, [_i64, _i64]),
    "vgt_hip_sdf_slab_carries_bytes": (_sz, [_i64, _i64]),
    "vgt_hip_sdf_slab_range": (_int, [_i64, ctypes.c_int32, ctypes.c_int32, _p, _p]),
    "vgt_hip_sdf_slab_begin_dev": (_int, [_p, _p, _i64, _i64, _i64, _i64, _int, _p, _sz, _p, _p]),
    "vgt_hip_sdf_slab_finish_dev": (_int, [_p, _i64, _i64, _i64, _i64, _i64, _f64, _int, _p, _p, _p,
                                           _sz, _p, _p]),
}


# exported by libvgt_hip_testing.so only (include/vgt_hip.h under VGT_HIP_TESTING)
TESTING_SIGNATURES = {
    "vgt_hip_set_edt_variant": (_int, [_p, _int]),
    "vgt_hip_debug_finalize_check": (_int, [_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_double, _p, _p]),
    "vgt_hip_testing_set_host_pipeline_min_voxels": (_int, [ctypes.c_int64]),
    "vgt_hip_testing_set_short_line_rows": (_int, [_int]),
    "vgt_hip_testing_class_record_bytes": (_sz, [_i64, _i64, _i64]),
    "vgt_hip_testing_class_records_dev": (_int, [_p, _p, _i64, _i64, _i64, _int, _i64, _p, _p]),
}


class VgtHipError(RuntimeError):
    """HIP / runtime failure reported by libvgt_hip (std::runtime_error in the C++ glue)."""


class VgtHipUnavailable(VgtHipError):
    """No usable device (helper->IsAvailable() == false in the C++ glue)."""


_TESTING_LIB = None
_ERRORS = threading.local()


def _bind(path, signatures):
    try:
        # One HIP runtime per process: if torch is around, let it load its bundled
        # libamdhip64 (same SONAME) first so the dynamic linker reuses it for us.
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the C ABI
        pass
    lib = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)

    def remember_error(result, func, args):
        # error messages are thread-local PER LIBRARY: fetch the message from the library that failed
        if isinstance(result, int) and result != 0:
            _ERRORS.message = lib.vgt_hip_last_error().decode("utf-8", "replace")
        return result

    for name, (restype, argtypes) in signatures.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
        if restype is _int and name != "vgt_hip_abi_version":
            fn.errcheck = remember_error
    return lib


def load(testing=False):
    """Loads libvgt_hip.so (or, testing=True, libvgt_hip_testing.so) once.  Raises if it has not been built (no fallback)."""
    global _LIB, _TESTING_LIB
    if testing:
        if _TESTING_LIB is None:
            if not os.path.exists(TESTING_LIB_PATH):
                raise VgtHipError("libvgt_hip_testing.so is not built (%s); run `make -C voxelized_geometry_tools_amd/csrc`"
                                  % TESTING_LIB_PATH)
            _TESTING_LIB = _bind(TESTING_LIB_PATH, dict(SIGNATURES, **TESTING_SIGNATURES))
        return _TESTING_LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise VgtHipError(
            "libvgt_hip.so is not built (%s); run `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C voxelized_geometry_tools_amd/csrc`" % LIB_PATH)
    _LIB = _bind(LIB_PATH, SIGNATURES)
    return _LIB


def last_error():
    """Message of the last failed call of this thread (whichever of the two libraries it went to)."""
    message = getattr(_ERRORS, "message", None)
    return message if message is not None else load().vgt_hip_last_error().decode("utf-8", "replace")


def check(rc):
    if rc == 0:
        return
    msg = last_error()
    if rc == 1:
        raise ValueError(msg)
    if rc == 3:
        raise VgtHipUnavailable(msg)
    raise VgtHipError(msg)


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(ctypes.c_void_p)
    return ctypes.c_void_p(int(a))


def device_count():
    n = _int(0)
    check(load().vgt_hip_device_count(ctypes.byref(n)))
    return n.value


def device_name(device):
    buf = ctypes.create_string_buffer(256)
    check(load().vgt_hip_device_name(device, buf, 256))
    return buf.value.decode()


class Context:
    """One device + one stream (vgt_hip_ctx)."""

    def __init__(self, device=0, threads_per_block=-1, testing=False):
        """testing=True: a context of libvgt_hip_testing.so (set_edt_variant, debug_finalize_check,
        set_host_pipeline_min_voxels exist there only)."""
        self._lib = load(testing)
        self.testing = bool(testing)
        h = _p()
        check(self._lib.vgt_hip_create(device, threads_per_block, ctypes.byref(h)))
        self.handle = h
        self._children = weakref.WeakSet()   # grids / filter grids / cell grids created from this context

    def _adopt(self, child):
        self._children.add(child)

    def close(self):
        """Destroys the context after the handles created from it (the library also tolerates the
        other order: it keeps the context record alive until its last handle is destroyed)."""
        if getattr(self, "handle", None):
            for child in list(self._children):
                child.close()
            self._lib.vgt_hip_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def sdf_estimate_distance(self, sdf, resolution, queries, grid_from_world=None):
        """EstimateLocationDistance for a batch of query points [N, 3] -> (distance [N] float64, has_value [N] bool)."""
        field = np.ascontiguousarray(sdf, dtype=np.float32)
        q = np.ascontiguousarray(queries, dtype=np.float64).reshape(-1, 3)
        xf = None if grid_from_world is None else np.ascontiguousarray(grid_from_world, dtype=np.float64).reshape(16)
        out = np.empty(len(q), dtype=np.float64)
        has = np.empty(len(q), dtype=np.uint8)
        check(self._lib.vgt_hip_sdf_estimate_distance(self.handle, _ptr(field), *field.shape, float(resolution), _ptr(xf),
                                                      _ptr(q), len(q), _ptr(out), _ptr(has)))
        return out, has.astype(bool)

    def sdf_fine_gradient(self, sdf, resolution, queries, window, grid_from_world=None):
        """GetLocationFineGradient for a batch of query points [N, 3] -> (gradient [N, 3] float64, has_value [N] bool)."""
        field = np.ascontiguousarray(sdf, dtype=np.float32)
        q = np.ascontiguousarray(queries, dtype=np.float64).reshape(-1, 3)
        xf = None if grid_from_world is None else np.ascontiguousarray(grid_from_world, dtype=np.float64).reshape(16)
        out = np.empty((len(q), 3), dtype=np.float64)
        has = np.empty(len(q), dtype=np.uint8)
        check(self._lib.vgt_hip_sdf_fine_gradient(self.handle, _ptr(field), *field.shape, float(resolution), _ptr(xf),
                                                  _ptr(q), len(q), float(window), _ptr(out), _ptr(has)))
        return out, has.astype(bool)

    def sdf_local_extrema_map(self, sdf, resolution, rotation=None):
        """ComputeLocalExtremaMap: [nx, ny, nz, 3] float64 (grid-frame extremum location per voxel, +inf = off the grid)."""
        field = np.ascontiguousarray(sdf, dtype=np.float32)
        rot = None if rotation is None else np.ascontiguousarray(rotation, dtype=np.float64).reshape(9)
        out = np.empty(field.shape + (3,), dtype=np.float64)
        check(self._lib.vgt_hip_sdf_local_extrema_map(self.handle, _ptr(field), *field.shape, float(resolution),
                                                      _ptr(rot), _ptr(out)))
        return out

    def trim(self):
        """Frees the device buffers the context caches between host-pointer calls."""
        check(self._lib.vgt_hip_trim(self.handle))

    def set_stream(self, stream_ptr):
        """Run on an external hipStream_t; 0 / None = HIP's legacy default stream (torch's default)."""
        check(self._lib.vgt_hip_set_stream(self.handle, _ptr(stream_ptr) if stream_ptr else None))

    def reset_stream(self):
        check(self._lib.vgt_hip_reset_stream(self.handle))

    def synchronize(self):
        check(self._lib.vgt_hip_synchronize(self.handle))

    def timing_start(self, max_calls):
        """Deferred per-kernel timing of the following SDF calls (no synchronisation until timing_stop)."""
        self._timing_capacity = int(max_calls)
        check(self._lib.vgt_hip_timing_start(self.handle, int(max_calls)))

    def timing_stop(self):
        """-> float32 array [calls, 3]: ms of (pass 1 [+ slab record fix-up], Y pass, X pass) per recorded call."""
        out = np.zeros((self._timing_capacity, 3), dtype=np.float32)
        n = ctypes.c_int32(0)
        check(self._lib.vgt_hip_timing_stop(self.handle, _ptr(out), ctypes.byref(n)))
        return out[:n.value].copy()

    def set_edt_variant(self, variant):
        """Testing library only: 0 default, 1 the cross-check pipeline (int16 Z scan + pruned search)."""
        if not self.testing:
            if int(variant) == 0:
                return
            raise VgtHipError("EDT variants other than 0 exist in libvgt_hip_testing.so only: Context(testing=True)")
        check(self._lib.vgt_hip_set_edt_variant(self.handle, int(variant)))

    def class_records(self, occ_ptr, shape, unknown_is_filled=True, z_offset=0, records_ptr=None, summary_ptr=None):
        """Testing library only: pass 1 alone (vgt_hip_testing_class_records_dev)."""
        nx, ny, nz = shape
        check(self._lib.vgt_hip_testing_class_records_dev(self.handle, _ptr(occ_ptr), nx, ny, nz,
                                                          int(bool(unknown_is_filled)), int(z_offset), _ptr(records_ptr),
                                                          _ptr(summary_ptr)))

    def set_short_line_rows(self, rows):
        """Testing library only: lines of at most `rows` rows take the short-line kernels whatever the item count
        (0 = sweeps everywhere, negative = back to the product's rule)."""
        check(self._lib.vgt_hip_testing_set_short_line_rows(int(rows)))

    def set_host_pipeline_min_voxels(self, min_voxels):
        """Testing library only (process-wide there): smallest grid the host-pointer SDF entry points pipeline."""
        check(self._lib.vgt_hip_testing_set_host_pipeline_min_voxels(int(min_voxels)))

    def debug_finalize_check(self, first_d2, count, resolution):
        """(mismatches, first mismatching d2 or None) of the fast vs exact final conversion."""
        bad = ctypes.c_uint64(0)
        first = ctypes.c_uint64(0)
        check(self._lib.vgt_hip_debug_finalize_check(self.handle, int(first_d2), int(count), float(resolution),
                                                     ctypes.byref(bad), ctypes.byref(first)))
        return int(bad.value), (None if first.value == 2 ** 64 - 1 else int(first.value))

    def sdf_coarse_gradient(self, sdf, resolution, enable_edge_gradients=False, rotation=None):
        """Grid-aligned (or rotated) coarse gradient of every voxel: (gradient [nx, ny, nz, 3] float64, has_value)."""
        field = np.ascontiguousarray(sdf, dtype=np.float32)
        nx, ny, nz = field.shape
        grad = np.empty((nx, ny, nz, 3), dtype=np.float64)
        has = np.empty((nx, ny, nz), dtype=np.uint8)
        rot = None if rotation is None else np.ascontiguousarray(rotation, dtype=np.float64).reshape(9)
        check(self._lib.vgt_hip_sdf_coarse_gradient(self.handle, _ptr(field), nx, ny, nz, float(resolution),
                                                    int(bool(enable_edge_gradients)), _ptr(rot), _ptr(grad),
                                                    _ptr(has)))
        return grad, has.astype(bool)

    def cells(self, records, shape, object_id_offset=4):
        """Uploads a grid of cell records (see Cells)."""
        return Cells(self, records, shape, object_id_offset)

    # ---- SDF ----
    def sdf_from_occupancy(self, occupancy, resolution, unknown_is_filled=True,
                           add_virtual_border=False, out=None):
        occ = np.ascontiguousarray(occupancy, dtype=np.float32)
        if occ.ndim != 3:
            raise ValueError("occupancy must be (nx, ny, nz)")
        nx, ny, nz = occ.shape
        if out is None:
            out = np.empty(occ.shape, dtype=np.float32)
        elif out.shape != occ.shape or out.dtype != np.float32 or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous float32 array of the occupancy's shape")
        lo, hi = _f32(), _f32()
        check(self._lib.vgt_hip_sdf_from_occupancy_f32(
            self.handle, _ptr(occ), nx, ny, nz, float(resolution), int(bool(unknown_is_filled)),
            int(bool(add_virtual_border)), _ptr(out), ctypes.byref(lo), ctypes.byref(hi)))
        return out, lo.value, hi.value

    def sdf_batch_from_occupancy(self, grids, resolution, unknown_is_filled=True, add_virtual_border=False):
        """vgt_hip_sdf_batch_from_occupancy_f32: a list of equal-shape float32 grids (any addresses) ->
        (list of fields, mins, maxs) from one batched extraction."""
        grids = [np.ascontiguousarray(g, dtype=np.float32) for g in grids]
        if not grids or any(g.ndim != 3 or g.shape != grids[0].shape for g in grids):
            raise ValueError("a batch is a non-empty list of (nx, ny, nz) grids of one shape")
        nx, ny, nz = grids[0].shape
        outs = [np.empty(g.shape, dtype=np.float32) for g in grids]
        batch = len(grids)
        in_ptrs = (ctypes.c_void_p * batch)(*[g.ctypes.data for g in grids])
        out_ptrs = (ctypes.c_void_p * batch)(*[o.ctypes.data for o in outs])
        lo = np.zeros(batch, dtype=np.float32)
        hi = np.zeros(batch, dtype=np.float32)
        check(self._lib.vgt_hip_sdf_batch_from_occupancy_f32(
            self.handle, ctypes.cast(in_ptrs, ctypes.c_void_p), batch, nx, ny, nz, float(resolution),
            int(bool(unknown_is_filled)), int(bool(add_virtual_border)), ctypes.cast(out_ptrs, ctypes.c_void_p),
            _ptr(lo), _ptr(hi)))
        return outs, lo, hi

    def sdf_batch_dev(self, occ_ptr, batch, shape, resolution, sdf_ptr, ws_ptr, ws_bytes, minmax_ptr=None,
                      unknown_is_filled=True, add_virtual_border=False):
        """vgt_hip_sdf_batch_dev: [batch][nx][ny][nz] device buffers in and out."""
        nx, ny, nz = shape
        check(self._lib.vgt_hip_sdf_batch_dev(
            self.handle, _ptr(occ_ptr), int(batch), nx, ny, nz, float(resolution), int(bool(unknown_is_filled)),
            int(bool(add_virtual_border)), _ptr(sdf_ptr), _ptr(ws_ptr), ws_bytes, _ptr(minmax_ptr)))

    def sdf_from_mask(self, mask, resolution, add_virtual_border=False):
        m = np.ascontiguousarray(mask, dtype=np.uint8)
        nx, ny, nz = m.shape
        out = np.empty(m.shape, dtype=np.float32)
        lo, hi = _f32(), _f32()
        check(self._lib.vgt_hip_sdf_from_mask_u8(
            self.handle, _ptr(m), nx, ny, nz, float(resolution), int(bool(add_virtual_border)),
            _ptr(out), ctypes.byref(lo), ctypes.byref(hi)))
        return out, lo.value, hi.value

    def sdf_dev(self, occ_ptr, shape, resolution, sdf_ptr, ws_ptr, ws_bytes, minmax_ptr=None,
                unknown_is_filled=True, add_virtual_border=False, kernel_ms=None):
        nx, ny, nz = shape
        if kernel_ms is None:
            check(self._lib.vgt_hip_sdf_dev(
                self.handle, _ptr(occ_ptr), nx, ny, nz, float(resolution),
                int(bool(unknown_is_filled)), int(bool(add_virtual_border)), _ptr(sdf_ptr),
                _ptr(ws_ptr), ws_bytes, _ptr(minmax_ptr)))
        else:
            check(self._lib.vgt_hip_sdf_dev_timed(
                self.handle, _ptr(occ_ptr), nx, ny, nz, float(resolution),
                int(bool(unknown_is_filled)), int(bool(add_virtual_border)), _ptr(sdf_ptr),
                _ptr(ws_ptr), ws_bytes, _ptr(minmax_ptr), _ptr(kernel_ms)))

    # ---- multi-GPU Z slabs ----
    def sdf_slab_begin(self, occ_ptr, local_shape, z_offset, ws_ptr, ws_bytes, summary_ptr,
                       unknown_is_filled=True, kernel_ms=None):
        nx, ny, nz = local_shape
        check(self._lib.vgt_hip_sdf_slab_begin_dev(
            self.handle, _ptr(occ_ptr), nx, ny, nz, int(z_offset), int(bool(unknown_is_filled)),
            _ptr(ws_ptr), ws_bytes, _ptr(summary_ptr), _ptr(kernel_ms)))

    def sdf_slab_carries(self, gathered_ptr, world, rank, nx, ny, nz_global, carries_ptr):
        check(self._lib.vgt_hip_sdf_slab_carries_dev(self.handle, _ptr(gathered_ptr), int(world), int(rank),
                                                     int(nx), int(ny), int(nz_global), _ptr(carries_ptr)))

    def sdf_slab_finish(self, local_shape, z_offset, nz_global, resolution, carries_ptr, sdf_ptr,
                        ws_ptr, ws_bytes, minmax_ptr=None, add_virtual_border=False,
                        kernel_ms=None):
        nx, ny, nz = local_shape
        check(self._lib.vgt_hip_sdf_slab_finish_dev(
            self.handle, nx, ny, nz, int(z_offset), int(nz_global), float(resolution),
            int(bool(add_virtual_border)), _ptr(carries_ptr), _ptr(sdf_ptr), _ptr(ws_ptr), ws_bytes,
            _ptr(minmax_ptr), _ptr(kernel_ms)))

    # ---- voxelizer ----
    def tracking_grids(self, num_cells, num_grids):
        return TrackingGrids(self, num_cells, num_grids)

    def filter_grid(self, occupancy):
        return FilterGrid(self, occupancy)


def sdf_multi(devices, occupancy, resolution, unknown_is_filled=True, add_virtual_border=False, out=None):
    """vgt_hipx_sdf_multi: one process, one Z slab per entry of `devices` (a device may repeat)."""
    lib = load()
    occ = np.ascontiguousarray(occupancy, dtype=np.float32)
    if occ.ndim != 3:
        raise ValueError("occupancy must be (nx, ny, nz)")
    nx, ny, nz = occ.shape
    devs = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
    if out is None:
        out = np.empty(occ.shape, dtype=np.float32)
    lo, hi = _f32(), _f32()
    check(lib.vgt_hipx_sdf_multi(devs, len(devices), _ptr(occ) if occ.size else None, nx, ny, nz, float(resolution),
                                 int(bool(unknown_is_filled)), int(bool(add_virtual_border)), _ptr(out),
                                 ctypes.byref(lo), ctypes.byref(hi)))
    return out, lo.value, hi.value


def point_share(num_points, shares, share):
    """vgt_hipx_point_share -> (first, count) of `share` among `shares` contiguous shares of a cloud."""
    first, count = ctypes.c_int64(0), ctypes.c_int64(0)
    check(load().vgt_hipx_point_share(int(num_points), int(shares), int(share), ctypes.byref(first), ctypes.byref(count)))
    return first.value, count.value


def sdf_multi_release():
    """Frees the device state vgt_hipx_sdf_multi keeps between calls."""
    load().vgt_hipx_release()


def sdf_multi_last_timing():
    """Phases of the last sdf_multi call in ms: setup, upload, compute, download (slowest slab each), total."""
    ms = (ctypes.c_float * 5)()
    check(load().vgt_hipx_last_timing(ms))
    return dict(zip(("setup_ms", "upload_ms", "compute_ms", "download_ms", "total_ms"), [float(v) for v in ms]))


def sdf_batch_workspace_bytes(batch, shape):
    return int(load().vgt_hip_sdf_batch_workspace_bytes(int(batch), *[int(v) for v in shape]))


def sdf_workspace_bytes(shape, variant=0):
    """Workspace of the device-resident SDF entry points (variant != 0: a cross-check pipeline of the testing library)."""
    lib = load(testing=int(variant) != 0)
    return int(lib.vgt_hip_sdf_workspace_bytes_for_variant(*[int(s) for s in shape], int(variant)))


class TrackingGrids:
    def __init__(self, ctx, num_cells, num_grids):
        self.ctx = ctx
        self._lib = ctx._lib
        h = _p()
        check(self._lib.vgt_hip_tracking_grids_create(ctx.handle, int(num_cells), int(num_grids),
                                                      ctypes.byref(h)))
        self.handle = h
        ctx._adopt(self)
        ctx._adopt(self)
        self.num_cells = int(num_cells)
        self.num_grids = int(num_grids)

    def close(self):
        if getattr(self, "handle", None):
            self._lib.vgt_hip_tracking_grids_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def offset(self, index):
        return int(self._lib.vgt_hip_tracking_grids_offset(self.handle, index))

    def dev_ptr(self, index):
        return self._lib.vgt_hip_tracking_grids_dev_ptr(self.handle, index)

    def clear(self):
        check(self._lib.vgt_hip_tracking_grids_clear(self.ctx.handle, self.handle))

    def raycast_f32(self, index, points, max_range, xform, voxel_size, inverse_voxel_size,
                    grid_sizes, counts):
        pts = np.ascontiguousarray(points, dtype=np.float32).reshape(-1)
        T = np.ascontiguousarray(xform, dtype=np.float32).reshape(16)
        check(self._lib.vgt_hip_raycast_points_f32(
            self.ctx.handle, self.handle, index, _ptr(pts) if pts.size else None, pts.size // 3,
            float(max_range), _ptr(T), float(voxel_size), float(inverse_voxel_size),
            float(grid_sizes[0]), float(grid_sizes[1]), float(grid_sizes[2]),
            int(counts[0]), int(counts[1]), int(counts[2])))

    def raycast_f32_split(self, index, helper_devices, points, max_range, xform, voxel_size, inverse_voxel_size,
                          grid_sizes, counts):
        """vgt_hipx_raycast_points_split: one cloud over this context's device + `helper_devices` (a device may
        repeat); the private grids are summed into grid `index`."""
        pts = np.ascontiguousarray(points, dtype=np.float32).reshape(-1)
        T = np.ascontiguousarray(xform, dtype=np.float32).reshape(16)
        devs = (ctypes.c_int * max(len(helper_devices), 1))(*[int(d) for d in helper_devices])
        check(self._lib.vgt_hipx_raycast_points_split(
            self.ctx.handle, self.handle, index, devs, len(helper_devices), _ptr(pts) if pts.size else None,
            pts.size // 3, float(max_range), _ptr(T), float(voxel_size), float(inverse_voxel_size),
            float(grid_sizes[0]), float(grid_sizes[1]), float(grid_sizes[2]),
            int(counts[0]), int(counts[1]), int(counts[2])))

    def raycast_pointcloud2(self, index, data, num_points, point_step, xyz_offset, max_range, xform, voxel_size,
                            inverse_voxel_size, grid_sizes, counts):
        """sensor_msgs/PointCloud2 data buffer (bytes) with x, y, z FLOAT32 at xyz_offset of every record."""
        buf = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray)
                                   else data.view(np.uint8).reshape(-1))
        if buf.size < int(num_points) * int(point_step):
            raise ValueError("data buffer shorter than num_points * point_step")
        T = np.ascontiguousarray(xform, dtype=np.float32).reshape(16)
        check(self._lib.vgt_hip_raycast_pointcloud2_f32(
            self.ctx.handle, self.handle, index, _ptr(buf) if buf.size else None, int(num_points),
            int(point_step), int(xyz_offset), float(max_range), _ptr(T), float(voxel_size),
            float(inverse_voxel_size), float(grid_sizes[0]), float(grid_sizes[1]), float(grid_sizes[2]),
            int(counts[0]), int(counts[1]), int(counts[2])))

    def raycast_f32_dev(self, index, points_ptr, num_points, max_range, xform, voxel_size,
                        inverse_voxel_size, grid_sizes, counts):
        T = np.ascontiguousarray(xform, dtype=np.float32).reshape(16)
        check(self._lib.vgt_hip_raycast_points_f32_dev(
            self.ctx.handle, self.handle, index, _ptr(points_ptr), int(num_points),
            float(max_range), _ptr(T), float(voxel_size), float(inverse_voxel_size),
            float(grid_sizes[0]), float(grid_sizes[1]), float(grid_sizes[2]),
            int(counts[0]), int(counts[1]), int(counts[2])))

    def raycast_f64(self, index, points, max_range, xform, voxel_size, inverse_voxel_size,
                    grid_sizes, counts):
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1)
        T = np.ascontiguousarray(xform, dtype=np.float64).reshape(16)
        check(self._lib.vgt_hip_raycast_points_f64(
            self.ctx.handle, self.handle, index, _ptr(pts) if pts.size else None, pts.size // 3,
            float(max_range), _ptr(T), float(voxel_size), float(inverse_voxel_size),
            float(grid_sizes[0]), float(grid_sizes[1]), float(grid_sizes[2]),
            int(counts[0]), int(counts[1]), int(counts[2])))

    def retrieve(self, index, counts=None):
        out = np.empty((self.num_cells, 2), dtype=np.int32)
        check(self._lib.vgt_hip_retrieve_tracking_grid(self.ctx.handle, self.handle, index,
                                                       _ptr(out)))
        if counts is not None:
            out = out.reshape(tuple(counts) + (2,))
        return out


class FilterGrid:
    def __init__(self, ctx, occupancy):
        self.ctx = ctx
        self._lib = ctx._lib
        occ = np.ascontiguousarray(occupancy, dtype=np.float32)
        self.shape = occ.shape
        h = _p()
        check(self._lib.vgt_hip_filter_grid_create(ctx.handle, occ.size,
                                                   _ptr(occ) if occ.size else None,
                                                   ctypes.byref(h)))
        self.handle = h
        ctx._adopt(self)

    def close(self):
        if getattr(self, "handle", None):
            self._lib.vgt_hip_filter_grid_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def dev_ptr(self):
        return self._lib.vgt_hip_filter_grid_dev_ptr(self.handle)

    def filter(self, grids, percent_seen_free=1.0, outlier_points_threshold=1,
               num_cameras_seen_free=1, ratio_in_double=False):
        if ratio_in_double:
            check(self._lib.vgt_hip_filter_tracking_grids_f64(
                self.ctx.handle, grids.handle, float(percent_seen_free),
                int(outlier_points_threshold), int(num_cameras_seen_free), self.handle))
        else:
            check(self._lib.vgt_hip_filter_tracking_grids(
                self.ctx.handle, grids.handle, float(percent_seen_free),
                int(outlier_points_threshold), int(num_cameras_seen_free), self.handle))

    def retrieve(self):
        out = np.empty(self.shape, dtype=np.float32)
        check(self._lib.vgt_hip_retrieve_filtered_grid(self.ctx.handle, self.handle, _ptr(out)))
        return out


# numpy record layouts of the reference's cell types (occupancy first, then uint32 fields)
OCCUPANCY_COMPONENT_CELL = np.dtype([("occupancy", np.float32), ("component", np.uint32)])
TAGGED_OBJECT_CELL = np.dtype([("occupancy", np.float32), ("object_id", np.uint32)])
TAGGED_OBJECT_COMPONENT_CELL = np.dtype([("occupancy", np.float32), ("object_id", np.uint32),
                                         ("component", np.uint32), ("spatial_segment", np.uint32)])


class Cells:
    """Device copy of the raw cell store of an OccupancyComponentMap / TaggedObjectOccupancyMap /
    TaggedObjectOccupancyComponentMap, for any number of SDF extractions (vgt_hip_cells_*)."""

    def __init__(self, ctx, records, shape, object_id_offset=4):
        self.ctx = ctx
        self._lib = ctx._lib
        rec = np.ascontiguousarray(records)
        self.shape = tuple(int(s) for s in shape)
        if rec.size != int(np.prod(self.shape)):
            raise ValueError("records do not match the grid shape")
        h = _p()
        check(self._lib.vgt_hip_cells_create(ctx.handle, _ptr(rec), self.shape[0], self.shape[1], self.shape[2],
                                             rec.dtype.itemsize, int(object_id_offset), ctypes.byref(h)))
        self.handle = h
        ctx._adopt(self)

    def close(self):
        if getattr(self, "handle", None):
            self._lib.vgt_hip_cells_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def object_ids(self, capacity=4096):
        ids = np.zeros(capacity, dtype=np.uint32)
        count = ctypes.c_int64(0)
        check(self._lib.vgt_hip_cells_object_ids(self.ctx.handle, self.handle, _ptr(ids), capacity,
                                                 ctypes.byref(count)))
        if count.value > capacity:
            return self.object_ids(int(count.value))
        return ids[:count.value].copy()

    def sdf(self, resolution, objects_to_use=(), unknown_is_filled=True, add_virtual_border=False):
        objs = np.ascontiguousarray(np.asarray(list(objects_to_use), dtype=np.uint32))
        out = np.empty(self.shape, dtype=np.float32)
        lo, hi = ctypes.c_float(0), ctypes.c_float(0)
        check(self._lib.vgt_hip_cells_sdf(self.ctx.handle, self.handle, _ptr(objs) if objs.size else None,
                                          objs.size, float(resolution), int(bool(unknown_is_filled)),
                                          int(bool(add_virtual_border)), _ptr(out), ctypes.byref(lo),
                                          ctypes.byref(hi)))
        return out, float(lo.value), float(hi.value)

    def separate_object_sdfs(self, resolution, object_ids, unknown_is_filled=True, add_virtual_border=False):
        """MakeSeparateObjectSDFs: {object id: (sdf, min, max)}, all objects in one batched extraction
        (vgt_hip_cells_object_sdfs)."""
        ids = np.ascontiguousarray(np.asarray(list(object_ids), dtype=np.uint32))
        if ids.size == 0:
            return {}
        outs = [np.empty(self.shape, dtype=np.float32) for _ in range(ids.size)]
        out_ptrs = (ctypes.c_void_p * ids.size)(*[o.ctypes.data for o in outs])
        lo = np.zeros(ids.size, dtype=np.float32)
        hi = np.zeros(ids.size, dtype=np.float32)
        check(self._lib.vgt_hip_cells_object_sdfs(
            self.ctx.handle, self.handle, _ptr(ids), ids.size, float(resolution), int(bool(unknown_is_filled)),
            int(bool(add_virtual_border)), ctypes.cast(out_ptrs, ctypes.c_void_p), _ptr(lo), _ptr(hi)))
        return {int(i): (outs[k], float(lo[k]), float(hi[k])) for k, i in enumerate(ids)}

    def separate_object_sdfs_one_by_one(self, resolution, object_ids, **kw):
        """The reference's own loop: one ExtractSignedDistanceField({id}) per object (vgt_hip_cells_sdf)."""
        return {int(i): self.sdf(resolution, [int(i)], **kw) for i in object_ids}

    def all_object_sdfs(self, resolution, **kw):
        """MakeAllObjectSDFs."""
        return self.separate_object_sdfs(resolution, self.object_ids(), **kw)

    def free_and_named_objects_sdf(self, resolution, unknown_is_filled=True, add_virtual_border=False):
        out = np.empty(self.shape, dtype=np.float32)
        lo, hi = ctypes.c_float(0), ctypes.c_float(0)
        check(self._lib.vgt_hip_cells_free_and_named_objects_sdf(
            self.ctx.handle, self.handle, float(resolution), int(bool(unknown_is_filled)),
            int(bool(add_virtual_border)), _ptr(out), ctypes.byref(lo), ctypes.byref(hi)))
        return out, float(lo.value), float(hi.value)
