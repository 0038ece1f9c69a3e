import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def sdf_kats():
    with open(os.path.join(GOLDEN, "sdf_reference_kats.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def sdf_scipy_cases():
    z = np.load(os.path.join(GOLDEN, "sdf_scipy.npz"))
    names = sorted({k.split("__")[0] for k in z.files})
    return {n: {f: z[n + "__" + f] for f in ("occ", "res", "uif", "sdf", "sdf_vb")} for n in names}


@pytest.fixture(scope="session")
def sdf_tagged_cases():
    """tests/golden/sdf_tagged_scipy.npz: {grid: {field: array}} (make_golden.make_tagged_sdf_fixture)."""
    z = np.load(os.path.join(GOLDEN, "sdf_tagged_scipy.npz"))
    out = {}
    for k in z.files:
        name, field = k.split("__", 1)
        out.setdefault(name, {})[field] = z[k]
    return out


def tagged_records(case, dtype):
    """Cell records of one tagged fixture grid in the layout of one of the reference's cell types."""
    rec = np.zeros(case["occ"].shape, dtype=dtype)
    rec["occupancy"] = case["occ"]
    if "object_id" in dtype.names:
        rec["object_id"] = case["ids"]
    if "component" in dtype.names:
        rec["component"] = (case["ids"] * 7 + 3).astype(np.uint32)      # must not influence any SDF
    if "spatial_segment" in dtype.names:
        rec["spatial_segment"] = 0xDEADBEEF
    return rec


@pytest.fixture(scope="session")
def voxelization_scene():
    z = np.load(os.path.join(GOLDEN, "voxelization_scene.npz"))
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def raycast_rays():
    return np.load(os.path.join(GOLDEN, "raycast_rays.npy"))


def kat_occupancy(case):
    """Build the occupancy grid described by one entry of sdf_reference_kats.json."""
    occ = np.full(case["shape"], case.get("background", 0.0), dtype=np.float32)
    box = case.get("filled_box")
    if box:
        x0, x1, y0, y1, z0, z1 = box
        occ[x0:x1, y0:y1, z0:z1] = 1.0
    return occ


def bits_equal(a, b):
    """Bit-exact float32 comparison (treats +0/-0 as different, inf == inf)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def ulp_diff_f32(a, b):
    """Max ULP distance between two float32 arrays of finite same-sign values."""
    ai = np.ascontiguousarray(a, dtype=np.float32).view(np.int32).astype(np.int64)
    bi = np.ascontiguousarray(b, dtype=np.float32).view(np.int32).astype(np.int64)
    return int(np.max(np.abs(ai - bi))) if ai.size else 0


# ---- predicates of the reference's test/pointcloud_voxelization_test.cpp:84-158 ----
def check_empty_voxelization(occ):
    occ = np.asarray(occ)
    assert np.all(occ[:, :, 0] == 1.0)
    assert np.all(occ[:, :, 1:] == 0.5)


def check_voxelization(occ):
    occ = np.asarray(occ)
    assert np.all(occ[:, :, 0] == 1.0)                 # bottom cells
    assert np.all(occ[3, 3:, 1:] == 0.0)               # seen empty
    assert np.all(occ[3:, 3, 1:] == 0.0)
    assert np.all(occ[4, 4:, 1:] == 1.0)               # seen filled
    assert np.all(occ[4:, 4, 1:] == 1.0)
    assert np.all(occ[5:, 5:, 1:] == 0.5)              # shadowed


def scene_clouds(scene, dtype):
    """Clouds of the reference test scene as (points, X_GC col-major) in the given precision."""
    clouds = []
    for cam in ("cam1", "cam2"):
        clouds.append((scene[cam + "_points"].astype(dtype),
                       scene[cam + "_X_GC_colmajor"].astype(dtype)))
    return clouds
