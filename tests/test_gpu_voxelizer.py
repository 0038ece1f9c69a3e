"""(gpu) Raycast voxelizer parity through the C ABI: tracking counts bit-exact against the
oracle of the same precision, the reference test scene's predicates, and the edge cases the
reference exercises (empty cloud list, empty cloud, NaN points, origin outside the grid)."""
import threading

import numpy as np
import pytest

from conftest import check_empty_voxelization, check_voxelization, scene_clouds
from voxelized_geometry_tools_amd import capi, synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as O
    return O


def _scene_grid(scene):
    occ = scene["static_occupancy"]
    vs = float(scene["voxel_size"])
    counts = occ.shape
    return occ, vs, counts


@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_reference_scene(ctx, oracle, voxelization_scene, precision):
    """test/pointcloud_voxelization_test.cpp: three clouds (third empty), filter (1.0, 1, 1)."""
    occ, vs, counts = _scene_grid(voxelization_scene)
    grids = ctx.tracking_grids(occ.size, 3)
    want = []
    if precision == "f32":
        vs32, ivs32 = np.float32(vs), np.float32(1.0 / vs)
        sizes = [np.float32(c * vs) for c in counts]
        for i, (pts, xf) in enumerate(scene_clouds(voxelization_scene, np.float32)):
            grids.raycast_f32(i, pts, np.float32(np.inf), xf, vs32, ivs32, sizes, counts)
            want.append(oracle.raycast_f32(pts, np.float32(np.inf), xf, vs32, ivs32, sizes, counts))
    else:
        sizes = [c * vs for c in counts]
        for i, (pts, xf) in enumerate(scene_clouds(voxelization_scene, np.float64)):
            grids.raycast_f64(i, pts, np.inf, xf, vs, 1.0 / vs, sizes, counts)
            want.append(oracle.raycast_f64(pts, np.inf, xf, vs, 1.0 / vs, sizes, counts))
    for i in range(2):
        assert np.array_equal(grids.retrieve(i, counts), want[i]), "tracking counts cloud %d" % i
    assert not grids.retrieve(2).any()
    fg = ctx.filter_grid(occ)
    fg.filter(grids, 1.0, 1, 1, ratio_in_double=(precision == "f64"))
    out = fg.retrieve()
    check_voxelization(out)
    want.append(np.zeros_like(want[0]))
    assert np.array_equal(out, oracle.filter_grids(np.stack(want), occ, 1.0, 1, 1, precision == "f64"))


def test_empty_voxelization(ctx, voxelization_scene):
    """No clouds: one zeroed grid is still filtered (device_pointcloud_voxelization.cpp:79-80)."""
    occ, _, _ = _scene_grid(voxelization_scene)
    grids = ctx.tracking_grids(occ.size, 1)
    fg = ctx.filter_grid(occ)
    fg.filter(grids, 1.0, 1, 1)
    check_empty_voxelization(fg.retrieve())


@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_raycasting_invariants(ctx, raycast_rays, precision):
    """test/voxel_raycasting_test.cpp: per ray, every count is 0/1 and never free and filled."""
    counts = (40, 40, 40)
    vs = 0.125
    grids = ctx.tracking_grids(40 ** 3, 1)
    for ray in raycast_rays[:250]:
        grids.clear()
        origin, point = ray[:3], ray[3:]
        xf = synthetic.translation_xform(*origin)
        local = (point - origin).reshape(1, 3)
        if precision == "f32":
            grids.raycast_f32(0, local, 10.0, xf, vs, 1.0 / vs, [5.0] * 3, counts)
        else:
            grids.raycast_f64(0, local, 10.0, xf, vs, 1.0 / vs, [5.0] * 3, counts)
        g = grids.retrieve(0, counts)
        assert g.min() >= 0 and g.max() <= 1
        assert not np.any((g[..., 0] > 0) & (g[..., 1] > 0))


@pytest.mark.parametrize("sensor", [(2.56, 2.56, 2.56), (-1.0, 2.56, 2.56)])
def test_synthetic_cloud_counts_bit_exact(ctx, oracle, sensor):
    """BASELINE config C3 shape at reduced size: 128^3 grid, 200k points, 1 % NaN, clipped and
    unclipped rays, sensor inside / outside the grid (slab-entry branch)."""
    counts = (128, 128, 128)
    vs = np.float32(0.04)
    ivs = np.float32(1.0) / vs
    sizes = [np.float32(c) * vs for c in counts]
    pts = synthetic.raycast_cloud(200_000, seed=42)
    xf = synthetic.translation_xform(*sensor).astype(np.float32)
    grids = ctx.tracking_grids(int(np.prod(counts)), 1)
    grids.raycast_f32(0, pts, 3.0, xf, vs, ivs, sizes, counts)
    got = grids.retrieve(0, counts)
    want = oracle.raycast_f32(pts, 3.0, xf, vs, ivs, sizes, counts)
    assert got.sum() == want.sum() and got.sum() > 0
    assert np.array_equal(got, want)


@pytest.mark.parametrize("sensor", [(2.56, 2.56, 2.56), (-1.0, 2.56, 2.56)])
def test_config3_full_size_counts_bit_exact(ctx, oracle, sensor):
    """BASELINE config 3 at full size: the 1M-point synthetic cloud (1 % NaN, clipped and unclipped rays) into a
    256^3 grid, sensor inside the grid (cloud A) and outside it (cloud B, slab-entry branch); every tracking
    count against the CPU oracle, and the filtered grid.  This is the size at which the direction sort and the
    per-workgroup LDS accumulation are active."""
    n = 256
    counts = (n, n, n)
    vs = np.float32(5.12 / n)
    ivs = np.float32(1.0) / vs
    sizes = [np.float32(c) * vs for c in counts]
    pts = synthetic.raycast_cloud(1_000_000, seed=42)
    xf = synthetic.translation_xform(*sensor).astype(np.float32)
    grids = ctx.tracking_grids(n ** 3, 1)
    grids.raycast_f32(0, pts, 3.0, xf, vs, ivs, sizes, counts)
    got = grids.retrieve(0, counts)
    want = oracle.raycast_f32(pts, 3.0, xf, vs, ivs, sizes, counts)
    assert got.sum() == want.sum() and got.sum() > 0
    assert np.array_equal(got, want)
    env = np.zeros(counts, dtype=np.float32)
    env[:, :, 0] = 1.0
    fg = ctx.filter_grid(env)
    fg.filter(grids, 1.0, 1, 1)
    assert np.array_equal(fg.retrieve(), oracle.filter_grids(want[None], env, 1.0, 1, 1, False))
    fg.close()
    grids.close()


@pytest.mark.parametrize("helpers", [[], [0], [0, 0, 0], [0] * 7])
@pytest.mark.parametrize("sensor", [(2.56, 2.56, 2.56), (-1.0, 2.56, 2.56)])
def test_one_cloud_split_over_devices(ctx, oracle, sensor, helpers):
    """vgt_hipx_raycast_points_split (single cloud): shares of the points in private grids, summed into
    the caller's grid -- bit-equal to the oracle on the whole cloud whatever the split; counts already in the grid
    stay.  One GPU here, so the helpers are the caller's own device (copy + add branch)."""
    counts = (96, 96, 96)
    vs = np.float32(5.12 / 96)
    ivs = np.float32(1.0) / vs
    sizes = [np.float32(c) * vs for c in counts]
    pts = synthetic.raycast_cloud(100_003, seed=7)
    xf = synthetic.translation_xform(*sensor).astype(np.float32)
    grids = ctx.tracking_grids(int(np.prod(counts)), 2)
    grids.raycast_f32_split(1, helpers, pts, 3.0, xf, vs, ivs, sizes, counts)
    want = oracle.raycast_f32(pts, 3.0, xf, vs, ivs, sizes, counts)
    assert np.array_equal(grids.retrieve(1, counts), want)
    assert not grids.retrieve(0).any()
    # a second cloud into the same grid accumulates, as RaycastPoints does
    grids.raycast_f32_split(1, helpers, pts[:1000], 3.0, xf, vs, ivs, sizes, counts)
    want2 = want + oracle.raycast_f32(pts[:1000], 3.0, xf, vs, ivs, sizes, counts)
    assert np.array_equal(grids.retrieve(1, counts), want2)
    # fewer points than shares; no points at all
    few = ctx.tracking_grids(int(np.prod(counts)), 1)
    few.raycast_f32_split(0, helpers, pts[:3], 3.0, xf, vs, ivs, sizes, counts)
    assert np.array_equal(few.retrieve(0, counts), oracle.raycast_f32(pts[:3], 3.0, xf, vs, ivs, sizes, counts))
    few.clear()
    few.raycast_f32_split(0, helpers, pts[:0], 3.0, xf, vs, ivs, sizes, counts)
    assert not few.retrieve(0).any()
    few.close()
    grids.close()
    capi.sdf_multi_release()


def test_one_cloud_split_errors(ctx):
    counts = (8, 8, 8)
    grids = ctx.tracking_grids(512, 1)
    pts = np.zeros((10, 3), dtype=np.float32)
    xf = np.eye(4, dtype=np.float32).T
    with pytest.raises(capi.VgtHipError, match="out of range"):
        grids.raycast_f32_split(0, [99], pts, 1.0, xf, 1.0, 1.0, [8.0] * 3, counts)
    with pytest.raises(ValueError, match="index out of range"):
        grids.raycast_f32_split(3, [0], pts, 1.0, xf, 1.0, 1.0, [8.0] * 3, counts)
    with pytest.raises(ValueError, match="do not match"):
        grids.raycast_f32_split(0, [0], pts, 1.0, xf, 1.0, 1.0, [8.0] * 3, (8, 8, 9))
    grids.close()


def test_ray_split_over_torch_distributed(ctx, oracle):
    """multi_gpu.RaySplit with a one-rank RCCL group: the library's tracking grid is the all-reduce's tensor
    (no copy), and the share rule is the library's."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from voxelized_geometry_tools_amd import multi_gpu
    for n, world in ((10, 3), (100_003, 8), (2, 5)):
        for r in range(world):
            assert multi_gpu.point_share(n, world, r) == capi.point_share(n, world, r)
    counts = (64, 64, 64)
    vs = np.float32(0.08)
    ivs = np.float32(1.0) / vs
    sizes = [np.float32(c) * vs for c in counts]
    pts = synthetic.raycast_cloud(50_000, seed=3)
    xf = synthetic.translation_xform(2.56, 2.56, 2.56).astype(np.float32)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        grids = ctx.tracking_grids(int(np.prod(counts)), 1)
        split = multi_gpu.RaySplit(torch, dist, grids, 0, dev)
        first, count = multi_gpu.point_share(pts.shape[0], 1, 0)
        share = torch.from_numpy(pts[first:first + count]).to(dev)
        total = split.run(share, 3.0, xf, vs, ivs, sizes, counts)
        torch.cuda.synchronize()
        want = oracle.raycast_f32(pts, 3.0, xf, vs, ivs, sizes, counts)
        assert np.array_equal(total.cpu().numpy().reshape(counts + (2,)), want)
        assert np.array_equal(grids.retrieve(0, counts), want)
        grids.close()
    finally:
        ctx.reset_stream()
        dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["one_direction", "two_points", "tiny_grid", "f64"])
def test_accumulation_table_corner_cases(ctx, oracle, kind):
    """Clouds that stress the per-workgroup LDS accumulation: every ray identical (one hot chain of cells),
    two alternating points, a grid with fewer cells than table slots, and the double-precision kernel."""
    rng = np.random.default_rng(5)
    npts = 70_000
    counts = (48, 40, 56)
    vs = np.float32(0.05)
    if kind == "one_direction":
        pts = np.tile(np.array([[1.1, 0.7, 0.9]], dtype=np.float32), (npts, 1))
    elif kind == "two_points":
        pts = np.tile(np.array([[1.1, 0.7, 0.9], [-0.8, 0.3, 1.2]], dtype=np.float32), (npts // 2, 1))
    elif kind == "tiny_grid":
        counts = (6, 5, 7)
        vs = np.float32(0.3)
        pts = (rng.standard_normal((npts, 3)) * 1.5).astype(np.float32)
    else:
        pts = (rng.standard_normal((npts, 3)) * 1.2).astype(np.float32)
    ivs = np.float32(1.0) / vs
    sizes = [np.float32(c) * vs for c in counts]
    xf = synthetic.translation_xform(float(sizes[0]) * 0.4, float(sizes[1]) * 0.5, float(sizes[2]) * 0.45)
    grids = ctx.tracking_grids(int(np.prod(counts)), 1)
    if kind == "f64":
        sizes64 = [float(c) * float(vs) for c in counts]
        grids.raycast_f64(0, pts.astype(np.float64), 2.0, xf.astype(np.float64), float(vs), 1.0 / float(vs), sizes64, counts)
        want = oracle.raycast_f64(pts.astype(np.float64), 2.0, xf.astype(np.float64), float(vs), 1.0 / float(vs), sizes64, counts)
    else:
        grids.raycast_f32(0, pts, 2.0, xf.astype(np.float32), vs, ivs, sizes, counts)
        want = oracle.raycast_f32(pts, 2.0, xf.astype(np.float32), vs, ivs, sizes, counts)
    assert np.array_equal(grids.retrieve(0, counts), want), kind
    grids.close()


def test_concurrent_raycasts_from_host_threads(ctx, oracle):
    """RaycastPoints is called concurrently on one helper with distinct grid indices
    (device_pointcloud_voxelization.cpp:147-149)."""
    counts = (64, 64, 64)
    vs = np.float32(0.05)
    ivs = np.float32(1.0) / vs
    sizes = [np.float32(c) * vs for c in counts]
    clouds = [synthetic.raycast_cloud(30_000, seed=100 + i) * np.float32(0.5) for i in range(6)]
    xfs = [synthetic.translation_xform(1.0 + 0.2 * i, 1.5, 1.6).astype(np.float32) for i in range(6)]
    grids = ctx.tracking_grids(64 ** 3, 6)
    errors = []

    def work(i):
        try:
            grids.raycast_f32(i, clouds[i], 1.5, xfs[i], vs, ivs, sizes, counts)
        except Exception as e:  # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors
    for i in range(6):
        want = oracle.raycast_f32(clouds[i], 1.5, xfs[i], vs, ivs, sizes, counts)
        assert np.array_equal(grids.retrieve(i, counts), want), i


def _hip_memcpy_htod(dev_ptr, array):
    """Test helper: writes a host array into library-owned device memory via the HIP runtime."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    a = np.ascontiguousarray(array)
    rc = hip.hipMemcpy(dev_ptr, a.ctypes.data_as(ctypes.c_void_p), a.nbytes, 1)
    assert rc == 0, "hipMemcpy failed: %d" % rc


def test_filter_options(ctx, oracle):
    """CountsSeenAs / FilterGrids over random counts and every option combination."""
    rng = np.random.default_rng(5)
    shape = (16, 16, 16)
    cells = 4096
    tracking = np.zeros((3,) + shape + (2,), dtype=np.int32)
    tracking[..., 0] = rng.integers(0, 6, (3,) + shape) * (rng.random((3,) + shape) < 0.5)
    tracking[..., 1] = rng.integers(0, 4, (3,) + shape) * (rng.random((3,) + shape) < 0.4)
    occ = rng.choice(np.array([0.0, 0.5, 1.0], dtype=np.float32), size=shape)
    grids = ctx.tracking_grids(cells, 3)
    ctx.synchronize()
    for g in range(3):
        assert grids.offset(g) == g * cells * 2
        _hip_memcpy_htod(grids.dev_ptr(g), tracking[g])
    for g in range(3):
        assert np.array_equal(grids.retrieve(g, shape), tracking[g])
    for dbl in (False, True):
        for pct, outlier, ncam in [(1.0, 1, 1), (0.6, 2, 1), (1.0, 1, 2), (0.3, 3, 3), (0.75, 1, 1)]:
            fg = ctx.filter_grid(occ)
            fg.filter(grids, pct, outlier, ncam, ratio_in_double=dbl)
            want = oracle.filter_grids(tracking, occ, pct, outlier, ncam, dbl)
            assert np.array_equal(fg.retrieve(), want), (pct, outlier, ncam, dbl)


def test_handle_errors(ctx):
    with pytest.raises(ValueError):
        ctx.tracking_grids(0, 1)                       # zero-element buffers are an error
    with pytest.raises(ValueError):
        ctx.tracking_grids(10, 0)
    grids = ctx.tracking_grids(8, 1)
    with pytest.raises(ValueError):                    # grid index out of range
        grids.raycast_f32(3, np.zeros((1, 3)), 1.0, np.eye(4).reshape(16), 1.0, 1.0, [2, 2, 2], (2, 2, 2))
    with pytest.raises(ValueError):                    # voxel counts do not match the handle
        grids.raycast_f32(0, np.zeros((1, 3)), 1.0, np.eye(4).reshape(16), 1.0, 1.0, [3, 3, 3], (3, 3, 3))
    grids.raycast_f32(0, np.zeros((0, 3)), 1.0, np.eye(4).reshape(16), 1.0, 1.0, [2, 2, 2], (2, 2, 2))
    assert not grids.retrieve(0).any()                 # empty cloud is a no-op
    with pytest.raises(capi.VgtHipUnavailable):
        capi.Context(9999)                             # device index out of range


@pytest.mark.parametrize("add_virtual_border", [False, True])
def test_device_resident_voxelize_then_sdf(ctx, oracle, add_virtual_border):
    """raycast -> filter -> SDF without leaving the device (the filtered grid's device buffer is
    the EDT's input, the extrema come back from the device instead of SignedDistanceField::Lock()'s host
    rescan), against the same chain through the oracle."""
    import torch
    counts = (96, 80, 64)
    vs = np.float32(0.05)
    ivs = np.float32(1.0) / vs
    sizes = [np.float32(c) * vs for c in counts]
    static = synthetic.make_occupancy(counts, "spheres", seed=5) * np.float32(0.0) + np.float32(0.5)  # all unknown
    clouds = [synthetic.raycast_cloud(40_000, seed=7 + i) * np.float32(0.6) for i in range(2)]
    xfs = [synthetic.translation_xform(2.0 + i, 2.0, 1.6).astype(np.float32) for i in range(2)]
    grids = ctx.tracking_grids(int(np.prod(counts)), 2)
    want_tracking = []
    for i in range(2):
        grids.raycast_f32(i, clouds[i], 4.0, xfs[i], vs, ivs, sizes, counts)
        want_tracking.append(oracle.raycast_f32(clouds[i], 4.0, xfs[i], vs, ivs, sizes, counts))
    fg = ctx.filter_grid(static)
    fg.filter(grids, 0.9, 1, 1)
    want_occ = oracle.filter_grids(np.stack(want_tracking), static, 0.9, 1, 1, False)
    assert len(np.unique(want_occ)) == 3          # free, unknown and filled cells all present
    want_sdf, wlo, whi = oracle.sdf_from_occupancy(want_occ, float(vs), True, add_virtual_border)

    sdf = torch.empty(counts, dtype=torch.float32, device="cuda")
    nbytes = capi.sdf_workspace_bytes(counts)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    minmax = torch.zeros(2, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    # same context => same stream as the raycast and the filter: no host synchronisation in between
    ctx.sdf_dev(fg.dev_ptr(), counts, float(vs), sdf.data_ptr(), ws.data_ptr(), nbytes, minmax.data_ptr(),
                unknown_is_filled=True, add_virtual_border=add_virtual_border)
    ctx.synchronize()
    got = sdf.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want_sdf.view(np.uint32))
    mm = minmax.cpu().numpy()
    assert (float(mm[0]), float(mm[1])) == (wlo, whi)


@pytest.mark.parametrize("layout", [(16, 0), (32, 4), (20, 8), (12, 0)], ids=["xyzi", "padded", "tail", "packed"])
def test_pointcloud2_ingestion(ctx, oracle, layout):
    """a PointCloud2 data buffer (point_step bytes per record, x/y/z FLOAT32 at an offset, other
    fields around them) raycast in place gives exactly the counts of the packed xyz path."""
    point_step, xyz_offset = layout
    counts = (64, 64, 64)
    vs = np.float32(0.05)
    ivs = np.float32(1.0) / vs
    sizes = [np.float32(c) * vs for c in counts]
    pts = synthetic.raycast_cloud(50_000, seed=3) * np.float32(0.5)
    xf = synthetic.translation_xform(1.6, 1.5, 1.4).astype(np.float32)
    rng = np.random.default_rng(5)
    data = rng.integers(0, 256, size=(len(pts), point_step), dtype=np.uint8)    # junk in the other fields
    data[:, xyz_offset:xyz_offset + 12] = pts.view(np.uint8).reshape(len(pts), 12)
    grids = ctx.tracking_grids(int(np.prod(counts)), 1)
    grids.raycast_pointcloud2(0, data, len(pts), point_step, xyz_offset, 2.0, xf, vs, ivs, sizes, counts)
    want = oracle.raycast_f32(pts, 2.0, xf, vs, ivs, sizes, counts)
    assert want.sum() > 0
    assert np.array_equal(grids.retrieve(0, counts), want)
    with pytest.raises(ValueError):
        grids.raycast_pointcloud2(0, data, len(pts), point_step, point_step - 8, 2.0, xf, vs, ivs, sizes, counts)
    if point_step % 4 == 0:
        with pytest.raises(ValueError):
            grids.raycast_pointcloud2(0, data, len(pts), point_step, 2, 2.0, xf, vs, ivs, sizes, counts)


def test_pooled_handle_buffers_come_back_clean(ctx, oracle):
    """Tracking-grid / filter-grid buffers of destroyed handles are kept for the next handle of the same size: a reused
    tracking grid starts from zero counts, a reused filter grid holds the new occupancy, vgt_hip_trim drops the pool."""
    counts = (48, 48, 48)
    cells = int(np.prod(counts))
    vs = np.float32(5.12 / 48)
    ivs = np.float32(1.0) / vs
    sizes = [np.float32(c) * vs for c in counts]
    pts = synthetic.raycast_cloud(20_000, seed=5)
    xf = synthetic.translation_xform(2.56, 2.56, 2.56).astype(np.float32)
    want = oracle.raycast_f32(pts, 3.0, xf, vs, ivs, sizes, counts)
    first_ptr = None
    for round_index in range(3):
        grids = ctx.tracking_grids(cells, 2)
        if round_index == 0:
            first_ptr = grids.dev_ptr(0)
        elif round_index == 1:
            assert grids.dev_ptr(0) == first_ptr, "the buffer of the destroyed handle was not reused"
        assert not grids.retrieve(0).any() and not grids.retrieve(1).any()
        grids.raycast_f32(1, pts, 3.0, xf, vs, ivs, sizes, counts)
        assert np.array_equal(grids.retrieve(1, counts), want)
        env = np.full(counts, 0.25 * round_index, dtype=np.float32)
        fg = ctx.filter_grid(env)
        assert np.array_equal(fg.retrieve().reshape(counts), env)
        fg.close()
        grids.close()
        if round_index == 1:
            ctx.trim()
def test_context_destroyed_before_its_handles():
    """A context destroyed while grids / filter grids / cell grids made from it are still alive (the RAII order
    of a caller that declares the context last) must not be touched after it is freed: the library keeps the
    context record until the last handle is destroyed."""
    c = capi.Context(0)
    grids = c.tracking_grids(512, 2)
    fg = c.filter_grid(np.zeros((8, 8, 8), dtype=np.float32))
    raw = c.handle
    c.handle = None                      # keep Context.close() from closing the children first
    c._lib.vgt_hip_destroy(raw)
    grids.close()
    fg.close()
    # and the ordinary order through the binding: close() closes the children, then the context
    c2 = capi.Context(0)
    g2 = c2.tracking_grids(64, 1)
    c2.close()
    assert g2.handle is None


def _rotation_xform(rng, translation):
    """Column-major rigid transform with a random rotation (the grid frame tilted against the cloud's)."""
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    m = np.eye(4)
    m[:3, :3] = q
    m[:3, 3] = translation
    return m.T.reshape(16).copy()


@pytest.mark.parametrize("threads", [64, 128, 192, 256, 512, 1024])
def test_table_kernel_at_every_workgroup_size(oracle, threads):
    """HIP_THREADS_PER_BLOCK given: the table kernel runs at that size (its table, the re-deal of rays by walk length
    and the flush all scale with it); 192 is not a power of two."""
    c = capi.Context(0, threads)
    counts = (100, 37, 250)     # no side a multiple of the table's 16-voxel window
    vs = np.float32(0.03)
    ivs = np.float32(1.0) / vs
    sizes = [np.float32(n) * vs for n in counts]
    pts = synthetic.raycast_cloud(90_000, seed=threads)
    xf = synthetic.translation_xform(1.4, 0.5, 3.1).astype(np.float32)
    grids = c.tracking_grids(int(np.prod(counts)), 1)
    grids.raycast_f32(0, pts, 2.5, xf, vs, ivs, sizes, counts)
    want = oracle.raycast_f32(pts, 2.5, xf, vs, ivs, sizes, counts)
    assert np.array_equal(grids.retrieve(0, counts), want)
    grids.close()
    c.close()


@pytest.mark.parametrize("case", ["rotated_inside", "rotated_outside", "far_sensor", "axis_aligned", "short_rays",
                                  "flat_grid", "line_grid", "f64_rotated"])
def test_walk_and_table_on_awkward_geometry(ctx, oracle, case):
    """The counter form of the walk (steps left per axis, steps to the grid's face) and the position-mapped table on
    geometry that exercises their edges: tilted frames, a sensor far outside (rays enter through a face, most miss),
    rays along the axes (ties between the axes' boundary times, zero components), rays shorter than a voxel, grids one
    voxel thick or one voxel wide (the table's window wraps in every step), and the double kernel (no re-deal)."""
    rng = np.random.default_rng(11)
    npts = 80_000
    counts, vs, max_range = (90, 70, 110), np.float32(0.04), 3.0
    pts = synthetic.raycast_cloud(npts, seed=3)
    centre = [float(c) * float(vs) * 0.5 for c in counts]
    xf = synthetic.translation_xform(*centre)
    if case == "rotated_inside":
        xf = _rotation_xform(rng, centre)
    elif case in ("rotated_outside", "f64_rotated"):
        xf = _rotation_xform(rng, [-0.7, centre[1] * 1.3, centre[2]])
    elif case == "far_sensor":
        xf = synthetic.translation_xform(-40.0, centre[1], centre[2])
        pts = (pts * np.float32(14.0)).astype(np.float32)
        max_range = 60.0
    elif case == "axis_aligned":
        axes = np.eye(3, dtype=np.float32)[rng.integers(0, 3, npts)] * rng.choice([-1.0, 1.0], (npts, 1)).astype(np.float32)
        diag = rng.choice([-1.0, 0.0, 1.0], (npts, 3)).astype(np.float32)
        pts = np.where(rng.random((npts, 1)) < 0.5, axes, diag) * rng.uniform(0.1, 4.0, (npts, 1)).astype(np.float32)
        pts = pts.astype(np.float32)
    elif case == "short_rays":
        pts = (pts * np.float32(0.01)).astype(np.float32)
    elif case == "flat_grid":
        counts = (120, 1, 90)
        centre = [float(c) * float(vs) * 0.5 for c in counts]
        xf = synthetic.translation_xform(*centre)
    elif case == "line_grid":
        counts = (1, 1, 300)
        centre = [float(c) * float(vs) * 0.5 for c in counts]
        xf = synthetic.translation_xform(*centre)
    ivs = np.float32(1.0) / vs
    sizes = [np.float32(c) * vs for c in counts]
    grids = ctx.tracking_grids(int(np.prod(counts)), 1)
    if case == "f64_rotated":
        sizes64 = [float(c) * float(vs) for c in counts]
        grids.raycast_f64(0, pts.astype(np.float64), max_range, xf.astype(np.float64), float(vs), 1.0 / float(vs), sizes64, counts)
        want = oracle.raycast_f64(pts.astype(np.float64), max_range, xf.astype(np.float64), float(vs), 1.0 / float(vs), sizes64, counts)
    else:
        xf32 = xf.astype(np.float32)
        grids.raycast_f32(0, pts, max_range, xf32, vs, ivs, sizes, counts)
        want = oracle.raycast_f32(pts, max_range, xf32, vs, ivs, sizes, counts)
    got = grids.retrieve(0, counts)
    assert got.sum() == want.sum(), case
    assert np.array_equal(got, want), case
    grids.close()


def test_random_scenes_counts_bit_exact(ctx, oracle):
    """Twelve random scenes (grid shape, voxel size, sensor pose with rotation, range law, max range, cloud size on either
    side of the sorted path's threshold): every tracking count against the oracle."""
    rng = np.random.default_rng(2026)
    for scene in range(12):
        counts = tuple(int(v) for v in rng.integers(8, 140, 3))
        vs = np.float32(rng.uniform(0.02, 0.09))
        ivs = np.float32(1.0) / vs
        sizes = [np.float32(c) * vs for c in counts]
        npts = int(rng.choice([5_000, 40_000, 120_000]))
        pts = synthetic.raycast_cloud(npts, seed=100 + scene)
        pts = (pts * np.float32(rng.uniform(0.3, 2.0))).astype(np.float32)
        where = [float(rng.uniform(-0.3, 1.3)) * float(s) for s in sizes]
        xf = _rotation_xform(rng, where).astype(np.float32)
        max_range = float(rng.uniform(0.5, 6.0))
        grids = ctx.tracking_grids(int(np.prod(counts)), 1)
        grids.raycast_f32(0, pts, max_range, xf, vs, ivs, sizes, counts)
        want = oracle.raycast_f32(pts, max_range, xf, vs, ivs, sizes, counts)
        assert np.array_equal(grids.retrieve(0, counts), want), (scene, counts, npts)
        grids.close()


@pytest.mark.parametrize("npts", [3, 50_000])
def test_zero_length_rays_outside_the_grid(ctx, oracle, npts):
    """Points AT the sensor with the sensor outside the grid (direction 0 / 0, entry point NaN): the float kernel starts
    them in voxel (0, 0, 0) like the reference's device kernels (the device cast of NaN is 0), the double kernel drops
    them like the CPU voxelizer (x86's cast); see tests/test_oracle_voxelization.py.  Small cloud: the kernel without
    the table; large: mixed into a sorted cloud."""
    from test_oracle_voxelization import _zero_length_ray_scene
    counts, vs, zero_pts, xf = _zero_length_ray_scene()
    pts = synthetic.raycast_cloud(npts, seed=8, nan_every=0)
    pts[::3] = 0.0
    sizes = [np.float32(c) * vs for c in counts]
    grids = ctx.tracking_grids(int(np.prod(counts)), 1)
    grids.raycast_f32(0, pts, 3.0, xf.astype(np.float32), vs, np.float32(1.0) / vs, sizes, counts)
    got = grids.retrieve(0, counts)
    want = oracle.raycast_f32(pts, 3.0, xf.astype(np.float32), vs, np.float32(1.0) / vs, sizes, counts)
    assert np.array_equal(got, want)
    assert got[0, 0, 0, 0] >= len(pts[::3])
    grids.clear()
    sizes64 = [float(c) * float(vs) for c in counts]
    call = (pts.astype(np.float64), 3.0, xf, float(vs), 1.0 / float(vs), sizes64, counts)
    grids.raycast_f64(0, *call)
    assert np.array_equal(grids.retrieve(0, counts), oracle.raycast_f64(*call))
    grids.close()
