"""(gpu) SDF parity: the HIP path through the C ABI vs the CPU oracle, the reference's
known answers and the committed independent-EDT fixtures.  Bar: bit-exact float32."""
import numpy as np
import pytest

from conftest import bits_equal, kat_occupancy
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


@pytest.fixture(scope="module")
def vctx():
    """A context of libvgt_hip_testing.so: the product's code plus the cross-check pipeline (EDT variant 1) and the
    testing hooks.  The product library (the `ctx` fixture, what every other test runs) contains none of them."""
    c = capi.Context(0, testing=True)
    yield c
    c.set_edt_variant(0)
    c.close()


# 0 = the default pipeline (class records + lane-per-line sweeps; run on the PRODUCT library), 1 = the independent
# cross-check pipeline of the testing library: an int16 distance field along Z as pass 1, then a pruned search from HBM
VARIANTS = [0, 1]


def context_for(variant, ctx, vctx):
    if variant == 0:
        return ctx
    vctx.set_edt_variant(variant)
    return vctx



@pytest.mark.parametrize("variant", VARIANTS)
def test_reference_known_answers(ctx, vctx, sdf_kats, variant):
    """test/sdf_generation_test.cpp extrema + exact cases through the HIP path."""
    ctx = context_for(variant, ctx, vctx)
    tol = sdf_kats["extrema_tolerance"]
    for case in sdf_kats["extrema_cases"]:
        occ = kat_occupancy(case)
        sdf, lo, hi = ctx.sdf_from_occupancy(occ, case["resolution"])
        exp_lo, exp_hi = float(case["min"]), float(case["max"])
        assert lo == exp_lo or abs(lo - exp_lo) <= tol, case["name"]
        assert hi == exp_hi or abs(hi - exp_hi) <= tol, case["name"]
        assert np.all(sdf[occ >= 0.5] < 0) and np.all(sdf[occ < 0.5] > 0), case["name"]
    for case in sdf_kats["exact_cases"]:
        occ = kat_occupancy(case)
        sdf, _, _ = ctx.sdf_from_occupancy(occ, case["resolution"])
        sq = np.array(case["expected_sq"], dtype=np.float32)
        expected = (np.sign(sq) * np.sqrt(np.abs(sq))).astype(np.float32).reshape(case["shape"])
        assert bits_equal(sdf, expected), case["name"]


@pytest.mark.parametrize("variant", VARIANTS)
def test_independent_edt_fixtures(ctx, vctx, sdf_scipy_cases, variant):
    ctx = context_for(variant, ctx, vctx)
    for name, c in sdf_scipy_cases.items():
        sdf, lo, hi = ctx.sdf_from_occupancy(c["occ"], float(c["res"]), bool(c["uif"]), False)
        assert bits_equal(sdf, c["sdf"]), name
        assert lo == c["sdf"].min() and hi == c["sdf"].max(), name
        vb, lo, hi = ctx.sdf_from_occupancy(c["occ"], float(c["res"]), bool(c["uif"]), True)
        assert bits_equal(vb, c["sdf_vb"]), name + " (virtual border)"
        assert lo == c["sdf_vb"].min() and hi == c["sdf_vb"].max(), name


SHAPES = [(1, 1, 1), (1, 1, 2), (2, 1, 1), (1, 70, 1), (3, 5, 64), (3, 5, 65), (7, 9, 130),
          (40, 33, 17), (64, 64, 64), (31, 130, 67), (130, 31, 67), (96, 80, 200)]


@pytest.mark.parametrize("shape", SHAPES)
def test_random_grids_vs_oracle(ctx, oracle, shape):
    rng = np.random.default_rng(abs(hash(shape)) % (2 ** 31))
    for p in (0.002, 0.05, 0.5, 0.97):
        occ = (rng.random(shape) < p).astype(np.float32)
        occ[rng.random(shape) < 0.02] = 0.5
        for uif in (True, False):
            got, lo, hi = ctx.sdf_from_occupancy(occ, 0.013, uif, False)
            want, wlo, whi = oracle.sdf_from_occupancy(occ, 0.013, uif, False)
            assert bits_equal(got, want), (shape, p, uif)
            assert (lo, hi) == (wlo, whi)


SHORT_SHAPES = [(1, 1, 1), (2, 7, 5), (8, 8, 8), (9, 16, 64), (17, 24, 65), (25, 33, 130), (32, 48, 40), (49, 63, 7),
                (64, 64, 64), (63, 65, 66), (65, 64, 3), (40, 40, 40), (64, 200, 20), (300, 48, 70), (100, 128, 64),
                (128, 96, 130), (97, 127, 5), (129, 128, 9)]


@pytest.mark.parametrize("shape", SHORT_SHAPES)
def test_short_line_kernels_and_sweeps_agree(vctx, oracle, shape):
    """Lines of at most 64 rows take the short-line kernels (csrc/edt_short_kernels.hip: whole line in registers,
    exhaustive search), longer ones the sweeps.  The testing library can move the limit: every shape here is extracted
    with the limit at 64 (the product's setting: short kernels up to 64 rows, up to 128 when the launch has few items),
    at 128 (every length up to 128 whatever the item count), at 0 (sweeps only) and at 24 (mixed), and all four must
    equal the oracle bit for bit -- extents around every instantiation's size (8, 16, 24, 32, 48, 64, 96, 128)."""
    rng = np.random.default_rng(sum(shape))
    vctx.set_edt_variant(0)
    try:
        for kind in ("spheres", "salt", "unknown_mix", "single", "full"):
            occ = synthetic.make_occupancy(shape, kind, seed=int(rng.integers(1, 1000)))
            for uif, vb in ((True, False), (False, True)):
                want, wlo, whi = oracle.sdf_from_occupancy(occ, 0.0213, uif, vb)
                for rows in (64, 128, 0, 24):
                    vctx.set_short_line_rows(rows)
                    got, lo, hi = vctx.sdf_from_occupancy(occ, 0.0213, uif, vb)
                    assert bits_equal(got, want), (shape, kind, uif, vb, rows)
                    assert (lo, hi) == (wlo, whi), (shape, kind, rows)
    finally:
        vctx.set_short_line_rows(-1)


# launches of fewer items than workgroup slots on lines of 128 rows and more (1100-row lines take 64-bit stack entries), and
# lines of 768 - 1024 rows with few items: smooth scenes and noisy ones
SWEEP_SHAPES = [(200, 150, 70), (129, 300, 64), (512, 130, 40), (128, 128, 128), (260, 1100, 20), (1100, 140, 65),
                (768, 9, 70), (769, 40, 64), (800, 33, 5), (1000, 20, 66), (1024, 12, 130), (1023, 300, 20)]


@pytest.mark.parametrize("shape", SWEEP_SHAPES)
def test_sweeps_on_few_item_and_long_line_shapes(ctx, oracle, shape):
    """The lane-per-line sweeps (csrc/edt_sweep_kernels.hip) where their bookkeeping differs from the cube case: one-round
    launches, partial last bands and sign words, both entry widths, ring spills on long lines."""
    for kind, seed in (("spheres", 3), ("salt", 4), ("unknown_mix", 5), ("single", 0), ("full", 0)):
        occ = synthetic.make_occupancy(shape, kind, seed=seed)
        for uif, vb in ((True, False), (False, True)):
            want, wlo, whi = oracle.sdf_from_occupancy(occ, 0.0171, uif, vb)
            got, lo, hi = ctx.sdf_from_occupancy(occ, 0.0171, uif, vb)
            assert bits_equal(got, want), (shape, kind, uif, vb)
            assert (lo, hi) == (wlo, whi), (shape, kind)


def test_degenerate_grids(ctx, oracle):
    for kind in ("empty", "full", "single"):
        occ = synthetic.occupancy_degenerate((48, 20, 70), kind)
        for vb in (False, True):
            got, lo, hi = ctx.sdf_from_occupancy(occ, 0.01, True, vb)
            want, wlo, whi = oracle.sdf_from_occupancy(occ, 0.01, True, vb)
            assert bits_equal(got, want), (kind, vb)
            assert (lo, hi) == (wlo, whi)
    got, lo, hi = ctx.sdf_from_occupancy(np.zeros((5, 6, 7), np.float32), 0.25)
    assert np.all(np.isposinf(got)) and np.isposinf(lo) and np.isposinf(hi)
    got, lo, hi = ctx.sdf_from_occupancy(np.ones((5, 6, 7), np.float32), 0.25)
    assert np.all(np.isneginf(got)) and np.isneginf(lo) and np.isneginf(hi)


@pytest.mark.parametrize("shape", [(40, 40, 40), (50, 50, 50), (64, 64, 64), (70, 64, 64)],
                         ids=["256KB-kernels-on-the-ring", "500KB-kernels-on-the-ring", "1MB-ring-and-DMA", "1.1MB-page-locked"])
def test_small_maps_back_to_back_with_changing_content(ctx, oracle, shape):
    """The host entry point moves small maps through ONE page-locked ring that every call reuses (up to 512 KiB the
    kernels read the map from it and write the field to it directly): calls of one shape with different content, one
    right behind the other, must each see their own map and return their own field -- nothing cached from the call
    before on either side of the link."""
    rng = np.random.default_rng(11)
    maps = [np.ascontiguousarray((rng.random(shape) < p).astype(np.float32)) for p in (0.002, 0.3, 0.0005, 0.9, 0.05, 0.5)]
    maps.append(np.zeros(shape, np.float32))
    maps.append(np.ones(shape, np.float32))
    want = [oracle.sdf_from_occupancy(m, 0.02) for m in maps]
    for rnd in range(3):
        for k in ([0, 1, 2, 3, 4, 5, 6, 7], [7, 3, 3, 0, 6, 1, 5, 2], [4, 4, 6, 7, 0, 2, 1, 3])[rnd]:
            got, lo, hi = ctx.sdf_from_occupancy(maps[k], 0.02)
            assert bits_equal(got, want[k][0]), (shape, rnd, k)
            assert (lo, hi) == want[k][1:], (shape, rnd, k)


def test_variants_agree_on_synthetic_distributions(ctx, vctx, oracle):
    shape = (72, 96, 160)
    for dist in ("spheres", "salt", "unknown_mix"):
        occ = synthetic.make_occupancy(shape, dist, seed=42)
        want, _, _ = oracle.sdf_from_occupancy(occ, 0.01)
        for variant in VARIANTS:
            got, _, _ = context_for(variant, ctx, vctx).sdf_from_occupancy(occ, 0.01)
            assert bits_equal(got, want), (dist, variant)
        # the testing library's own build of the default pipeline, too
        vctx.set_edt_variant(0)
        got, _, _ = vctx.sdf_from_occupancy(occ, 0.01)
        assert bits_equal(got, want), (dist, "testing library, default")


def test_mask_entry_point(ctx, oracle):
    """vgt_hip_sdf_from_mask_u8: the path for map types whose predicate is evaluated on the host."""
    rng = np.random.default_rng(11)
    # Z extents that take the generic scan (55), the four-voxels-per-lane scan with 1, 2, 4 and 8 chunks per line
    # (56, 260, 1024, 2048) and its partial last chunk (260)
    for shape, p in (((21, 34, 55), 0.1), ((21, 34, 56), 0.1), ((5, 9, 260), 0.02), ((9, 7, 1024), 0.004),
                     ((3, 3, 2048), 0.002), ((2, 2, 2048), 0.0)):
        mask = (rng.random(shape) < p).astype(np.uint8)
        if p == 0.0:
            mask[0, 0, 0] = 1
            mask[1, 1, 2047] = 1
        got, lo, hi = ctx.sdf_from_mask(mask, 0.05)
        want = oracle.sdf_from_mask(mask, 0.05)
        assert bits_equal(got, want), shape
        assert lo == want.min() and hi == want.max()


def test_pipelined_host_entry_point(vctx, oracle):
    """Large grids through the host-pointer entry points are uploaded in X chunks, scanned and swept along Y chunk
    by chunk, swept along X in ranges of Y and downloaded range by range on separate streams.  The size threshold
    is lowered so that small grids take that path: ragged chunk sizes, the virtual border (whose Y coordinate is
    the range's offset + the tile's), masks, and a grid just below the minimum extents (falls back).  The threshold is
    a hook of the testing library (the product pipelines from 2^27 voxels: bench.py's host_path runs that at 1024^3 and
    checks that pageable and pinned calls give the same field)."""
    ctx = vctx
    ctx.set_edt_variant(0)
    ctx.set_host_pipeline_min_voxels(1)
    try:
        _pipelined_host_cases(ctx, oracle)
    finally:
        ctx.set_host_pipeline_min_voxels(2 ** 27)


def _pipelined_host_cases(ctx, oracle):
    rng = np.random.default_rng(77)
    for shape in ((32, 32, 8), (33, 47, 20), (100, 61, 36), (70, 130, 17), (31, 64, 12), (64, 600, 8), (1030, 40, 8)):
        occ = (rng.random(shape) < 0.03).astype(np.float32)
        occ[rng.random(shape) < 0.01] = 0.5
        for vb in (False, True):
            want, wlo, whi = oracle.sdf_from_occupancy(occ, 0.07, True, vb)
            got, lo, hi = ctx.sdf_from_occupancy(occ, 0.07, True, vb)
            assert bits_equal(got, want), (shape, vb)
            assert (lo, hi) == (wlo, whi), (shape, vb)
        mask = (occ > 0.4).astype(np.uint8)
        got, lo, hi = ctx.sdf_from_mask(mask, 0.07)
        want = oracle.sdf_from_mask(mask, 0.07)
        assert bits_equal(got, want), shape
        assert lo == want.min() and hi == want.max()
    # the same call with the pipeline turned off gives the same field
    ctx.set_host_pipeline_min_voxels(-1)
    occ = (rng.random((100, 61, 36)) < 0.03).astype(np.float32)
    plain = ctx.sdf_from_occupancy(occ, 0.07, True, True)
    ctx.set_host_pipeline_min_voxels(1)
    piped = ctx.sdf_from_occupancy(occ, 0.07, True, True)
    assert bits_equal(plain[0], piped[0]) and plain[1:] == piped[1:]


def test_argument_errors(ctx):
    with pytest.raises(ValueError):
        ctx.sdf_from_occupancy(np.zeros((4, 4, 4), np.float32), 0.0)      # non-positive resolution
    with pytest.raises(ValueError):
        ctx.sdf_from_occupancy(np.zeros((4, 4, 4), np.float32), float("nan"))
    with pytest.raises(ValueError):
        ctx.sdf_from_occupancy(np.zeros((0, 4, 4), np.float32), 0.1)      # empty grid


def test_full_size_properties(ctx):
    """512^3 (BASELINE config C2) through the host entry point, checked by size-independent
    properties: sign follows occupancy, |sdf| >= res, a sphere's interior/exterior distances
    are analytic, and translating the scene by whole voxels translates the field."""
    n = 512
    res = 0.01
    occ = np.zeros((n, n, n), dtype=np.float32)
    c, r = np.array([200, 260, 310]), 57.0
    ax = np.arange(n)
    d2 = ((ax - c[0]) ** 2)[:, None, None] + ((ax - c[1]) ** 2)[None, :, None] + ((ax - c[2]) ** 2)[None, None, :]
    occ[d2 <= r * r] = 1.0
    sdf, lo, hi = ctx.sdf_from_occupancy(occ, res)
    assert np.all(sdf[occ > 0.5] < 0) and np.all(sdf[occ <= 0.5] > 0)
    assert np.min(np.abs(sdf)) >= np.float32(res)
    assert lo == sdf.min() and hi == sdf.max()
    # exact EDT of a digital ball is within one voxel diagonal of the analytic distance
    analytic = (np.sqrt(d2) - r) * res
    assert np.max(np.abs(sdf - analytic)) <= 1.8 * res
    # translation by (8, -16, 24) voxels: interior region away from the borders must match
    occ2 = np.roll(occ, (8, -16, 24), axis=(0, 1, 2))
    sdf2, _, _ = ctx.sdf_from_occupancy(occ2, res)
    core = (slice(120, 300), slice(180, 340), slice(230, 390))
    moved = np.roll(sdf2, (-8, 16, -24), axis=(0, 1, 2))
    assert bits_equal(sdf[core], moved[core])


@pytest.mark.parametrize("nslabs", [2, 3, 8])
def test_z_slab_pipeline_matches_single_device(ctx, oracle, nslabs):
    """The multi-GPU path (slab scan + summaries -> carries -> fix-up -> Y/X passes), with all
    slabs run on this one device and the exchange done in-process, is bit-identical to the
    full-grid result and to the oracle."""
    import torch
    from voxelized_geometry_tools_amd import multi_gpu
    shape = (40, 56, 72)
    for dist, vb in (("spheres", False), ("salt", True), ("unknown_mix", False), ("single", True),
                     ("empty", False), ("full", False)):
        occ = synthetic.make_occupancy(shape, dist, seed=9)
        want, wlo, whi = oracle.sdf_from_occupancy(occ, 0.02, True, vb)
        occ_dev = torch.from_numpy(occ).cuda()
        got, lo, hi = multi_gpu.sdf_slabs_single_device(ctx, torch, occ_dev, nslabs, 0.02, True, vb)
        assert bits_equal(got.cpu().numpy(), want), (dist, nslabs)
        assert (lo, hi) == (wlo, whi), (dist, nslabs)


def _host_ram_ok(voxels, bytes_per_voxel=40.0):
    try:
        with open("/proc/meminfo") as fh:
            avail_kb = next(int(l.split()[1]) for l in fh if l.startswith("MemAvailable"))
    except (OSError, StopIteration):
        return False
    return bytes_per_voxel * voxels < 0.7 * avail_kb * 1024.0


@pytest.mark.parametrize("dist", ["spheres", "salt"])
def test_headline_config_bit_exact_vs_oracle(ctx, oracle, dist):
    """BASELINE config 4 exactly as bench.py times it (1024^3, D1 spheres seed 42 and D2 salt p = 0.01, res 0.01,
    device-resident) against the CPU oracle, every voxel bit for bit, and the extrema."""
    import torch
    import bench
    shape = (1024, 1024, 1024)
    if not _host_ram_ok(float(np.prod(shape))):
        pytest.skip("the oracle needs ~40 GiB of host RAM at 1024^3")
    occ = bench.device_occupancy(torch, shape, dist, 42, torch.device("cuda", 0))
    sdf = torch.empty(shape, dtype=torch.float32, device="cuda")
    nbytes = capi.sdf_workspace_bytes(shape)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    minmax = torch.zeros(2, dtype=torch.float32, device="cuda")
    ctx.set_stream(None)
    try:
        ctx.sdf_dev(occ.data_ptr(), shape, 0.01, sdf.data_ptr(), ws.data_ptr(), nbytes, minmax.data_ptr())
        torch.cuda.synchronize()
    finally:
        ctx.reset_stream()
    del ws
    occ_host = occ.cpu().numpy()
    del occ
    want, wlo, whi = oracle.sdf_from_occupancy(occ_host, 0.01)
    del occ_host
    got = sdf.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), dist
    mm = minmax.cpu().numpy()
    assert (float(mm[0]), float(mm[1])) == (wlo, whi)


@pytest.mark.parametrize("shape,nslabs", [((1100, 600, 384), 3), ((600, 1300, 96), 2), ((2048, 520, 64), 4)])
def test_z_slab_pipeline_long_axes(ctx, oracle, shape, nslabs):
    """The slab pipeline on long axes (lines of more than 1024 rows: 64-bit stack entries in the sweep passes; a global Z
    extent that decides the entry kind where the slab's own would not), all slabs on this device, against the oracle."""
    import torch
    from voxelized_geometry_tools_amd import multi_gpu
    if not _host_ram_ok(float(np.prod(shape))):
        pytest.skip("not enough host RAM for the oracle")
    for dist, vb in (("spheres", False), ("salt", True)):
        occ = synthetic.make_occupancy(shape, dist, seed=5)
        want, wlo, whi = oracle.sdf_from_occupancy(occ, 0.02, True, vb)
        got, lo, hi = multi_gpu.sdf_slabs_single_device(ctx, torch, torch.from_numpy(occ).cuda(), nslabs, 0.02, True, vb)
        assert bits_equal(got.cpu().numpy(), want), (dist, shape)
        assert (lo, hi) == (wlo, whi), (dist, shape)


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0], [0] * 7])
def test_multi_device_entry_point(oracle, devices):
    """vgt_hipx_sdf_multi (one process, one Z slab per listed device, host buffers) against the oracle.  On one
    GPU the slabs share device 0: [0] runs the RCCL communicator path with a single rank, repeated devices the
    slab-to-slab copies; more slabs than Z voxels are clamped."""
    for shape, dist, vb in (((33, 47, 70), "spheres", False), ((40, 24, 96), "unknown_mix", True),
                            ((9, 11, 5), "salt", False), ((16, 16, 40), "empty", False), ((1, 1, 3), "single", False)):
        occ = synthetic.make_occupancy(shape, dist, seed=21)
        want, wlo, whi = oracle.sdf_from_occupancy(occ, 0.03, True, vb)
        got, lo, hi = capi.sdf_multi(devices, occ, 0.03, True, vb)
        assert bits_equal(got, want), (shape, dist, devices)
        assert (lo, hi) == (wlo, whi), (shape, dist, devices)


def test_multi_device_argument_errors():
    occ = np.zeros((4, 4, 4), dtype=np.float32)
    with pytest.raises(ValueError):
        capi.sdf_multi([0], occ, 0.0)
    with pytest.raises(ValueError):
        capi.sdf_multi([], occ, 0.1)
    with pytest.raises(capi.VgtHipError):
        capi.sdf_multi([0, 4096], occ, 0.1)


@pytest.mark.parametrize("shape", [(40, 33, 2048), (40, 33, 2049), (3, 1024, 1775), (3, 1024, 1776), (1024, 3, 1775),
                                   (1024, 3, 1776), (1024, 64, 1774), (1024, 65, 1774), (3, 1025, 40), (1025, 3, 40),
                                   (3, 2048, 1449), (2048, 30, 1449)])
def test_packed_entry_limits(ctx, oracle, shape):
    """The sweep passes keep a stack entry (G = F + row^2, row) in one 32-bit word (22 + 10 bits) while the line has at
    most 1024 rows and every G stays below 2^22 - 3, i.e. max input + (n - 1)^2 < 4194301 with max input = (nz - 1)^2 in
    the Y pass and (nz - 1)^2 + (ny - 1)^2 in the X pass, and 64-bit entries beyond: shapes on both sides of the limits
    (1023^2 + 1774^2 < 4194301 <= 1023^2 + 1775^2; 1023^2 + 1773^2 + 63^2 < 4194301 <= 1023^2 + 1773^2 + 64^2; 1024
    and 1025 rows), with fields whose distances reach the largest magnitudes (a single site in a corner) and dense ones."""
    rng = np.random.default_rng(sum(shape))
    fields = []
    occ = np.zeros(shape, dtype=np.float32)
    occ[0, 0, 0] = 1.0
    fields.append(occ)
    occ = np.zeros(shape, dtype=np.float32)
    occ[-1, -1, -1] = 1.0
    occ[0, shape[1] // 2, 0] = 1.0
    fields.append(occ)
    fields.append((rng.random(shape) < 0.002).astype(np.float32))
    for occ in fields:
        want, wlo, whi = oracle.sdf_from_occupancy(occ, 0.05)
        got, lo, hi = ctx.sdf_from_occupancy(occ, 0.05)
        assert bits_equal(got, want), shape
        assert (lo, hi) == (wlo, whi)


@pytest.mark.parametrize("shape", [(1100, 6, 40), (5, 1500, 33), (2048, 4, 16), (3, 2049, 20),
                                   (2100, 3, 8), (4, 5, 1100), (2, 3, 2500)])
def test_long_axes(ctx, vctx, oracle, shape):
    """Long axes in every pipeline.  Default: lines of more than 1024 rows take 64-bit stack entries, Z lines of more
    than 1024 voxels the 32- and 64-word record kernels.  Cross-checks: axes in (1024, 2048] use the tiled envelope's
    16-line tiles, longer ones fall back to the pruned search, Z lines beyond 1024 use the generic int16 scan."""
    rng = np.random.default_rng(sum(shape))
    occ = (rng.random(shape) < 0.01).astype(np.float32)
    occ[rng.random(shape) < 0.005] = 0.5
    want, wlo, whi = oracle.sdf_from_occupancy(occ, 0.05)
    for variant in VARIANTS:
        got, lo, hi = context_for(variant, ctx, vctx).sdf_from_occupancy(occ, 0.05)
        assert bits_equal(got, want), (shape, variant)
        assert (lo, hi) == (wlo, whi)


@pytest.mark.parametrize("shape", [(16384, 2, 70), (3, 16384, 65), (2, 3, 16384)])
def test_longest_axis(ctx, oracle, shape):
    """The per-axis limit itself (16384 voxels: rows up to 16383, the bound the 64-bit stack entries, the "no site" hull
    point and the 24-bit multiplies of the sweeps are sized for), on every axis, with sites at the far ends, lines without
    any site and a sparse random field."""
    rng = np.random.default_rng(sum(shape))
    fields = []
    occ = np.zeros(shape, dtype=np.float32)
    occ[0, 0, 0] = 1.0
    fields.append(occ)
    occ = np.zeros(shape, dtype=np.float32)
    occ[-1, -1, -1] = 1.0
    occ[0, 0, shape[2] // 2] = 1.0
    fields.append(occ)
    occ = (rng.random(shape) < 0.0005).astype(np.float32)
    occ[:, 0, :] = 0.0  # (lines without any site next to lines with some)
    fields.append(occ)
    for occ in fields:
        want, wlo, whi = oracle.sdf_from_occupancy(occ, 0.05)
        got, lo, hi = ctx.sdf_from_occupancy(occ, 0.05)
        assert bits_equal(got, want), shape
        assert (lo, hi) == (wlo, whi)


@pytest.mark.parametrize("resolution", [0.01, 0.25, 1.0, 1.0 / 3.0, 0.1, 0.05, 2.5e-3, 7.0, 1.0e-20, 3.0e25, 1.0e-42])
def test_fast_finalize_matches_exact_for_every_d2(vctx, resolution):
    """The final conversion float(sqrt(double(d2)) * res) (signed_distance_field_generation.hpp:98-105) has a
    fast evaluation with an exact fallback; both are run on the device for EVERY d2 in [0, 2^31)."""
    bad, first = vctx.debug_finalize_check(0, 2 ** 31, resolution)  # (a hook of the testing library)
    assert bad == 0, "first mismatch at d2 = %s" % first


def test_exact_finalize_matches_numpy(ctx):
    """Anchors the device's exact conversion itself: a one-voxel-thick grid whose distances sweep a range of
    d2 values, compared with numpy's float32(sqrt(float64) * res)."""
    n = 1500
    occ = np.zeros((n, 1, 1), dtype=np.float32)
    occ[0, 0, 0] = 1.0
    for res in (0.01, 1.0 / 3.0, 0.123456789):
        sdf, _, _ = ctx.sdf_from_occupancy(occ, res)
        d2 = np.arange(n, dtype=np.float64) ** 2
        expect = (np.sqrt(d2) * res).astype(np.float32)
        expect[0] = -np.float32(np.sqrt(1.0) * res)
        assert bits_equal(sdf[:, 0, 0], expect)


def _check_few_site_field(torch, sdf, shape, sites, res):
    """A grid whose only filled voxels are a few isolated `sites` has an analytic field: float32(sqrt(float64(min over
    the sites of the squared point distance)) * res), and -res on the sites themselves (their nearest free voxel is a
    neighbour).  Checks EVERY voxel of the device tensor `sdf` against it, 64 X slices at a time; returns the largest
    value seen."""
    ay = torch.arange(shape[1], device="cuda", dtype=torch.int64)
    az = torch.arange(shape[2], device="cuda", dtype=torch.int64)
    worst = 0
    hi = 0.0
    for x0 in range(0, shape[0], 64):
        ax = torch.arange(x0, min(x0 + 64, shape[0]), device="cuda", dtype=torch.int64)
        d2 = None
        for (sx, sy, sz) in sites:
            d = ((ax - sx) ** 2)[:, None, None] + ((ay - sy) ** 2)[None, :, None] + ((az - sz) ** 2)[None, None, :]
            d2 = d if d2 is None else torch.minimum(d2, d)
        del d
        want = (torch.sqrt(d2.to(torch.float64)) * res).to(torch.float32)
        got = sdf[x0:x0 + 64]
        want = torch.where(d2 == 0, torch.full_like(want, -float(np.float32(res))), want)
        del d2
        diff = (got.view(torch.int32) - want.view(torch.int32)).abs().max().item()
        worst = max(worst, diff)
        hi = max(hi, got.max().item())
        del want
    assert worst == 0
    return hi


def test_two_giga_voxel_slab_is_indexed_with_64_bits(ctx):
    """One rank's share of BASELINE config 5 at two GPUs (2048 x 2048 x 512 = 2^31 voxels), device-resident.
    Three filled voxels in far corners make the field analytic (min over three point distances), so every
    voxel is checked, chunk by chunk, against float32(sqrt(float64(d2)) * res) computed with torch."""
    import torch
    free = torch.cuda.mem_get_info()[0]
    if free < 70 * 2 ** 30:
        pytest.skip("needs ~60 GiB of free HBM")
    shape = (2048, 2048, 512)
    res = 0.01
    sites = [(2047, 2047, 511), (0, 2040, 3), (1999, 1, 500)]
    occ = torch.zeros(shape, dtype=torch.float32, device="cuda")
    for s in sites:
        occ[s] = 1.0
    sdf = torch.empty(shape, dtype=torch.float32, device="cuda")
    nbytes = capi.sdf_workspace_bytes(shape)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    minmax = torch.zeros(2, dtype=torch.float32, device="cuda")
    ctx.set_stream(None)
    try:
        ctx.sdf_dev(occ.data_ptr(), shape, res, sdf.data_ptr(), ws.data_ptr(), nbytes, minmax.data_ptr())
        torch.cuda.synchronize()
    finally:
        ctx.reset_stream()
    del ws, occ
    hi = _check_few_site_field(torch, sdf, shape, sites, res)
    mm = minmax.cpu().numpy()
    assert mm[0] == -np.float32(res) and mm[1] == np.float32(hi)


# BASELINE config 5 at its OWN size: 2048 x 2048 x 1024 = 2^32 voxels -- exactly where a 32-bit voxel index wraps
# (the reference's device kernels index with int32: S/cuda_voxelization_helpers.cu:85,365,377).  Sites sit on the
# last voxel (linear index 2^32 - 1), on index 2^31 exactly, just above 2^31 and 2^32 - 2^22, in the first and the last
# Z slab of an 8-way cut and in the middle ones.
CONFIG5_SHAPE = (2048, 2048, 1024)
CONFIG5_SITES = [(2047, 2047, 1023), (1024, 0, 0), (1024, 1, 130), (2046, 2047, 900), (0, 2040, 3), (1999, 1, 1000),
                 (600, 1000, 517), (1400, 700, 300)]


@pytest.mark.parametrize("path", ["plain", "slabs8"])
def test_config5_at_its_own_size(ctx, path):
    """Every voxel of a 2^32-voxel field, device-resident, through the plain pipeline (the one-GPU reference of the
    scaling series) and through the Z-slab pipeline with the 8 slabs of the 8-GPU run, all on this device
    (begin -> summaries of all slabs -> carries -> finish per slab), against the analytic few-site field; extrema too."""
    import torch
    from voxelized_geometry_tools_amd import multi_gpu
    shape, sites, res = CONFIG5_SHAPE, CONFIG5_SITES, 0.01
    n = int(np.prod(shape))
    assert n == 2 ** 32 and all(0 <= c < e for s in sites for c, e in zip(s, shape))
    linear = sorted((x * shape[1] + y) * shape[2] + z for x, y, z in sites)
    assert linear[-1] == 2 ** 32 - 1 and 2 ** 31 in linear and sum(i >= 2 ** 31 for i in linear) >= 4
    ws_bytes = capi.sdf_workspace_bytes(shape)
    slab_ws = 8 * capi.sdf_workspace_bytes((shape[0], shape[1], shape[2] // 8))
    # occupancy + field + workspace(s) + the checker's temporaries (+ the slabs' contiguous copies and fields)
    need = 8 * n + (ws_bytes if path == "plain" else slab_ws + 8 * n) + 12 * 2 ** 30
    free = torch.cuda.mem_get_info()[0]
    if free < need:
        pytest.skip("needs %.0f GiB of free HBM, %.0f free" % (need / 2 ** 30, free / 2 ** 30))
    occ = torch.zeros(shape, dtype=torch.float32, device="cuda")
    for s in sites:
        occ[s] = 1.0
    if path == "plain":
        sdf = torch.empty(shape, dtype=torch.float32, device="cuda")
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
        minmax = torch.zeros(2, dtype=torch.float32, device="cuda")
        ctx.set_stream(None)
        try:
            ctx.sdf_dev(occ.data_ptr(), shape, res, sdf.data_ptr(), ws.data_ptr(), ws_bytes, minmax.data_ptr())
            torch.cuda.synchronize()
        finally:
            ctx.reset_stream()
        del ws, occ
        lo, hi_reported = (float(v) for v in minmax.cpu().numpy())
    else:
        sdf, lo, hi_reported = multi_gpu.sdf_slabs_single_device(ctx, torch, occ, 8, res)
        del occ
    torch.cuda.empty_cache()
    hi = _check_few_site_field(torch, sdf, shape, sites, res)
    assert lo == -np.float32(res) and hi_reported == np.float32(hi)
    # spot values at the index boundaries, spelled out: the last voxel is a site; its Z neighbour is one voxel away
    assert sdf[2047, 2047, 1023].item() == -np.float32(res) and sdf[2047, 2047, 1022].item() == np.float32(res)
    assert sdf[1024, 0, 0].item() == -np.float32(res) and sdf[1023, 2047, 1023].item() > 0
    del sdf
    torch.cuda.empty_cache()


@pytest.mark.parametrize("shape", [(5, 6, 1024), (4, 3, 516), (3, 4, 260), (2, 2, 2048), (2, 3, 768), (3, 2, 1028),
                                   (6, 5, 4), (2, 2, 252), (3, 3, 1023), (2, 2, 2044)])
def test_z_scan_chunkings(ctx, oracle, shape):
    """Every chunk count of the four-voxels-per-lane Z scan (nz % 4 == 0: 1, 2, 4, 8 chunks of 256 voxels, full and
    ragged last chunk) and the one-voxel-per-lane kernels it falls back to, for occupancy and mask inputs, with
    long runs, isolated voxels and empty / full lines."""
    rng = np.random.default_rng(sum(shape) * 7 + shape[2])
    nx, ny, nz = shape
    occ = np.zeros(shape, dtype=np.float32)
    for x in range(nx):
        for y in range(ny):
            kind = (x * ny + y) % 5
            if kind == 0:
                occ[x, y] = (rng.random(nz) < 0.01)
            elif kind == 1:
                occ[x, y] = (rng.random(nz) < 0.6)
            elif kind == 2:
                a, b = sorted(rng.integers(0, nz, size=2))
                occ[x, y, a:b + 1] = 1.0                       # one long run
            elif kind == 3:
                occ[x, y, :] = 1.0 if (x + y) % 2 else 0.0     # full / empty line
            else:
                occ[x, y, rng.integers(0, nz)] = 1.0           # a single voxel
                occ[x, y, nz - 1] = 1.0
    want, wlo, whi = oracle.sdf_from_occupancy(occ, 0.03)
    got, lo, hi = ctx.sdf_from_occupancy(occ, 0.03)
    assert bits_equal(got, want), shape
    assert (lo, hi) == (wlo, whi)
    got, _, _ = ctx.sdf_from_mask((occ > 0.5).astype(np.uint8), 0.03)
    assert bits_equal(got, want), shape


def test_deferred_kernel_timing(ctx):
    """vgt_hip_timing_start / _stop: per-call kernel durations without per-call synchronisation."""
    import torch
    shape = (128, 128, 128)
    occ = torch.zeros(shape, dtype=torch.float32, device="cuda")
    occ[40:60, 50:70, 30:90] = 1.0
    sdf = torch.empty(shape, dtype=torch.float32, device="cuda")
    nbytes = capi.sdf_workspace_bytes(shape)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    ctx.set_stream(None)
    try:
        ctx.timing_start(3)
        for _ in range(5):                                   # two calls beyond the capacity are not recorded
            ctx.sdf_dev(occ.data_ptr(), shape, 0.01, sdf.data_ptr(), ws.data_ptr(), nbytes)
        ms = ctx.timing_stop()
    finally:
        ctx.reset_stream()
    assert ms.shape == (3, 3)
    assert np.all(ms > 0.0) and np.all(ms < 50.0)
    with pytest.raises(ValueError):
        ctx.timing_start(0)


def test_slab_carry_kernel_matches_torch_reduction(ctx):
    """vgt_hip_sdf_slab_carries_dev vs multi_gpu.carries_from_summaries (the torch restatement the gloo tests use), on
    packed 4-byte summaries of random slabs with an uneven split (203 = 5 x 40 + 3)."""
    import torch
    from voxelized_geometry_tools_amd import multi_gpu
    rng = np.random.default_rng(17)
    world, nx, ny, nz = 5, 23, 31, 203
    lines = nx * ny
    filled = rng.random((nx, ny, nz)) < 0.02
    filled[rng.random((nx, ny)) < 0.3] = False          # many lines without any filled voxel
    filled[rng.random((nx, ny)) < 0.05] = True          # some without any free one
    records = []
    for r in range(world):
        local_shape, z0 = multi_gpu.slab_of((nx, ny, nz), r, world)
        records.append(multi_gpu.summary_reference(filled[:, :, z0:z0 + local_shape[2]], z0))
    gathered = torch.from_numpy(np.stack(records)).cuda()
    ctx.set_stream(None)
    try:
        for rank in range(world):
            want = multi_gpu.carries_from_summaries(torch, gathered, rank, nz)
            got = torch.empty((lines, 4), dtype=torch.int16, device="cuda")
            ctx.sdf_slab_carries(gathered.data_ptr(), world, rank, nx, ny, nz, got.data_ptr())
            torch.cuda.synchronize()
            assert torch.equal(got, want), rank
    finally:
        ctx.reset_stream()


def test_slab_finish_rejects_carries_of_another_slab(ctx):
    """The carries are decoded with the slab ranges of vgt_hip_sdf_slab_range: finishing a slab that is not the one the
    carries were computed for (a custom or uneven split) is an argument error, not a silently wrong field."""
    import torch
    from voxelized_geometry_tools_amd import multi_gpu
    nx, ny, nz, world = 6, 7, 40, 3
    occ = torch.zeros((nx, ny, nz), dtype=torch.float32, device="cuda")
    occ[2, 3, 17] = 1.0
    ctx.set_stream(None)
    try:
        gathered = torch.empty((world, nx * ny, 2), dtype=torch.int16, device="cuda")
        slabs = []
        for r in range(world):
            local_shape, z0 = multi_gpu.slab_of((nx, ny, nz), r, world)
            local = occ[:, :, z0:z0 + local_shape[2]].contiguous()
            nbytes = capi.sdf_workspace_bytes(local_shape)
            ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
            ctx.sdf_slab_begin(local.data_ptr(), local_shape, z0, ws.data_ptr(), nbytes, gathered[r].data_ptr(), True)
            slabs.append((local_shape, z0, ws, nbytes))
        carries = torch.empty((nx * ny, 4), dtype=torch.int16, device="cuda")
        ctx.sdf_slab_carries(gathered.data_ptr(), world, 1, nx, ny, nz, carries.data_ptr())
        local_shape, z0, ws, nbytes = slabs[1]
        sdf = torch.empty(local_shape, dtype=torch.float32, device="cuda")
        with pytest.raises(ValueError, match="another slab"):  # the carries of rank 1 for a slab that starts elsewhere
            ctx.sdf_slab_finish(local_shape, z0 + 1, nz, 0.1, carries.data_ptr(), sdf.data_ptr(), ws.data_ptr(), nbytes)
        ctx.sdf_slab_finish(local_shape, z0, nz, 0.1, carries.data_ptr(), sdf.data_ptr(), ws.data_ptr(), nbytes)
        torch.cuda.synchronize()
    finally:
        ctx.reset_stream()


def _records_reference(filled, mark_no_site):
    """numpy restatement of the class-record format (csrc/vgt_internal.hpp): filled bool [nx, ny, nz] ->
    uint32 [nx, nwords, ny, 4] = (mask_lo, mask_hi, below2, above2)."""
    bias, none_below, none_above, no_site = 1 << 17, 0, 2 << 17, 0xFFFFFFF0
    nx, ny, nz = filled.shape
    nwords = (nz + 63) // 64
    padded = np.concatenate([filled, np.repeat(filled[:, :, -1:], nwords * 64 - nz, axis=2)], axis=2)
    bits = padded.reshape(nx, ny, nwords, 64).astype(np.uint64)
    weights = (np.uint64(1) << np.arange(64, dtype=np.uint64))
    masks = (bits * weights).sum(axis=3, dtype=np.uint64)                       # [nx, ny, nwords]
    out = np.zeros((nx, nwords, ny, 4), dtype=np.uint32)
    out[..., 0] = (masks & np.uint64(0xFFFFFFFF)).astype(np.uint32).transpose(0, 2, 1)
    out[..., 1] = (masks >> np.uint64(32)).astype(np.uint32).transpose(0, 2, 1)
    trans = filled[:, :, :-1] != filled[:, :, 1:] if nz > 1 else np.zeros((nx, ny, 0), dtype=bool)
    t = np.arange(max(nz - 1, 0))
    last = np.maximum.accumulate(np.where(trans, t, -1), axis=2) if nz > 1 else trans.astype(np.int64)
    nxt = np.minimum.accumulate(np.where(trans, t, 1 << 30)[:, :, ::-1], axis=2)[:, :, ::-1] if nz > 1 else last
    any_transition = trans.any(axis=2)
    for w in range(nwords):
        begin = 64 * w
        below = last[:, :, begin - 1] if begin >= 1 and nz > 1 else np.full((nx, ny), -1)
        above = nxt[:, :, begin + 63] if begin + 63 <= nz - 2 else np.full((nx, ny), 1 << 30)
        b2 = np.where(below < 0, none_below, 2 * (below - begin) + bias)
        a2 = np.where(above >= (1 << 30), none_above, 2 * (above - begin) + bias)
        if mark_no_site:
            a2 = np.where(any_transition, a2, no_site)
        out[:, w, :, 2] = b2.astype(np.uint32)
        out[:, w, :, 3] = a2.astype(np.uint32)
    return out


@pytest.mark.parametrize("shape", [(2, 3, 1), (3, 5, 63), (3, 5, 64), (2, 7, 65), (5, 3, 128), (2, 9, 200), (3, 4, 1000),
                                   (2, 6, 1024), (2, 5, 1025), (1, 3, 2048), (2, 2, 4096), (1, 3, 4097), (2, 2, 5000),
                                   (37, 41, 130)])
def test_class_records_against_the_format_definition(vctx, shape):
    """Pass 1 alone (a hook of the testing library): the records of every kernel geometry -- 1 to 64 words per line, the
    long-line kernel beyond 4096 voxels, partial last words, steps that straddle X planes -- against a numpy restatement
    of the record format, with and without the one-class marks, plus the slab summaries that come with the latter."""
    import torch
    from voxelized_geometry_tools_amd import multi_gpu
    rng = np.random.default_rng(sum(shape) * 7 + 1)
    nx, ny, nz = shape
    occ = np.zeros(shape, dtype=np.float32)
    runs = rng.random(shape) < (8.0 / max(nz, 8))                 # class changes every ~nz / 8 voxels
    occ[np.cumsum(runs, axis=2) % 2 == 1] = 1.0
    occ[rng.random((nx, ny)) < 0.3] = 0.0                         # lines of one class ...
    occ[rng.random((nx, ny)) < 0.1] = 1.0
    occ[rng.random(shape) < 0.01] = 0.5                           # ... and unknown cells (filled here)
    filled = occ >= 0.5
    occ_dev = torch.from_numpy(occ).cuda()
    nbytes = capi.load(testing=True).vgt_hip_testing_class_record_bytes(nx, ny, nz)
    nwords = (nz + 63) // 64
    for with_summary in (False, True):
        rec_dev = torch.zeros(nbytes // 4, dtype=torch.int32, device="cuda")
        summary = torch.zeros((nx * ny, 2), dtype=torch.int16, device="cuda") if with_summary else None
        vctx.class_records(occ_dev.data_ptr(), shape, True, 5 if with_summary else 0, rec_dev.data_ptr(),
                           summary.data_ptr() if with_summary else None)
        got = rec_dev.cpu().numpy().view(np.uint32)[:nx * nwords * ny * 4].reshape(nx, nwords, ny, 4)
        want = _records_reference(filled, mark_no_site=not with_summary)
        assert np.array_equal(got, want), (shape, with_summary, np.argwhere(got != want)[:5])
        if with_summary:
            assert np.array_equal(summary.cpu().numpy(), multi_gpu.summary_reference(filled, 5)), shape
