"""(gpu) SDF entry points of the map types with tagged cells (vgt_hip_cells_*): the HIP path
against the committed scipy fixture and, on larger random grids, against the oracle.  Bit-exact."""
import numpy as np
import pytest

from conftest import bits_equal, tagged_records
from voxelized_geometry_tools_amd import capi

pytestmark = pytest.mark.gpu

DTYPES = [capi.TAGGED_OBJECT_CELL, capi.TAGGED_OBJECT_COMPONENT_CELL]


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as O
    return O


@pytest.mark.parametrize("dtype", DTYPES, ids=["tagged8", "tagged16"])
def test_fixture_objects_to_use_and_free_and_named(ctx, sdf_tagged_cases, dtype):
    for name, case in sdf_tagged_cases.items():
        rec = tagged_records(case, dtype)
        res = float(case["res"])
        cells = ctx.cells(rec, rec.shape)
        assert np.array_equal(cells.object_ids(), case["object_ids"]), name
        for uif in (0, 1):
            tag = "uif%d__" % uif
            got, lo, hi = cells.sdf(res, (), bool(uif))
            assert bits_equal(got, case[tag + "all"]), (name, uif)
            assert (lo, hi) == (got.min(), got.max())
            got, _, _ = cells.sdf(res, (), bool(uif), True)
            assert bits_equal(got, case[tag + "all_vb"]), (name, uif)
            for k in range(4):
                got, lo, hi = cells.sdf(res, case[tag + "objs%d" % k], bool(uif))
                assert bits_equal(got, case[tag + "sdf%d" % k]), (name, uif, k)
                assert (lo, hi) == (got.min(), got.max())
            got, lo, hi = cells.free_and_named_objects_sdf(res, bool(uif))
            assert bits_equal(got, case[tag + "free_and_named"]), (name, uif)
            assert (lo, hi) == (got.min(), got.max())
        cells.close()


def _kat_records(occ, dtype):
    """The reference test's cells for an occupancy scene (test/sdf_generation_test.cpp:279-294, 387-418): filled cells
    carry object id 1, empty ones 0; components / segments are left at their constructors' 0."""
    rec = np.zeros(occ.shape, dtype=dtype)
    rec["occupancy"] = occ
    if "object_id" in dtype.names:
        rec["object_id"] = (occ != 0.0).astype(np.uint32)
    return rec


KAT_TYPES = [(capi.OCCUPANCY_COMPONENT_CELL, -1), (capi.TAGGED_OBJECT_CELL, 4), (capi.TAGGED_OBJECT_COMPONENT_CELL, 4)]


@pytest.mark.parametrize("dtype,id_offset", KAT_TYPES, ids=["component8", "tagged8", "tagged16"])
def test_reference_known_answers_on_every_map_type(ctx, sdf_kats, dtype, id_offset):
    """test/sdf_generation_test.cpp asserts its five extrema cases, the sign of every voxel (:140-256) and the three
    exact-value grids (:586-1055) on all FOUR map types; tests/test_gpu_sdf.py replays them through the OccupancyMap
    entry point, this test through vgt_hip_cells_sdf with the other three cell layouts (empty object list, as
    GenerateSignedDistanceFields calls them, :72-81)."""
    from conftest import kat_occupancy
    tol = sdf_kats["extrema_tolerance"]
    for case in sdf_kats["extrema_cases"]:
        occ = kat_occupancy(case)
        cells = ctx.cells(_kat_records(occ, dtype), occ.shape, object_id_offset=id_offset)
        sdf, lo, hi = cells.sdf(case["resolution"])
        cells.close()
        exp_lo, exp_hi = float(case["min"]), float(case["max"])
        assert lo == exp_lo or abs(lo - exp_lo) <= tol, case["name"]
        assert hi == exp_hi or abs(hi - exp_hi) <= tol, case["name"]
        assert np.all(sdf[occ >= 0.5] < 0) and np.all(sdf[occ < 0.5] > 0), case["name"]
        # and the same field as the OccupancyMap entry point, bit for bit
        plain, plo, phi = ctx.sdf_from_occupancy(occ, case["resolution"])
        assert bits_equal(sdf, plain) and (lo, hi) == (plo, phi), case["name"]
    for case in sdf_kats["exact_cases"]:
        occ = kat_occupancy(case)
        cells = ctx.cells(_kat_records(occ, dtype), occ.shape, object_id_offset=id_offset)
        sdf, _, _ = cells.sdf(case["resolution"])
        cells.close()
        sq = np.array(case["expected_sq"], dtype=np.float32)
        expected = (np.sign(sq) * np.sqrt(np.abs(sq))).astype(np.float32).reshape(case["shape"])
        assert bits_equal(sdf, expected), case["name"]


def test_component_map(ctx, sdf_tagged_cases):
    for name, case in sdf_tagged_cases.items():
        rec = tagged_records(case, capi.OCCUPANCY_COMPONENT_CELL)
        cells = ctx.cells(rec, rec.shape, object_id_offset=-1)
        got, _, _ = cells.sdf(float(case["res"]))
        assert bits_equal(got, case["uif1__all"]), name
        assert cells.object_ids().size == 0
        with pytest.raises(ValueError):
            cells.sdf(float(case["res"]), [1])
        with pytest.raises(ValueError):
            cells.free_and_named_objects_sdf(float(case["res"]))
        cells.close()


def test_all_object_sdfs_vs_oracle(ctx, oracle):
    """MakeAllObjectSDFs on a 96 x 64 x 80 scene of box-shaped objects, ids up to 0xffffffff, long object lists
    (bisection path) and duplicated ids in the list."""
    rng = np.random.default_rng(77)
    shape = (96, 64, 80)
    rec = np.zeros(shape, dtype=capi.TAGGED_OBJECT_CELL)
    ids = [1, 2, 5, 40, 41, 1000, 70000, 0xffffffff]
    for oid in ids:
        lo = [int(rng.integers(0, s - 12)) for s in shape]
        ext = [int(rng.integers(3, 12)) for _ in shape]
        box = tuple(slice(a, a + e) for a, e in zip(lo, ext))
        rec["occupancy"][box] = 1.0
        rec["object_id"][box] = oid
    rec["occupancy"][rng.random(shape) < 0.002] = 0.5          # unknown cells of object 0
    cells = ctx.cells(rec, shape)
    found = cells.object_ids()
    assert np.array_equal(found, np.unique(rec["object_id"][rec["object_id"] > 0]))
    per_object = cells.all_object_sdfs(0.05)
    assert sorted(per_object) == [int(i) for i in found]
    for oid, (sdf, lo, hi) in per_object.items():
        want, wlo, whi = oracle.sdf_from_cells(rec, shape, 0.05, [oid])
        assert bits_equal(sdf, want), oid
        assert (lo, hi) == (wlo, whi)
    many = list(range(2, 60)) + [70000, 70000, 5]
    got, lo, hi = cells.sdf(0.05, many, unknown_is_filled=False, add_virtual_border=True)
    want, wlo, whi = oracle.sdf_from_cells(rec, shape, 0.05, many, False, True)
    assert bits_equal(got, want) and (lo, hi) == (wlo, whi)
    got, lo, hi = cells.free_and_named_objects_sdf(0.05, True, True)
    want, wlo, whi = oracle.free_and_named_objects_sdf(rec, shape, 0.05, True, True)
    assert bits_equal(got, want) and (lo, hi) == (wlo, whi)
    cells.close()


def test_argument_errors(ctx):
    rec = np.zeros((4, 4, 4), dtype=capi.TAGGED_OBJECT_CELL)
    with pytest.raises(ValueError):
        ctx.cells(rec, (4, 4, 4), object_id_offset=6)            # misaligned
    with pytest.raises(ValueError):
        ctx.cells(rec, (4, 4, 4), object_id_offset=8)            # outside the record
    cells = ctx.cells(rec, (4, 4, 4))
    with pytest.raises(ValueError):
        cells.sdf(0.0)                                            # resolution must be positive
    cells.close()


def test_object_ids_one_pass_many_ids(ctx):
    """vgt_hip_cells_object_ids collects the distinct ids in one pass (a device hash set): many ids, every cell its
    own id, the extreme id values, and a grid without any."""
    rng = np.random.default_rng(12)
    shape = (24, 20, 30)
    for kind in ("many", "all_distinct", "extremes", "none"):
        rec = np.zeros(shape, dtype=np.dtype([("occupancy", np.float32), ("object_id", np.uint32)]))
        rec["occupancy"] = (rng.random(shape) < 0.3).astype(np.float32)
        if kind == "many":
            rec["object_id"] = rng.integers(0, 5000, size=shape, dtype=np.uint32)
        elif kind == "all_distinct":
            rec["object_id"] = (np.arange(rec.size, dtype=np.uint32) * np.uint32(2654435761) | np.uint32(1)).reshape(shape)
        elif kind == "extremes":
            rec["object_id"] = rng.choice(np.array([0, 1, 2, 0x7fffffff, 0x80000000, 0xfffffffe, 0xffffffff], dtype=np.uint32),
                                          size=shape)
        cells = capi.Cells(ctx, rec, shape)
        want = np.unique(rec["object_id"])
        want = want[want > 0]
        got = cells.object_ids()
        assert got.dtype == np.uint32 and np.array_equal(got, want), kind
        cells.close()
