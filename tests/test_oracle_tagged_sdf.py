"""Oracle restatement of the tagged map types' SDF entry points against the committed
fixture built from the reference's definitions + scipy's EDT (tests/golden/make_golden.py).  The reference's
own tests hold no known answers for these methods, so this fixture is what pins them."""
import numpy as np
import pytest

from conftest import bits_equal, tagged_records
from oracle import oracle as O
from voxelized_geometry_tools_amd import capi

DTYPES = [capi.TAGGED_OBJECT_CELL, capi.TAGGED_OBJECT_COMPONENT_CELL]


@pytest.mark.parametrize("dtype", DTYPES, ids=["tagged8", "tagged16"])
def test_objects_to_use(sdf_tagged_cases, dtype):
    for name, case in sdf_tagged_cases.items():
        rec = tagged_records(case, dtype)
        res = float(case["res"])
        for uif in (0, 1):
            tag = "uif%d__" % uif
            got, lo, hi = O.sdf_from_cells(rec, rec.shape, res, (), bool(uif))
            assert bits_equal(got, case[tag + "all"]), (name, uif)
            assert (lo, hi) == (got.min(), got.max())
            got, _, _ = O.sdf_from_cells(rec, rec.shape, res, (), bool(uif), True)
            assert bits_equal(got, case[tag + "all_vb"]), (name, uif)
            for k in range(4):
                got, _, _ = O.sdf_from_cells(rec, rec.shape, res, case[tag + "objs%d" % k], bool(uif))
                assert bits_equal(got, case[tag + "sdf%d" % k]), (name, uif, k)


def test_component_map_ignores_the_component(sdf_tagged_cases):
    """OccupancyComponentMap::ExtractSignedDistanceField looks at the occupancy only."""
    for name, case in sdf_tagged_cases.items():
        rec = tagged_records(case, capi.OCCUPANCY_COMPONENT_CELL)
        got, _, _ = O.sdf_from_cells(rec, rec.shape, float(case["res"]), (), True, False, None, -1)
        assert bits_equal(got, case["uif1__all"]), name


@pytest.mark.parametrize("dtype", DTYPES, ids=["tagged8", "tagged16"])
def test_free_and_named_objects(sdf_tagged_cases, dtype):
    for name, case in sdf_tagged_cases.items():
        rec = tagged_records(case, dtype)
        for uif in (0, 1):
            got, lo, hi = O.free_and_named_objects_sdf(rec, rec.shape, float(case["res"]), bool(uif))
            assert bits_equal(got, case["uif%d__free_and_named" % uif]), (name, uif)
            assert (lo, hi) == (got.min(), got.max())
