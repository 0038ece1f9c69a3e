"""Oracle restatement of GetGridAlignedIndexCoarseGradient against values derived by hand from
the reference's exact-value SDF cases (test/sdf_generation_test.cpp) and against a numpy evaluation of the same
formulas.  The reference's tests hold no gradient known answers."""
import numpy as np

from oracle import oracle as O


def test_planar_case_by_hand():
    # PlanarExactTest field (res 1.0), padded to 3 x 4 x 4 so that there is an interior
    s2, s5, s8 = np.sqrt(np.float32(2)), np.sqrt(np.float32(5)), np.sqrt(np.float32(8))
    plane = np.array([[-2, -1, 1, 2], [-1, -1, 1, 2], [1, 1, s2, s5], [2, 2, s5, s8]], dtype=np.float32)
    sdf = np.stack([plane, plane, plane])
    grad, has = O.coarse_gradient(sdf, 1.0)
    assert has.sum() == 1 * 2 * 2 and has[1, 1:3, 1:3].all()
    assert np.array_equal(grad[1, 1, 1], [0.0, (1.0 - (-1.0)) / 2.0, (1.0 - (-1.0)) / 2.0])
    assert np.array_equal(grad[1, 2, 2], [0.0, np.float64(np.float32(s5 - np.float32(1.0))) * 0.5,
                                          np.float64(np.float32(s5 - np.float32(1.0))) * 0.5])
    assert np.isnan(grad[0]).all() and not has[0].any()
    # edge gradients: one-sided differences in double, zero along an axis of extent 1
    grad, has = O.coarse_gradient(sdf[:1], 1.0, True)
    assert has.all()
    assert np.array_equal(grad[0, 0, 0], [0.0, (-1.0 - (-2.0)) / 1.0, (-1.0 - (-2.0)) / 1.0])
    assert np.array_equal(grad[0, 3, 3], [0.0, float(s8) - float(s5), float(s8) - float(s5)])


def test_matches_numpy_formulas():
    rng = np.random.default_rng(3)
    sdf = (rng.standard_normal((7, 9, 8)) * 3).astype(np.float32)
    sdf[2, 3, 4] = np.inf
    sdf[5, 5, 5] = -np.inf
    res = 0.37
    grad, has = O.coarse_gradient(sdf, res)
    inv = 1.0 / (2.0 * res)
    with np.errstate(invalid="ignore"):
        want_x = (sdf[2:, 1:-1, 1:-1] - sdf[:-2, 1:-1, 1:-1]).astype(np.float64) * inv
        want_y = (sdf[1:-1, 2:, 1:-1] - sdf[1:-1, :-2, 1:-1]).astype(np.float64) * inv
        want_z = (sdf[1:-1, 1:-1, 2:] - sdf[1:-1, 1:-1, :-2]).astype(np.float64) * inv
    core = grad[1:-1, 1:-1, 1:-1]
    for got, want in ((core[..., 0], want_x), (core[..., 1], want_y), (core[..., 2], want_z)):
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    assert has[1:-1, 1:-1, 1:-1].all() and has.sum() == 5 * 7 * 6
