"""(not gpu) The oracle's restatement of SignedDistanceField::EstimateLocationDistance and
::GetLocationFineGradient (signed_distance_field.hpp:808-833, 1050-1091), pinned by what the reference's
definitions imply: the estimate equals the corrected centre distance at cell centres, reproduces trilinear
interpolation of the corrected field (scipy.ndimage.map_coordinates, order 1) between centres, extrapolates
linearly in the half cell next to a face of the grid (GetAxisInterpolationIndices), and a field that is linear
in the location has its coefficients as fine gradient."""
import numpy as np
import pytest
from scipy import ndimage

from oracle import oracle as O


def _corrected(sdf, res):
    s = sdf.astype(np.float64)
    return np.where(s >= 0.0, s - 0.5 * res, s + 0.5 * res)


def _random_field(shape, seed):
    rng = np.random.default_rng(seed)
    f = rng.normal(size=shape).astype(np.float32)
    f[np.abs(f) < 0.05] = 0.07          # keep away from the sign flip of the centre correction
    return f


def test_estimate_at_cell_centres_and_outside():
    res = 0.25
    sdf = _random_field((5, 6, 7), 1)
    idx = np.stack(np.meshgrid(*[np.arange(n) for n in sdf.shape], indexing="ij"), axis=-1).reshape(-1, 3)
    centres = (idx + 0.5) * res
    got, has = O.estimate_distance(sdf, res, centres)
    assert has.all()
    assert np.allclose(got, _corrected(sdf, res).reshape(-1), rtol=0, atol=1e-12)
    outside = np.array([[-0.01, 0.3, 0.3], [0.3, 6 * res + 1e-9, 0.3], [0.3, 0.3, 7 * res], [np.nan, 0.1, 0.1]])
    got, has = O.estimate_distance(sdf, res, outside)
    assert not has.any() and np.isnan(got).all()


def test_estimate_is_trilinear_between_centres():
    res = 0.1
    sdf = _random_field((9, 8, 10), 2)
    rng = np.random.default_rng(3)
    # points at least half a cell inside the grid: the eight surrounding centres exist
    lo, hi = 0.5 * res, (np.array(sdf.shape) - 0.5) * res
    q = lo + rng.random((4000, 3)) * (hi - lo - 1e-9)
    got, has = O.estimate_distance(sdf, res, q)
    assert has.all()
    want = ndimage.map_coordinates(_corrected(sdf, res), (q / res - 0.5).T, order=1, mode="nearest")
    assert np.allclose(got, want, rtol=0, atol=1e-9)


def test_estimate_extrapolates_in_the_border_half_cell_and_linear_fields_are_reproduced():
    res = 0.2
    shape = (6, 5, 7)
    a, b, c, d = 0.7, -0.4, 1.3, 5.0                       # positive everywhere: one sign of the correction
    idx = np.stack(np.meshgrid(*[np.arange(n) for n in shape], indexing="ij"), axis=-1)
    centres = (idx + 0.5) * res
    sdf = (a * centres[..., 0] + b * centres[..., 1] + c * centres[..., 2] + d).astype(np.float32)
    rng = np.random.default_rng(4)
    q = rng.random((5000, 3)) * (np.array(shape) * res) * (1 - 1e-12)   # anywhere in the grid, border half cells too
    got, has = O.estimate_distance(sdf, res, q)
    assert has.all()
    want = a * q[:, 0] + b * q[:, 1] + c * q[:, 2] + d - 0.5 * res
    assert np.allclose(got, want, rtol=0, atol=2e-6)       # float32 storage of the field
    grad, has, too_large = O.fine_gradient(sdf, res, q, 0.03)
    assert has.all() and not too_large
    assert np.allclose(grad, [a, b, c], rtol=0, atol=2e-4)


def test_world_frame_queries_and_window_error():
    res = 0.5
    sdf = _random_field((4, 4, 4), 5)
    # grid_from_world: rotation by 90 degrees about z then a shift (column-major 4x4)
    M = np.array([[0.0, 1.0, 0.0, 0.3], [-1.0, 0.0, 0.0, 1.9], [0.0, 0.0, 1.0, -0.2], [0.0, 0.0, 0.0, 1.0]])
    world = np.array([[1.0, 0.2, 1.0], [0.4, 0.1, 0.9]])
    got, has = O.estimate_distance(sdf, res, world, M.T.reshape(-1))
    grid = (M[:3, :3] @ world.T).T + M[:3, 3]
    want, has2 = O.estimate_distance(sdf, res, grid)
    assert np.array_equal(has, has2) and np.allclose(got, want, rtol=0, atol=1e-12, equal_nan=True)
    # a window wider than the grid leaves it on both sides: the reference throws
    _, has, too_large = O.fine_gradient(sdf, res, np.array([[1.0, 1.0, 1.0]]), 5.0)
    assert too_large and not has[0]
    # one-sided windows near a face are fine
    grad, has, too_large = O.fine_gradient(sdf, res, np.array([[0.05, 1.0, 1.0]]), 0.2)
    assert has[0] and not too_large and np.isfinite(grad).all()
