"""(not gpu) The oracle's literal restatement of SignedDistanceField::ComputeLocalExtremaMap
(signed_distance_field.hpp:1205-1231 over :385-541) on fields whose answer follows from the definition."""
import numpy as np

from oracle import oracle as O


def _centres(shape, res):
    idx = np.stack(np.meshgrid(*[np.arange(n) for n in shape], indexing="ij"), axis=-1)
    return (idx + 0.5) * res


def test_constant_field_every_cell_is_its_own_extremum():
    res, shape = 0.25, (4, 5, 6)
    out = O.local_extrema_map(np.full(shape, 0.7, dtype=np.float32), res)
    assert np.array_equal(out, _centres(shape, res))


def test_ramps_run_off_the_grid():
    res, shape = 0.5, (6, 3, 3)
    x = np.arange(shape[0], dtype=np.float32)[:, None, None]
    free_ramp = np.broadcast_to(x + 1.0, shape).astype(np.float32)        # outside obstacles: uphill = +x
    assert np.all(np.isposinf(O.local_extrema_map(free_ramp, res)))
    filled_ramp = -free_ramp                                              # inside: downhill = +x as well
    assert np.all(np.isposinf(O.local_extrema_map(filled_ramp, res)))


def test_tent_collects_at_the_ridge():
    res, shape = 0.1, (9, 1, 1)
    x = np.arange(9, dtype=np.float32)
    sdf = (5.0 - np.abs(x - 4.0)).reshape(shape).astype(np.float32)
    out = O.local_extrema_map(sdf, res)
    assert np.allclose(out[:, 0, 0, 0], 4.5 * res) and np.allclose(out[:, 0, 0, 1:], 0.5 * res)


def test_two_cycle_takes_the_entry_cell_of_the_first_walk():
    # cells 1 and 2 point at each other; the walk from cell 0 (first in X-major order) enters the cycle at cell 1
    res = 1.0
    sdf = np.array([0.1, 1.0, 1.0, 0.1], dtype=np.float32).reshape(4, 1, 1)
    out = O.local_extrema_map(sdf, res)
    assert np.allclose(out[:, 0, 0, 0], 1.5)
    # mirrored along the line, the first walk (from cell 0 again) enters at what is now cell 1 as well; put the
    # basin's smallest cell on the other side by making cell 0 flat: the first walk then starts at cell 1
    sdf2 = np.array([1.0, 1.0, 0.1, 1.0, 1.0, 0.1], dtype=np.float32).reshape(6, 1, 1)
    out2 = O.local_extrema_map(sdf2, res)
    # cell 0: one-sided gradient 0 -> flat, its own extremum; cells 1..: 1 -> 0? (central (0.1-1)/2 < 0 -> cell 0)
    assert np.allclose(out2[0, 0, 0], [0.5, 0.5, 0.5])
    assert np.allclose(out2[1, 0, 0], [0.5, 0.5, 0.5])


def test_every_value_is_a_cell_centre_or_infinity_and_fixed_cells_keep_their_location():
    rng = np.random.default_rng(0)
    res, shape = 0.2, (7, 6, 8)
    sdf = rng.normal(size=shape).astype(np.float32)
    out = O.local_extrema_map(sdf, res)
    assert not np.any(np.isneginf(out)) and not np.any(np.isnan(out))
    finite = np.isfinite(out[..., 0])
    idx = out[finite] / res - 0.5
    assert np.allclose(idx, np.rint(idx), atol=1e-9)
    assert np.all((np.rint(idx) >= 0) & (np.rint(idx) < np.array(shape)))
    assert np.array_equal(np.isposinf(out[..., 0]), np.isposinf(out[..., 2]))
