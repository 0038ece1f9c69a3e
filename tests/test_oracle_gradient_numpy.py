"""(not gpu) The oracle's coarse gradient (GetGridAlignedIndexCoarseGradient, signed_distance_field.hpp:923-1016) against
an independent numpy evaluation of the same definition: central differences of the float field scaled by
1 / (2 res) in the interior, one-sided differences over the available span on the faces (edge gradients on), and no
value on the faces otherwise."""
import numpy as np

from oracle import oracle as O


def test_coarse_gradient_matches_numpy_differences():
    rng = np.random.default_rng(8)
    res = 0.2
    sdf = rng.normal(size=(9, 7, 11)).astype(np.float32)
    grad, has = O.coarse_gradient(sdf, res, enable_edge_gradients=False)
    interior = np.zeros(sdf.shape, dtype=bool)
    interior[1:-1, 1:-1, 1:-1] = True
    assert np.array_equal(has, interior)
    assert np.isnan(grad[~interior]).all()
    inv = 1.0 / (2.0 * res)
    for axis in range(3):
        hi = np.roll(sdf, -1, axis=axis)
        lo = np.roll(sdf, 1, axis=axis)
        want = (hi - lo).astype(np.float64) * inv          # float difference, double scale
        assert np.array_equal(grad[..., axis][interior], want[interior])
    # edge gradients: np.gradient uses one-sided first differences on the boundary and central ones inside;
    # the reference computes them in double from the float values
    grad_e, has_e = O.coarse_gradient(sdf, res, enable_edge_gradients=True)
    assert has_e.all()
    edge = ~interior
    for axis in range(3):
        want = np.gradient(sdf.astype(np.float64), res, axis=axis)
        assert np.allclose(grad_e[..., axis][edge], want[edge], rtol=0, atol=1e-12)
        assert np.array_equal(grad_e[..., axis][interior], grad[..., axis][interior])
