"""(gpu) Coarse gradient of every voxel (vgt_hip_sdf_coarse_gradient) vs the oracle: bit-exact
doubles (NaN patterns included), with and without edge gradients, and with a rotation."""
import numpy as np
import pytest

from voxelized_geometry_tools_amd import capi, synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def same_doubles(a, b):
    a, b = np.ascontiguousarray(a, dtype=np.float64), np.ascontiguousarray(b, dtype=np.float64)
    nan_a, nan_b = np.isnan(a), np.isnan(b)
    return np.array_equal(nan_a, nan_b) and np.array_equal(a[~nan_a].view(np.uint64), b[~nan_b].view(np.uint64))


@pytest.mark.parametrize("shape", [(24, 20, 28), (1, 9, 7), (2, 2, 2), (5, 1, 1), (33, 3, 64)])
def test_gradient_field_matches_oracle(ctx, shape):
    from oracle import oracle as O
    if min(shape) > 2:
        occ = synthetic.make_occupancy(shape, "spheres", seed=4)
    else:
        occ = (np.random.default_rng(1).random(shape) < 0.3).astype(np.float32)
    sdf, _, _ = ctx.sdf_from_occupancy(occ, 0.04)
    for edges in (False, True):
        got, has = ctx.sdf_coarse_gradient(sdf, 0.04, edges)
        want, whas = O.coarse_gradient(sdf, 0.04, edges)
        assert np.array_equal(has, whas), (shape, edges)
        assert same_doubles(got, want), (shape, edges)
    # all-free grid: +inf everywhere, inf - inf = NaN in the interior exactly as the reference's arithmetic gives
    inf_field = np.full(shape, np.inf, dtype=np.float32)
    got, has = ctx.sdf_coarse_gradient(inf_field, 0.04, True)
    want, whas = O.coarse_gradient(inf_field, 0.04, True)
    assert np.array_equal(has, whas) and same_doubles(got, want)


def test_rotation_is_applied_to_valid_gradients(ctx):
    from oracle import oracle as O
    occ = synthetic.make_occupancy((16, 18, 20), "spheres", seed=2)
    sdf, _, _ = ctx.sdf_from_occupancy(occ, 0.1)
    c, s = np.cos(0.3), np.sin(0.3)
    rot = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
    got, has = ctx.sdf_coarse_gradient(sdf, 0.1, False, rot)
    base, _ = O.coarse_gradient(sdf, 0.1, False)
    g = base[has]
    want = np.stack([rot[0, 0] * g[:, 0] + rot[0, 1] * g[:, 1] + rot[0, 2] * g[:, 2],
                     rot[1, 0] * g[:, 0] + rot[1, 1] * g[:, 1] + rot[1, 2] * g[:, 2],
                     rot[2, 0] * g[:, 0] + rot[2, 1] * g[:, 1] + rot[2, 2] * g[:, 2]], axis=1)
    assert np.allclose(got[has], want, rtol=1e-15, atol=0.0)
    assert np.isnan(got[~has]).all()
