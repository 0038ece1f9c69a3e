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


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as O
    return O


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


def _query_cloud(shape, res, n, seed):
    """Query points in and around the grid (outside points, border half cells, exact cell boundaries, NaN)."""
    rng = np.random.default_rng(seed)
    extent = np.array(shape, dtype=np.float64) * res
    q = (rng.random((n, 3)) * 1.3 - 0.15) * extent
    q[::17] = np.floor(q[::17] / res) * res                    # on cell boundaries
    q[5] = [np.nan, 0.1, 0.1]
    q[6] = [np.inf, 0.1, 0.1]
    return q


@pytest.mark.parametrize("shape", [(7, 9, 11), (1, 6, 5), (24, 20, 33)])
def test_estimate_distance_matches_oracle(ctx, oracle, shape):
    """vgt_hip_sdf_estimate_distance (EstimateLocationDistance for a batch of points) against the oracle's
    restatement: same documented operation order in double -> bit-identical; BASELINE's tolerance is 1e-5."""
    rng = np.random.default_rng(sum(shape))
    res = 0.125
    sdf = rng.normal(size=shape).astype(np.float32)
    q = _query_cloud(shape, res, 5000, 11)
    for xf in (None, np.array([[0.0, -1.0, 0.0, 0.4], [1.0, 0.0, 0.0, 0.1], [0.0, 0.0, 1.0, 0.05], [0, 0, 0, 1.0]]).T.reshape(-1)):
        got, has = ctx.sdf_estimate_distance(sdf, res, q, xf)
        want, whas = oracle.estimate_distance(sdf, res, q, xf)
        assert np.array_equal(has, whas)
        assert np.array_equal(np.isnan(got), np.isnan(want))
        assert np.array_equal(got[has].view(np.uint64), want[whas].view(np.uint64))
        assert np.nanmax(np.abs(got - want), initial=0.0) <= 1e-5


def test_fine_gradient_matches_oracle(ctx, oracle):
    rng = np.random.default_rng(3)
    shape, res = (12, 10, 14), 0.1
    sdf = rng.normal(size=shape).astype(np.float32)
    q = _query_cloud(shape, res, 4000, 5)
    got, has = ctx.sdf_fine_gradient(sdf, res, q, 0.04)
    want, whas, too_large = oracle.fine_gradient(sdf, res, q, 0.04)
    assert not too_large
    assert np.array_equal(has, whas)
    assert np.array_equal(got[has].view(np.uint64), want[whas].view(np.uint64))
    assert np.isnan(got[~has]).all()
    # the reference throws when the window leaves the grid on both sides of a query inside it
    with pytest.raises(ValueError, match="Window size"):
        ctx.sdf_fine_gradient(sdf, res, np.array([[0.5, 0.5, 0.5]]), 10.0)
    with pytest.raises(ValueError):
        ctx.sdf_fine_gradient(sdf, res, q, 0.0)


def test_estimate_distance_on_an_extracted_field(ctx, oracle):
    """End to end: occupancy -> SDF on the device -> estimates at points near a sphere's surface are within a voxel
    of the analytic distance (the estimate rounds corners but must not be off by more than the grid can resolve)."""
    n, res = 48, 0.05
    ax = np.arange(n)
    c = np.array([24, 22, 25])
    d2 = ((ax - c[0]) ** 2)[:, None, None] + ((ax - c[1]) ** 2)[None, :, None] + ((ax - c[2]) ** 2)[None, None, :]
    occ = (d2 <= 10.0 ** 2).astype(np.float32)
    sdf, _, _ = ctx.sdf_from_occupancy(occ, res)
    rng = np.random.default_rng(9)
    dirs = rng.normal(size=(2000, 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    radius = rng.uniform(5.0, 18.0, size=(2000, 1))
    q = ((c + 0.5) + dirs * radius) * res
    got, has = ctx.sdf_estimate_distance(sdf, res, q)
    assert has.all()
    analytic = (radius[:, 0] - 10.0) * res
    assert np.max(np.abs(got - analytic)) <= 1.5 * res
    want, _ = oracle.estimate_distance(sdf, res, q)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))


@pytest.mark.parametrize("kind", ["noise", "spheres", "two_cycle_line", "rotated", "plateaus"])
def test_local_extrema_map_matches_oracle(ctx, oracle, kind):
    """vgt_hip_sdf_local_extrema_map against the oracle's literal, sequential ComputeLocalExtremaMap: chains that end
    at flat cells, leave the grid, and -- the order-dependent part -- run into cycles (noise fields are full of them)."""
    rng = np.random.default_rng(7)
    rotation = None
    res = 0.1
    if kind == "noise":
        sdf = rng.normal(size=(17, 13, 21)).astype(np.float32)
    elif kind == "spheres":
        occ = synthetic.make_occupancy((40, 36, 44), "spheres", seed=4)
        sdf, _, _ = ctx.sdf_from_occupancy(occ, res)
    elif kind == "two_cycle_line":
        sdf = np.array([0.1, 1.0, 1.0, 0.1, 0.5, 0.5, 0.9, 0.2], dtype=np.float32).reshape(8, 1, 1)
    elif kind == "rotated":
        sdf = rng.normal(size=(12, 14, 10)).astype(np.float32)
        c, s = np.cos(0.7), np.sin(0.7)
        rotation = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
    else:
        sdf = np.round(rng.normal(size=(15, 15, 15)) * 2.0).astype(np.float32) * np.float32(res)   # many exactly flat cells
    got = ctx.sdf_local_extrema_map(sdf, res, rotation)
    want = oracle.local_extrema_map(sdf, res, rotation)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), kind


def test_local_extrema_map_medium_grid(ctx, oracle):
    """A 96^3 field of an extracted SDF plus small noise: long chains and many cycles."""
    occ = synthetic.make_occupancy((96, 96, 96), "spheres", seed=2)
    sdf, _, _ = ctx.sdf_from_occupancy(occ, 0.05)
    sdf = sdf + (np.random.default_rng(1).normal(size=sdf.shape) * 0.004).astype(np.float32)
    got = ctx.sdf_local_extrema_map(sdf, 0.05)
    want = oracle.local_extrema_map(sdf, 0.05)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
