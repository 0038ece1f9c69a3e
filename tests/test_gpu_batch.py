"""(gpu) Batched SDF extraction (vgt_hip_sdf_batch_*, vgt_hip_cells_object_sdfs): many grids of one shape, or many
object masks of one tagged map, through ONE run of the three EDT passes.  The reference has no batched entry point: it
loops ExtractSignedDistanceField (tagged_object_occupancy_map.hpp:249-290), so the bar is "bit-equal to the loop of
single calls", and the single calls are pinned against the oracle elsewhere (test_gpu_sdf.py, test_gpu_tagged_sdf.py);
the smaller cases here are checked against the oracle directly as well."""
import numpy as np
import pytest

from conftest import bits_equal
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


def _mixed_batch(shape, batch, seed):
    """Grids of every kind in one batch: spheres, salt, unknown mix, empty, full, one voxel, one-class lines."""
    kinds = ["spheres", "salt", "unknown_mix", "empty", "full", "single"]
    grids = []
    for b in range(batch):
        kind = kinds[b % len(kinds)]
        grids.append(np.ascontiguousarray(synthetic.make_occupancy(shape, kind, seed=seed + b)))
    return grids


# (short-line kernels on both axes, on one, on none: lines of more than 64 rows take the sweeps, whose X pass deals
# (grid, y, segment) items and keeps extrema per grid)
BATCH_CASES = [((64, 64, 64), 7), ((13, 17, 40), 5), ((9, 70, 130), 3), ((1, 1, 5), 4), ((40, 40, 40), 64),
               ((3, 5, 64), 2), ((1, 70, 1), 3), ((24, 8, 200), 1), ((70, 9, 130), 3), ((130, 66, 65), 2),
               ((300, 20, 33), 5), ((65, 129, 20), 4)]


@pytest.mark.parametrize("shape,batch", BATCH_CASES)
def test_batch_equals_single_calls_and_oracle(ctx, oracle, shape, batch):
    grids = _mixed_batch(shape, batch, seed=100 + batch)
    for uif, vb in ((True, False), (False, True)):
        fields, lo, hi = ctx.sdf_batch_from_occupancy(grids, 0.037, uif, vb)
        assert len(fields) == batch
        for b in range(batch):
            single, slo, shi = ctx.sdf_from_occupancy(grids[b], 0.037, uif, vb)
            assert bits_equal(fields[b], single), (shape, b, uif, vb)
            assert (float(lo[b]), float(hi[b])) == (slo, shi), (shape, b)
            if np.prod(shape) * batch <= 64 ** 3 * 8:
                want, wlo, whi = oracle.sdf_from_occupancy(grids[b], 0.037, uif, vb)
                assert bits_equal(fields[b], want), (shape, b, uif, vb)
                assert (float(lo[b]), float(hi[b])) == (wlo, whi)


def test_batch_device_resident(ctx):
    """vgt_hip_sdf_batch_dev on [batch][nx][ny][nz] device buffers against vgt_hip_sdf_dev grid by grid; per-grid
    extrema; the workspace size is checked."""
    import torch
    shape, batch = (48, 56, 72), 11
    grids = _mixed_batch(shape, batch, seed=9)
    occ = torch.from_numpy(np.stack(grids)).cuda()
    sdf = torch.empty_like(occ)
    nbytes = capi.sdf_batch_workspace_bytes(batch, shape)
    assert nbytes > batch * int(np.prod(shape)) * 4
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    mm = torch.zeros((batch, 2), dtype=torch.float32, device="cuda")
    ctx.set_stream(None)
    try:
        ctx.sdf_batch_dev(occ.data_ptr(), batch, shape, 0.02, sdf.data_ptr(), ws.data_ptr(), nbytes, mm.data_ptr())
        torch.cuda.synchronize()
        with pytest.raises(ValueError):
            ctx.sdf_batch_dev(occ.data_ptr(), batch, shape, 0.02, sdf.data_ptr(), ws.data_ptr(), nbytes - 1024, None)
        one_bytes = capi.sdf_workspace_bytes(shape)
        ws1 = torch.empty(one_bytes, dtype=torch.uint8, device="cuda")
        one = torch.empty(shape, dtype=torch.float32, device="cuda")
        mm1 = torch.zeros(2, dtype=torch.float32, device="cuda")
        for b in range(batch):
            ctx.sdf_dev(occ[b].data_ptr(), shape, 0.02, one.data_ptr(), ws1.data_ptr(), one_bytes, mm1.data_ptr())
            torch.cuda.synchronize()
            assert torch.equal(one.view(torch.int32), sdf[b].view(torch.int32)), b
            assert torch.equal(mm1, mm[b]), b
    finally:
        ctx.reset_stream()


def test_batch_is_cut_into_launches_when_large(ctx):
    """More grids than one launch takes (2 GiB of device buffers): the host entry point cuts the batch itself."""
    shape, batch = (128, 128, 128), 130
    rng = np.random.default_rng(3)
    base = synthetic.make_occupancy(shape, "spheres", seed=1)
    grids = []
    for b in range(batch):
        g = np.roll(base, shift=(b * 3, b * 5, b * 7), axis=(0, 1, 2)).copy()
        g[tuple(int(v) for v in rng.integers(0, 128, 3))] = 1.0
        grids.append(g)
    fields, lo, hi = ctx.sdf_batch_from_occupancy(grids, 0.01)
    for b in (0, 1, 63, 112, 113, 114, 129):
        single, slo, shi = ctx.sdf_from_occupancy(grids[b], 0.01)
        assert bits_equal(fields[b], single), b
        assert (float(lo[b]), float(hi[b])) == (slo, shi)


@pytest.mark.parametrize("shape", [(160, 128, 144), (208, 208, 208)], ids=["12MB-two-ring-slots", "36MB-page-locked"])
def test_host_copy_paths_by_array_size(ctx, shape):
    """The host entry points move the caller's arrays by size: below 32 MiB through the context's ring of page-locked 8 MiB
    slots (one slot, several slots per array, both directions), from 32 MiB page-locked in place with a fresh output
    faulted in by helper threads first.  Every path bit-equal to the plain single call."""
    grids = [np.ascontiguousarray(synthetic.make_occupancy(shape, kind, seed=9 + k))
             for k, kind in enumerate(["spheres", "salt", "unknown_mix"])]
    fields, lo, hi = ctx.sdf_batch_from_occupancy(grids, 0.02)
    singles = [ctx.sdf_from_occupancy(g, 0.02) for g in grids]
    for b, (single, slo, shi) in enumerate(singles):
        assert bits_equal(fields[b], single), b
        assert (float(lo[b]), float(hi[b])) == (slo, shi), b
    # one field of a tagged map into a fresh array (ring / faulted in + page-locked), and its objects as a batch
    rec = np.zeros(shape, dtype=capi.TAGGED_OBJECT_CELL)
    rec["occupancy"] = grids[0]
    rec["object_id"][grids[0] > 0.5] = 7
    rec["object_id"][: shape[0] // 2][grids[0][: shape[0] // 2] > 0.5] = 3
    cells = ctx.cells(rec, shape)
    got, glo, ghi = cells.sdf(0.02)
    assert bits_equal(got, singles[0][0]) and (glo, ghi) == singles[0][1:]
    batched = cells.all_object_sdfs(0.02)
    one_by_one = cells.separate_object_sdfs_one_by_one(0.02, cells.object_ids())
    assert sorted(batched) == [3, 7]
    for oid in batched:
        assert bits_equal(batched[oid][0], one_by_one[oid][0]) and batched[oid][1:] == one_by_one[oid][1:], oid
    cells.close()


def test_all_object_sdfs_as_one_batch(ctx, oracle):
    """A 32-object 128^3 tagged map: the batched MakeAllObjectSDFs against 32 vgt_hip_cells_sdf calls and the oracle."""
    rng = np.random.default_rng(5)
    shape = (128, 128, 128)
    rec = np.zeros(shape, dtype=capi.TAGGED_OBJECT_CELL)
    ids = [int(v) for v in rng.choice(np.arange(1, 5000), size=30, replace=False)] + [0x7fffffff, 0xffffffff]
    for oid in ids:
        lo = [int(rng.integers(0, s - 20)) for s in shape]
        ext = [int(rng.integers(3, 20)) for _ in shape]
        box = tuple(slice(a, a + e) for a, e in zip(lo, ext))
        rec["occupancy"][box] = 1.0
        rec["object_id"][box] = oid
    rec["occupancy"][rng.random(shape) < 0.001] = 0.5
    cells = ctx.cells(rec, shape)
    found = cells.object_ids()
    batched = cells.all_object_sdfs(0.02)
    assert sorted(batched) == [int(i) for i in found]
    one_by_one = cells.separate_object_sdfs_one_by_one(0.02, found)
    for oid in batched:
        assert bits_equal(batched[oid][0], one_by_one[oid][0]), oid
        assert batched[oid][1:] == one_by_one[oid][1:], oid
    for oid in [int(found[0]), int(found[7]), 0xffffffff]:
        want, wlo, whi = oracle.sdf_from_cells(rec, shape, 0.02, [oid])
        assert bits_equal(batched[oid][0], want) and batched[oid][1:] == (wlo, whi), oid
    # predicate variants, an id that no cell carries (an empty mask: +inf everywhere), id 0, duplicates
    got = cells.separate_object_sdfs(0.02, [int(found[3]), 4999999, 0, int(found[3])], unknown_is_filled=False,
                                     add_virtual_border=True)
    for oid in (int(found[3]), 4999999, 0):
        want, wlo, whi = oracle.sdf_from_cells(rec, shape, 0.02, [oid], False, True)
        assert bits_equal(got[oid][0], want) and got[oid][1:] == (wlo, whi), oid
    cells.close()


@pytest.mark.parametrize("dtype", [capi.TAGGED_OBJECT_CELL, capi.TAGGED_OBJECT_COMPONENT_CELL], ids=["tagged8", "tagged16"])
def test_object_batch_on_the_fixture_grids(ctx, sdf_tagged_cases, dtype):
    from conftest import tagged_records
    for name, case in sdf_tagged_cases.items():
        rec = tagged_records(case, dtype)
        cells = ctx.cells(rec, rec.shape)
        ids = cells.object_ids()
        batched = cells.separate_object_sdfs(float(case["res"]), ids)
        for oid in ids:
            single = cells.sdf(float(case["res"]), [int(oid)])
            assert bits_equal(batched[int(oid)][0], single[0]) and batched[int(oid)][1:] == single[1:], (name, oid)
        cells.close()


def test_batch_limits(ctx):
    """batch * nx * ny must stay below 2^28 lines for the device entry point (the host one cuts the batch itself)."""
    import torch
    shape = (512, 512, 1)
    batch = 1024          # 2^28 lines exactly: refused
    occ = torch.zeros(8, dtype=torch.float32, device="cuda")
    with pytest.raises(ValueError):
        ctx.sdf_batch_dev(occ.data_ptr(), batch, shape, 0.1, occ.data_ptr(), occ.data_ptr(), 1 << 40, None)
    with pytest.raises(ValueError):
        ctx.sdf_batch_dev(occ.data_ptr(), 0, shape, 0.1, occ.data_ptr(), occ.data_ptr(), 1 << 40, None)


def test_batch_argument_errors(ctx):
    g = np.zeros((4, 4, 4), dtype=np.float32)
    with pytest.raises(ValueError):
        ctx.sdf_batch_from_occupancy([g, np.zeros((4, 4, 5), np.float32)], 0.1)
    with pytest.raises(ValueError):
        ctx.sdf_batch_from_occupancy([g], 0.0)
    assert capi.sdf_batch_workspace_bytes(0, (4, 4, 4)) == 0
    assert capi.sdf_batch_workspace_bytes(1, (64, 64, 64)) == capi.sdf_workspace_bytes((64, 64, 64))
    rec = np.zeros((4, 4, 4), dtype=capi.OCCUPANCY_COMPONENT_CELL)
    cells = ctx.cells(rec, rec.shape, object_id_offset=-1)
    with pytest.raises(ValueError):
        cells.separate_object_sdfs(0.1, [1])          # this cell type carries no object id
    cells.close()
