"""(not gpu) The N > 1 host path on CPU: two gloo ranks exchange slab summaries and derive the
carries; the result must reproduce the full-grid nearest-other-class distance along Z."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _z_distance_reference(filled):
    """Signed distance along Z to the nearest voxel of the other class (32767 = none)."""
    nx, ny, nz = filled.shape
    z = np.arange(nz)
    out = np.empty(filled.shape, dtype=np.int32)
    for cls in (True, False):
        other = filled != cls  # voxels of the other class
        # nearest other-class position below / above every z, per line
        below = np.maximum.accumulate(np.where(other, z, -10 ** 6), axis=2)
        above = np.minimum.accumulate(np.where(other, z, 10 ** 6)[:, :, ::-1], axis=2)[:, :, ::-1]
        d = np.minimum(z - below, above - z)
        d = np.where(d > 30000, 32767, d)
        sel = filled == cls
        out[sel] = np.where(cls, -d, d)[sel]
    return out


def _worker(rank, world, port, shape, seed, queue):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from voxelized_geometry_tools_amd import multi_gpu
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(seed)
        filled = rng.random(shape) < 0.03
        filled[0, 0, :] = True           # a line with no free voxel anywhere
        filled[1, 1, :] = False          # a line with no filled voxel anywhere
        local_shape, z0 = multi_gpu.slab_of(shape, rank, world)
        local = filled[:, :, z0:z0 + local_shape[2]]
        summary = torch.from_numpy(multi_gpu.summary_reference(local, z0))
        assert summary.shape == (shape[0] * shape[1], 2)  # 4 bytes per line
        carries = multi_gpu.exchange_carries(torch, dist, summary, rank, world, shape[2]).numpy()
        # fold the carries into the slab-local distances exactly as SlabFixupKernel does
        d_local = _z_distance_reference(local)
        zg = np.arange(z0, z0 + local_shape[2])[None, None, :]
        c = carries.reshape(shape[0], shape[1], 4).astype(np.int32)
        mag = np.abs(d_local)
        for cls, prev_col, next_col in ((True, multi_gpu.PREV_FREE, multi_gpu.NEXT_FREE),
                                        (False, multi_gpu.PREV_FILLED, multi_gpu.NEXT_FILLED)):
            prev_o, next_o = c[..., prev_col][..., None], c[..., next_col][..., None]
            cand = np.minimum(np.where(prev_o >= 0, zg - prev_o, 32767),
                              np.where(next_o >= 0, next_o - zg, 32767))
            sel = local == cls
            mag = np.where(sel, np.minimum(mag, cand), mag)
        got = np.where(local, -mag, mag)
        want = _z_distance_reference(filled)[:, :, z0:z0 + local_shape[2]]
        ok = bool(np.array_equal(got, want))
        # extrema of the whole field: one two-element collective (multi_gpu.reduce_extrema)
        per_rank = [(-0.5 - r, 2.0 + 3 * r) for r in range(world)]
        per_rank[world - 1] = (float("-inf"), 1.0)
        mm = torch.tensor(per_rank[rank], dtype=torch.float32)
        multi_gpu.reduce_extrema(dist, mm)
        ok = ok and mm.tolist() == [float("-inf"), max(h for _, h in per_rank)]
        queue.put((rank, ok, None))
    except Exception as exc:  # pragma: no cover
        queue.put((rank, False, repr(exc)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,shape", [(2, (6, 7, 40)), (3, (5, 4, 31))])
def test_slab_exchange_over_gloo(world, shape):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, shape, 123, queue)) for r in range(world)]
    for p in procs:
        p.start()
    results = [queue.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, err in results:
        assert err is None, "rank %d: %s" % (rank, err)
        assert ok, "rank %d: slab distances differ from the full-grid scan" % rank


def test_slab_bounds_and_carries():
    import torch
    from voxelized_geometry_tools_amd import multi_gpu
    assert multi_gpu.slab_bounds(10, 3) == [(0, 4), (4, 7), (7, 10)]
    assert multi_gpu.slab_of((2, 3, 1024), 5, 8) == ((2, 3, 128), 640)
    with pytest.raises(ValueError):
        multi_gpu.slab_of((2, 2, 2), 2, 3)
    # two lines, three slabs of 10 voxels; per slab (first_filled, last_filled, first_free, last_free) is
    #   line 0: (3, 5, 0, 9) | (-, -, 10, 19) | (25, 29, 20, 28)      line 1: (-, -, 0, 9) | (12, 12, 10, 19) | (-, -, 20, 29)
    # as packed records (class bit of the first / last voxel, position of the first / last voxel of the other class)
    F, N = multi_gpu.FILLED_BIT, multi_gpu.OTHER_NONE
    g = torch.tensor([[[3, 5], [N, N]],
                      [[N, N], [12, 12]],
                      [[25, F | 28], [N, N]]], dtype=torch.int32).to(torch.int16)
    full = multi_gpu.decode_summaries(torch, g, 30).tolist()
    assert full[0] == [[3, 5, 0, 9], [-1, -1, 0, 9]]
    assert full[1] == [[-1, -1, 10, 19], [12, 12, 10, 19]]
    assert full[2] == [[25, 29, 20, 28], [-1, -1, 20, 29]]
    c1 = multi_gpu.carries_from_summaries(torch, g, 1, 30).tolist()
    assert c1 == [[5, 25, 9, 20], [-1, -1, 9, 20]]
    c0 = multi_gpu.carries_from_summaries(torch, g, 0, 30).tolist()
    assert c0 == [[-1, 25, -1, 10], [-1, 12, -1, 10]]
    c2 = multi_gpu.carries_from_summaries(torch, g, 2, 30).tolist()
    assert c2 == [[5, -1, 19, -1], [12, -1, 19, -1]]


def test_summary_reference_packs_both_classes():
    """The 4-byte record against the plain four-value summary on random slabs (uneven splits included)."""
    import torch
    from voxelized_geometry_tools_amd import multi_gpu
    rng = np.random.default_rng(5)
    shape, world = (4, 5, 23), 4
    filled = rng.random(shape) < 0.2
    filled[0, 0, :] = True
    filled[1, 1, :] = False
    records = []
    for r in range(world):
        local_shape, z0 = multi_gpu.slab_of(shape, r, world)
        records.append(torch.from_numpy(multi_gpu.summary_reference(filled[:, :, z0:z0 + local_shape[2]], z0)))
    full = multi_gpu.decode_summaries(torch, torch.stack(records), shape[2]).numpy()
    z = np.arange(shape[2])
    for r, (z0, z1) in enumerate(multi_gpu.slab_bounds(shape[2], world)):
        local = filled[:, :, z0:z1].reshape(-1, z1 - z0)
        zz = z[z0:z1]
        for col, mask, first in ((0, local, True), (1, local, False), (2, ~local, True), (3, ~local, False)):
            if first:
                want = np.where(mask.any(axis=1), np.where(mask, zz, 10 ** 6).min(axis=1), -1)
            else:
                want = np.where(mask.any(axis=1), np.where(mask, zz, -1).max(axis=1), -1)
            assert np.array_equal(full[r, :, col], want), (r, col)


def _ray_worker(rank, world, port, queue):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import oracle as O
    from voxelized_geometry_tools_amd import multi_gpu, synthetic
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        counts = (24, 24, 24)
        vs = np.float32(5.12 / 24)
        ivs = np.float32(1.0) / vs
        sizes = [np.float32(c) * vs for c in counts]
        pts = synthetic.raycast_cloud(5_001, seed=11)
        xf = synthetic.translation_xform(-1.0, 2.56, 2.56).astype(np.float32)
        first, count = multi_gpu.point_share(pts.shape[0], world, rank)
        mine = O.raycast_f32(pts[first:first + count], 3.0, xf, vs, ivs, sizes, counts)
        want = O.raycast_f32(pts, 3.0, xf, vs, ivs, sizes, counts)
        everywhere = multi_gpu.sum_counts(dist, torch.from_numpy(mine.copy()).view(-1))
        ok = bool(np.array_equal(everywhere.numpy().reshape(want.shape), want))
        on_root = multi_gpu.sum_counts(dist, torch.from_numpy(mine.copy()).view(-1), root=0)
        if rank == 0:
            ok = ok and bool(np.array_equal(on_root.numpy().reshape(want.shape), want))
        queue.put((rank, ok, None))
    except Exception as exc:  # pragma: no cover
        queue.put((rank, False, repr(exc)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_one_cloud_ray_split_over_gloo(world):
    """multi_gpu.point_share + sum_counts: the ranks' private tracking counts (here from the oracle) add up to the
    whole cloud's counts bit for bit -- the property vgt_hipx_raycast_points_split and RaySplit rest on."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ray_worker, args=(r, world, port, queue)) for r in range(world)]
    for p in procs:
        p.start()
    results = [queue.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, err in results:
        assert err is None, "rank %d: %s" % (rank, err)
        assert ok, "rank %d: summed counts differ from the whole cloud's" % rank


def test_point_share_matches_library():
    from voxelized_geometry_tools_amd import capi, multi_gpu
    for n, world in ((0, 1), (1, 4), (10, 3), (1_000_000, 8), (7, 7), (100_003, 6)):
        shares = [multi_gpu.point_share(n, world, r) for r in range(world)]
        assert shares == [capi.point_share(n, world, r) for r in range(world)]
        assert sum(c for _, c in shares) == n
        assert all(shares[r][0] + shares[r][1] == shares[r + 1][0] for r in range(world - 1))
    # invalid shares are argument errors (a share count of zero used to divide by zero inside the library)
    for n, world, r in ((10, 0, 0), (10, 3, 3), (10, 3, -1), (-1, 2, 0)):
        with pytest.raises(ValueError):
            capi.point_share(n, world, r)
