"""Z-slab partitioned SDF extraction: one process (rank) per GPU, one RCCL exchange.

The grid (nx, ny, nz) is cut along Z into `world` slabs.  Lines along Y and X are local to a
slab, so the two envelope passes need no communication.  Only pass 1 -- distance along Z to the
nearest voxel of the other class -- crosses slabs, and all it needs from the other slabs is, per
(x, y) line, the nearest filled / free voxel below and above the slab.  Each rank therefore
publishes 4 BYTES per line: a slab's first voxel is filled or free, so "first filled" and "first
free" are the slab's first voxel and the first voxel of the other class; the record holds, for the
first and for the last voxel, the class (bit 15) and the global z of the first / last voxel of the
other class (bits 0-14, 0x7fff = none).  The ranks all-gather these summaries once
(torch.distributed: backend "nccl" is RCCL over xGMI; "gloo" in the CPU tests) -- 4 B x nx x ny
x world received per rank, 128 MiB at 2048 x 2048 lines and 8 ranks -- and every rank reduces
them to per-line carries for its own slab.  Exact for any slab count: the first pass is a
nearest-site scan, not a k-voxel halo.

The exchange helpers are plain torch code and run on CPU tensors too (tests/test_multi_gpu_gloo.py).

Raycast voxelization of ONE point cloud over several ranks (RaySplit below): rays are independent but
scatter into one grid, so every rank casts a contiguous share of the points into a private tracking grid
and the grids are summed with one int32 all-reduce (or a reduce onto the rank that runs the filter).
Tracking counts are integers: the sum equals the counts of the whole cloud bit for bit, whatever the split.
Whole clouds (the reference's own unit of dispatch, S/device_pointcloud_voxelization.cpp:147-149) need no
collective at all: one cloud per rank, then the filter reads every rank's grid.
"""
import numpy as np

NONE = -1
_BIG = 32767
FILLED_BIT = 0x8000   # summary halves: class of the slab's first / last voxel
OTHER_NONE = 0x7FFF   # ... and "no voxel of the other class in this slab"

# column layout of the per-line records (must match SlabLineSummary / SlabLineCarry in
# csrc/vgt_internal.hpp)
FIRST, LAST = 0, 1
PREV_FILLED, NEXT_FILLED, PREV_FREE, NEXT_FREE = 0, 1, 2, 3


def slab_bounds(nz, world):
    """Global z ranges [z0, z1) of the slabs: as equal as possible, earlier ranks get the extras."""
    base, extra = divmod(nz, world)
    bounds, z = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        bounds.append((z, z + n))
        z += n
    return bounds


def slab_of(full_shape, rank, world):
    """(local_shape, z_offset) of `rank`'s slab."""
    nx, ny, nz = full_shape
    z0, z1 = slab_bounds(nz, world)[rank]
    if z1 <= z0:
        raise ValueError("more slabs than Z voxels")
    return (nx, ny, z1 - z0), z0


def decode_summaries(torch, gathered, nz):
    """gathered: int16 tensor [world, lines, 2] of packed slab summaries (the slabs are slab_bounds(nz, world)) ->
    int32 tensor [world, lines, 4] = global z of (first_filled, last_filled, first_free, last_free), -1 = none."""
    world = gathered.shape[0]
    rec = gathered.to(torch.int32) & 0xFFFF
    filled = (rec & FILLED_BIT) != 0
    other = rec & OTHER_NONE
    other = torch.where(other == OTHER_NONE, torch.full_like(other, NONE), other)
    bounds = torch.tensor(slab_bounds(nz, world), dtype=torch.int32, device=gathered.device)
    begin = bounds[:, 0].view(world, 1).expand(-1, gathered.shape[1])
    end = (bounds[:, 1] - 1).view(world, 1).expand(-1, gathered.shape[1])
    out = torch.empty(gathered.shape[:2] + (4,), dtype=torch.int32, device=gathered.device)
    out[..., 0] = torch.where(filled[..., FIRST], begin, other[..., FIRST])   # first filled
    out[..., 1] = torch.where(filled[..., LAST], end, other[..., LAST])       # last filled
    out[..., 2] = torch.where(filled[..., FIRST], other[..., FIRST], begin)   # first free
    out[..., 3] = torch.where(filled[..., LAST], other[..., LAST], end)       # last free
    return out


def carries_from_summaries(torch, gathered, rank, nz):
    """gathered: int16 tensor [world, lines, 2] of packed slab summaries -> int16 [lines, 4] carries
    (prev_filled, next_filled, prev_free, next_free as global z, -1 = none) for `rank`."""
    world, lines, _ = gathered.shape
    full = decode_summaries(torch, gathered, nz)
    out = torch.full((lines, 4), NONE, dtype=torch.int32, device=gathered.device)
    if rank > 0:
        below = full[:rank]
        out[:, PREV_FILLED] = below[:, :, 1].max(dim=0).values
        out[:, PREV_FREE] = below[:, :, 3].max(dim=0).values
    if rank + 1 < world:
        above = full[rank + 1:]
        for dst, src in ((NEXT_FILLED, 0), (NEXT_FREE, 2)):
            v = above[:, :, src]
            v = torch.where(v < 0, torch.full_like(v, _BIG), v).min(dim=0).values
            out[:, dst] = torch.where(v == _BIG, torch.full_like(v, NONE), v)
    return out.to(torch.int16)


def exchange_carries(torch, dist, summary, rank, world, nz, gathered=None):
    """All-gathers the [lines, 2] int16 summaries of every rank and returns this rank's carries."""
    if gathered is None:
        gathered = torch.empty((world,) + tuple(summary.shape), dtype=summary.dtype,
                               device=summary.device)
    # the records are 4 bytes per line; move them as int32 words (gloo has no int16 collectives)
    dist.all_gather_into_tensor(gathered.view(torch.int32).view(-1),
                                summary.contiguous().view(torch.int32).view(-1))
    return carries_from_summaries(torch, gathered, rank, nz)


def reduce_extrema(dist, minmax):
    """In place: minmax = (min over ranks of minmax[0], max over ranks of minmax[1]) with one collective
    (min(a) = -max(-a); exact for floats, infinities included)."""
    minmax[0:1].neg_()
    dist.all_reduce(minmax, op=dist.ReduceOp.MAX)
    minmax[0:1].neg_()
    return minmax


def summary_reference(filled, z_offset):
    """numpy restatement of the packed per-line slab summary (tests): filled = bool (nx, ny, nzl) -> int16 [lines, 2]."""
    nx, ny, nzl = filled.shape
    z = np.arange(nzl)
    out = np.zeros((nx, ny, 2), dtype=np.uint16)
    for col, boundary, pick in ((FIRST, 0, "first"), (LAST, nzl - 1, "last")):
        cls = filled[:, :, boundary]
        other = filled != cls[:, :, None]                    # voxels of the other class
        has = other.any(axis=2)
        if pick == "first":
            pos = np.where(other, z, nzl).min(axis=2)
        else:
            pos = np.where(other, z, -1).max(axis=2)
        value = np.where(has, pos + z_offset, OTHER_NONE).astype(np.uint16)
        out[..., col] = value | np.where(cls, FILLED_BIT, 0).astype(np.uint16)
    return out.view(np.int16).reshape(nx * ny, 2)


class SlabSdf:
    """One rank's share of a Z-slab partitioned SDF extraction (device-resident).

    The context must run on torch's current stream (ctx.set_stream(torch.cuda.current_stream()
    .cuda_stream)) so that the library's kernels, the torch reductions and the collective are
    stream-ordered."""

    PHASES = ("scan", "exchange", "carries+fixup", "Y", "X")

    def __init__(self, ctx, torch, dist, full_shape, rank, world, device):
        from . import capi
        self.ctx, self.torch, self.dist = ctx, torch, dist
        self.full_shape = tuple(int(s) for s in full_shape)
        self.rank, self.world = rank, world
        self.local_shape, self.z_offset = slab_of(self.full_shape, rank, world)
        lines = self.local_shape[0] * self.local_shape[1]
        lib = capi.load()
        assert lib.vgt_hip_sdf_slab_summary_bytes(*self.local_shape[:2]) == lines * 4
        assert lib.vgt_hip_sdf_slab_carries_bytes(*self.local_shape[:2]) == lines * 8
        self.summary = torch.empty((lines, 2), dtype=torch.int16, device=device)
        self.gathered = torch.empty((world, lines, 2), dtype=torch.int16, device=device)
        self.carries = torch.empty((lines, 4), dtype=torch.int16, device=device)
        self.exchange_bytes_received = int(self.gathered.numel() * 2)
        self.ms_begin = np.zeros(1, dtype=np.float32)
        self.ms_finish = np.zeros(3, dtype=np.float32)

    def run(self, occ, sdf, ws, minmax, resolution, kernel_ms=None, unknown_is_filled=True,
            add_virtual_border=False, events=None):
        """events (optional): 4 torch.cuda.Event(enable_timing=True) recorded on the current stream at the start, after
        the slab scan, after the all-gather and at the end of the step; the carries / fix-up / Y / X split of the last
        interval comes from the library's own events (ctx.timing_start / timing_stop)."""
        torch = self.torch
        timed = kernel_ms is not None
        if events:
            events[0].record()
        self.ctx.sdf_slab_begin(occ.data_ptr(), self.local_shape, self.z_offset, ws.data_ptr(),
                                ws.numel(), self.summary.data_ptr(), unknown_is_filled,
                                self.ms_begin if timed else None)
        if events:
            events[1].record()
        # one all-gather of the 4-byte records (as int32 words), then the library's carry kernel
        self.dist.all_gather_into_tensor(self.gathered.view(torch.int32).view(-1),
                                         self.summary.view(torch.int32).view(-1))
        if events:
            events[2].record()
        carries = self.carries
        self.ctx.sdf_slab_carries(self.gathered.data_ptr(), self.world, self.rank, self.local_shape[0],
                                  self.local_shape[1], self.full_shape[2], carries.data_ptr())
        self.ctx.sdf_slab_finish(self.local_shape, self.z_offset, self.full_shape[2], resolution,
                                 carries.data_ptr(), sdf.data_ptr(), ws.data_ptr(), ws.numel(),
                                 minmax.data_ptr(), add_virtual_border,
                                 self.ms_finish if timed else None)
        # the field's extrema are those of all slabs: ONE two-element all-reduce, MAX over (-min, max)
        reduce_extrema(self.dist, minmax)
        if events:
            events[3].record()
        if timed:
            kernel_ms[0] = self.ms_begin[0] + self.ms_finish[0]
            kernel_ms[1] = self.ms_finish[1]
            kernel_ms[2] = self.ms_finish[2]
        return carries


def sdf_slabs_single_device(ctx, torch, occ, nslabs, resolution, unknown_is_filled=True,
                            add_virtual_border=False):
    """Runs the slab pipeline for every slab on ONE device, exchanging in-process.  Used by the
    GPU parity tests to exercise the multi-GPU kernels where only one GPU is available."""
    from . import capi
    nx, ny, nz = occ.shape
    dev = occ.device
    # torch ops and the library's kernels must be ordered: run both on torch's current stream
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    parts = []
    for r in range(nslabs):
        local_shape, z0 = slab_of((nx, ny, nz), r, nslabs)
        local = occ[:, :, z0:z0 + local_shape[2]].contiguous()
        nbytes = capi.sdf_workspace_bytes(local_shape)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        summary = torch.empty((nx * ny, 2), dtype=torch.int16, device=dev)
        ctx.sdf_slab_begin(local.data_ptr(), local_shape, z0, ws.data_ptr(), nbytes,
                           summary.data_ptr(), unknown_is_filled)
        parts.append((local_shape, z0, local, ws, nbytes, summary))
    ctx.synchronize()
    gathered = torch.stack([p[5] for p in parts]).contiguous()
    out = torch.empty((nx, ny, nz), dtype=torch.float32, device=dev)
    extrema = []
    for r, (local_shape, z0, local, ws, nbytes, summary) in enumerate(parts):
        carries = torch.empty((nx * ny, 4), dtype=torch.int16, device=dev)
        ctx.sdf_slab_carries(gathered.data_ptr(), nslabs, r, nx, ny, nz, carries.data_ptr())
        sdf = torch.empty(local_shape, dtype=torch.float32, device=dev)
        mm = torch.zeros(2, dtype=torch.float32, device=dev)
        ctx.sdf_slab_finish(local_shape, z0, nz, resolution, carries.data_ptr(), sdf.data_ptr(),
                            ws.data_ptr(), nbytes, mm.data_ptr(), add_virtual_border)
        ctx.synchronize()
        out[:, :, z0:z0 + local_shape[2]] = sdf
        extrema.append(mm.cpu().numpy())
    extrema = np.array(extrema)
    ctx.reset_stream()
    return out, float(extrema[:, 0].min()), float(extrema[:, 1].max())


def point_share(num_points, world, rank):
    """(first, count) of `rank`'s contiguous share of a cloud: the rule of slab_bounds (and of vgt_hipx_point_share)."""
    first, end = slab_bounds(num_points, world)[rank]
    return first, end - first


def sum_counts(dist, counts, root=None):
    """Sums the ranks' private tracking counts (int32 tensor, in place): everywhere, or onto `root` only (the other
    ranks' tensors are then undefined)."""
    if root is None:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    else:
        dist.reduce(counts, dst=root, op=dist.ReduceOp.SUM)
    return counts


class _DeviceInts:
    """int32 device memory owned by the library, described for torch.as_tensor (no copy)."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<i4", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def tracking_grid_tensor(torch, grids, index, device):
    """int32 tensor [2 * cells] aliasing tracking grid `index` of `grids` (seen_free, seen_filled per cell); valid
    while `grids` is open."""
    return torch.as_tensor(_DeviceInts(grids.dev_ptr(index), 2 * grids.num_cells), device=device)


class RaySplit:
    """One rank's share of the raycast of ONE point cloud (device-resident points).

    The context must run on torch's current stream (ctx.set_stream(torch.cuda.current_stream().cuda_stream)) so
    that the raycast kernel and the collective are stream-ordered."""

    def __init__(self, torch, dist, grids, index, device):
        self.torch, self.dist, self.grids, self.index = torch, dist, grids, index
        self.counts = tracking_grid_tensor(torch, grids, index, device)

    def run(self, share_points, max_range, xform, voxel_size, inverse_voxel_size, grid_sizes, counts, root=None):
        """share_points: this rank's point_share of the cloud, float32 [n, 3] on the device.  Afterwards grid `index`
        holds the counts of the WHOLE cloud on every rank (root=None) or on `root`."""
        n = int(share_points.shape[0])
        if n:
            self.grids.raycast_f32_dev(self.index, share_points.data_ptr(), n, max_range, xform, voxel_size,
                                       inverse_voxel_size, grid_sizes, counts)
        return sum_counts(self.dist, self.counts, root)
