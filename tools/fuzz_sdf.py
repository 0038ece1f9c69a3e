#!/usr/bin/env python3
"""Randomised parity run on the GPU box: the product library's SDF entry points against the CPU oracle, bit for bit, on
many small random grids -- shapes around the 64-voxel word and band boundaries, runs / salt / one-class lines, both
unknown_is_filled settings, the virtual border, the u8 mask entry point, and the Z-slab pipeline with random (also
uneven) slab counts.  Test infrastructure (uses oracle/); prints one line per failure and a summary.

    python tools/fuzz_sdf.py [cases] [seed]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def random_shape(rng):
    def axis():
        kind = rng.integers(0, 6)
        if kind == 0:
            return int(rng.integers(1, 5))
        if kind == 1:
            return int(rng.choice([15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129]))
        if kind == 2:
            return int(rng.integers(1, 40))
        if kind == 3:
            return int(rng.integers(40, 200))
        if kind == 4:
            return int(rng.choice([255, 256, 257, 300]))
        return int(rng.integers(1, 90))
    while True:
        shape = (axis(), axis(), axis())
        if np.prod(shape) <= 400_000:
            return shape


def random_occupancy(rng, shape):
    kind = rng.integers(0, 6)
    nx, ny, nz = shape
    if kind == 0:
        occ = (rng.random(shape) < rng.choice([0.001, 0.01, 0.1, 0.5, 0.9])).astype(np.float32)
    elif kind == 1:  # runs along Z
        flips = rng.random(shape) < rng.choice([0.01, 0.05, 0.3])
        occ = (np.cumsum(flips, axis=2) % 2).astype(np.float32)
    elif kind == 2:  # runs along Y, lines of one class along Z
        flips = rng.random((nx, ny, 1)) < 0.2
        occ = np.broadcast_to((np.cumsum(flips, axis=1) % 2).astype(np.float32), shape).copy()
    elif kind == 3:  # blobs
        occ = np.zeros(shape, dtype=np.float32)
        for _ in range(int(rng.integers(1, 6))):
            c = [rng.integers(0, s) for s in shape]
            r = rng.integers(1, max(2, min(shape) // 2 + 2))
            g = np.ogrid[:nx, :ny, :nz]
            occ[(g[0] - c[0]) ** 2 + (g[1] - c[1]) ** 2 + (g[2] - c[2]) ** 2 <= r * r] = 1.0
    elif kind == 4:  # a single voxel / a plane
        occ = np.zeros(shape, dtype=np.float32)
        if rng.random() < 0.5:
            occ[tuple(rng.integers(0, s) for s in shape)] = 1.0
        else:
            occ[:, :, rng.integers(0, nz)] = 1.0
    else:
        occ = np.full(shape, float(rng.choice([0.0, 1.0])), dtype=np.float32)
    occ[rng.random(shape) < rng.choice([0.0, 0.0, 0.02])] = 0.5
    if rng.random() < 0.3:  # lines of one class
        occ[rng.random((nx, ny)) < 0.5] = float(rng.choice([0.0, 1.0]))
    return occ


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    import torch
    from oracle import oracle as O
    from voxelized_geometry_tools_amd import capi, multi_gpu
    rng = np.random.default_rng(seed)
    ctx = capi.Context(0)
    bad = 0
    kinds = {"host": 0, "mask": 0, "slab": 0}
    for case in range(cases):
        shape = random_shape(rng)
        occ = random_occupancy(rng, shape)
        res = float(rng.choice([0.01, 0.25, 1.0 / 3.0, 2.5]))
        uif = bool(rng.integers(0, 2))
        vb = bool(rng.integers(0, 2))
        mode = rng.choice(["host", "host", "mask", "slab"])
        kinds[mode] += 1
        if mode == "mask":
            mask = (occ > 0.5).astype(np.uint8)
            want = O.sdf_from_mask(mask, res)  # (the oracle's mask entry has no virtual border)
            got, lo, hi = ctx.sdf_from_mask(mask, res)
            extrema_ok = lo == want.min() and hi == want.max()
        elif mode == "slab":
            nslabs = int(rng.integers(1, min(shape[2], 6) + 1))
            want, wlo, whi = O.sdf_from_occupancy(occ, res, uif, vb)
            out, lo, hi = multi_gpu.sdf_slabs_single_device(ctx, torch, torch.from_numpy(occ).cuda(), nslabs, res, uif, vb)
            got = out.cpu().numpy()
            extrema_ok = (lo, hi) == (wlo, whi)
        else:
            want, wlo, whi = O.sdf_from_occupancy(occ, res, uif, vb)
            got, lo, hi = ctx.sdf_from_occupancy(occ, res, uif, vb)
            extrema_ok = (lo, hi) == (wlo, whi)
        same = np.array_equal(got.view(np.uint32), want.view(np.uint32))
        if not (same and extrema_ok):
            bad += 1
            print("MISMATCH case %d: %s shape %s res %g uif %d border %d (field %s, extrema %s)" % (
                case, mode, shape, res, uif, vb, same, extrema_ok), flush=True)
    print("%d cases (%s), %d mismatches" % (cases, ", ".join("%s %d" % kv for kv in kinds.items()), bad))
    ctx.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
