#!/usr/bin/env python3
"""Soak test on the GPU: the default pipeline of the PRODUCT library (class records + lane-per-line sweeps) against the
independent cross-check pipeline of the testing library (EDT variant 1: an int16 Z scan, then a pruned outward search per
voxel straight from HBM) on large device-resident grids, bit for bit, over many seeds, distributions and shapes -- sizes at
which the CPU oracle would take minutes per case.  The two pipelines share no kernel, and tests/ pins both to the oracle at
sizes the oracle finishes.  (The search costs O(distance) per voxel: "single" -- one filled voxel, distances of hundreds of
voxels -- runs with one seed only.)

Usage: python tools/soak_variants.py [seeds]     (prints one line per case, exits non-zero on a mismatch)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

import bench
from voxelized_geometry_tools_amd import capi

SHAPES = [(512, 512, 512), (1024, 512, 256), (300, 700, 1000), (2048, 256, 512), (1024, 1024, 128), (40, 2048, 2048),
          (96, 300, 4100), (200, 1500, 320)]
CASES = [("spheres", 0.0), ("unknown_mix", 0.0), ("salt", 1e-4), ("salt", 1e-2), ("salt", 0.3), ("single", 0.0)]


def extract(ctx, occ, shape, variant, vb):
    sdf = torch.empty(shape, dtype=torch.float32, device=occ.device)
    nbytes = capi.sdf_workspace_bytes(shape, variant)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=occ.device)
    minmax = torch.empty(2, dtype=torch.float32, device=occ.device)
    ctx.set_edt_variant(variant)
    ctx.sdf_dev(occ.data_ptr(), shape, 0.01, sdf.data_ptr(), ws.data_ptr(), nbytes, minmax.data_ptr(),
                True, vb)
    torch.cuda.synchronize()
    return sdf, minmax


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    dev = torch.device("cuda", 0)
    product = capi.Context(0)
    product.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx = capi.Context(0, testing=True)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    bad = 0
    for shape in SHAPES:
        for dist, p in CASES:
            for seed in range(1 if dist == "single" else seeds):
                occ = bench.device_occupancy(torch, shape, dist, 1000 + seed, dev, salt_p=p or 0.01)
                vb = bool(seed & 1)
                a, ma = extract(product, occ, shape, 0, vb)
                b, mb = extract(ctx, occ, shape, 1, vb)
                same = torch.equal(a.view(torch.int32), b.view(torch.int32)) and torch.equal(
                    ma.view(torch.int32), mb.view(torch.int32))
                bad += 0 if same else 1
                print("%-18s %-12s p=%-7g seed=%d border=%d %s" % (shape, dist, p, seed, vb, "ok" if same else "MISMATCH"),
                      flush=True)
                del occ, a, b
    ctx.set_edt_variant(0)
    print("mismatches:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
