#!/usr/bin/env python3
"""Short-line kernels (csrc/edt_short_kernels.hip) against the sweeps on the same grids: per-kernel ms of a batch of B cubes
of edge n and of one cube, with the limit of the short kernels at 64 (short) and at 0 (sweeps), testing library.
Usage: python tools/short_vs_sweep.py [edges...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from voxelized_geometry_tools_amd import capi


def main():
    edges = [int(a) for a in sys.argv[1:]] or [8, 16, 24, 32, 40, 48, 64]
    dev = torch.device("cuda", 0)
    ctx = capi.Context(0, testing=True)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    for n in edges:
        shape = (n, n, n)
        for batch in (1, max(1, (64 ** 3 * 64) // n ** 3 // 4)):
            grids = torch.stack([bench.device_occupancy(torch, shape, "spheres", 42 + b, dev) for b in range(min(batch, 64))])
            if batch > 64:
                grids = grids.repeat((batch + 63) // 64, 1, 1, 1)[:batch].contiguous()
            fields = torch.empty_like(grids)
            nbytes = capi.sdf_batch_workspace_bytes(batch, shape)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            row = []
            ref = None
            for rows in (64, 0):
                ctx.set_short_line_rows(rows)
                for _ in range(3):
                    ctx.sdf_batch_dev(grids.data_ptr(), batch, shape, 0.01, fields.data_ptr(), ws.data_ptr(), nbytes, None)
                torch.cuda.synchronize()
                ctx.timing_start(10)
                t0 = time.perf_counter()
                for _ in range(10):
                    ctx.sdf_batch_dev(grids.data_ptr(), batch, shape, 0.01, fields.data_ptr(), ws.data_ptr(), nbytes, None)
                torch.cuda.synchronize()
                wall = (time.perf_counter() - t0) / 10 * 1e3
                k = ctx.timing_stop().mean(axis=0)
                row.append((rows, wall, k))
                if ref is None:
                    ref = fields.clone()
                else:
                    assert torch.equal(ref.view(torch.int32), fields.view(torch.int32))
            ctx.set_short_line_rows(64)
            print("n=%3d batch=%5d | short: wall %.4f ms  p1 %.4f Y %.4f X %.4f | sweeps: wall %.4f ms  p1 %.4f Y %.4f X %.4f"
                  % (n, batch, row[0][1], *row[0][2], row[1][1], *row[1][2]))


if __name__ == "__main__":
    main()
