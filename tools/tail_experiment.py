#!/usr/bin/env python3
"""How the sweep passes' time follows the number of work items: X-pass items = ny * ceil(nz / 64), Y-pass items =
nx * ceil(nz / 64), 4096 waves resident.  Prints per shape the kernel times and the time per round of items.
    python tools/tail_experiment.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from voxelized_geometry_tools_amd import capi


def main():
    dev = torch.device("cuda", 0)
    ctx = capi.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    for ny in (512, 768, 896, 960, 1024, 1088, 1152, 1280, 1536):
        shape = (1024, ny, 1024)
        occ = bench.device_occupancy(torch, shape, "spheres", 42, dev)
        sdf = torch.empty(shape, dtype=torch.float32, device=dev)
        nbytes = capi.sdf_workspace_bytes(shape)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        mm = torch.empty(2, dtype=torch.float32, device=dev)
        ms = np.zeros(3, dtype=np.float32)
        runs = []
        for _ in range(6):
            ctx.sdf_dev(occ.data_ptr(), shape, 0.01, sdf.data_ptr(), ws.data_ptr(), nbytes, mm.data_ptr(), True, False, ms)
            runs.append(ms.copy())
        z, y, x = np.median(np.array(runs[2:]), axis=0)
        x_items = ny * 16
        print("ny %5d: Z %.3f Y %.3f X %.3f ms | X items %6d = %.2f rounds of 4096, %.3f ms per round, %.1f ns per voxel-row"
              % (ny, z, y, x, x_items, x_items / 4096.0, x / (x_items / 4096.0), x * 1e6 / (x_items * 1024.0)))
        del occ, sdf, ws
        torch.cuda.empty_cache()
    ctx.close()


if __name__ == "__main__":
    main()
