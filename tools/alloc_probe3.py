#!/usr/bin/env python3
"""Per-step pass-1 times right after (re)allocating the buffers: constant within a placement, or drifting with time?"""
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
    dev = torch.device("cuda", 0)
    ctx = capi.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    shape = (1024,) * 3
    nbytes = capi.sdf_workspace_bytes(shape, 0)
    hold = []
    for trial in range(8):
        occ = bench.device_occupancy(torch, shape, "spheres", 42, dev)
        sdf = torch.empty(shape, dtype=torch.float32, device=dev)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        minmax = torch.zeros(2, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        if trial >= 4:
            time.sleep(3.0)  # idle before the first step
        ctx.timing_start(60)
        for _ in range(60):
            ctx.sdf_dev(occ.data_ptr(), shape, 0.01, sdf.data_ptr(), ws.data_ptr(), nbytes, minmax.data_ptr())
        torch.cuda.synchronize()
        k = ctx.timing_stop().astype(np.float64)
        print("trial %d pass1 per step:" % trial, " ".join("%.2f" % v for v in k[:, 0]), flush=True)
        print("trial %d X     per step:" % trial, " ".join("%.2f" % v for v in k[::4, 2]), flush=True)
        del occ, sdf, ws, minmax
        torch.cuda.empty_cache()
        hold.append(torch.empty((trial + 1) * 29 * 1024 * 1024, dtype=torch.uint8, device=dev))


if __name__ == "__main__":
    main()
