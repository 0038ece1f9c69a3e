#!/usr/bin/env python3
"""Does the speed of the 1024^3 SDF depend on WHERE its buffers were allocated?  Allocates occupancy / result / workspace
again and again in one process (torch caching allocator emptied in between, sometimes with a spacer allocation held to move
the next ones elsewhere) and prints the kernel times of each placement together with the buffers' addresses.

Usage: python tools/alloc_probe.py [trials]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

import bench
from voxelized_geometry_tools_amd import capi


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device("cuda", 0)
    ctx = capi.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    shape = (1024,) * 3
    nbytes = capi.sdf_workspace_bytes(shape, 0)
    spacers = []
    for trial in range(trials):
        occ = bench.device_occupancy(torch, shape, "spheres", 42, dev)
        sdf = torch.empty(shape, dtype=torch.float32, device=dev)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        minmax = torch.zeros(2, dtype=torch.float32, device=dev)
        for _ in range(3):
            ctx.sdf_dev(occ.data_ptr(), shape, 0.01, sdf.data_ptr(), ws.data_ptr(), nbytes, minmax.data_ptr())
        torch.cuda.synchronize()
        ctx.timing_start(20)
        for _ in range(20):
            ctx.sdf_dev(occ.data_ptr(), shape, 0.01, sdf.data_ptr(), ws.data_ptr(), nbytes, minmax.data_ptr())
        torch.cuda.synchronize()
        k = ctx.timing_stop().astype(np.float64).mean(axis=0)
        print("trial %2d  occ %#x sdf %#x ws %#x   pass1 %.3f  Y %.3f  X %.3f  sum %.3f" % (
            trial, occ.data_ptr(), sdf.data_ptr(), ws.data_ptr(), k[0], k[1], k[2], k.sum()), flush=True)
        del occ, sdf, ws, minmax
        torch.cuda.empty_cache()
        if trial % 2 == 1:
            # hold an odd-sized block so that the next placement starts somewhere else
            spacers.append(torch.empty((trial + 1) * 37 * 1024 * 1024 + 4096 * trial, dtype=torch.uint8, device=dev))


if __name__ == "__main__":
    main()
