#!/usr/bin/env python3
"""Which buffer's placement decides pass 1's speed?  (A) occupancy kept, workspace allocated again; (B) workspace kept,
occupancy allocated again; each time: pass-1 / Y / X kernel times and the time of a plain read of the occupancy (torch sum).
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
    dev = torch.device("cuda", 0)
    ctx = capi.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    shape = (1024,) * 3
    nbytes = capi.sdf_workspace_bytes(shape, 0)
    sdf = torch.empty(shape, dtype=torch.float32, device=dev)
    minmax = torch.zeros(2, dtype=torch.float32, device=dev)

    def measure(tag, occ, ws):
        for _ in range(3):
            ctx.sdf_dev(occ.data_ptr(), shape, 0.01, sdf.data_ptr(), ws.data_ptr(), nbytes, minmax.data_ptr())
        torch.cuda.synchronize()
        ctx.timing_start(20)
        for _ in range(20):
            ctx.sdf_dev(occ.data_ptr(), shape, 0.01, sdf.data_ptr(), ws.data_ptr(), nbytes, minmax.data_ptr())
        torch.cuda.synchronize()
        k = ctx.timing_stop().astype(np.float64).mean(axis=0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        flat = occ.view(-1)
        flat.sum()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            flat.sum()
        e1.record()
        torch.cuda.synchronize()
        print("%s  occ %#x ws %#x  pass1 %.3f  Y %.3f  X %.3f   read-only sum %.3f ms" % (
            tag, occ.data_ptr(), ws.data_ptr(), k[0], k[1], k[2], e0.elapsed_time(e1) / 10), flush=True)

    hold = []
    occ = bench.device_occupancy(torch, shape, "spheres", 42, dev)
    for i in range(6):
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        measure("A%d occ kept, ws new" % i, occ, ws)
        del ws
        torch.cuda.empty_cache()
        hold.append(torch.empty((i + 1) * 53 * 1024 * 1024, dtype=torch.uint8, device=dev))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    for i in range(6):
        measure("B%d ws kept, occ new" % i, occ, ws)
        del occ
        torch.cuda.empty_cache()
        hold.append(torch.empty((i + 1) * 41 * 1024 * 1024, dtype=torch.uint8, device=dev))
        occ = bench.device_occupancy(torch, shape, "spheres", 42, dev)


if __name__ == "__main__":
    main()
