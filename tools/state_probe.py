#!/usr/bin/env python3
"""How the box's speed drifts over time: the 1024^3 D1 SDF in a loop for a few minutes, per-kernel times (the library's
events) averaged per block of steps, printed with the wall-clock second -- continuous load first, then with idle gaps.

Usage: python tools/state_probe.py [seconds_busy] [seconds_gapped]
"""
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
    busy = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
    gapped = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
    dev = torch.device("cuda", 0)
    ctx = capi.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    shape = (1024,) * 3
    occ = bench.device_occupancy(torch, shape, "spheres", 42, dev)
    sdf = torch.empty(shape, dtype=torch.float32, device=dev)
    nbytes = capi.sdf_workspace_bytes(shape, 0)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    minmax = torch.zeros(2, dtype=torch.float32, device=dev)

    def block(steps):
        ctx.timing_start(steps)
        for _ in range(steps):
            ctx.sdf_dev(occ.data_ptr(), shape, 0.01, sdf.data_ptr(), ws.data_ptr(), nbytes, minmax.data_ptr())
        torch.cuda.synchronize()
        return ctx.timing_stop().astype(np.float64).mean(axis=0)

    t0 = time.time()
    print("# second  pass1  Y  X  (ms, mean of 20 steps); continuous load")
    while time.time() - t0 < busy:
        k = block(20)
        print("%7.1f  %.3f %.3f %.3f" % (time.time() - t0, k[0], k[1], k[2]), flush=True)
    print("# 20 steps, then 2 s idle")
    t1 = time.time()
    while time.time() - t1 < gapped:
        k = block(20)
        print("%7.1f  %.3f %.3f %.3f" % (time.time() - t0, k[0], k[1], k[2]), flush=True)
        time.sleep(2.0)


if __name__ == "__main__":
    main()
